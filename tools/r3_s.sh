#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
per_wave=1 timeout -k 10 300 python tools/tn_stamps.py stamps 2>&1 | grep -v amdgpu | tee $OUT/s_tn_stamps.log
