#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
timeout -k 10 400 python tools/reserve_ab.py 0 8 16 32 2>&1 | grep -v amdgpu | tee $OUT/ab_reserve.log
