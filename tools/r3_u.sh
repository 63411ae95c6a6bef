#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python tools/torch_op_sites.py 2>&1 | grep -v amdgpu | tee $OUT/u_op_sites.log
