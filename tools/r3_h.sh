#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
LIB=${1:-s2}
timeout -k 10 300 python tools/nt_check.py $LIB 2>&1 | grep -v amdgpu | tee $OUT/h_check_$LIB.log
only=nt timeout -k 10 400 python tools/gemm_bench.py product,$LIB 5 2>&1 | grep -v amdgpu | tee $OUT/h_bench_$LIB.log
