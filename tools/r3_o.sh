#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python tools/nt_check.py ${1:-nap} 2>&1 | grep -v amdgpu | tee $OUT/o_check_$1.log || exit 1
only=nt timeout -k 10 400 python tools/gemm_bench.py base,product,${1:-nap} 5 2>&1 | grep -v amdgpu | tee $OUT/o_bench_$1.log
