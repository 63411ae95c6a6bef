#!/bin/bash
# round-3 pass x: phased K loops with and without the wave-group stagger
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
LIB=${1:-nostag}
timeout -k 10 300 python tools/nt_check.py $LIB 2>&1 | grep -v amdgpu | tail -3 | tee $OUT/x_check_$LIB.log || exit 1
timeout -k 10 300 python tools/tn_check.py $LIB 2>&1 | grep -v amdgpu | tail -3 | tee -a $OUT/x_check_$LIB.log || exit 1
timeout -k 10 600 python tools/gemm_bench.py product,$LIB 5 2>&1 | grep -v amdgpu | grep "NT\|TN" | tee $OUT/x_bench_$LIB.log
