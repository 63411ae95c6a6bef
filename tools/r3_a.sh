#!/bin/bash
# round-3 GPU pass A: GEMM NT parity + A/B of the phased main loop against the round-2 library
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
LIBS=${1:-base,product}
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_nt" > $OUT/a_tests.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/a_tests.log
only=nt timeout -k 10 500 python tools/gemm_bench.py $LIBS 5 > $OUT/a_bench.log 2>&1; echo "bench exit $?"; cat $OUT/a_bench.log
