#!/bin/bash
# round-3 GPU pass A: GEMM parity (pytest -k $2) + A/B of library builds on the GEMM shapes (only=$3)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
LIBS=${1:-base,product}
KEXPR=${2:-gemm}
ONLY=${3:-}
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "$KEXPR" > $OUT/a_tests.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/a_tests.log
only=$ONLY timeout -k 10 500 python tools/gemm_bench.py $LIBS 5 > $OUT/a_bench.log 2>&1; echo "bench exit $?"; cat $OUT/a_bench.log
