#!/bin/bash
# round-3 GPU pass C: full GPU suite + NT phase stamps of the instrumented build
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/c_tests.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/c_tests.log
timeout -k 10 300 python tools/gemm_stamps.py stamps > $OUT/c_stamps.log 2>&1; echo "stamps exit $?"; cat $OUT/c_stamps.log
