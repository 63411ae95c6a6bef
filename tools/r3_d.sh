#!/bin/bash
# round-3 GPU pass D: phase stamps + A/B of schedule variants
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
LIBS=${1:-product}
timeout -k 10 300 python tools/gemm_stamps.py stamps > $OUT/d_stamps.log 2>&1; echo "stamps exit $?"; cat $OUT/d_stamps.log
only=nt timeout -k 10 600 python tools/gemm_bench.py $LIBS 5 > $OUT/d_bench.log 2>&1; echo "bench exit $?"; cat $OUT/d_bench.log
