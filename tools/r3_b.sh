#!/bin/bash
# round-3 GPU pass B: full GPU test suite + in-process step A/B + headline bench
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
LIBS=${1:-base,product}
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/b_tests.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/b_tests.log
timeout -k 10 400 python tools/step_ab.py $LIBS > $OUT/b_step_ab.log 2>&1; echo "step_ab exit $?"; cat $OUT/b_step_ab.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --breakdown > $OUT/b_bench.json 2> $OUT/b_bench.err; echo "bench exit $?"; head -c 1200 $OUT/b_bench.json
