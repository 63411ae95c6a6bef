#!/bin/bash
# round-3 GPU pass E: new-feature tests (streaming attention, dim_head, interpolation, vit_h_14 geometry) then the whole suite
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "${1:-streaming or head_dim or interpolate or vit_h_14 or 384px}" -rP > $OUT/e_new.log 2>&1; echo "new tests exit $?"; grep -E "passed|failed|^E  |Error" $OUT/e_new.log | head -20
if [ "$2" != "only" ]; then
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/e_all.log 2>&1; echo "all tests exit $?"; tail -3 $OUT/e_all.log
fi
