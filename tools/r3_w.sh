#!/bin/bash
# round-3 pass w: what the driver runs at round end -- smoke(), the GPU suite, the default bench line
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -3 | tee $OUT/w_smoke.log || exit 1
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -v amdgpu | tail -5 | tee $OUT/w_pytest.log || exit 1
timeout -k 10 400 python bench.py 2>&1 | grep -v amdgpu | tee $OUT/w_bench.json | cut -c1-700
