#!/bin/bash
# round-3 final pass: GPU suite + fp32 element checks of multi-tile NT launches against the previous round's kernel build
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -v amdgpu | tail -4 | tee $OUT/ac_pytest.log || exit 1
for shape in "50432 768 768" "50432 3072 768" "25216 1024 4096" "12544 768 3072"; do
  timeout -k 10 200 python tools/nt_diag.py product base $shape 2>&1 | grep -v amdgpu | grep "product" | tee -a $OUT/ac_diag.log
done
