#!/bin/bash
# round-3 pass r: full GPU suite with the persistent NT kernel + headline bench + whole-step A/B on the other configs
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -v amdgpu | tail -15 | tee $OUT/r_pytest.log || exit 1
rm -f $OUT/r_step_ab.log
for a in vit_b_16 vit_s_16 vit_l_16 mae_b_16; do
  arch=$a timeout -k 10 400 python tools/step_ab.py base,product 5 6 2>&1 | grep -v amdgpu | tee -a $OUT/r_step_ab.log
done
