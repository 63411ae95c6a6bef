#!/bin/bash
# round-3 pass z: whole-step A/B of the working tree against HEAD (tools/build_dev.py base --rev HEAD)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
rm -f $OUT/z_step_ab.log
for a in ${ARCHS:-vit_b_16}; do
  arch=$a timeout -k 10 400 python tools/step_ab.py base,product 5 6 2>&1 | grep -v amdgpu | tee -a $OUT/z_step_ab.log
done
