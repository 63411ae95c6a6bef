#!/bin/bash
# round-3 pass j: weight gradients on a second stream, A/B on the whole step
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
for a in ${ARCHS:-vit_b_16 vit_s_16}; do
  arch=$a timeout -k 10 300 python tools/wgrad_ab.py 5 6 2>&1 | grep -v amdgpu | tee -a $OUT/j_wgrad_ab.log || exit 1
done
