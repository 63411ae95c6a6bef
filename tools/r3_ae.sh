#!/bin/bash
# round-3 pass ae: graph-replayed step with the final kernels (ViT-B/16, ViT-S/16)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
rm -f $OUT/ae_graph.log
for a in vit_b_16 vit_s_16; do
  timeout -k 10 300 python bench.py --arch $a --graph --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -v amdgpu | cut -c1-160 | tee -a $OUT/ae_graph.log || exit 1
done
