#!/bin/bash
# round-3 pass k: full GPU suite + headline bench + ViT-S / ViT-L / MAE bench lines with the side stream in the product
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -v amdgpu | tail -15 | tee $OUT/k_pytest.log
timeout -k 10 300 python bench.py 2>&1 | grep -v amdgpu | tee $OUT/k_bench_vit_b_16.json
for a in vit_s_16 vit_l_16 mae_b_16; do
  timeout -k 10 300 python bench.py --arch $a --no-cpu-baseline 2>&1 | grep -v amdgpu | tee $OUT/k_bench_$a.json
done
