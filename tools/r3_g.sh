#!/bin/bash
# round-3 GPU pass G: LayerNorm parity + A/B at D = 384 and 768, then ViT-S step A/B
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "layernorm" > $OUT/g_tests.log 2>&1; echo "pytest exit $?"; tail -2 $OUT/g_tests.log
for D in 384 768; do D=$D only=misc timeout -k 10 300 python tools/gemm_bench.py head,product 5 2>&1 | grep "ln_"; done
arch=vit_s_16 timeout -k 10 300 python tools/step_ab.py head,product 2>&1 | grep -v amdgpu
