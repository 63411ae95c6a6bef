#!/bin/bash
# round-3 pass p: persistent NT kernel in the product: bit-equality vs the previous commit, GEMM bench, whole step A/B
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python tools/nt_check.py base 2>&1 | grep -v amdgpu | tee $OUT/p_check.log || exit 1
only=nt timeout -k 10 400 python tools/gemm_bench.py base,product 5 2>&1 | grep -v amdgpu | tee $OUT/p_bench.log
for a in ${ARCHS:-}; do
  arch=$a timeout -k 10 400 python tools/step_ab.py base,product 5 6 2>&1 | grep -v amdgpu | tee -a $OUT/p_step_ab.log
done
