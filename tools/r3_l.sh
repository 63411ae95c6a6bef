#!/bin/bash
# round-3 pass l: persistent NT kernel -- bit-equality against the previous commit, GEMM tests, A/B
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python tools/nt_check.py base 2>&1 | grep -v amdgpu | tee $OUT/l_check.log || exit 1
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | grep -v amdgpu | tail -5 | tee $OUT/l_pytest.log || exit 1
only=nt timeout -k 10 400 python tools/gemm_bench.py base,one,product 5 2>&1 | grep -v amdgpu | tee $OUT/l_bench.log
