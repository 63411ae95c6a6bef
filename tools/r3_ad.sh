#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "lucid or mae" 2>&1 | grep -v amdgpu | tail -12 | tee $OUT/ad_pytest.log
