#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -v amdgpu | tail -8 | tee $OUT/v_pytest.log
