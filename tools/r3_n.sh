#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
for shape in "9000 2048 192" "9000 2048 192" "257 264 256"; do
  timeout -k 10 120 python tools/nt_diag.py product base $shape 2>&1 | grep -v amdgpu | tee -a $OUT/n_diag.log
done
