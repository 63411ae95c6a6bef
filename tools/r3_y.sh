#!/bin/bash
# round-3 pass y: one-kernel attention backward (product) against the two-kernel form (base)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
B=${B:-256} timeout -k 10 300 python tools/attn_ab.py base product ${NS:-197} 2>&1 | grep -v amdgpu | tee $OUT/y_attn_ab.log
