#!/bin/bash
# round-3 pass m: phase stamps of the persistent NT kernel launched one tile per workgroup (what does a tile cost now?)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python tools/gemm_stamps.py ${1:-stamps1} 2>&1 | grep -v amdgpu | tee $OUT/m_stamps_${1:-stamps1}.log
