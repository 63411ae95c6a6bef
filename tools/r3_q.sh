#!/bin/bash
# round-3 pass q: late-start spread of the persistent NT kernel, 0 / 30 / 60 % of a tile time for every epilogue
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
only=nt timeout -k 10 600 python tools/gemm_bench.py base,nap0,nap30,nap60,product 5 2>&1 | grep -v amdgpu | tee $OUT/q_bench.log
