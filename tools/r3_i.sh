#!/bin/bash
# round-3 pass i: NT tile-height sweep with the phased K loop (does the cost model still pick the fastest height?)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
libs=t128,t192,t256,t320,t1384 timeout -k 10 900 python tools/tile_sweep.py 4 2>&1 | grep -v amdgpu | tee $OUT/i_tile_sweep.log
