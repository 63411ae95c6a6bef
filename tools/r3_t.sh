#!/bin/bash
# round-3 pass t: PMC traffic of the other configs, then their bench lines (carrying that traffic) and rocprof kernel stats
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
mkdir -p gpurun_out/r03
TAG=r03 bash tools/traffic_others.sh || exit 1
for f in gpurun_out/r03/traffic_per_launch_*_b*.json; do cp $f profiles/r03_$(basename $f); done
bash tools/gpu_round.sh r03 others
