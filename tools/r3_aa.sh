#!/bin/bash
# round-3 pass aa: the N > 1 defaults of bench.py (RCCL CU budget 8, GEMM launches planned for 248 CUs) on a one-rank RCCL group
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --rccl-max-ctas 8 --reserve-cus 8 2>&1 | grep -v amdgpu | tail -3 | tee $OUT/aa_one_rank_rccl_budget.log
