#!/bin/bash
# rocprofv3 --pmc passes over tools/du_pmc.py (program directly behind "--"), reduced per kernel name.  bash tools/du_pmc.sh TAG
# SETS="A B;C D" replaces the default counter sets (one pass per set; the TA / TCP blocks take two or three counters per pass)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT; export TMPDIR=/tmp
i=0
if [ -n "$SETS" ]; then IFS=';' read -ra LIST <<< "$SETS"; else LIST=("SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS"); fi
for SET in "${LIST[@]}"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pmc_$i -o c -- python3 $ROOT/tools/du_pmc.py > $OUT/pmc_$i.log 2>&1) || { tail -5 $OUT/pmc_$i.log; exit 1; }
  F=$(find $OUT/pmc_$i -name "*counter_collection.csv" | head -1)
  python $ROOT/tools/pmc_summary.py $F > $OUT/pmc_${i}_summary.txt 2>&1; cat $OUT/pmc_${i}_summary.txt
  rm -rf $OUT/pmc_$i
done
