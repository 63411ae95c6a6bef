#!/bin/bash
# One gpurun call = a sequence of named steps, stopping at the first failure (no GPU step is started behind a failed one).
#   bash tools/gpu_pass.sh TAG step1 step2 ...       outputs under gpurun_out/TAG/
# steps: traffic[:ARCH[:BATCH]]  tests[:KEXPR]  bench[:ARCH[:extra flags]]  graphnodes:ARCH  gemmab:LIBS[:only]  stepab:LIBS[:ARCHS]  prof:ARCH  pmcsq:ARCH
#        py:SCRIPT[:args]  (python tools/SCRIPT args)      prof:ARCH[:extra bench flags[:suffix]]
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
for STEP in "$@"; do
  IFS=: read -r NAME A1 A2 A3 <<< "$STEP"
  echo "== $STEP"
  case $NAME in
    tests)
      timeout -k 10 1000 python -m pytest tests -m gpu -q -x ${A1:+-k "$A1"} > $OUT/tests_${A1// /_}.log 2>&1; RC=$?
      tail -4 $OUT/tests_${A1// /_}.log ;;
    bench)
      A=${A1:-vit_b_16}
      timeout -k 10 400 python bench.py --arch $A $A2 > $OUT/bench_$A${A3:+_$A3}.json 2> $OUT/bench_$A${A3:+_$A3}.err; RC=$?
      head -c 600 $OUT/bench_$A${A3:+_$A3}.json; echo; tail -12 $OUT/bench_$A${A3:+_$A3}.err ;;
    graphnodes)
      timeout -k 10 300 python tools/graph_nodes.py $A1 > $OUT/graphnodes_$A1.log 2>&1; RC=$?
      tail -70 $OUT/graphnodes_$A1.log ;;
    gemmab)
      only=$A2 timeout -k 10 500 python tools/gemm_bench.py $A1 5 > $OUT/gemmab.log 2>&1; RC=$?; cat $OUT/gemmab.log ;;
    stepab)
      rm -f $OUT/stepab.log; RC=0
      for a in ${A2:-vit_b_16}; do arch=$a timeout -k 10 400 python tools/step_ab.py $A1 5 6 2>&1 | grep -v amdgpu | tee -a $OUT/stepab.log; RC=$?; [ $RC -ne 0 ] && break; done ;;
    prof)
      A=${A1:-vit_b_16}
      (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$A${A3:+_$A3} -o p -- python3 $ROOT/bench.py --arch $A $A2 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/prof_$A${A3:+_$A3}.log 2>&1); RC=$?
      find $OUT/prof_$A${A3:+_$A3} -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$A${A3:+_$A3}.csv
      find $OUT/prof_$A${A3:+_$A3} -type f ! -name "*kernel_stats.csv" -delete
      head -12 $OUT/kernel_stats_$A${A3:+_$A3}.csv | cut -c1-200 ;;
    pmcsq)
      A=${A1:-vit_b_16}; RC=0
      for SET in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" \
                 "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
        N=$(echo $SET | cut -d' ' -f1)
        (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pmc_$N -o c -- python3 $ROOT/bench.py --arch $A $A2 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/pmc_$N.log 2>&1); RC=$?
        [ $RC -ne 0 ] && { tail -5 $OUT/pmc_$N.log; break; }
        F=$(find $OUT/pmc_$N -name "*counter_collection.csv" | head -1)
        python tools/pmc_summary.py $F > $OUT/pmc_${N}_summary.txt 2>&1; head -40 $OUT/pmc_${N}_summary.txt
        rm -rf $OUT/pmc_$N
      done ;;
    traffic)
      A=${A1:-vit_b_16}; BATCH=${A2:-256}; RC=0
      for C in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -o c -- python3 $ROOT/bench.py --arch $A --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/pmc_$C.log 2>&1); RC=$?
        [ $RC -ne 0 ] && { tail -5 $OUT/pmc_$C.log; break; }
      done
      if [ $RC -eq 0 ]; then
        F=$(find $OUT/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
        python tools/traffic_from_pmc.py $F $W $OUT/traffic_per_launch_${A}_b$BATCH.json > $OUT/traffic_$A.log 2>&1; tail -30 $OUT/traffic_$A.log
        rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
      fi ;;
    dbgagent)      # one command under the ROCm debug agent: on a GPU memory fault it prints the faulting waves (kernel, pc, registers)
      HSA_TOOLS_LIB=/opt/rocm/lib/librocm-debug-agent.so.2 HSA_ENABLE_DEBUG=1 timeout -k 10 240 python $A1 $A2 > $OUT/dbgagent.log 2>&1; RC=$?
      grep -v "^  node" $OUT/dbgagent.log | head -150 | cut -c1-300 ;;
    py)
      timeout -k 10 600 python tools/$A1 $A2 $A3 > $OUT/py_${A1%.py}.log 2>&1; RC=$?; tail -60 $OUT/py_${A1%.py}.log ;;
    *) echo "unknown step $NAME"; RC=9 ;;
  esac
  echo "== $STEP exit $RC"
  [ $RC -ne 0 ] && exit $RC
done
echo done
