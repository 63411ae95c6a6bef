#!/bin/bash
# One GPU-box pass for the round's evidence: tests, headline bench, rocprofv3 kernel stats, the two PMC traffic passes and
# the bench lines of the other BASELINE configs.  Usage (through gpurun): bash tools/gpu_round.sh <tag> [quick|notests|others]
# (quick: tests + headline evidence only; others: only the other configs)
set -o pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
if [ "$2" != "notests" ] && [ "$2" != "others" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -q -x -rP > $OUT/t_all.log 2>&1; echo "pytest exit $?" | tee -a $OUT/t_all.log
  grep -E "passed|failed" $OUT/t_all.log | tail -2
fi
if [ "$2" != "others" ]; then
timeout -k 10 400 python bench.py --breakdown > $OUT/bench_vit_b_16.json 2> $OUT/bench_vit_b_16.err || exit 1
cat $OUT/bench_vit_b_16.json | head -c 1500; echo
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_b -o b -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/prof_b.log 2>&1 || { tail -5 $OUT/prof_b.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/pmc_fetch.log 2>&1 || { tail -5 $OUT/pmc_fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/pmc_write.log 2>&1 || { tail -5 $OUT/pmc_write.log; exit 1; }
cd $ROOT
find $OUT/prof_b -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_vit_b_16.csv
F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
python tools/traffic_from_pmc.py $F $W $OUT/traffic_per_launch_vit_b_16_b256.json > $OUT/traffic.log 2>&1; tail -3 $OUT/traffic.log
rm -rf $OUT/pmc_fetch $OUT/pmc_write            # raw per-dispatch CSVs are large; the reduction is kept
find $OUT/prof_b -type f ! -name "*kernel_stats.csv" -delete
fi
if [ "$2" != "quick" ]; then
  for A in vit_s_16 vit_l_16 mae_b_16 simplevit_b_16; do
    timeout -k 10 300 python bench.py --arch $A --steps 20 --warmup 5 --no-cpu-baseline --breakdown > $OUT/bench_$A.json 2> $OUT/bench_$A.err; echo "$A exit $?"; head -c 400 $OUT/bench_$A.json; echo
  done
  timeout -k 10 300 python bench.py --robust --steps 20 --warmup 5 --no-cpu-baseline --breakdown > $OUT/bench_vit_b_16_robust.json 2> $OUT/bench_vit_b_16_robust.err; echo "robust exit $?"
  timeout -k 10 300 python bench.py --noise-std 0.1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/bench_vit_b_16_noise.json 2> $OUT/bench_vit_b_16_noise.err; echo "noise exit $?"
  cd /tmp
  for A in vit_s_16 vit_l_16 mae_b_16; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$A -o p -- python3 $ROOT/bench.py --arch $A --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/prof_$A.log 2>&1
    find $OUT/prof_$A -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_$A.csv
    find $OUT/prof_$A -type f ! -name "*kernel_stats.csv" -delete
  done
fi
echo done
