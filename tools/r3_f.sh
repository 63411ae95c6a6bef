#!/bin/bash
# round-3 GPU pass F: bench lines of every BASELINE config, eager and graph-replayed
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
for A in ${1:-vit_s_16 mae_b_16 vit_b_16 vit_l_16}; do
  for G in "" "--graph"; do
    T=$OUT/f_bench_${A}${G:+_graph}.json
    timeout -k 10 300 python bench.py --arch $A --steps 20 --warmup 5 --no-cpu-baseline --no-roofline $G > $T 2> ${T%.json}.err; echo "$A $G exit $?"
    python - "$T" <<'PY'
import json,sys
try:
    r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("   ", r["config"]["arch"], r["config"].get("launch"), r["value"], "img/s", r["ms_per_step"], "ms/step (median", r["ms_per_step_median"], ")")
except Exception as e:
    print("   parse error", e); print(open(sys.argv[1][:-5]+".err").read()[-1500:])
PY
  done
done
