set -o pipefail
R=$PWD; OUT=$R/gpurun_out/${TAG:-r03}; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for spec in "vit_s_16 256" "vit_l_16 128" "mae_b_16 256"; do
  set -- $spec; A=$1; B=$2
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pf_$A -o f -- python3 $R/bench.py --arch $A --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/pf_$A.log 2>&1 || { tail -3 $OUT/pf_$A.log; exit 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pw_$A -o w -- python3 $R/bench.py --arch $A --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/pw_$A.log 2>&1 || { tail -3 $OUT/pw_$A.log; exit 1; }
  F=$(find $OUT/pf_$A -name "*counter_collection.csv" | head -1); W=$(find $OUT/pw_$A -name "*counter_collection.csv" | head -1)
  python3 $R/tools/traffic_from_pmc.py $F $W $OUT/traffic_per_launch_${A}_b$B.json > $OUT/traffic_$A.log 2>&1; tail -2 $OUT/traffic_$A.log
  rm -rf $OUT/pf_$A $OUT/pw_$A
done
echo done
