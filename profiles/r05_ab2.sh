#!/bin/bash
# Round 5: interleaved A/B of library builds on a few workloads.   bash profiles/r05_ab2.sh <tag> "<variants>" "<workload:steps> ..." [rounds]
# variant "main" = csrc/libycnr_als.so, anything else = ablibs/lib_<variant>.so
TAG=$1; VARS=$2; WLS=$3; ROUNDS=${4:-2}
mkdir -p gpurun_out
for r in $(seq 1 $ROUNDS); do
  for wls in $WLS; do
    wl=${wls%%:*}; st=${wls##*:}
    for v in $VARS; do
      lib=""; [ $v != main ] && lib="$PWD/ablibs/lib_$v.so"
      YCNR_ALS_LIB=$lib timeout -k 10 600 python bench.py --workload $wl --steps $st --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_ab_${v}_${wl}_$r.json 2> gpurun_out/${TAG}_ab_${v}_${wl}_$r.err
      python3 -c "
import json
d=json.loads(open('gpurun_out/${TAG}_ab_${v}_${wl}_$r.json').read().strip().splitlines()[-1]); it=d['roofline']['iteration']
print('round $r %-8s %-8s ms %.3f user %.3f item %.3f' % ('$wl', '$v', d['ms_per_step'], it['byUser_ms'], it['byItem_ms']))" || tail -n 3 gpurun_out/${TAG}_ab_${v}_${wl}_$r.err
    done
  done
done
