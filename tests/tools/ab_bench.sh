#!/bin/bash
# A/B of two builds of libycnr_als.so on the GPU box (YCNR_ALS_LIB selects the build):
#   bash tests/tools/ab_bench.sh <tag> <libA> <libB> [workload ...]
# AB_ENV="YCNR_NO_OVERLAP=1" runs the kernels in stream order (each kernel's own time); AB_STEPS sets the steps.
# per workload: interleaved bench lines (ms per iteration and per half-step) and a bit-for-bit comparison of
# the factors both builds produce.  Output: gpurun_out/ab_<tag>.log
TAG=$1; A=$2; B=$3; shift 3
WLS=${@:-mal}
OUT=gpurun_out/ab_$TAG.log
mkdir -p gpurun_out; : > $OUT
for wl in $WLS; do
  for rep in 1 2; do
    for L in $A $B; do
      env $AB_ENV YCNR_ALS_LIB=$PWD/$L timeout 600 python bench.py --workload $wl --steps ${AB_STEPS:-8} --warmup 2 --no-cpu-baseline 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); it = d['roofline']['iteration']
print('$wl $L ms %.3f user %.3f item %.3f' % (d['ms_per_step'], it['byUser_ms'], it['byItem_ms']), ' '.join('%s=%.3f' % (k['kernel'].replace('als_','').replace('_kernel',''), k['avg_launch_ms']) for k in d['roofline']['kernels']))" >> $OUT
    done
  done
  YCNR_ALS_LIB=$PWD/$A timeout 600 python bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --dump-factors /tmp/ab_a.npz > /dev/null 2>&1
  YCNR_ALS_LIB=$PWD/$B timeout 600 python bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --dump-factors /tmp/ab_b.npz > /dev/null 2>&1
  python - >> $OUT <<PY
import numpy as np
a, b = np.load('/tmp/ab_a.npz'), np.load('/tmp/ab_b.npz')
for n in 'UV':
    d = np.abs(a[n].astype(np.float64) - b[n]).max(1) / np.maximum(np.abs(a[n]).max(1), 1e-30)
    print('$wl', n, 'rows differing', int((d > 0).sum()), 'of', len(d), 'max rel', float(d.max()))
PY
done
cat $OUT
