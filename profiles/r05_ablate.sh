#!/bin/bash
# Round 5: what the two phases of the fused planes row kernel cost alone (MAL scale, stream order).
#   bash profiles/r05_ablate.sh <tag>
# ablibs/lib_nogram.so / lib_nosolve.so: `make -C you-can-not-recommend_amd/csrc OUT=../../ablibs/lib_nogram.so EXTRA=-DYCNR_ABLATE_GRAM`
# (als_gram_solve_x6p_kernel without its Gramian) and `... lib_nosolve.so EXTRA=-DYCNR_ABLATE_SOLVE` (without its solve): results
# are wrong by construction, only the times count.
TAG=$1
mkdir -p gpurun_out
for v in base nogram nosolve; do
  lib=""; [ $v != base ] && lib="$PWD/ablibs/lib_$v.so"
  YCNR_ALS_LIB=$lib YCNR_NO_ROW_PAIR=1 YCNR_NO_OVERLAP=1 YCNR_IGNORE_NUMERIC=1 timeout -k 10 300 python bench.py --workload mal --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_ablate_$v.json 2> gpurun_out/${TAG}_ablate_$v.err
  python3 -c "
import json
d=json.loads(open('gpurun_out/${TAG}_ablate_$v.json').read().strip().splitlines()[-1]); it=d['roofline']['iteration']
print('$v ms %.3f user %.3f item %.3f' % (d['ms_per_step'], it['byUser_ms'], it['byItem_ms']), ' '.join('%s=%.3f' % (k['kernel'].replace('als_','').replace('_kernel',''), k['avg_launch_ms']) for k in d['roofline']['kernels']))" || tail -n 5 gpurun_out/${TAG}_ablate_$v.err
done
