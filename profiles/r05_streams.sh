#!/bin/bash
# Round 5: how the one-piece half-step's kernels are spread over streams, eight ranks emulated (bands) and one GPU.
#   bash profiles/r05_streams.sh <tag> <lib or ""> name:VAR=1,VAR2=x ...
TAG=$1; LIB=$2; shift 2
mkdir -p gpurun_out
for spec in "$@"; do
  name=${spec%%:*}; envs=$(echo "${spec#*:}" | tr ',' ' ')
  env $envs YCNR_ALS_LIB=$LIB timeout -k 10 600 python bench.py --workload mal --emulate-world 8 --item-sharding bands --steps 3 --warmup 1 > gpurun_out/${TAG}_st_${name}_em.json 2> gpurun_out/${TAG}_st_${name}_em.err
  python3 -c "
import json
d=json.loads(open('gpurun_out/${TAG}_st_${name}_em.json').read().strip().splitlines()[-1]); b=d['bands_cut']; u=b['byUser']; i=b['byItem']
print('%-12s emulate8 bands: user %.3f..%.3f item %.3f..%.3f slowest %.3f (kernels %.3f)' % ('$name', min(u['compute_ms']), max(u['compute_ms']), min(i['compute_ms']), max(i['compute_ms']), b['iteration_ms_slowest_rank'], max(a+c for a,c in zip(u['compute_ms'],i['compute_ms']))))" || tail -n 3 gpurun_out/${TAG}_st_${name}_em.err
  env $envs YCNR_ALS_LIB=$LIB timeout -k 10 600 python bench.py --workload mal --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_st_${name}_mal.json 2> gpurun_out/${TAG}_st_${name}_mal.err
  python3 -c "
import json
d=json.loads(open('gpurun_out/${TAG}_st_${name}_mal.json').read().strip().splitlines()[-1]); it=d['roofline']['iteration']
print('%-12s one GPU: ms %.3f user %.3f item %.3f' % ('$name', d['ms_per_step'], it['byUser_ms'], it['byItem_ms']))" || tail -n 3 gpurun_out/${TAG}_st_${name}_mal.err
done
