#!/bin/bash
# Round 5: interleaved A/B runs of bench.py on one box under different environments.
#   bash profiles/r05_ab.sh <tag> <workload> <steps> name1:VAR=1,VAR2=x name2: ...
TAG=$1; WL=$2; ST=$3; shift 3
mkdir -p gpurun_out
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  envs=$(echo "$envs" | tr ',' ' ')
  env $envs timeout -k 10 400 python bench.py --workload $WL --steps $ST --warmup 2 --no-cpu-baseline --dump-step-info $BENCH_ARGS > gpurun_out/${TAG}_ab_$name.json 2> gpurun_out/${TAG}_ab_$name.err
  python3 -c "
import json
d=json.loads(open('gpurun_out/${TAG}_ab_$name.json').read().strip().splitlines()[-1]); it=d['roofline']['iteration']
print('%-16s ms %.3f user %.3f item %.3f' % ('$name', d['ms_per_step'], it['byUser_ms'], it['byItem_ms']), ' '.join('%s=%.3f' % (k['kernel'].replace('als_','').replace('_kernel',''), k['avg_launch_ms']) for k in d['roofline']['kernels']))" || tail -n 5 gpurun_out/${TAG}_ab_$name.err
done
