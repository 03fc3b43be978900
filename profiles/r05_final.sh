#!/bin/bash
# Round 5: the measurement pass over the final library sources, on the GPU box from the repo root:  bash profiles/r05_final.sh <tag>
# (bench lines of every workload, eight emulated ranks in both layouts, kernel statistics + PMC passes at MAL scale, prep bench)
TAG=$1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python bench.py > gpurun_out/${TAG}_default_bench.json 2> gpurun_out/${TAG}_default_bench.err; cut -c1-300 gpurun_out/${TAG}_default_bench.json
bash profiles/r05_run.sh $TAG bench mal 20 bench c3 20 bench ml1m 100 bench ml100k 100 bench c5shard 5
bash profiles/r05_run.sh $TAG emulate mal
timeout -k 10 900 python bench.py --workload mal --emulate-world 8 --item-sharding bands --steps 3 --warmup 1 > gpurun_out/${TAG}_mal_emulate8_bands.json 2> gpurun_out/${TAG}_mal_emulate8_bands.err
python3 -c "
import json
d=json.loads(open('gpurun_out/${TAG}_mal_emulate8_bands.json').read().strip().splitlines()[-1])
for c in ('cost_model_cut','after_feedback_recut'):
    if c in d:
        x=d[c]; print('bands', c, 'user', x['byUser']['compute_ms'], 'item', x['byItem']['compute_ms'], 'slowest', x['iteration_ms_slowest_rank'])" || tail -n 3 gpurun_out/${TAG}_mal_emulate8_bands.err
bash profiles/r05_run.sh $TAG prof mal
bash profiles/r05_run.sh $TAG prep
bash profiles/r05_run.sh $TAG bench c5 3
