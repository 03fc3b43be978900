#!/bin/bash
# Round-5 GPU session script (run from the repo root on the GPU box):
#   bash profiles/r05_run.sh <tag> [tests|bench <wl> <steps>|emulate|trace_emulate|...]
TAG=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
while [ $# -gt 0 ]; do
  what=$1; shift
  case $what in
    tests) timeout -k 10 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --durations=10 > gpurun_out/${TAG}_gputests.log 2>&1; tail -8 gpurun_out/${TAG}_gputests.log ;;
    pytest) sel=$1; shift; timeout -k 10 1100 python -m pytest $sel -m gpu -q --timeout 900 -p no:cacheprovider -x > gpurun_out/${TAG}_pytest.log 2>&1; tail -15 gpurun_out/${TAG}_pytest.log ;;
    bench) wl=$1; st=$2; shift 2
      timeout -k 10 900 python bench.py --workload $wl --steps $st --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_${wl}_bench.json 2> gpurun_out/${TAG}_${wl}_bench.err
      python3 -c "
import json
d=json.loads(open('gpurun_out/${TAG}_${wl}_bench.json').read().strip().splitlines()[-1]); it=d['roofline']['iteration']
print('$wl ms %.3f user %.3f item %.3f' % (d['ms_per_step'], it['byUser_ms'], it['byItem_ms']), ' '.join('%s=%.3f/%.2f' % (k['kernel'].replace('als_','').replace('_kernel',''), k['avg_launch_ms'], k['mfma_frac']) for k in d['roofline']['kernels']))" || tail -5 gpurun_out/${TAG}_${wl}_bench.err ;;
    fullbench) wl=$1; st=$2; shift 2
      timeout -k 10 1100 python bench.py --workload $wl --steps $st --warmup 2 > gpurun_out/${TAG}_${wl}_bench.json 2> gpurun_out/${TAG}_${wl}_bench.err
      cut -c1-400 gpurun_out/${TAG}_${wl}_bench.json ;;
    emulate) wl=${1:-mal}; shift
      timeout -k 10 1100 python bench.py --workload $wl --emulate-world 8 --steps 3 --warmup 1 > gpurun_out/${TAG}_${wl}_emulate8.json 2> gpurun_out/${TAG}_${wl}_emulate8.err
      python3 -c "
import json
d=json.loads(open('gpurun_out/${TAG}_${wl}_emulate8.json').read().strip().splitlines()[-1])
for c in ('cost_model_cut','after_feedback_recut'):
    x=d[c]; print(c, 'user', x['byUser']['compute_ms'], 'item', x['byItem']['compute_ms'], 'slowest', x['iteration_ms_slowest_rank'])" || tail -3 gpurun_out/${TAG}_${wl}_emulate8.err ;;
    prof) wl=$1; shift; bash profiles/collect.sh ${TAG}_$wl $wl ;;
    stats) name=$1; shift; args=$1; shift
      # per-kernel averages of one bench command (args quoted as one word)
      R=$PWD; (cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats_$name -o p -- python3 $R/bench.py $args --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG}_stats_$name.log 2>&1)
      python3 profiles/kernel_times.py gpurun_out/${TAG}_stats_$name ;;
    prep) timeout -k 10 900 python prep_bench.py > gpurun_out/${TAG}_prep_bench.json 2> gpurun_out/${TAG}_prep_bench.err; cut -c1-700 gpurun_out/${TAG}_prep_bench.json ;;
    trace) name=$1; shift; args=$1; shift
      # kernel trace with timestamps of one bench command (args quoted as one word)
      rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_trace_$name -o t -- python3 bench.py $args --no-cpu-baseline > gpurun_out/${TAG}_trace_$name.json 2> gpurun_out/${TAG}_trace_$name.err
      ls gpurun_out/${TAG}_trace_$name/*/ 2>/dev/null | head ;;
  esac
done
