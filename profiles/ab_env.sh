# interleaved A/B of one environment toggle: bash profiles/ab_env.sh VAR workload [steps] [rounds]
var=$1; wl=${2:-mal}; st=${3:-10}; rounds=${4:-3}
for r in $(seq 1 $rounds); do for e in 0 1; do
  if [ $e = 1 ]; then export $var=1; else unset $var; fi
  timeout -k 10 400 python bench.py --workload $wl --steps $st --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); x=d['exchange']; print('$var=$e', round(d['ms_per_step'],3), 'byUser', x['byUser']['compute_ms'], 'byItem', x['byItem']['compute_ms'], 'rmse', d['rmse_in_sample_after_iters'])"
done; done
