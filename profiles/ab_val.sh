# interleaved runs of one environment variable over several values ("-" = unset): bash profiles/ab_val.sh VAR workload steps rounds v1 v2 ...
var=$1; wl=$2; st=$3; rounds=$4; shift 4
for r in $(seq 1 $rounds); do for v in "$@"; do
  if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
  timeout -k 10 400 python bench.py --workload $wl --steps $st --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); x=d['exchange']; print('$var=$v', round(d['ms_per_step'],3), 'byUser', x['byUser']['compute_ms'], 'byItem', x['byItem']['compute_ms'], 'rmse', d['rmse_in_sample_after_iters'])"
done; done
