# small uploads: what one launch per half-step would buy (no chunks, no dual classes): bash profiles/ab_small_policy.sh [workload]
wl=${1:-ml100k}
for r in 1 2; do for cfg in "0 -" "4096 -" "4096 2" "0 2"; do set -- $cfg
  if [ "$2" = "-" ]; then unset YCNR_EXTRA_FLAGS; else export YCNR_EXTRA_FLAGS=$2; fi
  timeout -k 10 300 python bench.py --workload $wl --steps 200 --warmup 20 --chunk $1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); x=d['exchange']; print('$wl chunk=$1 flags=$2', round(d['ms_per_step'],4), 'byUser', x['byUser']['compute_ms'], x['byUser']['wall_ms'], 'byItem', x['byItem']['compute_ms'], x['byItem']['wall_ms'], 'rmse', round(d['rmse_in_sample_after_iters'],6))"
done; done
