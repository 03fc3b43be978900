# small uploads: ms per iteration over --chunk values (0 = automatic): bash profiles/ab_chunk_small.sh workload v1 v2 ...
wl=$1; shift
for r in 1 2; do for c in "$@"; do
  timeout -k 10 300 python bench.py --workload $wl --steps 200 --warmup 20 --chunk $c --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); x=d['exchange']; print('$wl chunk=$c', round(d['ms_per_step'],4), 'byUser', x['byUser']['compute_ms'], x['byUser']['wall_ms'], 'byItem', x['byItem']['compute_ms'], x['byItem']['wall_ms'])"
done; done
