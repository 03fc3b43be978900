# interleaved A/B of two builds of the library on one workload: bash profiles/ab_libs.sh workload steps rounds libA.so libB.so
wl=$1; st=$2; rounds=$3; shift 3
for r in $(seq 1 $rounds); do for lib in "$@"; do
  YCNR_ALS_LIB=$PWD/$lib timeout -k 10 400 python bench.py --workload $wl --steps $st --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); x=d['exchange']; print('$lib', round(d['ms_per_step'],3), 'byUser', x['byUser']['compute_ms'], 'byItem', x['byItem']['compute_ms'])"
done; done
