# per-iteration time across the kernel families' boundaries, 200 K x 20 K, 20 M ratings, float32
for k in 64 100 112 116 128 132 192 256 260 512; do
  timeout -k 10 300 python bench.py --workload c3 --factors $k --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); it=d['roofline']['iteration']; print('k $k', round(d['ms_per_step'],2), 'user', round(it['byUser_ms'],2), 'item', round(it['byItem_ms'],2))"
done
