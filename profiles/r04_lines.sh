# Is the item half-step of the MAL shape bound by cache lines per gathered row?  k = 96 (384-byte rows: 3 lines of 128 bytes),
# 100 (400 bytes: always 4), 128 (512 bytes: 4, and 29 % more tiles than k = 100).  bash profiles/r04_lines.sh
for k in 96 100 112 128; do
  timeout -k 10 300 python bench.py --workload mal --factors $k --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); it=d['roofline']['iteration']; print('k $k', round(d['ms_per_step'],2), 'user', round(it['byUser_ms'],2), 'item', round(it['byItem_ms'],2), ' '.join('%s=%.3f' % (q['kernel'].replace('als_','').replace('_kernel',''), q['avg_launch_ms']) for q in d['roofline']['kernels']))"
done
