# k = 256: where should a row stop taking the one-wave dual form (n x n) and take the workgroup Gramian + four-wave solve?
for m in 12 10 8 6; do
  YCNR_DUAL_MAX_BLOCKS=$m timeout -k 10 300 python bench.py --workload c5shard --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); it=d['roofline']['iteration']; print('dual up to $m blocks:', round(d['ms_per_step'],2), 'user', round(it['byUser_ms'],2), 'item', round(it['byItem_ms'],2))"
done
