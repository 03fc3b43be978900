# float32 factorsCount % 4 != 0: kernels on matrices padded to a multiple of 4 columns (YCNR_NO_KPAD_SMALL=1: the plain float32-MFMA kernels)
for k in 99 50 30; do
  for e in "" "YCNR_NO_KPAD_SMALL=1"; do
    env $e timeout -k 10 300 python bench.py --workload mal --factors $k --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); it=d['roofline']['iteration']; print('k $k [$e]', round(d['ms_per_step'],2), 'user', round(it['byUser_ms'],2), 'item', round(it['byItem_ms'],2))"
  done
done
