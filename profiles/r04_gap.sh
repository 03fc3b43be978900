# both half-steps of an iteration enqueued before the host waits (EmfLord.alsTrainIter -> AlsDevice.iteration) against two awaited steps
for wl in ml100k ml1m c3 mal; do
  st=200; [ $wl = mal ] && st=20
  for e in 1 0 1 0; do
      YCNR_PIPELINE_ITER=$e timeout -k 10 200 python bench.py --workload $wl --steps $st --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); it=d['roofline']['iteration']; print('$wl in flight $e', round(d['ms_per_step'],4), 'kernels', it['kernel_ms_per_step'], 'user', it['byUser_ms'], 'item', it['byItem_ms'])"
  done
done
