# small shapes: the row kernel second among a graph's kernels (YCNR_ROW_KERNEL_LAST: behind the dual classes, the former order)
L=you-can-not-recommend_amd/csrc/devtest/ablibs/libycnr_next.so
for wl in ml1m ml100k; do
  for e in "" "YCNR_ROW_KERNEL_LAST=1" "" "YCNR_ROW_KERNEL_LAST=1"; do
      env YCNR_ALS_LIB=$PWD/$L $e timeout -k 10 200 python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); it=d['roofline']['iteration']; print('$wl [$e]', round(d['ms_per_step'],4), 'kernels', it['kernel_ms_per_step'], 'user', it['byUser_ms'], 'item', it['byItem_ms'])"
  done
done
