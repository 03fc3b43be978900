# A/B runs of the 240 < k <= 256 Gramian kernels on one GPU's eighth of C5: bash profiles/ab_g32.sh [lib.so ...]
# without arguments: the shipped library without and with YCNR_G32=1, interleaved
run() { timeout -k 10 400 python bench.py --workload c5shard --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2), [(k['kernel'][4:], round(k['avg_launch_ms'],2)) for k in d['roofline']['kernels']])"; }
if [ $# -gt 0 ]; then
  for lib in "$@"; do YCNR_ALS_LIB=$PWD/$lib run $lib; done
else
  for e in 0 1 0 1; do if [ $e = 1 ]; then export YCNR_G32=1; else unset YCNR_G32; fi; run g32=$e; done
fi
