#!/bin/bash
# Round 5: what the dual classes of the k = 256 path cost per class, in stream order (one GPU's eighth of C5).
#   bash profiles/r05_c5dual.sh <tag> <variant> [<variant> ...]
# variant "main" = csrc/libycnr_als.so, anything else = ablibs/lib_<variant>.so, built with
#   make -C you-can-not-recommend_amd/csrc OUT=../../ablibs/lib_<variant>.so EXTRA=-D...
# (YCNR_DUAL_XDEPTH=n, YCNR_DUAL_ABLATE_G / _SOLVE / _X: the ablated builds solve wrong rows by construction, only times count).
TAG=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
for v in "$@"; do
  lib=""; [ $v != main ] && lib="$R/ablibs/lib_$v.so"
  export YCNR_ALS_LIB=$lib YCNR_NO_OVERLAP=1 YCNR_IGNORE_NUMERIC=1
  (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_c5dual_$v -o p -- python3 $R/bench.py --workload c5shard --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG}_c5dual_$v.log 2>&1) || echo "($v: bench.py refused its line -- expected for an ablated build)"
  echo "== $v"
  python3 profiles/kernel_times.py gpurun_out/${TAG}_c5dual_$v | grep -E "dual|wg_gram|slab_solve2" | sort
  grep -m1 '^{"metric' gpurun_out/${TAG}_c5dual_$v.log | python3 -c "
import json,sys
t=sys.stdin.read()
if t.strip():
    d=json.loads(t); it=d['roofline']['iteration']; print('   ms %.2f user %.2f item %.2f' % (d['ms_per_step'], it['byUser_ms'], it['byItem_ms']))"
done
