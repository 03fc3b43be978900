#!/bin/bash
# Round 5: per-kernel times of one MAL-scale iteration in stream order for several library builds.
#   bash profiles/r05_perkernel.sh <tag> <workload> <variant> ...      (variant "main" = csrc/libycnr_als.so, else ablibs/lib_<variant>.so)
TAG=$1; WL=$2; shift 2
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
for v in "$@"; do
  lib=""; [ $v != main ] && lib="$R/ablibs/lib_$v.so"
  export YCNR_ALS_LIB=$lib YCNR_NO_OVERLAP=1
  (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_pk_$v -o p -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG}_pk_$v.log 2>&1)
  echo "== $v"
  python3 profiles/kernel_times.py gpurun_out/${TAG}_pk_$v | grep -E "dual|gram|reduce" | sort
done
