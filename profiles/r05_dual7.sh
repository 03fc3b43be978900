#!/bin/bash
# Round 5: the 7-block dual class at two waves per SIMD (devtest/dual7/README.md) with the probe builds of this round.
#   bash profiles/r05_dual7.sh <tag> <variant> ...      (ablibs/lib_d7w2_<variant>.so: EXTRA="-DYCNR_DUAL7_WAVES=2 [-DYCNR_PIVOT_PROBE=n]")
TAG=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  echo "== $v"
  YCNR_ALS_LIB=$PWD/ablibs/lib_d7w2_$v.so timeout -k 10 300 python tests/tools/dual_probe.py 7 256 300 2>&1 | tee gpurun_out/${TAG}_dual7_$v.log | tail -n 6
done
