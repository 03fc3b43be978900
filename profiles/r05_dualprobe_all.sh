#!/bin/bash
# Every dual class ALONE on the chip (tests/tools/dual_probe.py: 300 rows of every length of the class, all against float64, twice):
# the lockstep situation in which the two-wave 7-block class of rounds 3-4 failed.   bash profiles/r05_dualprobe_all.sh <tag>
TAG=$1
mkdir -p gpurun_out
for k in 256 100; do
  top=12; [ $k = 100 ] && top=5
  for m in $(seq 1 $top); do
    echo "== k $k, $m blocks"
    timeout -k 10 300 python tests/tools/dual_probe.py $m $k 300 2>&1 | grep -E "^run|same" | tee -a gpurun_out/${TAG}_dualprobe_all.log
  done
done
