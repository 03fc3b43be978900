#!/bin/bash
# Collects the rocprofv3 passes profiles/README.md lists, on the GPU box, from the repo root:
#   bash profiles/collect.sh <tag> [workload]
# Writes gpurun_out/prof_<tag>/{trace,fetch,write,sq}/ (scratch); aggregate_pmc.py and a copy of
# trace/*kernel_stats.csv turn them into the committed summaries.  The kernel trace and every
# counter set are separate runs (never --pmc together with a trace domain other than
# --kernel-trace), and the profiled program is python3 itself.
set -u
TAG=$1
WL=${2:-mal}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --workload $WL --steps 1 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o p -- $B > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- $B > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- $B > $OUT/write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq -o p -- $B > $OUT/sq.log 2>&1
for p in trace fetch write sq; do tail -1 $OUT/$p.log | cut -c1-160; done
python3 $R/profiles/kernel_times.py $OUT/trace
