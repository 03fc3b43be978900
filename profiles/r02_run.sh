#!/bin/bash
# Round-2 GPU session script (run from the repo root on the GPU box):
#   bash profiles/r02_run.sh <tag> [tests|bench|prof ...]
# tests: the whole -m gpu suite; bench: bench lines of the four single-GPU configs; prof: collect.sh on c3 / ml1m / c5shard
TAG=$1; shift
mkdir -p gpurun_out
for what in "$@"; do
  case $what in
    tests) timeout 2400 python -m pytest tests -m gpu -x -q -s > gpurun_out/${TAG}_gputests.log 2>&1; tail -3 gpurun_out/${TAG}_gputests.log ;;
    cfgtests) timeout 2400 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -s > gpurun_out/${TAG}_cfgtests.log 2>&1; tail -3 gpurun_out/${TAG}_cfgtests.log ;;
    bench)
      for wl in ml1m c3 c5shard mal; do
        st=10; [ $wl = c5shard ] && st=3
        timeout 900 python bench.py --workload $wl --steps $st --warmup 2 > gpurun_out/${TAG}_${wl}_bench.json 2> gpurun_out/${TAG}_${wl}_bench.err
        cut -c1-400 gpurun_out/${TAG}_${wl}_bench.json
      done ;;
    prof)
      for wl in c3 ml1m c5shard; do bash profiles/collect.sh ${TAG}_$wl $wl; done ;;
    profmal) bash profiles/collect.sh ${TAG}_mal mal ;;
  esac
done
