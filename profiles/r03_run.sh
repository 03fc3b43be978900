#!/bin/bash
# Round-3 GPU session script (run from the repo root on the GPU box):
#   bash profiles/r03_run.sh <tag> [tests|cfgtests|bench|benchc5|emulate|emulatec5|ipc2|prof <wl>|profc5 ...]
TAG=$1; shift
mkdir -p gpurun_out
while [ $# -gt 0 ]; do
  what=$1; shift
  case $what in
    tests) timeout -k 10 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --durations=10 > gpurun_out/${TAG}_gputests.log 2>&1; tail -8 gpurun_out/${TAG}_gputests.log ;;
    cfgtests) timeout -k 10 1500 python -m pytest tests/test_gpu_configs.py -m gpu -q -p no:cacheprovider > gpurun_out/${TAG}_cfgtests.log 2>&1; tail -5 gpurun_out/${TAG}_cfgtests.log ;;
    failed) timeout -k 10 900 python -m pytest "tests/test_gpu_configs.py::test_full_size_iteration[c5]" tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "c5 or every_row_length_class or ml100k_shape" > gpurun_out/${TAG}_failedtests.log 2>&1; tail -5 gpurun_out/${TAG}_failedtests.log ;;
    bench)
      for wl in ml100k ml1m c3 c5shard mal; do
        st=10; [ $wl = c5shard ] && st=3
        timeout -k 10 900 python bench.py --workload $wl --steps $st --warmup 2 > gpurun_out/${TAG}_${wl}_bench.json 2> gpurun_out/${TAG}_${wl}_bench.err
        cut -c1-300 gpurun_out/${TAG}_${wl}_bench.json
      done ;;
    benchc5)
      timeout -k 10 1100 python bench.py --workload c5 --steps 3 --warmup 1 > gpurun_out/${TAG}_c5_bench.json 2> gpurun_out/${TAG}_c5_bench.err
      cut -c1-400 gpurun_out/${TAG}_c5_bench.json; tail -3 gpurun_out/${TAG}_c5_bench.err ;;
    emulate)
      timeout -k 10 900 python bench.py --workload mal --emulate-world 8 --steps 3 --warmup 1 > gpurun_out/${TAG}_mal_emulate8.json 2> gpurun_out/${TAG}_mal_emulate8.err
      cut -c1-600 gpurun_out/${TAG}_mal_emulate8.json; tail -3 gpurun_out/${TAG}_mal_emulate8.err ;;
    emulatec5)
      timeout -k 10 1100 python bench.py --workload c5 --emulate-world 8 --steps 2 --warmup 1 > gpurun_out/${TAG}_c5_emulate8.json 2> gpurun_out/${TAG}_c5_emulate8.err
      cut -c1-600 gpurun_out/${TAG}_c5_emulate8.json; tail -3 gpurun_out/${TAG}_c5_emulate8.err ;;
    ipc2)
      # two ranks sharing the one GPU, MAL scale, the device-to-device transport: the pipelined exchange with real sizes
      for tr in ipc shm; do
        timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29577 bench.py \
          --gpus 2 --backend gloo --same-device --transport $tr --workload mal --steps 5 --warmup 2 --no-cpu-baseline \
          > gpurun_out/${TAG}_mal_2ranks_${tr}.json 2> gpurun_out/${TAG}_mal_2ranks_${tr}.err
        python3 -c "import json,sys; d=json.loads(open('gpurun_out/${TAG}_mal_2ranks_${tr}.json').read().strip().splitlines()[-1]); print('$tr', d['ms_per_step'], json.dumps(d['exchange'])[:700])" || tail -5 gpurun_out/${TAG}_mal_2ranks_${tr}.err
      done ;;
    prof) wl=$1; shift; bash profiles/collect.sh ${TAG}_$wl $wl ;;
  esac
done
