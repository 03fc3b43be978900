# power and shader clock of the GPU while bench.py runs a workload: bash profiles/smi_sample.sh [workload] [steps]
wl=${1:-c5shard}; st=${2:-60}
(timeout -k 10 300 python bench.py --workload $wl --steps $st --warmup 2 --no-cpu-baseline > gpurun_out/smi_bench_$wl.json 2>/dev/null &)
for i in $(seq 1 22); do
  sleep 2
  rocm-smi --showpower --showclocks 2>&1 | grep -E "sclk|Package Power" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ' '; echo
done
wait
sleep 5
cut -c1-160 gpurun_out/smi_bench_$wl.json
