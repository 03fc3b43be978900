"""How many host CPUs does this process really get?  Prints cpu_count, the affinity mask size,
the cgroup quota, and the oracle's row-solve rate at several OpenMP thread counts (k = 100,
rows of 64 ratings), so that bench.py's cpu_baseline can say which thread count it used."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import oracle as orc

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except OSError:
        pass
orc.build()
k, rows, per = 100, 8192, 64
rng = np.random.default_rng(1)
fixed = rng.standard_normal((4096, k)).astype(np.float32) * 0.1
rp = np.arange(rows + 1, dtype=np.int64) * per
indx = rng.integers(0, 4096, rows * per).astype(np.int32)
vals = rng.integers(1, 11, rows * per).astype(np.float32)
for th in (1, 4, 8, 16, 32, 64, 128, 256):
    if th > (os.cpu_count() or 1):
        break
    out = np.zeros((rows, k), np.float32)
    t = time.perf_counter()
    n = orc.als_step_csr(0.05, k, rp, indx, vals, fixed, out, 0, rows, threads=th)
    dt = time.perf_counter() - t
    print("threads %3d  %.2f s  %.2f M ratings/s" % (th, dt, n / dt / 1e6), flush=True)
