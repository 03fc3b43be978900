"""Timing probe for the workgroup-per-row kernels (k > 128): per-step and per-solve cost from rows of
equal length.  python tests/tools/wgtime.py [k]   (on the GPU box, from the repo root)"""
import sys
import numpy as np
sys.path.insert(0, "you-can-not-recommend_amd/python")
import torch  # noqa: F401
import ycnr_als
from ycnr_als import _lib

k = int(sys.argv[1]) if len(sys.argv) > 1 else 256
items = 20000
rng = np.random.default_rng(1)
V = (rng.standard_normal((items, k)) / np.sqrt(k)).astype(np.float32)


def run(rows, n, chunk, reps=3):
    rowPtr = np.arange(rows + 1, dtype=np.int64) * n
    indx = rng.integers(0, items, rows * n).astype(np.int32)
    vals = rng.integers(1, 11, rows * n).astype(np.float32)
    dev = ycnr_als.AlsDevice(k, rows, items, chunkRatings=chunk, flags=_lib.FLAG_NO_DUAL)
    dev.set_ratings("byUser", rowPtr, indx, vals)
    dev.set_factors("byItem", V)
    best = None
    for _ in range(reps):
        i = dev.step("byUser")
        t = (i.gramSlabMs, i.gramSolveMs, i.reduceSolveMs)
        best = t if best is None else tuple(min(a, b) for a, b in zip(best, t))
    dev.destroy()
    return best


cus = 256
for n in (192, 1792, 3392):
    rows = cus * 8
    slab, fused, red = run(rows, n, 0)
    print(f"whole rows  n={n:5d} ({(n + 31) // 32:3d} steps): fused {fused:8.3f} ms = {fused * 1e3 / 8:8.1f} us per row")
rows = cus * 8
slab, fused, red = run(rows, 1792, 896)
print(f"2 chunks of 896: slab {slab:8.3f} ms = {slab * 1e3 / 16:8.1f} us per chunk (28 steps); reduce+solve {red:8.3f} ms = {red * 1e3 / 8:8.1f} us per row")
