"""Prints the in-kernel stamps of a -DYCNR_WG_STAMPS build for the 32 x 32 Gramian kernel (als_gram32_kernels.hip.h):
shader-clock cycles per 32-rating step of workgroup 0's first rows, and the clock the kernel held (shader cycles per
tick of the 100 MHz real-time counter).
  YCNR_ALS_LIB=<stamps build> YCNR_DUMP_STAMPS=/tmp/stamps.bin python tests/tools/g32stamps.py [ratings per row]"""
import os
import sys
import numpy as np
sys.path.insert(0, "you-can-not-recommend_amd/python")
import torch  # noqa: F401
import ycnr_als
from ycnr_als import _lib

k = 256
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
items, rows = 200000, 256 * 24
rng = np.random.default_rng(1)
V = (rng.standard_normal((items, k)) / np.sqrt(k)).astype(np.float32)
rowPtr = np.arange(rows + 1, dtype=np.int64) * n
indx = rng.integers(0, items, rows * n).astype(np.int32)
vals = rng.integers(1, 11, rows * n).astype(np.float32)
dev = ycnr_als.AlsDevice(k, rows, items, flags=_lib.FLAG_NO_DUAL)
dev.set_ratings("byUser", rowPtr, indx, vals)
dev.set_factors("byItem", V)
for _ in range(3):
    i = dev.step("byUser")
print("half-step ms", i.gramSolveMs, " rows", rows, " steps per row", (n + 31) // 32)
st = np.fromfile(os.environ["YCNR_DUMP_STAMPS"], np.uint64)[8:]
for w in (0, 1):
    s = st[256 * w:256 * w + 256].astype(np.int64)
    d = np.diff(s[:240])
    per_row = (n + 31) // 32
    inrow = np.array([x for j, x in enumerate(d) if (j + 1) % (per_row + 1) != 0])  # drop the gaps between rows
    gaps = d[per_row::per_row + 1]
    ghz = (s[239] - s[0]) / ((s[241] - s[240]) * 10.0)
    print("wave %d: step cycles median %d  min %d  max %d;  between rows median %d;  clock %.2f GHz -> %.2f us per step"
          % (w, np.median(inrow), inrow.min(), inrow.max(), np.median(gaps), ghz, np.median(inrow) / ghz / 1e3))
