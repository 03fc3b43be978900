"""Prints the in-kernel stamps of a -DYCNR_WG_STAMPS build (als_wg_kernels.hip.h): shader-clock cycles
between the phase boundaries of one row's solve, for wave 0 (the factoring wave) and wave 1.
  YCNR_ALS_LIB=<stamps build> YCNR_DUMP_STAMPS=/tmp/stamps.bin python tests/tools/wgstamps.py [k]"""
import os
import sys
import numpy as np
sys.path.insert(0, "you-can-not-recommend_amd/python")
import torch  # noqa: F401
import ycnr_als
from ycnr_als import _lib

k = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nb = (k + 15) // 16
items, rows, n = 20000, 512, 320
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
print("fused ms", i.gramSolveMs, "per row us", i.gramSolveMs * 1e3 / (rows / 256))
st = np.fromfile(os.environ["YCNR_DUMP_STAMPS"], np.uint64)[8:]
w0, w1 = st[:256].astype(np.int64), st[256:512].astype(np.int64)
t0 = w0[0]
print("slot meanings: 0 start, 1 after diag barrier, 2 factor(0) done, 3 barrier; per J: 4+4J panel done, 5+4J barrier A, 6+4J trailing done, 7+4J barrier B; 80 loop end; 81 back-subst end")
print("wave0: start->diag barrier %d, factor0 %d, barrier %d" % (w0[1] - w0[0], w0[2] - w0[1], w0[3] - w0[2]))
for J in range(nb):
    b = 4 + 4 * J
    prev = w0[b - 1]
    line = "J=%2d  w0: panel %5d  wait %5d" % (J, w0[b] - prev, w0[b + 1] - w0[b])
    line1 = "   w1: panel %5d  wait %5d" % (w1[b] - w1[b - 1], w1[b + 1] - w1[b])
    if J + 1 < nb:
        line += "  update+factor %5d  wait %5d" % (w0[b + 2] - w0[b + 1], w0[b + 3] - w0[b + 2])
        line1 += "  trailing      %5d  wait %5d" % (w1[b + 2] - w1[b + 1], w1[b + 3] - w1[b + 2])
    print(line + line1)
print("loop total", w0[80] - w0[3], " back substitution", w0[81] - w0[80], " whole solve", w0[81] - t0)
