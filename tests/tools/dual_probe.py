"""Rows of every length n of one dual class (16 (m - 1) + 1 .. 16 m ratings, k = 256 by default) through the dual-form
kernel of the library named in YCNR_ALS_LIB, against float64: failures per n, twice (is the set of wrong rows the same?).
  YCNR_ALS_LIB=<lib.so> python tests/tools/dual_probe.py [blocks=7] [k=256] [rows_per_n=300]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "you-can-not-recommend_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import ycnr_als  # noqa: E402
from ycnr_als.data import Csr  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 7
k = int(sys.argv[2]) if len(sys.argv) > 2 else 256
per = int(sys.argv[3]) if len(sys.argv) > 3 else 300
items = 3000
rng = np.random.default_rng(5)
lens = np.repeat(np.arange(16 * (m - 1) + 1, 16 * m + 1), per).astype(np.int64)
rng.shuffle(lens)
users = len(lens)
rowPtr = np.zeros(users + 1, np.int64)
np.cumsum(lens, out=rowPtr[1:])
indx = np.concatenate([np.sort(rng.choice(items, n, replace=False)) for n in lens]).astype(np.int32)
vals = (rng.standard_normal(rowPtr[-1]) * 2.0 + 5.0).astype(np.float32)
V = (rng.standard_normal((items, k)) / np.sqrt(k)).astype(np.float32)
want = np.zeros((users, k))
V64 = V.astype(np.float64)
for u in range(users):
    Y = V64[indx[rowPtr[u]:rowPtr[u + 1]]]
    n = len(Y)
    want[u] = np.linalg.solve(Y.T @ Y + 0.05 * n * np.eye(k), Y.T @ vals[rowPtr[u]:rowPtr[u + 1]].astype(np.float64))
runs = []
for rep in range(2):
    dev = ycnr_als.AlsDevice(k, users, items)
    dev.set_ratings("byUser", rowPtr, indx, vals)
    dev.set_factors("byUser", np.zeros((users, k), np.float32))
    dev.set_factors("byItem", V)
    info = dev.step("byUser")
    got = dev.get_factors("byUser").astype(np.float64)
    dev.destroy()
    err = np.linalg.norm(got - want, axis=1) / np.linalg.norm(want, axis=1)
    runs.append(err > 1e-4)
    print(f"run {rep}: dualRows {info.dualRows} of {users}, wrong rows {int(runs[-1].sum())}, worst {err.max():.3g}")
    print("  wrong per n:", {int(n): int(runs[-1][lens == n].sum()) for n in np.unique(lens) if runs[-1][lens == n].any()})
print("same set of wrong rows in both runs:", bool(np.array_equal(runs[0], runs[1])), " lib:", os.environ.get("YCNR_ALS_LIB", "shipped"))
