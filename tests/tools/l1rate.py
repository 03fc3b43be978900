"""Portions per second of the level-1 op ycnr_sAlsCalcPortion through plain ctypes (works with any build of
the library, e.g. round 1's): 10 000-rating portions of consecutive users, k = 100, MAL-like row lengths.
  YCNR_ALS_LIB=<lib.so> python tests/tools/l1rate.py"""
import ctypes as C
import os
import sys
import time
import numpy as np
import torch  # noqa: F401  (its HIP runtime first)

L = C.CDLL(os.environ.get("YCNR_ALS_LIB", "you-can-not-recommend_amd/csrc/libycnr_als.so"))
L.ycnr_sAlsCalcPortion.restype = C.c_int64
L.ycnr_sAlsCalcPortion.argtypes = [C.c_double, C.c_int] + [C.c_void_p] * 4 + [C.c_int64, C.c_void_p, C.c_int64]
k, items, users = 100, 12700, 12000
rng = np.random.default_rng(1)
lens = np.clip(rng.lognormal(np.log(45), 1.2, users).astype(np.int64), 1, 4000)
rp = np.concatenate([[0], np.cumsum(lens)])
indx = rng.integers(0, items, rp[-1]).astype(np.int32)
vals = rng.integers(1, 11, rp[-1]).astype(np.float32)
V = (rng.standard_normal((items, k)) / 10).astype(np.float32)
U = np.zeros((users, k), np.float32)
portions, lo = [], 0
while lo < users and len(portions) < 48:
    hi = max(int(np.searchsorted(rp, rp[lo] + 10_000, side="right")) - 1, lo + 1)
    hi = min(hi, users)
    rows = np.empty(1 + 2 * (hi - lo), np.int32)
    rows[0] = hi - lo
    rows[1::2] = np.arange(lo, hi)
    rows[2::2] = lens[lo:hi]
    portions.append((rows, np.ascontiguousarray(indx[rp[lo]:rp[hi]]), np.ascontiguousarray(vals[rp[lo]:rp[hi]])))
    lo = hi


def call(p):
    n = L.ycnr_sAlsCalcPortion(0.05, k, p[0].ctypes.data, p[1].ctypes.data, p[2].ctypes.data, V.ctypes.data, items, U.ctypes.data, users)
    assert n > 0, n
    return n


def run(label):
    call(portions[0])
    t = time.perf_counter()
    n = sum(call(p) for p in portions)
    dt = time.perf_counter() - t
    print(f"{label}: {len(portions)} portions, {n} ratings: {dt / len(portions) * 1e3:.3f} ms per portion, {len(portions) / dt:.1f} portions/s, {n / dt / 1e6:.2f} M ratings/s")


run("fixed rows compacted and uploaded per portion")
if hasattr(L, "ycnr_sAlsPinFixedFactors"):
    L.ycnr_sAlsPinFixedFactors.restype = C.c_int
    L.ycnr_sAlsPinFixedFactors.argtypes = [C.c_void_p, C.c_int64, C.c_int]
    assert L.ycnr_sAlsPinFixedFactors(V.ctypes.data, items, k) == 0
    run("fixed matrix pinned on the device")
