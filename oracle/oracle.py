"""ctypes loader for the CPU oracle (oracle/als_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package.  PARITY UNPINNED: see als_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# YCNR_ORACLE_LIB: another build of the same sources (`make -C oracle asan`: AddressSanitizer + UBSan, run with the
# sanitizer runtimes preloaded -- see the Makefile)
_SO = os.environ.get("YCNR_ORACLE_LIB") or os.path.join(_HERE, "_build", "libals_oracle.so")
_lib = None

i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile the oracle with gcc if the .so is missing or stale."""
    srcs = [os.path.join(_HERE, f) for f in ("als_oracle.c", "als_oracle_impl.h")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if os.environ.get("YCNR_ORACLE_LIB"):
        return _SO
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()  # compiles when the .so is missing or older than its sources
    L = C.CDLL(_SO)
    for pfx, fp in (("s", f32p), ("d", f64p)):
        f = getattr(L, f"oracle_{pfx}AlsCalcPortion")
        f.restype = C.c_int64
        f.argtypes = [C.c_double, C.c_int, i32p, i32p, fp, fp, fp, C.c_int]
        f = getattr(L, f"oracle_{pfx}AlsStepCsr")
        f.restype = C.c_int64
        f.argtypes = [C.c_double, C.c_int, C.c_int64, C.c_int64, i64p, i32p, fp, fp, fp, C.c_int]
        f = getattr(L, f"oracle_{pfx}RmsePortion")
        f.restype = None
        f.argtypes = [C.c_int, i32p, i32p, fp, fp, fp, C.c_double, f64p]
        f = getattr(L, f"oracle_{pfx}RmseCsr")
        f.restype = None
        f.argtypes = [C.c_int, C.c_int64, C.c_int64, i64p, i32p, fp, fp, fp, C.c_double, f64p]
        f = getattr(L, f"oracle_{pfx}PackPortion")
        f.restype = C.c_int
        f.argtypes = [C.c_int, i32p, i32p, fp, i32p, i32p, fp, C.c_int]
    L.oracle_split_to_portions.restype = C.c_int
    L.oracle_split_to_portions.argtypes = [i32p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int,
                                           C.c_int, i32p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.oracle_split_to_sets.restype = C.c_int
    L.oracle_split_to_sets.argtypes = [C.c_int64, i64p, np.ctypeslib.ndpointer(np.int8, flags="C_CONTIGUOUS"), i32p, C.c_uint32]
    for pfx, fp in (("s", f32p), ("d", f64p)):
        f = getattr(L, f"oracle_{pfx}RatingStats")
        f.restype = None
        f.argtypes = [C.c_int64, i64p, fp, C.c_void_p, i32p, f64p]
    for pfx, fp in (("s", f32p), ("d", f64p)):
        f = getattr(L, f"oracle_{pfx}Recommend")
        f.restype = C.c_int
        f.argtypes = [C.c_int, fp, C.c_int64, fp, C.c_int64, i32p, C.c_double, C.c_double, C.c_int, i32p, f64p]
    L.oracle_csr_from_triplets.restype = C.c_int
    L.oracle_csr_from_triplets.argtypes = [C.c_int, C.c_int64, i32p, i32p, C.c_void_p, C.c_int64, i64p, i32p, C.c_void_p]
    _lib = L
    return L


def _pfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "s"
    if dtype == np.float64:
        return "d"
    raise TypeError("invalid type!")  # cpp_utils/cpp_utils.js:12


def als_calc_portion(lam, k, alsRows, alsIndx, alsVals, fixed, solved, threads=1):
    """In-place portion solve (EmfWorker.js:176-251). Returns ratingsInPortion."""
    f = getattr(lib(), f"oracle_{_pfx(alsVals.dtype)}AlsCalcPortion")
    n = f(float(lam), int(k), alsRows, alsIndx, alsVals, fixed, solved, int(threads))
    if n < 0:
        raise FloatingPointError(f"oracle gesv failed, info={-n}")
    return n


def als_step_csr(lam, k, rowPtr, indx, vals, fixed, solved, row_begin=0, row_end=None, threads=1):
    """In-place half-step over CSR rows [row_begin, row_end)."""
    if row_end is None:
        row_end = len(rowPtr) - 1
    f = getattr(lib(), f"oracle_{_pfx(vals.dtype)}AlsStepCsr")
    n = f(float(lam), int(k), int(row_begin), int(row_end), rowPtr, indx, vals, fixed, solved, int(threads))
    if n < 0:
        raise FloatingPointError(f"oracle gesv failed, info={-n}")
    return n


def rmse_portion(k, rows, indx, vals, U, I, shift=0.0):
    out = np.zeros(3, np.float64)
    getattr(lib(), f"oracle_{_pfx(vals.dtype)}RmsePortion")(int(k), rows, indx, vals, U, I, float(shift), out)
    return out


def rmse_csr(k, rowPtr, indx, vals, U, I, shift=0.0, row_begin=0, row_end=None):
    if row_end is None:
        row_end = len(rowPtr) - 1
    out = np.zeros(3, np.float64)
    getattr(lib(), f"oracle_{_pfx(vals.dtype)}RmseCsr")(int(k), int(row_begin), int(row_end), rowPtr, indx, vals,
                                                       U, I, float(shift), out)
    return out


def pack_portion(r1, c1, rating, compat=True, max_rows=None):
    """EmfMaster.js:571-614 packer. r1/c1 1-based, sorted by row. Returns (rows, indx, vals)."""
    n = len(r1)
    rating = np.ascontiguousarray(rating)
    max_rows = n if max_rows is None else max_rows
    bufRows = np.zeros(1 + 2 * (max_rows + 1), np.int32)
    bufIndx = np.zeros(max(n, 1), np.int32)
    bufVals = np.zeros(max(n, 1), rating.dtype)
    f = getattr(lib(), f"oracle_{_pfx(rating.dtype)}PackPortion")
    f(n, np.ascontiguousarray(r1, np.int32), np.ascontiguousarray(c1, np.int32), rating, bufRows, bufIndx, bufVals,
      1 if compat else 0)
    return bufRows, bufIndx, bufVals


def split_to_portions(cnt_per_row, rows_cnt, ratings_in_portion, num_threads, pct=0):
    """EmfLord.js:510-612. Returns (portionsRowIdTo[1-based], maxRatingsInPortion, maxRowsInPortion)."""
    cnt = np.ascontiguousarray(cnt_per_row, np.int32)
    out = np.zeros(max(len(cnt), 1), np.int32)
    mr, mw = C.c_int(0), C.c_int(0)
    pos = cnt[cnt > 0]
    p = lib().oracle_split_to_portions(cnt, len(cnt), int(rows_cnt), int(pos.max()) if len(pos) else 0,
                                       int(pos.sum()), int(ratings_in_portion), int(num_threads), int(pct), out,
                                       C.byref(mr), C.byref(mw))
    return out[:p].copy(), mr.value, mw.value


def split_to_sets(rowPtr, types, pcts=(85, 10, 5), seed=1):
    """EmfLord.doSplitToSets (EmfLord.js:402-505) with the keyed shuffle of include/ycnr_als.h.
    types: int8 per rating, 0 = unassigned; returns the completed copy."""
    rp = np.ascontiguousarray(rowPtr, np.int64)
    t = np.array(types, np.int8, copy=True)
    rc = lib().oracle_split_to_sets(len(rp) - 1, rp, t, np.asarray(pcts, np.int32), int(seed) & 0xFFFFFFFF)
    if rc:
        raise MemoryError("oracle_split_to_sets")
    return t


def rating_stats(rowPtr, vals, types=None):
    """count and double sum per row of the ratings of type 1..3 (all when types is None)."""
    rp = np.ascontiguousarray(rowPtr, np.int64)
    v = np.ascontiguousarray(vals)
    rows = len(rp) - 1
    cnt = np.zeros(rows, np.int32)
    sm = np.zeros(rows, np.float64)
    t = None if types is None else np.ascontiguousarray(types, np.int8)
    getattr(lib(), f"oracle_{_pfx(v.dtype)}RatingStats")(rows, rp, v, None if t is None else t.ctypes.data, cnt, sm)
    return cnt, sm


def csr_from_triplets(row, col, vals, rows):
    """(rowPtr, indx, vals) ordered by (row, col, input position)."""
    r = np.ascontiguousarray(row, np.int32)
    c = np.ascontiguousarray(col, np.int32)
    v = np.ascontiguousarray(vals)
    rp = np.zeros(rows + 1, np.int64)
    ix = np.zeros(len(r), np.int32)
    ov = np.zeros_like(v)
    rc = lib().oracle_csr_from_triplets(v.dtype.itemsize, len(r), r, c, v.ctypes.data, rows, rp, ix, ov.ctypes.data)
    if rc:
        raise MemoryError("oracle_csr_from_triplets")
    return rp, ix, ov


def recommend(user_row, items, skip, shift, min_rating, limit):
    """YcnrController.recommendItemsForUser (lib/YcnrController.js:255-274) for one user.
    Returns (ids, predicts) of recItems, best first."""
    it = np.ascontiguousarray(items)
    u = np.ascontiguousarray(user_row, it.dtype)
    sk = np.ascontiguousarray(skip, np.int32)
    ids = np.zeros(limit + 1, np.int32)
    pr = np.zeros(limit + 1, np.float64)
    n = getattr(lib(), f"oracle_{_pfx(it.dtype)}Recommend")(it.shape[1], u, it.shape[0], it, len(sk), sk, float(shift), float(min_rating),
                                                          int(limit), ids, pr)
    return ids[:n].copy(), pr[:n].copy()
