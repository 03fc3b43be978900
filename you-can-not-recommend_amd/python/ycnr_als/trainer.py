"""Thin object wrapper over the level-1 and level-2 C ABI (include/ycnr_als.h).

Accepts numpy arrays (host memory) or torch CUDA tensors (device memory) for ratings and
factors.  All arithmetic happens in libycnr_als.so on the GPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import BY_ITEM, BY_USER, F32, F64, MEM_DEVICE, MEM_HOST, RMSE_TEST, RMSE_VALIDATE, check

SIDES = {"byUser": BY_USER, "byItem": BY_ITEM, BY_USER: BY_USER, BY_ITEM: BY_ITEM}
TRANSPORTS = {"rccl": _lib.COMM_RCCL, "shm": _lib.COMM_SHM, "ipc": _lib.COMM_IPC, "stub": _lib.COMM_STUB,
              _lib.COMM_RCCL: _lib.COMM_RCCL, _lib.COMM_SHM: _lib.COMM_SHM, _lib.COMM_IPC: _lib.COMM_IPC, _lib.COMM_STUB: _lib.COMM_STUB}
RMSE_SETS = {"rmseValidate": RMSE_VALIDATE, "rmseTest": RMSE_TEST, RMSE_VALIDATE: RMSE_VALIDATE,
             RMSE_TEST: RMSE_TEST}


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _ptr_kind(x, dtype=None):
    """(address, memKind) of a numpy array or torch tensor; checks dtype and contiguity."""
    if _is_torch(x):
        import torch
        want = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64,
                np.dtype(np.int32): torch.int32, np.dtype(np.int64): torch.int64}
        if dtype is not None and x.dtype != want[np.dtype(dtype)]:
            raise TypeError("invalid type!")  # cpp_utils/cpp_utils.js:12
        if not x.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return x.data_ptr(), (MEM_DEVICE if x.is_cuda else MEM_HOST)
    a = x
    if dtype is not None and a.dtype != np.dtype(dtype):
        raise TypeError("invalid type!")
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("array must be C-contiguous")
    return a.ctypes.data, MEM_HOST


def _portion_prefix(vals):
    if vals.dtype == np.float32:
        return "s"
    if vals.dtype == np.float64:
        return "d"
    raise TypeError("invalid type!")


def als_calc_portion(lam, k, alsRows, alsIndx, alsVals, fixedFactors, solvedFactors):
    """Level 1: drop-in for the body of EmfWorker.mw_calcTrainAlsPortion (EmfWorker.js:176-251).

    Host numpy buffers; solvedFactors rows named in alsRows are overwritten in place.
    Returns ratingsInPortion."""
    L = _lib.load()
    p = _portion_prefix(alsVals)
    for a, dt in ((alsRows, np.int32), (alsIndx, np.int32), (fixedFactors, alsVals.dtype),
                  (solvedFactors, alsVals.dtype)):
        if a.dtype != np.dtype(dt) or not a.flags["C_CONTIGUOUS"]:
            raise TypeError("invalid type!")
    f = getattr(L, f"ycnr_{p}AlsCalcPortion")
    return check(f(float(lam), int(k), alsRows.ctypes.data, alsIndx.ctypes.data, alsVals.ctypes.data,
                   fixedFactors.ctypes.data, fixedFactors.size // k, solvedFactors.ctypes.data,
                   solvedFactors.size // k))


def pin_fixed_factors(fixedFactors, k):
    """Level 1, once per half-step: keep the step's fixed factor matrix on the device for the portion
    calls that follow (ycnr_{s,d}AlsPinFixedFactors)."""
    L = _lib.load()
    p = _portion_prefix(fixedFactors)
    if not fixedFactors.flags["C_CONTIGUOUS"]:
        raise TypeError("invalid type!")
    check(getattr(L, f"ycnr_{p}AlsPinFixedFactors")(fixedFactors.ctypes.data, fixedFactors.size // k, int(k)))


def unpin_fixed_factors():
    """Level 1: forget the pinned matrix (ycnr_AlsUnpinFixedFactors)."""
    check(_lib.load().ycnr_AlsUnpinFixedFactors())


def release_portion_state():
    check(_lib.load().ycnr_AlsReleasePortionState())


def rmse_portion(k, rmseRows, rmseIndx, rmseVals, userFactors, itemFactors, globalAvgShift=0.0):
    """Level 1: drop-in for EmfWorker.mw_calcRmsePortion (EmfWorker.js:266-315).
    Returns np.array([rSumDiff2, rCnt, rSum])."""
    L = _lib.load()
    p = _portion_prefix(rmseVals)
    out = np.zeros(3, np.float64)
    f = getattr(L, f"ycnr_{p}RmsePortion")
    check(f(int(k), rmseRows.ctypes.data, rmseIndx.ctypes.data, rmseVals.ctypes.data, userFactors.ctypes.data,
            userFactors.size // k, itemFactors.ctypes.data, itemFactors.size // k, float(globalAvgShift),
            out.ctypes.data))
    return out


def split_to_sets(rowPtr, types, dataSetDistr=(85, 10, 5), seed=1):
    """EmfLord.doSplitToSets (lib/emf/EmfLord.js:402-505) on the GPU: completes the int8 array of
    dataset types (0 = unassigned) of a CSR-by-user matrix.  Returns (types, kernel ms)."""
    L = _lib.load()
    rp = np.ascontiguousarray(rowPtr, np.int64)
    t = np.array(types, np.int8, copy=True)
    pc = np.asarray(dataSetDistr, np.int32)
    ms = C.c_double(0.0)
    _lib.check(L.ycnr_split_to_sets(len(rp) - 1, rp.ctypes.data, t.ctypes.data, pc.ctypes.data, int(seed) & 0xFFFFFFFF, C.byref(ms)))
    return t, ms.value


def rating_stats(rowPtr, vals, types=None):
    """Per-row count and double sum of the ratings of type 1..3 (all when types is None): the
    ratings_count / avg_rating columns of EmfLord.js:252-396.  Returns (cnt, sum, kernel ms)."""
    L = _lib.load()
    rp = np.ascontiguousarray(rowPtr, np.int64)
    v = np.ascontiguousarray(vals)
    if v.dtype not in (np.float32, np.float64):
        raise TypeError("invalid type!")
    rows = len(rp) - 1
    cnt = np.zeros(rows, np.int32)
    sm = np.zeros(rows, np.float64)
    t = None if types is None else np.ascontiguousarray(types, np.int8)
    ms = C.c_double(0.0)
    _lib.check(L.ycnr_rating_stats(_lib.F64 if v.dtype == np.float64 else _lib.F32, rows, rp.ctypes.data, v.ctypes.data,
                                   None if t is None else t.ctypes.data, cnt.ctypes.data, sm.ctypes.data, C.byref(ms)))
    return cnt, sm, ms.value


def recommend_items(userRows, itemFactors, skipPtr, skipIds, globalAvgShift=0.0, minRecommendRating=7.0, limit=20):
    """YcnrController.recommendItemsForUser (lib/YcnrController.js:227-284) for a batch of users on
    the GPU.  userRows: [nUsers x k]; skipPtr / skipIds: CSR of 0-based item ids to leave out
    (rated + unrated_items), ascending per user.  Returns (ids [nUsers x limit] with -1 padding,
    predicts, counts, kernel ms); like the reference, at most limit - 1 items per user."""
    L = _lib.load()
    it = np.ascontiguousarray(itemFactors)
    if it.dtype not in (np.float32, np.float64):
        raise TypeError("invalid type!")
    ur = np.ascontiguousarray(userRows, it.dtype).reshape(-1, it.shape[1])
    sp = np.ascontiguousarray(skipPtr, np.int64)
    sk = np.ascontiguousarray(skipIds, np.int32)
    n = ur.shape[0]
    ids = np.full((n, limit), -1, np.int32)
    pred = np.zeros((n, limit), np.float64)
    cnt = np.zeros(n, np.int32)
    ms = C.c_double(0.0)
    _lib.check(L.ycnr_recommend_items(_lib.F64 if it.dtype == np.float64 else _lib.F32, it.shape[1], n, ur.ctypes.data, it.shape[0],
                                      it.ctypes.data, sp.ctypes.data, sk.ctypes.data if len(sk) else None, float(globalAvgShift),
                                      float(minRecommendRating), int(limit), ids.ctypes.data, pred.ctypes.data, cnt.ctypes.data,
                                      C.byref(ms)))
    return ids, pred, cnt, ms.value


class AlsDevice:
    """Level 2: resident trainer handle (one per GPU / process)."""

    def __init__(self, factorsCount, totalUsersCount, totalItemsCount, useDoublePrecision=False,
                 userFactReg=0.05, itemFactReg=0.05, device=0, chunkRatings=0, flags=0):
        self._L = _lib.load()
        self.k = int(factorsCount)
        self.users = int(totalUsersCount)
        self.items = int(totalItemsCount)
        self.dtype = np.float64 if useDoublePrecision else np.float32
        o = _lib.Options()
        o.struct_size = C.sizeof(_lib.Options)
        o.device = int(device)
        o.dtype = F64 if useDoublePrecision else F32
        o.factorsCount = self.k
        o.totalUsersCount = self.users
        o.totalItemsCount = self.items
        o.userFactReg = float(userFactReg)
        o.itemFactReg = float(itemFactReg)
        o.chunkRatings = int(chunkRatings)
        o.flags = int(flags)
        h = C.c_void_p()
        check(self._L.ycnr_als_create(C.byref(o), C.byref(h)))
        self._h = h
        self._keep = {}  # bound external tensors stay alive with the handle

    # -- lifetime -------------------------------------------------------------------
    def destroy(self):
        if getattr(self, "_h", None):
            self._L.ycnr_als_destroy(self._h)
            self._h = None
            self._keep = {}

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def rows(self, side):
        return self.users if SIDES[side] == BY_USER else self.items

    # -- data -----------------------------------------------------------------------
    def set_stream(self, hip_stream):
        """hip_stream: a hipStream_t value (0 = the default stream), or None for the handle's own stream."""
        check(self._L.ycnr_als_set_stream(self._h, C.c_void_p(-1 if hip_stream is None else int(hip_stream))))

    def _upload(self, fn, key, rowPtr, indx, vals, rowBegin, rowEnd):
        n = (len(rowPtr) if not _is_torch(rowPtr) else rowPtr.numel()) - 1
        if rowEnd is None:
            rowEnd = n
        p0, k0 = _ptr_kind(rowPtr, np.int64)
        p1, k1 = _ptr_kind(indx, np.int32)
        p2, k2 = _ptr_kind(vals, self.dtype)
        if not (k0 == k1 == k2):
            raise ValueError("rowPtr, indx and vals must live in the same kind of memory")
        check(fn(self._h, key, p0, p1, p2, int(rowBegin), int(rowEnd), k0))

    def set_ratings(self, side, rowPtr, indx, vals, rowBegin=0, rowEnd=None):
        self._upload(self._L.ycnr_als_set_ratings, SIDES[side], rowPtr, indx, vals, rowBegin, rowEnd)

    def set_ratings_sharded(self, side, rowPtr, indx, vals, bounds, nChunks=None):
        """Sharded upload for every rank of the communicator: bounds[r] = ascending row ids
        [begin, cut, ..., end] of rank r (all ranks the same number of pieces).  Afterwards step(side)
        includes the exchange of the solved rows (ycnr_als_set_ratings_sharded)."""
        b = np.ascontiguousarray(bounds, np.int64)
        if b.ndim != 2:
            raise ValueError("bounds must be [world, nChunks + 1]")
        p0, k0 = _ptr_kind(rowPtr, np.int64)
        p1, k1 = _ptr_kind(indx, np.int32)
        p2, k2 = _ptr_kind(vals, self.dtype)
        if not (k0 == k1 == k2):
            raise ValueError("rowPtr, indx and vals must live in the same kind of memory")
        check(self._L.ycnr_als_set_ratings_sharded(self._h, SIDES[side], p0, p1, p2, k0, b.shape[1] - 1, b.shape[0], b.ctypes.data))

    def set_ratings_banded(self, side, rowPtr, indx, vals, bandBounds, rankBands, ownerBounds):
        """The side's half-step sharded by bands of COLUMNS (ycnr_als_set_ratings_banded): rowPtr / indx / vals hold every row of
        the side with only the ratings whose column lies in this rank's bands."""
        bb = np.ascontiguousarray(bandBounds, np.int64)
        rb = np.ascontiguousarray(rankBands, np.int64)
        ob = np.ascontiguousarray(ownerBounds, np.int64)
        p0, k0 = _ptr_kind(rowPtr, np.int64)
        p1, k1 = _ptr_kind(indx, np.int32)
        p2, k2 = _ptr_kind(vals, self.dtype)
        if not (k0 == k1 == k2):
            raise ValueError("rowPtr, indx and vals must live in the same kind of memory")
        check(self._L.ycnr_als_set_ratings_banded(self._h, SIDES[side], p0, p1, p2, k0, len(bb) - 1, bb.ctypes.data, rb.ctypes.data, ob.ctypes.data))

    # -- multi-GPU exchange -----------------------------------------------------------
    @staticmethod
    def comm_unique_id(transport="rccl"):
        """128 bytes made by ONE rank; every rank passes the same bytes to comm_init."""
        L = _lib.load()
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        check(L.ycnr_comm_unique_id(TRANSPORTS[transport], buf))
        return buf.raw

    def comm_init(self, unique_id, rank, world, transport="rccl"):
        if len(unique_id) != _lib.COMM_ID_BYTES:
            raise ValueError("unique id must be 128 bytes")
        check(self._L.ycnr_als_comm_init(self._h, TRANSPORTS[transport], C.c_char_p(bytes(unique_id)), int(rank), int(world)))

    def comm_destroy(self):
        check(self._L.ycnr_als_comm_destroy(self._h))

    def exchange(self, side):
        check(self._L.ycnr_als_exchange(self._h, SIDES[side]))

    def defer_exchange(self, side, deferred=True):
        check(self._L.ycnr_als_defer_exchange(self._h, SIDES[side], 1 if deferred else 0))

    def broadcast_factors(self, side, root=0):
        check(self._L.ycnr_als_broadcast_factors(self._h, SIDES[side], int(root)))

    def allreduce_sum(self, arr):
        """In-place sum over ranks of a float64 numpy array."""
        a = np.ascontiguousarray(arr, np.float64)
        check(self._L.ycnr_als_allreduce_sum(self._h, a.ctypes.data, a.size))
        return a

    def comm_selftest(self, nFloats=1 << 20):
        check(self._L.ycnr_als_comm_selftest(self._h, int(nFloats)))

    def comm_info(self):
        """{transport, rank, world, rccl_ranks}: rccl_ranks is what RCCL itself counts (ncclCommCount), None off RCCL."""
        out = np.zeros(4, np.int32)
        check(self._L.ycnr_als_comm_info(self._h, out.ctypes.data))
        names = {v: k for k, v in TRANSPORTS.items() if isinstance(k, str)}
        return {"transport": names.get(int(out[0]), "none"), "rank": int(out[1]), "world": int(out[2]),
                "rccl_ranks": int(out[3]) if out[3] >= 0 else None}

    def last_rmse_ms(self):
        ms = C.c_double(0.0)
        check(self._L.ycnr_als_last_rmse_ms(self._h, C.byref(ms)))
        return ms.value

    def set_rmse_ratings(self, which, rowPtr, indx, vals, rowBegin=0, rowEnd=None):
        self._upload(self._L.ycnr_als_set_rmse_ratings, RMSE_SETS[which], rowPtr, indx, vals, rowBegin, rowEnd)

    def set_factors(self, side, src):
        p, kind = _ptr_kind(src, self.dtype)
        n = src.numel() if _is_torch(src) else src.size
        if n != self.rows(side) * self.k:
            raise ValueError("factor matrix has the wrong size")
        check(self._L.ycnr_als_set_factors(self._h, SIDES[side], p, kind))

    def get_factors(self, side, rowBegin=0, rowCount=None):
        if rowCount is None:
            rowCount = self.rows(side) - rowBegin
        out = np.empty((rowCount, self.k), self.dtype)
        check(self._L.ycnr_als_get_factors(self._h, SIDES[side], out.ctypes.data, int(rowBegin), int(rowCount),
                                           MEM_HOST))
        return out

    def factors_ptr(self, side):
        p = C.c_void_p()
        check(self._L.ycnr_als_factors_ptr(self._h, SIDES[side], C.byref(p)))
        return p.value

    def bind_factors(self, side, tensor):
        """Adopt a torch CUDA tensor [rows, k] as the side's factor matrix (no copy)."""
        p, kind = _ptr_kind(tensor, self.dtype)
        if kind != MEM_DEVICE or tensor.numel() != self.rows(side) * self.k:
            raise ValueError("bind_factors needs a CUDA tensor of shape [rows, k]")
        check(self._L.ycnr_als_bind_factors(self._h, SIDES[side], p))
        self._keep[SIDES[side]] = tensor

    # -- compute --------------------------------------------------------------------
    def step(self, side):
        """One half-step = EmfLord.alsTrainStep(stepType) (EmfLord.js:963-984). Returns StepInfo."""
        check(self._L.ycnr_als_step(self._h, SIDES[side]))
        return self.last_step_info()

    def step_async(self, side):
        check(self._L.ycnr_als_step_async(self._h, SIDES[side]))

    def sync(self):
        check(self._L.ycnr_als_sync(self._h))

    def step_info(self, side):
        """StepInfo of the last completed half-step of one side (after sync)."""
        info = _lib.StepInfo()
        check(self._L.ycnr_als_step_info_of(self._h, SIDES[side], C.byref(info)))
        return info

    def iteration(self):
        """EmfLord.alsTrainIter (EmfLord.js:954-958) without the host between the two half-steps: both enqueued, one sync.
        Returns (StepInfo byUser, StepInfo byItem)."""
        self.step_async("byUser")
        self.step_async("byItem")
        self.sync()
        return self.step_info("byUser"), self.step_info("byItem")

    def last_step_info(self):
        info = _lib.StepInfo()
        check(self._L.ycnr_als_last_step_info(self._h, C.byref(info)))
        return info

    def rmse(self, which, globalAvgShift=0.0, portionRowEnd=None):
        """Partial sums per portion: array [nPortions, 3] of {rSumDiff2, rCnt, rSum}."""
        if portionRowEnd is None or len(portionRowEnd) == 0:
            out = np.zeros((1, 3), np.float64)
            check(self._L.ycnr_als_rmse(self._h, RMSE_SETS[which], float(globalAvgShift), 0, None, out.ctypes.data))
            return out
        ends = np.ascontiguousarray(portionRowEnd, np.int64)
        out = np.zeros((len(ends), 3), np.float64)
        check(self._L.ycnr_als_rmse(self._h, RMSE_SETS[which], float(globalAvgShift), len(ends), ends.ctypes.data,
                                    out.ctypes.data))
        return out
