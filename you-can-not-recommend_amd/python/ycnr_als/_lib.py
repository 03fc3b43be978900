"""ctypes binding of libycnr_als.so (include/ycnr_als.h).

There is no fallback: if the HIP library is missing this module raises at load time, and
every call that fails raises YcnrError carrying ycnr_last_error().
"""
import ctypes as C
import os

# The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and reads the variable when
# it starts: with 8, the row kernel and every dual class of a half-step get a queue of their own.  libycnr_als.so looks at the
# variable when it is LOADED (it never sets it) and uses one side stream per dual class only if it says >= 8 then -- so it is
# set here, before the library is loaded, unless torch has initialised HIP already (the runtime then runs with what it read at
# its start, and five side streams on four queues would be the slow configuration).
def _hip_started():
    import sys
    t = sys.modules.get("torch")
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:  # noqa: BLE001
        return False


if not _hip_started():
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.normpath(os.path.join(_HERE, "..", "..", "csrc"))
# YCNR_ALS_LIB selects another build of the same library (A/B tests of kernel variants)
SO_PATH = os.environ.get("YCNR_ALS_LIB") or os.path.join(CSRC, "libycnr_als.so")

OK = 0
ERR_INVALID, ERR_HIP, ERR_NOMEM, ERR_UNSUPPORTED, ERR_NUMERIC, ERR_STATE = -1, -2, -3, -4, -5, -6
BY_USER, BY_ITEM = 0, 1
F32, F64 = 0, 1
MEM_HOST, MEM_DEVICE = 0, 1
RMSE_VALIDATE, RMSE_TEST = 0, 1
FLAG_LDS_SOLVER, FLAG_NO_DUAL, FLAG_LOCALITY_SORT, FLAG_NO_VALU_EDGE, FLAG_NO_BF16X6, FLAG_NO_BANDS = 1, 2, 4, 8, 16, 32
FLAG_NO_OVERLAP = 64
FLAG_NO_GRAPH = 128

# every symbol include/ycnr_als.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "ycnr_last_error", "ycnr_version", "ycnr_device_count",
    "ycnr_sAlsCalcPortion", "ycnr_dAlsCalcPortion", "ycnr_sRmsePortion", "ycnr_dRmsePortion",
    "ycnr_sAlsPinFixedFactors", "ycnr_dAlsPinFixedFactors", "ycnr_AlsUnpinFixedFactors", "ycnr_AlsReleasePortionState",
    "ycnr_als_create", "ycnr_als_destroy", "ycnr_als_set_stream", "ycnr_als_set_ratings",
    "ycnr_als_set_rmse_ratings", "ycnr_als_set_factors", "ycnr_als_get_factors", "ycnr_als_factors_ptr",
    "ycnr_als_bind_factors", "ycnr_als_step", "ycnr_als_step_async", "ycnr_als_sync",
    "ycnr_als_last_step_info", "ycnr_als_step_info_of", "ycnr_als_rmse",
    "ycnr_split_to_sets", "ycnr_rating_stats", "ycnr_csr_from_triplets", "ycnr_csr_transpose",
    "ycnr_recommend_items",
    "ycnr_comm_unique_id", "ycnr_als_comm_init", "ycnr_als_comm_destroy", "ycnr_als_set_ratings_sharded",
    "ycnr_als_exchange", "ycnr_als_broadcast_factors", "ycnr_als_allreduce_sum", "ycnr_als_comm_selftest",
    "ycnr_als_comm_info", "ycnr_als_last_rmse_ms", "ycnr_als_set_ratings_banded", "ycnr_als_defer_exchange",
]
COMM_NONE, COMM_RCCL, COMM_SHM, COMM_IPC, COMM_STUB = 0, 1, 2, 3, 4
COMM_ID_BYTES = 128
ABI_VERSION = 4


class YcnrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ycnr_als error {code}: {msg}")
        self.code = code


class Options(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("dtype", C.c_int32),
                ("factorsCount", C.c_int32), ("totalUsersCount", C.c_int64), ("totalItemsCount", C.c_int64),
                ("userFactReg", C.c_double), ("itemFactReg", C.c_double), ("chunkRatings", C.c_int32),
                ("flags", C.c_int32)]


class StepInfo(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("side", C.c_int32), ("rows", C.c_int64), ("ratings", C.c_int64),
                ("units", C.c_int64), ("splitRows", C.c_int64), ("fusedRows", C.c_int64),
                ("fusedRatings", C.c_int64), ("dualRows", C.c_int64), ("dualRatings", C.c_int64),
                ("gramSlabMs", C.c_float), ("gramSolveMs", C.c_float), ("dualSolveMs", C.c_float),
                ("reduceSolveMs", C.c_float), ("totalMs", C.c_float), ("numericErrors", C.c_int32),
                ("dualOverlapped", C.c_int32), ("parts", C.c_int32), ("reserved0", C.c_int32),
                ("exchangeBytes", C.c_int64), ("exchangeMs", C.c_float), ("exposedExchangeMs", C.c_float),
                ("dualFlops", C.c_double)]


_lib = None


def load():
    """Load the shared library and declare prototypes. Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(
            f"{SO_PATH} is missing: build it with `make -C {CSRC}` (or __graft_entry__.build()). "
            "The ALS path has no CPU fallback.")
    # PyTorch bundles its own HIP runtime under the same SONAME (libamdhip64.so.7).  Whichever copy
    # is loaded first serves the whole process, and torch does not see the GPU through the system's
    # copy (observed: torch.cuda.is_available() == False when this library was loaded first).  The
    # host classes of this package use torch for device memory and collectives, so let it go first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(SO_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_double
    L.ycnr_last_error.restype = C.c_char_p
    L.ycnr_last_error.argtypes = []
    L.ycnr_version.restype = i32
    L.ycnr_device_count.restype = i32
    for p in "sd":
        f = getattr(L, f"ycnr_{p}AlsCalcPortion")
        f.restype = i64
        f.argtypes = [dbl, i32, vp, vp, vp, vp, i64, vp, i64]
        f = getattr(L, f"ycnr_{p}RmsePortion")
        f.restype = i32
        f.argtypes = [i32, vp, vp, vp, vp, i64, vp, i64, dbl, vp]
    for p in "sd":
        f = getattr(L, f"ycnr_{p}AlsPinFixedFactors")
        f.restype = i32
        f.argtypes = [vp, i64, i32]
    L.ycnr_AlsUnpinFixedFactors.restype = i32
    L.ycnr_AlsUnpinFixedFactors.argtypes = []
    L.ycnr_AlsReleasePortionState.restype = i32
    L.ycnr_AlsReleasePortionState.argtypes = []
    L.ycnr_als_create.restype = i32
    L.ycnr_als_create.argtypes = [C.POINTER(Options), C.POINTER(vp)]
    L.ycnr_als_destroy.restype = i32
    L.ycnr_als_destroy.argtypes = [vp]
    L.ycnr_als_set_stream.restype = i32
    L.ycnr_als_set_stream.argtypes = [vp, vp]
    for name in ("ycnr_als_set_ratings", "ycnr_als_set_rmse_ratings"):
        f = getattr(L, name)
        f.restype = i32
        f.argtypes = [vp, i32, vp, vp, vp, i64, i64, i32]
    L.ycnr_als_set_factors.restype = i32
    L.ycnr_als_set_factors.argtypes = [vp, i32, vp, i32]
    L.ycnr_als_get_factors.restype = i32
    L.ycnr_als_get_factors.argtypes = [vp, i32, vp, i64, i64, i32]
    L.ycnr_als_factors_ptr.restype = i32
    L.ycnr_als_factors_ptr.argtypes = [vp, i32, C.POINTER(vp)]
    L.ycnr_als_bind_factors.restype = i32
    L.ycnr_als_bind_factors.argtypes = [vp, i32, vp]
    for name in ("ycnr_als_step", "ycnr_als_step_async"):
        f = getattr(L, name)
        f.restype = i32
        f.argtypes = [vp, i32]
    L.ycnr_als_sync.restype = i32
    L.ycnr_als_sync.argtypes = [vp]
    L.ycnr_als_last_step_info.restype = i32
    L.ycnr_als_last_step_info.argtypes = [vp, C.POINTER(StepInfo)]
    L.ycnr_als_step_info_of.restype = i32
    L.ycnr_als_step_info_of.argtypes = [vp, i32, C.POINTER(StepInfo)]
    L.ycnr_als_rmse.restype = i32
    L.ycnr_als_rmse.argtypes = [vp, i32, dbl, i32, vp, vp]
    L.ycnr_split_to_sets.restype = i32
    L.ycnr_split_to_sets.argtypes = [i64, vp, vp, vp, C.c_uint32, C.POINTER(dbl)]
    L.ycnr_rating_stats.restype = i32
    L.ycnr_rating_stats.argtypes = [i32, i64, vp, vp, vp, vp, vp, C.POINTER(dbl)]
    L.ycnr_csr_from_triplets.restype = i32
    L.ycnr_csr_from_triplets.argtypes = [i32, i64, vp, vp, vp, i64, i64, vp, vp, vp, C.POINTER(dbl)]
    L.ycnr_recommend_items.restype = i32
    L.ycnr_recommend_items.argtypes = [i32, i32, i64, vp, i64, vp, vp, vp, dbl, dbl, i32, vp, vp, vp, C.POINTER(dbl)]
    L.ycnr_csr_transpose.restype = i32
    L.ycnr_csr_transpose.argtypes = [i32, i64, i64, vp, vp, vp, vp, vp, vp, C.POINTER(dbl)]
    L.ycnr_comm_unique_id.restype = i32
    L.ycnr_comm_unique_id.argtypes = [i32, vp]
    L.ycnr_als_comm_init.restype = i32
    L.ycnr_als_comm_init.argtypes = [vp, i32, vp, i32, i32]
    L.ycnr_als_comm_destroy.restype = i32
    L.ycnr_als_comm_destroy.argtypes = [vp]
    L.ycnr_als_set_ratings_sharded.restype = i32
    L.ycnr_als_set_ratings_sharded.argtypes = [vp, i32, vp, vp, vp, i32, i32, i32, vp]
    L.ycnr_als_set_ratings_banded.restype = i32
    L.ycnr_als_set_ratings_banded.argtypes = [vp, i32, vp, vp, vp, i32, i32, vp, vp, vp]
    L.ycnr_als_defer_exchange.restype = i32
    L.ycnr_als_defer_exchange.argtypes = [vp, i32, i32]
    L.ycnr_als_exchange.restype = i32
    L.ycnr_als_exchange.argtypes = [vp, i32]
    L.ycnr_als_broadcast_factors.restype = i32
    L.ycnr_als_broadcast_factors.argtypes = [vp, i32, i32]
    L.ycnr_als_allreduce_sum.restype = i32
    L.ycnr_als_allreduce_sum.argtypes = [vp, vp, i64]
    L.ycnr_als_comm_selftest.restype = i32
    L.ycnr_als_comm_selftest.argtypes = [vp, i64]
    L.ycnr_als_comm_info.restype = i32
    L.ycnr_als_comm_info.argtypes = [vp, vp]
    L.ycnr_als_last_rmse_ms.restype = i32
    L.ycnr_als_last_rmse_ms.argtypes = [vp, C.POINTER(dbl)]
    _lib = L
    return L


def check(rc):
    """Raise YcnrError for a negative return code; pass the value through otherwise."""
    if rc < 0:
        raise YcnrError(rc, load().ycnr_last_error().decode("utf-8", "replace"))
    return rc
