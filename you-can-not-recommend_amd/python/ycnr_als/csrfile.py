"""Ratings ingestion: MovieLens text files, (user, item, rating) triplets ->
CSR on the GPU, and the binary CSR file pair a training run starts from.

File layout (include/ycnr_als.h): "YCSR", uint32 version, uint32 dtype, uint32 flags,
int64 rows, cols, nnz, int64 rowPtr[rows + 1], int32 indx[nnz], T vals[nnz]; little endian.
lib/CsrFile.js writes the same bytes.
"""
import ctypes as C
import struct

import numpy as np

from . import _lib
from .data import Csr

MAGIC = b"YCSR"
VERSION = 1
FLAG_SORTED = 1
_HEADER = struct.Struct("<4sIIIqqq")


def read_movielens(path):
    """'u.data' (tab separated: user item rating timestamp, data/db-schema.sql:459-476) or
    'ratings.dat' (user::item::rating::timestamp, lib/YcnrController.js:126-133); ids 1-based in
    the file, 0-based here (EmfMaster.js:584-586).  Returns (users, items, user, item, rating)."""
    sep = "::" if path.endswith(".dat") else "\t"
    user, item, rating = [], [], []
    with open(path, "r") as f:
        for line in f:
            if not line.strip():
                continue
            p = line.rstrip("\n").split(sep)
            user.append(int(p[0]) - 1)
            item.append(int(p[1]) - 1)
            rating.append(float(p[2]))
    user = np.asarray(user, np.int32)
    item = np.asarray(item, np.int32)
    return int(user.max()) + 1, int(item.max()) + 1, user, item, np.asarray(rating, np.float32)


def csr_from_triplets(row, col, vals, rows, cols):
    """CSR ordered by (row, col, input position), built on the GPU.  Returns (Csr, kernel ms)."""
    L = _lib.load()
    r = np.ascontiguousarray(row, np.int32)
    c = np.ascontiguousarray(col, np.int32)
    v = np.ascontiguousarray(vals)
    if v.dtype not in (np.float32, np.float64):
        raise TypeError("invalid type!")
    rp = np.zeros(rows + 1, np.int64)
    ix = np.zeros(len(r), np.int32)
    ov = np.zeros_like(v)
    ms = C.c_double(0.0)
    _lib.check(L.ycnr_csr_from_triplets(_lib.F64 if v.dtype == np.float64 else _lib.F32, len(r), r.ctypes.data, c.ctypes.data,
                                        v.ctypes.data, rows, cols, rp.ctypes.data, ix.ctypes.data, ov.ctypes.data, C.byref(ms)))
    return Csr(rows, cols, rp, ix, ov), ms.value


def transpose(a):
    """The same ratings by column (CSR by user -> CSR by item), on the GPU.  Returns (Csr, kernel ms)."""
    L = _lib.load()
    rp = np.ascontiguousarray(a.rowPtr, np.int64)
    ix = np.ascontiguousarray(a.indx, np.int32)
    v = np.ascontiguousarray(a.vals)
    op = np.zeros(a.cols + 1, np.int64)
    oi = np.zeros(len(ix), np.int32)
    ov = np.zeros_like(v)
    ms = C.c_double(0.0)
    _lib.check(L.ycnr_csr_transpose(_lib.F64 if v.dtype == np.float64 else _lib.F32, a.rows, a.cols, rp.ctypes.data, ix.ctypes.data,
                                    v.ctypes.data, op.ctypes.data, oi.ctypes.data, ov.ctypes.data, C.byref(ms)))
    return Csr(a.cols, a.rows, op, oi, ov), ms.value


def write_csr(path, a, sorted_rows=True):
    v = np.ascontiguousarray(a.vals)
    dtype = 1 if v.dtype == np.float64 else 0
    with open(path, "wb") as f:
        f.write(_HEADER.pack(MAGIC, VERSION, dtype, FLAG_SORTED if sorted_rows else 0, a.rows, a.cols, len(v)))
        f.write(np.ascontiguousarray(a.rowPtr, "<i8").tobytes())
        f.write(np.ascontiguousarray(a.indx, "<i4").tobytes())
        f.write(v.astype("<f8" if dtype else "<f4", copy=False).tobytes())


def read_csr(path):
    with open(path, "rb") as f:
        magic, version, dtype, flags, rows, cols, nnz = _HEADER.unpack(f.read(_HEADER.size))
        if magic != MAGIC or version != VERSION or dtype not in (0, 1):
            raise ValueError(f"{path}: not a YCSR version {VERSION} file")
        rp = np.frombuffer(f.read(8 * (rows + 1)), "<i8").astype(np.int64)
        ix = np.frombuffer(f.read(4 * nnz), "<i4").astype(np.int32)
        v = np.frombuffer(f.read((8 if dtype else 4) * nnz), "<f8" if dtype else "<f4").astype(np.float64 if dtype else np.float32)
    if len(rp) != rows + 1 or len(ix) != nnz or len(v) != nnz or rp[0] != 0 or rp[-1] != nnz:
        raise ValueError(f"{path}: truncated or inconsistent")
    return Csr(rows, cols, rp, ix, v)
