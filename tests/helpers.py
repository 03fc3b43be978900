"""Shared helpers for the test-suite: small seeded problems, an independent float64 numpy
solve, and a backend that runs the oracle behind the product's host classes (used only to
test host logic and the sharded exchange on CPU -- never shipped)."""
import numpy as np

from oracle import oracle as orc

EPS32 = float(np.finfo(np.float32).eps)


def host_cpus():
    """CPUs this process can really use: the affinity mask capped by the cgroup CPU quota (the GPU
    boxes report os.cpu_count() = 256 under a 16-CPU quota)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def make_problem(users, items, k, density=0.1, seed=0, dtype=np.float32, max_rating=5, min_per_row=0,
                 empty_rows=()):
    """Random CSR by user + the same ratings by item, and random factor matrices."""
    rng = np.random.default_rng(seed)
    mask = rng.random((users, items)) < density
    for u in range(users):
        if mask[u].sum() < min_per_row:
            mask[u, rng.choice(items, min_per_row, replace=False)] = True
    for u in empty_rows:
        mask[u] = False
    R = np.where(mask, rng.integers(1, max_rating + 1, (users, items)), 0).astype(dtype)
    U = (rng.standard_normal((users, k)) / np.sqrt(k)).astype(dtype)
    V = (rng.standard_normal((items, k)) / np.sqrt(k)).astype(dtype)
    return csr_of(R, mask), csr_of(R.T, mask.T), U, V


def csr_of(R, mask):
    from ycnr_als.data import Csr
    rows, cols = R.shape
    rowPtr = np.zeros(rows + 1, np.int64)
    rowPtr[1:] = np.cumsum(mask.sum(1))
    r, c = np.nonzero(mask)
    return Csr(rows, cols, rowPtr, c.astype(np.int32), np.ascontiguousarray(R[r, c]))


def numpy_row_solve(lam, k, cols_idx, vals, fixed):
    """Independent second opinion (LAPACK through numpy, float64): returns x and the
    amplification factor of a working-precision solve of this row,
        amp = cond(A) * || |Y|^T |r| || / || Y^T r ||,
    i.e. the conditioning of the solve times the cancellation in forming b = Y^T r
    (forward error of x <= c * amp * eps, Higham, Accuracy and Stability, thm 7.2 + 3.5)."""
    Y = fixed[cols_idx].astype(np.float64)
    r = vals.astype(np.float64)
    n = len(cols_idx)
    A = Y.T @ Y + (lam * n) * np.eye(k)
    b = Y.T @ r
    kb = np.linalg.norm(np.abs(Y).T @ np.abs(r)) / max(np.linalg.norm(b), 1e-300)
    return np.linalg.solve(A, b), np.linalg.cond(A) * max(kb, 1.0)


def numpy_step(lam, k, csr, fixed, solved):
    """float64 reference of a half-step; returns (new matrix, cond per row)."""
    out = solved.astype(np.float64).copy()
    conds = np.ones(csr.rows)
    for r in range(csr.rows):
        b, e = csr.rowPtr[r], csr.rowPtr[r + 1]
        if e > b:
            out[r], conds[r] = numpy_row_solve(lam, k, csr.indx[b:e], csr.vals[b:e], fixed)
    return out, conds


def row_rel_err(a, ref):
    num = np.linalg.norm(a.astype(np.float64) - ref.astype(np.float64), axis=1)
    den = np.maximum(np.linalg.norm(ref.astype(np.float64), axis=1), 1e-300)
    return num / den


class OracleBackend:
    """Backend with the interface of ycnr_als.emf.HipBackend, computing with the CPU oracle.
    TEST ONLY: lets the host classes and the gloo exchange be exercised without a GPU."""

    def __init__(self, opts, users, items, device=0, threads=1, dtype=None):
        import torch
        self.torch = torch
        self.threads = threads
        self.k = opts["factorsCount"]
        self.dt = np.float64 if opts["useDoublePrecision"] else np.float32
        if dtype is not None:  # e.g. the float64 oracle behind a float32 run's host options
            self.dt = dtype
        self.lam = [opts["als"]["userFactReg"], opts["als"]["itemFactReg"]]
        self.fac_np = [np.zeros((users, self.k), self.dt), np.zeros((items, self.k), self.dt)]
        self.fac = [torch.from_numpy(a) for a in self.fac_np]  # shared memory views
        self.ratings, self.rm = {}, {}

    def factors(self, side):
        return self.fac[side]

    def set_factors(self, side, arr):
        self.fac_np[side][...] = arr

    def get_factors(self, side):
        return self.fac_np[side].copy()

    def set_ratings(self, side, csr, rb, re):
        self.ratings[side] = (csr.numpy().astype(self.dt), rb, re)

    def set_rmse_ratings(self, which, csr, rb, re):
        self.rm[which] = (csr.numpy().astype(self.dt), rb, re)

    def step(self, side):
        c, rb, re = self.ratings[side]
        orc.als_step_csr(self.lam[side], self.k, c.rowPtr, c.indx, c.vals, self.fac_np[1 - side], self.fac_np[side],
                         rb, re, threads=self.threads)
        return None

    def rmse(self, which, shift, ends):
        c, rb, re = self.rm[which]
        out = np.zeros((max(len(ends), 1), 3))
        prev = rb
        ends = [re] if len(ends) == 0 else ends
        for p, e in enumerate(ends):
            e = int(min(max(e, rb), re))
            lo = max(prev, rb)
            if e > lo:
                out[p] = orc.rmse_csr(self.k, c.rowPtr, c.indx, c.vals, self.fac_np[0], self.fac_np[1], shift, lo, e)
            prev = max(prev, e)
        return out

    def destroy(self):
        pass
