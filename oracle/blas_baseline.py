"""CPU baseline, BLAS / LAPACK flavour: the reference's per-row
structure -- gather (EmfBase.js:537-555), A = Y^T Y through gemm (EmfWorker.js:231-232),
A += lambda n I (:233-235), b = Y^T r (:238-245), general square solve through gesv (:246) -- with
the arithmetic in the BLAS / LAPACK this box offers (PyTorch's CPU build: MKL), one worker process
per host core like the reference's numThreadsForTrain worker processes (EmfMaster.js:44-98), one
BLAS thread each.

TEST INFRASTRUCTURE ONLY, like the rest of oracle/: run by bench.py's cpu_baseline leg as a child
process (python oracle/blas_baseline.py <sample.npz> <workers>), never imported by the product.
Prints one JSON line: {"ratings": n, "seconds": t, "rows": r, "workers": w}.
"""
import json
import multiprocessing as mp
import sys
import time

import numpy as np


def solve_rows(args):
    path, lo, hi = args
    import torch
    torch.set_num_threads(1)
    z = np.load(path)
    k, lam = int(z["k"]), float(z["lam"])
    rp, indx, vals = z["rowPtr"], torch.from_numpy(z["indx"].astype(np.int64)), torch.from_numpy(z["vals"])
    fixed = torch.from_numpy(z["fixed"])
    out = torch.zeros(hi - lo, k, dtype=fixed.dtype)
    eye = torch.eye(k, dtype=fixed.dtype)
    t = time.perf_counter()
    n = 0
    for r in range(lo, hi):
        b, e = int(rp[r]), int(rp[r + 1])
        if e == b:
            continue
        Y = fixed[indx[b:e]]                      # copySubFixedFactors
        A = Y.t() @ Y                             # BLAS.gemm(Trans, NoTrans)
        A += (lam * (e - b)) * eye                # lambda.diagonal(lambda * n); A.add(lambda)
        rhs = Y.t() @ vals[b:e]                   # subFixedFactors.transposed().multiply(rat)
        out[r - lo] = torch.linalg.solve(A, rhs)  # Matrix.solveSquare: LAPACK gesv
        n += e - b
    return n, time.perf_counter() - t


def main():
    path, workers = sys.argv[1], int(sys.argv[2])
    rp = np.load(path)["rowPtr"]
    rows = len(rp) - 1
    # contiguous row ranges with equal ratings, one per worker
    cuts = np.searchsorted(rp, np.linspace(0, rp[-1], workers + 1)).astype(int)
    cuts[0], cuts[-1] = 0, rows
    jobs = [(path, int(cuts[i]), int(cuts[i + 1])) for i in range(workers) if cuts[i + 1] > cuts[i]]
    t = time.perf_counter()
    with mp.get_context("fork").Pool(len(jobs)) as pool:
        res = pool.map(solve_rows, jobs)
    wall = time.perf_counter() - t
    print(json.dumps({"ratings": int(sum(r[0] for r in res)), "seconds": wall, "rows": rows, "workers": len(jobs),
                      "busy_seconds_max": max(r[1] for r in res)}))


if __name__ == "__main__":
    main()
