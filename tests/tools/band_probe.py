#!/usr/bin/env python3
"""What the item half-step of ONE rank of eight costs when the side is sharded by USER BANDS instead of by items:
rank r accumulates the Gramians of ALL items over the ratings of its own users.  Stand-in built from the existing API: a handle whose
byItem ratings are the MAL-scale CSR-by-item restricted to the users of band r (every item a row), one byItem half-step; the
chunk kernel's time (gramSlabMs) is the rank's Gramian cost, reduceSolveMs here solves all 12.7 K items (a rank would solve an eighth).
Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "you-can-not-recommend_amd", "python"))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench import WORKLOADS  # noqa: E402
from ycnr_als.data import Csr, init_factors, select_csr, synth_ratings  # noqa: E402
from ycnr_als.emf import shard_ranges  # noqa: E402
from ycnr_als.trainer import AlsDevice  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "mal"
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    users, items, nnz, k, max_rating, zipf_a, sigma, desc = WORKLOADS[wl]
    dev = torch.device("cuda", 0)
    by_user, by_item = synth_ratings(users, items, nnz, max_rating=max_rating, seed=20260004, device=dev, degree_sigma=sigma, zipf_a=zipf_a)
    cu = by_user.counts().cpu().numpy()
    bands = shard_ranges(cu, world, k)
    U = torch.from_numpy(init_factors(users, k, 1)).to(dev)
    V = torch.from_numpy(init_factors(items, k, 2)).to(dev)
    out = {"workload": desc, "world": world, "bands": bands.tolist(), "ranks": []}
    for r in range(world):
        lo, hi = int(bands[r]), int(bands[r + 1])
        mask = (by_item.indx >= lo) & (by_item.indx < hi)
        sub = select_csr(by_item, mask)
        h = AlsDevice(k, users, items)
        h.bind_factors(0, U)
        h.bind_factors(1, V.clone())
        h.set_ratings("byItem", sub.rowPtr, sub.indx, sub.vals)
        ms = []
        for it in range(4):
            torch.cuda.synchronize()
            info = h.step("byItem")
            if it:
                ms.append((info.gramSlabMs, info.gramSolveMs, info.dualSolveMs, info.reduceSolveMs, info.totalMs))
        m = np.mean(np.asarray(ms), axis=0)
        out["ranks"].append({"rank": r, "ratings": int(sub.nnz), "chunk_gramian_ms": round(float(m[0]), 4), "row_kernel_ms": round(float(m[1]), 4),
                             "dual_ms": round(float(m[2]), 4), "reduce_solve_all_items_ms": round(float(m[3]), 4), "total_ms": round(float(m[4]), 4),
                             "splitRows": int(info.splitRows), "fusedRows": int(info.fusedRows)})
        print("band %d: %s" % (r, out["ranks"][-1]), file=sys.stderr, flush=True)
        h.destroy()
        del h, sub, mask
        torch.cuda.empty_cache()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
