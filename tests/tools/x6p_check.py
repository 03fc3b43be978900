"""The fused row kernel on pre-split planes (GramX6P) against the LDS-DMA float kernel (GramX6D): same products in the same
order, so every row must be bit-identical -- except at k = 16 m + 4, where the corner tile of the packed last block (k = 4: GramX6D's packed form
against the plain one here) adds its nine plane products in another order (float32 rounding of a 5 x 5 corner: rows may differ in the last bits).  Runs itself
twice (child processes: the toggles are read once per process) and compares.   python tests/tools/x6p_check.py"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "you-can-not-recommend_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
KS = (100, 20, 36, 52, 68, 84, 108, 44, 4, 8, 12)


def child(tag):
    import ycnr_als
    from helpers import make_problem
    for k in KS:
        bu, bi, U, V = make_problem(3000, 700, k, density=0.25, seed=k, max_rating=10)
        outs = []
        for flags, dens in ((ycnr_als._lib.FLAG_NO_DUAL, 0.25), (0, 0.07)):  # whole rows through the row kernel; short rows through the dual classes
            if flags == 0:
                bu, bi, U, V = make_problem(3000, 700, k, density=dens, seed=k + 1, max_rating=10)
            dev = ycnr_als.AlsDevice(k, 3000, 700, flags=flags)
            dev.set_ratings("byUser", bu.rowPtr, bu.indx, bu.vals)
            dev.set_factors("byUser", U)
            dev.set_factors("byItem", V)
            info = dev.step("byUser")
            outs.append(dev.get_factors("byUser"))
            dev.destroy()
            print(tag, k, "flags", flags, "numericErrors", info.numericErrors, "rows", info.rows, "dualRows", info.dualRows, "ms", round(info.totalMs, 3), flush=True)
        np.save(f"/tmp/x6p_{tag}_{k}.npy", np.concatenate(outs))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
        sys.exit(0)
    for tag, env in (("p", {}), ("d", {"YCNR_NO_X6P": "1"})):
        subprocess.check_call([sys.executable, os.path.abspath(__file__), tag], env=dict(os.environ, **env))
    bad = 0
    for k in KS:
        a, b = np.load(f"/tmp/x6p_p_{k}.npy"), np.load(f"/tmp/x6p_d_{k}.npy")
        d = np.abs(a.astype(np.float64) - b).max(1) / np.maximum(np.abs(b).max(1), 1e-30)
        print(k, "rows differing", int((d > 0).sum()), "of", len(d), "max rel", float(np.nanmax(d)), "NaN rows", int(np.isnan(d).sum()),
              "first bad rows", np.flatnonzero(~(d <= 2e-5))[:12].tolist())
        bad += int((~(d <= (2e-5 if k % 16 == 4 else 0))).sum())
    sys.exit(1 if bad else 0)
