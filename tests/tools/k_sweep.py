"""One half-step of each side against the float64 numpy reference for a sweep of factorsCount values (float32), on a GPU box:
python tests/tools/k_sweep.py [k ...].  Guards the kernel selection: padded sizes (k % 4 != 0), the eight-block bf16x6
kernels (116 ... 128), the workgroup path (132 ... 256) and the any-k path with and without the bf16x6 Gramian."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))  # (helpers imports the oracle package: test infrastructure)
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "you-can-not-recommend_amd", "python"))
from helpers import EPS32, make_problem, numpy_step, row_rel_err  # noqa: E402
import ycnr_als as als  # noqa: E402

ks = [int(x) for x in sys.argv[1:]] or [2, 3, 5, 6, 9, 10, 13, 15, 17, 21, 30, 50, 66, 99, 101, 113, 117, 118, 121, 125, 127, 130, 131, 253, 255, 258, 261, 300, 577, 580]
bad = 0
for k in ks:
    users, items = 90, 260
    bu, bi, U, V = make_problem(users, items, k, density=0.45, seed=300 + k, dtype=np.float32, empty_rows=(3,))
    dev = als.AlsDevice(k, users, items)
    dev.set_ratings("byUser", bu.rowPtr, bu.indx, bu.vals)
    dev.set_ratings("byItem", bi.rowPtr, bi.indx, bi.vals)
    dev.set_factors("byUser", U)
    dev.set_factors("byItem", V)
    iu = dev.step("byUser")
    U1 = dev.get_factors("byUser")
    want, conds = numpy_step(0.05, k, bu, V, U)
    eu = row_rel_err(U1, want) / np.maximum(conds * EPS32, 1e-30)
    ii = dev.step("byItem")
    V1 = dev.get_factors("byItem")
    want_i, conds_i = numpy_step(0.05, k, bi, U1, V)
    ei = row_rel_err(V1, want_i) / np.maximum(conds_i * EPS32, 1e-30)
    ok = eu.max() < 16 and ei.max() < 16 and iu.numericErrors == 0 and ii.numericErrors == 0 and np.array_equal(U1[3], U[3])
    bad += not ok
    print("k %4d  user %.2f  item %.2f  (worst error in units of cond x eps32)  dual rows %d/%d  %s" % (k, eu.max(), ei.max(), iu.dualRows, ii.dualRows, "ok" if ok else "FAIL"))
    dev.destroy()
sys.exit(1 if bad else 0)
