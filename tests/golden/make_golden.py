"""Generates the fixtures in tests/golden/ from the CPU oracle (oracle/als_oracle.c).

The reference has no golden vectors and cannot run in this container, so
these are NOT reference outputs: they freeze the restated algorithm's results on fixed
seeded inputs, so that (a) the oracle cannot drift silently and (b) the HIP path is
checked against committed numbers, not only against a library built from the same tree.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "you-can-not-recommend_amd", "python"))

from oracle import oracle as orc  # noqa: E402
from helpers import make_problem  # noqa: E402
from ycnr_als.data import csr_to_portion  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def portion_fixture(name, users, items, k, dt, density, seed, lam=0.05, shift=0.125):
    bu, bi, U, V = make_problem(users, items, k, density=density, seed=seed, dtype=dt, empty_rows=(1,))
    rows, indx, vals = csr_to_portion(bu, 0, users)
    solved = U.copy()
    n = orc.als_calc_portion(lam, k, rows, indx, vals, V, solved)
    rm = orc.rmse_portion(k, rows, indx, vals, solved, V, shift)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), lam=lam, k=k, shift=shift, ratings=n, alsRows=rows,
                        alsIndx=indx, alsVals=vals, fixed=V, solved_in=U, solved_out=solved, rmse_out=rm)


if __name__ == "__main__":
    orc.build()
    portion_fixture("portion_f32_k20", 30, 40, 20, np.float32, 0.3, 101)
    portion_fixture("portion_f64_k20", 30, 40, 20, np.float64, 0.3, 102)
    portion_fixture("portion_f32_k100", 12, 150, 100, np.float32, 0.5, 103)
    portion_fixture("portion_f64_k7", 25, 16, 7, np.float64, 0.4, 104)
    print("written", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))
