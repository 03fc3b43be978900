"""Fixtures recorded from an EXECUTION of the reference's own hot-path source.

tests/golden/reference_harness.json was written by tests/golden/make_reference_fixtures.js, which loads
/root/reference/lib/emf/EmfWorker.js under Node with the un-vendored third-party modules stubbed and runs
EmfWorker.mw_calcTrainAlsPortion / mw_calcRmsePortion (and the EmfBase methods they call) verbatim on
synthetic portion buffers.  The BLAS / LAPACK arithmetic underneath is a plain-JS stand-in, so these
fixtures pin the reference's data flow -- buffer parsing, row offsets, lambda * n, the regularisation per step
type, in-place placement of the solved row, untouched rows, the 'completedPortion' fields, the RMSE sums --
not its BLAS (parity stays "unpinned" there, oracle/als_oracle.c).

CPU: the oracle against the fixtures.  GPU (-m gpu): the level-1 C ABI against the same fixtures.
"""
import json
import os

import numpy as np
import pytest

from helpers import row_rel_err

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_harness.json")
CASES = json.load(open(FIX))["cases"]


def arrays(c):
    dt = np.float64 if c["useDoublePrecision"] else np.float32
    k = c["k"]
    U0 = np.array(c["before"]["U"], dt).reshape(c["users"], k)
    V0 = np.array(c["before"]["V"], dt).reshape(c["items"], k)
    U1 = np.array(c["after"]["U"], dt).reshape(c["users"], k)
    V1 = np.array(c["after"]["V"], dt).reshape(c["items"], k)
    rows, indx, vals = np.array(c["alsRows"], np.int32), np.array(c["alsIndx"], np.int32), np.array(c["alsVals"], dt)
    return dt, k, U0, V0, U1, V1, rows, indx, vals


def check_case(c, calc_portion, rmse_portion):
    dt, k, U0, V0, U1, V1, rows, indx, vals = arrays(c)
    by_user = c["stepType"] == "byUser"
    fixed, solved, want = (V0, U0.copy(), U1) if by_user else (U0, V0.copy(), V1)
    n = calc_portion(c["lambda"], k, rows, indx, vals, fixed, solved)
    msg = c["completedPortion"]
    assert n == msg["ratingsInPortion"] == len(indx)
    assert msg["rowsRange"] == {"from": int(rows[1]), "cnt": int(rows[0])} and msg["portionNo"] == 3
    ids = rows[1::2]
    untouched = np.setdiff1d(np.arange(len(solved)), ids)
    assert np.array_equal(solved[untouched], want[untouched])          # only the portion's rows are written ...
    assert np.array_equal((V1 if by_user else U1), (V0 if by_user else U0))  # ... and the fixed side not at all
    err = row_rel_err(solved[ids], want[ids])
    assert err.max() <= (1e-11 if dt == np.float64 else 3e-5), err.max()
    if c["rmseCompletedPortion"]:
        m = c["rmseCompletedPortion"]
        out = rmse_portion(k, rows, indx, vals, want, V0, c["globalAvgShift"])   # the factors the reference's pass saw
        assert out[1] == m["rCnt"] == len(indx)
        assert abs(out[0] - m["rSumDiff2"]) <= (1e-10 if dt == np.float64 else 2e-4) * max(m["rSumDiff2"], 1.0)
        assert abs(out[2] - m["rSum"]) <= (1e-10 if dt == np.float64 else 2e-4) * max(abs(m["rSum"]), 1.0)
    return float(err.max())


@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_oracle_against_the_executed_reference(oracle, c):
    check_case(c, oracle.als_calc_portion, oracle.rmse_portion)


@pytest.mark.gpu
@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_hip_path_against_the_executed_reference(c):
    import ycnr_als
    L = ycnr_als._lib.load()
    assert L.ycnr_device_count() >= 1, L.ycnr_last_error()
    check_case(c, ycnr_als.als_calc_portion, ycnr_als.rmse_portion)
