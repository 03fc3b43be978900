"""Pins the CPU oracle (oracle/als_oracle.c).  The reference ships no tests or golden
vectors (package.json:30), so the oracle is checked against analytic known answers, an
independent float64 LAPACK solve, and the committed fixtures in tests/golden/."""
import os

import numpy as np
import pytest

from helpers import EPS32, make_problem, numpy_row_solve, numpy_step, row_rel_err
from ycnr_als.data import csr_to_portion

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def portion_of_rows(rows):
    """[(rowId, idx array, val array)] -> alsRows / alsIndx / alsVals"""
    alsRows = [len(rows)]
    indx, vals = [], []
    for rid, i, v in rows:
        alsRows += [rid, len(i)]
        indx += list(i)
        vals += list(v)
    return np.array(alsRows, np.int32), np.array(indx, np.int32), np.array(vals)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_single_rating_row(oracle, dt):
    # n = 1: x = r * y / (|y|^2 + lambda)
    k, lam = 8, 0.05
    rng = np.random.default_rng(1)
    V = rng.standard_normal((5, k)).astype(dt)
    U = np.zeros((3, k), dt)
    rows, indx, vals = portion_of_rows([(2, [3], [4.0])])
    n = oracle.als_calc_portion(lam, k, rows, indx, vals.astype(dt), V, U)
    assert n == 1
    y = V[3].astype(np.float64)
    want = 4.0 * y / (y @ y + lam)
    cond = (y @ y + lam) / lam  # A = y y^T + lam I
    tol = 8 * cond * EPS32 if dt == np.float32 else 1e-12
    assert row_rel_err(U[2:3], want[None])[0] <= tol
    assert not U[0].any() and not U[1].any()  # other rows untouched


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_orthonormal_rows(oracle, dt):
    # Y orthonormal (n = k): x = Y^T r / (1 + lambda * n)
    k, lam = 6, 0.05
    rng = np.random.default_rng(2)
    Q, _ = np.linalg.qr(rng.standard_normal((k, k)))
    V = Q.astype(dt)
    r = rng.integers(1, 6, k).astype(dt)
    U = np.zeros((1, k), dt)
    rows, indx, vals = portion_of_rows([(0, list(range(k)), r)])
    oracle.als_calc_portion(lam, k, rows, indx, vals.astype(dt), V, U)
    want = Q.T @ r.astype(np.float64) / (1 + lam * k)
    assert np.allclose(U[0], want, rtol=2e-5 if dt == np.float32 else 1e-12, atol=1e-6 if dt == np.float32 else 1e-13)


def test_large_lambda_limit(oracle):
    # lambda -> large: x -> Y^T r / (lambda n)
    k, lam = 5, 1e9
    rng = np.random.default_rng(3)
    V = rng.standard_normal((7, k))
    U = np.zeros((1, k))
    idx = [0, 2, 5]
    r = np.array([1.0, 5.0, 3.0])
    rows, indx, vals = portion_of_rows([(0, idx, r)])
    oracle.als_calc_portion(lam, k, rows, indx, vals, V, U)
    want = V[idx].T @ r / (lam * 3)
    assert np.allclose(U[0], want, rtol=1e-6)


def test_weighted_lambda_is_lambda_times_n(oracle):
    # the regulariser is lambda * n_row, not lambda (EmfWorker.js:233-235)
    k, lam = 4, 0.05
    rng = np.random.default_rng(4)
    V = rng.standard_normal((9, k))
    idx = np.array([1, 3, 4, 6, 8], np.int32)
    r = rng.integers(1, 6, 5).astype(np.float64)
    U = np.zeros((1, k))
    rows, indx, vals = portion_of_rows([(0, idx, r)])
    oracle.als_calc_portion(lam, k, rows, indx, vals, V, U)
    want, _ = numpy_row_solve(lam, k, idx, r, V)
    assert np.allclose(U[0], want, rtol=1e-10)
    Y = V[idx]
    plain = np.linalg.solve(Y.T @ Y + lam * np.eye(k), Y.T @ r)
    assert not np.allclose(U[0], plain, rtol=1e-3)


def test_permutation_invariance(oracle):
    k, lam = 6, 0.05
    rng = np.random.default_rng(5)
    V = rng.standard_normal((20, k))
    idx = rng.choice(20, 9, replace=False).astype(np.int32)
    r = rng.integers(1, 6, 9).astype(np.float64)
    U1, U2 = np.zeros((1, k)), np.zeros((1, k))
    oracle.als_calc_portion(lam, k, *portion_of_rows([(0, idx, r)]), V, U1)
    p = rng.permutation(9)
    oracle.als_calc_portion(lam, k, *portion_of_rows([(0, idx[p], r[p])]), V, U2)
    assert np.allclose(U1, U2, rtol=1e-11)


@pytest.mark.parametrize("k", [3, 20, 33])
def test_step_matches_numpy_float64(oracle, k):
    bu, bi, U, V = make_problem(40, 30, k, density=0.3, seed=k, dtype=np.float64, empty_rows=(7,))
    before = U.copy()
    n = oracle.als_step_csr(0.05, k, bu.rowPtr, bu.indx, bu.vals, V, U, threads=2)
    assert n == bu.nnz
    want, _ = numpy_step(0.05, k, bu, V, before)
    assert row_rel_err(U, want).max() < 1e-10
    assert np.array_equal(U[7], before[7])  # a row without ratings is never written


def test_float32_error_is_bounded_by_conditioning(oracle):
    k = 20
    bu, bi, U, V = make_problem(60, 50, k, density=0.4, seed=11, dtype=np.float32)
    U64 = U.astype(np.float64)
    oracle.als_step_csr(0.05, k, bu.rowPtr, bu.indx, bu.vals, V, U)
    want, conds = numpy_step(0.05, k, bu, V, U64)
    err = row_rel_err(U, want)
    assert (err <= 8 * conds * EPS32).all(), (err / (conds * EPS32)).max()


def test_portion_and_csr_entry_points_agree(oracle):
    k = 10
    bu, bi, U, V = make_problem(25, 18, k, density=0.35, seed=6, dtype=np.float32, empty_rows=(0, 24))
    U1, U2 = U.copy(), U.copy()
    oracle.als_step_csr(0.05, k, bu.rowPtr, bu.indx, bu.vals, V, U1)
    for lo, hi in ((0, 9), (9, 17), (17, 25)):
        rows, indx, vals = csr_to_portion(bu, lo, hi)
        oracle.als_calc_portion(0.05, k, rows, indx, vals, V, U2)
    assert np.array_equal(U1, U2)


def test_rmse_sums(oracle):
    k = 7
    bu, bi, U, V = make_problem(12, 9, k, density=0.5, seed=8, dtype=np.float64)
    out = oracle.rmse_csr(k, bu.rowPtr, bu.indx, bu.vals, U, V, 0.25)
    pred = np.array([U[u] @ V[i] + 0.25 for u in range(12) for i in bu.indx[bu.rowPtr[u]:bu.rowPtr[u + 1]]])
    assert np.isclose(out[0], ((bu.vals - pred) ** 2).sum(), rtol=1e-12)
    assert out[1] == bu.nnz
    assert np.isclose(out[2], pred.sum(), rtol=1e-12)
    rows, indx, vals = csr_to_portion(bu, 0, 12)
    out2 = oracle.rmse_portion(k, rows, indx, vals, U, V, 0.25)
    assert np.allclose(out, out2, rtol=1e-14)


def test_packer_reference_quirk(oracle):
    # EmfMaster.js:582-609: the row open at the last triplet is recorded before the triplet is
    # counted, so the last rating of a portion is dropped; compat=False keeps everything.
    r1 = np.array([1, 1, 1, 2, 2, 4], np.int32)
    c1 = np.array([2, 5, 7, 1, 2, 9], np.int32)
    v = np.array([5, 4, 3, 2, 1, 5], np.float32)
    rows, indx, vals = oracle.pack_portion(r1, c1, v, compat=True)
    assert rows[0] == 2  # the trailing single-rating row (id 3) is never recorded
    assert list(rows[1:5]) == [0, 3, 1, 2]
    assert list(indx[:6]) == [1, 4, 6, 0, 1, 8]
    # a one-rating portion yields one row with cols = 0 (the singular case the HIP path skips)
    rows1, _, _ = oracle.pack_portion(r1[:1], c1[:1], v[:1], compat=True)
    assert rows1[0] == 1 and list(rows1[1:3]) == [0, 0]
    rows, indx, vals = oracle.pack_portion(r1, c1, v, compat=False)
    assert rows[0] == 3 and list(rows[1:7]) == [0, 3, 1, 2, 3, 1]
    # a portion that ends inside a row loses that row's last rating in compat mode
    rows, _, _ = oracle.pack_portion(r1[:5], c1[:5], v[:5], compat=True)
    assert rows[0] == 2 and list(rows[1:5]) == [0, 3, 1, 1]


def test_split_to_portions_restatement_matches_host_mirror(oracle):
    from ycnr_als.emf import split_to_portions
    rng = np.random.default_rng(9)
    for trial in range(20):
        n = int(rng.integers(1, 300))
        cnt = rng.integers(0, 60, n).astype(np.int32)
        if trial % 3 == 0:
            cnt[rng.integers(0, n)] = 500
        if cnt.sum() == 0:
            cnt[0] = 1
        rip = int(rng.integers(5, 400))
        nt = int(rng.integers(1, 9))
        pct = [0, 11, 6][trial % 3]
        a = oracle.split_to_portions(cnt, int((cnt > 0).sum()), rip, nt, pct)
        b = split_to_portions(cnt, int((cnt > 0).sum()), rip, nt, pct)
        assert list(a[0]) == list(b[0]) and a[1] == b[1] and a[2] == b[2]
        # every row with ratings is covered exactly once, portions ascend
        ends = list(a[0])
        assert ends == sorted(ends) and ends[-1] == np.nonzero(cnt)[0][-1] + 1


@pytest.mark.parametrize("name", ["portion_f32_k20", "portion_f64_k20", "portion_f32_k100", "portion_f64_k7"])
def test_golden_fixtures(oracle, name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    solved = z["solved_in"].copy()
    n = oracle.als_calc_portion(float(z["lam"]), int(z["k"]), z["alsRows"], z["alsIndx"], z["alsVals"], z["fixed"], solved)
    assert n == int(z["ratings"])
    assert np.array_equal(solved, z["solved_out"])  # the oracle is deterministic: bit-exact
    out = oracle.rmse_portion(int(z["k"]), z["alsRows"], z["alsIndx"], z["alsVals"], z["solved_out"], z["fixed"],
                              float(z["shift"]))
    assert np.array_equal(out, z["rmse_out"])
    # and the fixture itself agrees with an independent float64 solve
    want, conds = numpy_step_portion(z)
    tol = 8 * conds * EPS32 if z["alsVals"].dtype == np.float32 else 1e-10
    ids = z["alsRows"][1::2]
    assert (row_rel_err(z["solved_out"][ids], want) <= tol).all()


def numpy_step_portion(z):
    k, lam = int(z["k"]), float(z["lam"])
    rows = z["alsRows"]
    off, out, conds = 0, [], []
    for r in range(rows[0]):
        cols = rows[2 + 2 * r]
        x, c = numpy_row_solve(lam, k, z["alsIndx"][off:off + cols], z["alsVals"][off:off + cols], z["fixed"])
        out.append(x)
        conds.append(c)
        off += cols
    return np.array(out), np.array(conds)
