"""Parity of the HIP path (through the C ABI) against the CPU oracle, on a real MI355X.

Tolerance policy (DESIGN.md "Parity"):
  * useDoublePrecision = true is the strict gate: per-row relative error of the factor
    vector <= 1e-5 against the float64 oracle (observed ~1e-13), RMSE sums <= 1e-9 relative;
  * float32 (the reference default) is checked against the float64 solve of the same
    normal equations with a conditioning-aware bound, err <= 8 * cond(A) * eps32 per row --
    the same bound the float32 oracle itself is held to in tests/test_oracle.py -- and the
    RMSE must agree with the oracle to 1e-6.
"""
import os

import numpy as np
import pytest

from helpers import EPS32, make_problem, numpy_step, row_rel_err
from ycnr_als.data import Csr, csr_to_portion, transpose_csr

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def als():
    import ycnr_als
    L = ycnr_als._lib.load()  # raises if libycnr_als.so is missing: no fallback
    assert L.ycnr_device_count() >= 1, L.ycnr_last_error()
    return ycnr_als


def check_rows(got, want64, conds, dt, rows=None):
    err = row_rel_err(got, want64)
    tol = np.full(len(err), 1e-5) if dt == np.float64 else np.maximum(8 * conds * EPS32, 1e-6)
    if rows is not None:
        err, tol = err[rows], tol[rows]
    bad = np.nonzero(err > tol)[0]
    assert len(bad) == 0, f"{len(bad)} rows out of tolerance, worst err/tol = {(err / tol).max():.3g}"
    return err


@pytest.mark.parametrize("name", ["portion_f32_k20", "portion_f64_k20", "portion_f32_k100", "portion_f64_k7"])
def test_golden_portion(als, oracle, name):
    """Level 1 entry point on the committed fixtures: in-place rows, untouched rows, count."""
    z = np.load(os.path.join(GOLD, name + ".npz"))
    k, lam = int(z["k"]), float(z["lam"])
    solved = z["solved_in"].copy()
    n = als.als_calc_portion(lam, k, z["alsRows"], z["alsIndx"], z["alsVals"], z["fixed"], solved)
    assert n == int(z["ratings"])
    ids = z["alsRows"][1::2]
    untouched = np.setdiff1d(np.arange(len(solved)), ids)
    assert np.array_equal(solved[untouched], z["solved_in"][untouched])
    dt = z["alsVals"].dtype.type
    rows, indx, vals = z["alsRows"], z["alsIndx"], z["alsVals"]
    csr = Csr(len(solved), len(z["fixed"]), np.zeros(len(solved) + 1, np.int64), indx, vals)
    cnt = np.zeros(len(solved), np.int64)
    cnt[ids] = rows[2::2]
    csr.rowPtr[1:] = np.cumsum(cnt)
    want, conds = numpy_step(lam, k, csr, z["fixed"], z["solved_in"])
    check_rows(solved, want, conds, dt, rows=ids)
    # against the frozen oracle output too (float32: same error class, not bitwise)
    err_vs_gold = row_rel_err(solved[ids], z["solved_out"][ids])
    assert (err_vs_gold <= (1e-5 if dt == np.float64 else np.maximum(16 * conds[ids] * EPS32, 1e-6))).all()
    out = als.rmse_portion(k, rows, indx, vals, z["solved_out"], z["fixed"], float(z["shift"]))
    assert out[1] == z["rmse_out"][1]
    assert np.allclose(out, z["rmse_out"], rtol=1e-6 if dt == np.float32 else 1e-12)


@pytest.mark.parametrize("k,dt", [(20, np.float32), (100, np.float32), (100, np.float64), (200, np.float32)])
def test_portion_ops_keep_state_and_pinned_matrix(als, k, dt):
    """Level 1 as the reference would drive it: many portions per half-step on one thread (stream and
    device buffers are reused), with and without the step's fixed matrix pinned on the device
    (ycnr_{s,d}AlsPinFixedFactors) -- bit-identical results either way, against float64, including
    k > 128 through the workgroup-per-row kernels; a changed matrix must be pinned again."""
    import ycnr_als
    users, items = 600, 900
    bu, _, U, V = make_problem(users, items, k, density=0.06, seed=5 + k, dtype=dt, empty_rows=(7,))
    want, conds = numpy_step(0.05, k, bu, V, U)
    cuts = [0, 50, 51, 200, 430, users]
    outs = []
    for pin in (False, True):
        solved = U.copy()
        if pin:
            ycnr_als.pin_fixed_factors(V, k)
        total = 0
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            rows, indx, vals = csr_to_portion(bu, lo, hi)
            total += als.als_calc_portion(0.05, k, rows, indx, vals, V, solved)
        assert total == bu.nnz
        check_rows(solved, want, conds, dt)
        assert np.array_equal(solved[7], U[7])
        outs.append(solved)
    assert np.array_equal(outs[0], outs[1])
    # the pinned copy is a snapshot: after the host changes the matrix it must pin again
    V2 = (V * dt(0.5)).astype(dt)
    ycnr_als.pin_fixed_factors(V2, k)
    rows, indx, vals = csr_to_portion(bu, 0, 50)
    s2 = U.copy()
    als.als_calc_portion(0.05, k, rows, indx, vals, V2, s2)
    w2, c2 = numpy_step(0.05, k, Csr(50, items, bu.rowPtr[:51].copy(), bu.indx[:bu.rowPtr[50]], bu.vals[:bu.rowPtr[50]]), V2, U[:50])
    check_rows(s2[:50], w2, c2, dt)
    ycnr_als.release_portion_state()


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("k", [1, 7, 16, 20, 33, 36, 64, 100, 116, 128])
def test_half_steps_all_k(als, oracle, k, dt):
    """Resident trainer, byUser then byItem, every MFMA tile count (k -> NB = ceil(k/16))."""
    users, items = 70, 50
    bu, bi, U, V = make_problem(users, items, k, density=0.3, seed=100 + k, dtype=dt, empty_rows=(3, 69))
    bi.vals = bi.vals.astype(dt)
    dev = als.AlsDevice(k, users, items, useDoublePrecision=(dt == np.float64))
    dev.set_ratings("byUser", bu.rowPtr, bu.indx, bu.vals)
    dev.set_ratings("byItem", bi.rowPtr, bi.indx, bi.vals)
    dev.set_factors("byUser", U)
    dev.set_factors("byItem", V)
    info = dev.step("byUser")
    assert info.ratings == bu.nnz and info.rows == users - 2 and info.numericErrors == 0
    U1 = dev.get_factors("byUser")
    want, conds = numpy_step(0.05, k, bu, V, U)
    check_rows(U1, want, conds, dt)
    assert np.array_equal(U1[3], U[3]) and np.array_equal(U1[69], U[69])  # rows without ratings untouched
    # Gauss-Seidel across half-steps: byItem must see the NEW user factors
    dev.step("byItem")
    V1 = dev.get_factors("byItem")
    want, conds = numpy_step(0.05, k, bi, U1, V)
    check_rows(V1, want, conds, dt)
    # oracle from the same start, both half-steps
    Uo, Vo = U.copy(), V.copy()
    oracle.als_step_csr(0.05, k, bu.rowPtr, bu.indx, bu.vals, Vo, Uo)
    oracle.als_step_csr(0.05, k, bi.rowPtr, bi.indx, bi.vals, Uo, Vo)
    if dt == np.float64:
        assert row_rel_err(U1, Uo).max() < 1e-9 and row_rel_err(V1, Vo).max() < 1e-9
    else:
        # float32 against the float32 oracle: two correct float32 solves of one row differ by what each is
        # allowed against float64 (check_rows' bound 8 cond(A) kappa_b eps32, floor 1e-6), so their distance is
        # held to twice that bound per row; the oracle's item step starts from ITS user factors, which differ
        # from U1 by the first bound, hence the factor 4 there
        _, cu = numpy_step(0.05, k, bu, V, U)
        _, ci = numpy_step(0.05, k, bi, U1, V)
        eu, ei = row_rel_err(U1, Uo), row_rel_err(V1, Vo)
        assert (eu <= 2 * np.maximum(8 * cu * EPS32, 1e-6)).all(), float((eu / np.maximum(8 * cu * EPS32, 1e-6)).max())
        assert (ei <= 4 * np.maximum(8 * ci * EPS32, 1e-6)).all(), float((ei / np.maximum(8 * ci * EPS32, 1e-6)).max())
    dev.destroy()


@pytest.mark.parametrize("k", [97, 99, 50, 37, 130])
def test_padded_sizes_without_regularisation(als, k):
    """float32 factorsCount % 4 != 0 runs on matrices padded to the next multiple of 4 (kPad): the padded columns take a UNIT
    diagonal, not lambda n -- with userFactReg = itemFactReg = 0, which ycnr_als_create accepts (the reference's defaults are
    options, lib/emf/EmfBase.js:65-70), lambda n there left a zero pivot and every primal row failed with YCNR_ERR_NUMERIC
    (round-4 review).  Rows have more ratings than factors, so the normal matrices are positive definite without lambda.
    k = 97: three padded columns inside the four edge columns that solve_edge4 eliminates first; k = 130: the workgroup path."""
    users, items = 340, 420
    bu, bi, U, V = make_problem(users, items, k, density=0.62, seed=300 + k, min_per_row=k + 30)
    dev = als.AlsDevice(k, users, items, userFactReg=0.0, itemFactReg=0.0)
    dev.set_ratings("byUser", bu.rowPtr, bu.indx, bu.vals)
    dev.set_ratings("byItem", bi.rowPtr, bi.indx, bi.vals)
    dev.set_factors("byUser", U)
    dev.set_factors("byItem", V)
    info = dev.step("byUser")
    assert info.numericErrors == 0 and info.rows == users
    U1 = dev.get_factors("byUser")
    want, conds = numpy_step(0.0, k, bu, V, U)
    check_rows(U1, want, conds, np.float32)
    info = dev.step("byItem")
    assert info.numericErrors == 0
    V1 = dev.get_factors("byItem")
    want, conds = numpy_step(0.0, k, bi, U1, V)
    check_rows(V1, want, conds, np.float32)
    dev.destroy()


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_split_rows_and_chunk_edges(als, dt):
    """Rows longer than a work unit are split over several waves and reduced in slab order;
    lengths straddle every boundary of the 4-rating MFMA step and of the chunk."""
    k, items = 20, 5000
    chunk = 64
    lens = [1, 2, 3, 4, 5, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1000, 2999, 0, 64 * 64, 64 * 64 + 1, 4999]
    lens = [min(n, items) for n in lens]
    rng = np.random.default_rng(5)
    rowPtr = np.zeros(len(lens) + 1, np.int64)
    rowPtr[1:] = np.cumsum(lens)
    indx = np.concatenate([np.sort(rng.choice(items, n, replace=False)) for n in lens]).astype(np.int32)
    vals = rng.integers(1, 11, rowPtr[-1]).astype(dt)
    bu = Csr(len(lens), items, rowPtr, indx, vals)
    U = (rng.standard_normal((len(lens), k)) / k).astype(dt)
    V = (rng.standard_normal((items, k)) / np.sqrt(k)).astype(dt)
    dev = als.AlsDevice(k, len(lens), items, useDoublePrecision=(dt == np.float64), chunkRatings=chunk)
    dev.set_ratings("byUser", bu.rowPtr, bu.indx, bu.vals)
    dev.set_factors("byUser", U)
    dev.set_factors("byItem", V)
    info = dev.step("byUser")
    assert info.splitRows == sum(1 for n in lens if n > chunk)
    got = dev.get_factors("byUser")
    want, conds = numpy_step(0.05, k, bu, V, U)
    check_rows(got, want, conds, dt)
    assert np.array_equal(got[16], U[16])
    # same problem, default chunk: fused where possible -> same results within tolerance,
    # and bitwise identical when repeated (fixed reduction order)
    dev2 = als.AlsDevice(k, len(lens), items, useDoublePrecision=(dt == np.float64), chunkRatings=chunk)
    dev2.set_ratings("byUser", bu.rowPtr, bu.indx, bu.vals)
    dev2.set_factors("byUser", U)
    dev2.set_factors("byItem", V)
    dev2.step("byUser")
    assert np.array_equal(dev2.get_factors("byUser"), got)
    dev.destroy()
    dev2.destroy()


@pytest.mark.parametrize("k", [20, 64, 100, 128, 129, 132, 150, 200, 201, 255, 256])
def test_every_row_length_class(als, k):
    """Rows of 1..130 ratings at one k: short rows take the dual (n x n) form, grouped by their
    number of 16-rating blocks, longer ones the primal form; all against float64, and the three
    float32 solver variants (dual+MFMA, MFMA only, LDS) against each other."""
    from ycnr_als import _lib
    items = 400
    lens = list(range(1, 131)) + [0, 16, 32, 48, 64, 80, 96, 97, 112, 300, 143, 144, 145, 159, 160, 161, 176, 177, 191, 192, 193]
    rng = np.random.default_rng(k)
    rowPtr = np.zeros(len(lens) + 1, np.int64)
    rowPtr[1:] = np.cumsum(lens)
    indx = np.concatenate([np.sort(rng.choice(items, n, replace=False)) for n in lens]).astype(np.int32)
    vals = rng.integers(1, 11, rowPtr[-1]).astype(np.float32)
    bu = Csr(len(lens), items, rowPtr, indx, vals)
    U = (rng.standard_normal((len(lens), k)) / k).astype(np.float32)
    V = (rng.standard_normal((items, k)) / np.sqrt(k)).astype(np.float32)
    want, conds = numpy_step(0.05, k, bu, V, U)
    got = {}
    variants = (("dual", 0), ("primal", _lib.FLAG_NO_DUAL), ("lds", _lib.FLAG_LDS_SOLVER),
                ("noedge", _lib.FLAG_NO_DUAL | _lib.FLAG_NO_VALU_EDGE))
    if k > 128:  # 4-wave kernels: with and without the dual form for short rows; no LDS-solver variant
        variants = (("dual", 0), ("primal", _lib.FLAG_NO_DUAL))
    for name, flags in variants:
        dev = als.AlsDevice(k, len(lens), items, flags=flags)
        dev.set_ratings("byUser", bu.rowPtr, bu.indx, bu.vals)
        dev.set_factors("byUser", U)
        dev.set_factors("byItem", V)
        info = dev.step("byUser")
        if name == "dual":
            nb = (k + 15) // 16
            dual_max = 16 * min(12 if k > 128 else 5, nb - 1)   # k > 128: up to 192 ratings stay one wave's n x n problem
            assert info.dualRows == sum(1 for n in lens if 0 < n <= dual_max)
        else:
            assert info.dualRows == 0
        got[name] = dev.get_factors("byUser")
        check_rows(got[name], want, conds, np.float32)
        assert np.array_equal(got[name][130], U[130])  # the empty row
        dev.destroy()
    for a, b in (("dual", "primal"), ("primal", "lds"), ("primal", "noedge")):
        if b not in got:
            continue
        err = row_rel_err(got[a], got[b])
        assert (err <= np.maximum(16 * conds * EPS32, 2e-6)).all()


@pytest.mark.parametrize("k,dt", [(129, np.float64), (200, np.float64), (256, np.float64), (320, np.float64), (512, np.float64),
                                  (257, np.float32), (320, np.float32), (333, np.float32), (512, np.float32), (600, np.float32)])
def test_any_factors_count(als, oracle, k, dt, monkeypatch):
    """The reference accepts any factorsCount in either precision (lib/emf/EmfBase.js:112, config/config-base.js:31;
    lib/emf/EmfWorker.js:200-246).  Beyond what registers and LDS hold -- float64 above 128 factors, float32 above
    256 -- the normal matrix lives in global memory (als_gen_kernels.hip.h): rows of every kind (empty, 1 rating,
    fewer ratings than factors -- the float32 dual classes --, more, split over several chunks and several
    BATCHES of the slab arena) against the float64 oracle, both half-steps, the level-1 portion op, bitwise
    repeatability.  float32 up to 512 factors and float64 up to 256 take the left-looking solve, the larger ones the
    right-looking one; k % 4 != 0 (float32) / k % 2 != 0 (float64) the element-wise panel loader."""
    monkeypatch.setenv("YCNR_GEN_ARENA_MB", "8")  # a k = 512 image is 0.5 / 1 MB: several batches
    users, items = 60, 400
    # (float32: rows of at most 192 ratings take the dual classes, whatever k; the longer ones the any-k kernels)
    lens = [0, 1, 2, 15, 16, 17, 40, 90, 33, 64, 5, 77, 176, 192, 193, 260, 333] + [int(x) for x in np.random.default_rng(k).integers(1, 91, users - 17)]
    rng = np.random.default_rng(5 * k)
    rowPtr = np.zeros(users + 1, np.int64)
    rowPtr[1:] = np.cumsum(lens)
    indx = np.concatenate([np.sort(rng.choice(items, n, replace=False)) for n in lens]).astype(np.int32)
    vals = rng.integers(1, 11, rowPtr[-1]).astype(dt)
    bu = Csr(users, items, rowPtr, indx, vals)
    # the same ratings by item
    order = np.lexsort((np.repeat(np.arange(users), lens), indx))
    bi = Csr(items, users, np.concatenate([[0], np.cumsum(np.bincount(indx, minlength=items))]).astype(np.int64),
             np.repeat(np.arange(users), lens)[order].astype(np.int32), vals[order])
    U = (rng.standard_normal((users, k)) / k).astype(dt)
    V = (rng.standard_normal((items, k)) / np.sqrt(k)).astype(dt)
    dbl = dt == np.float64
    dev = als.AlsDevice(k, users, items, useDoublePrecision=dbl, chunkRatings=32)  # rows above 32 ratings: several chunks
    dev.set_ratings("byUser", bu.rowPtr, bu.indx, bu.vals)
    dev.set_ratings("byItem", bi.rowPtr, bi.indx, bi.vals)
    dev.set_factors("byUser", U)
    dev.set_factors("byItem", V)
    iu = dev.step("byUser")
    assert iu.numericErrors == 0 and iu.rows == users - 1 and iu.ratings == bu.nnz
    U1 = dev.get_factors("byUser")
    assert np.array_equal(U1[0], U[0])  # the row without ratings
    Uo = U.astype(np.float64)
    oracle.als_step_csr(0.05, k, bu.rowPtr, bu.indx, bu.vals.astype(np.float64), V.astype(np.float64), Uo)
    want, conds = numpy_step(0.05, k, bu, V, U)
    assert row_rel_err(want, Uo).max() < 1e-9          # the two float64 references agree
    check_rows(U1, want, conds, dt)
    ii = dev.step("byItem")  # sees the new user factors
    assert ii.numericErrors == 0
    V1 = dev.get_factors("byItem")
    want_i, conds_i = numpy_step(0.05, k, bi, U1, V)
    check_rows(V1, want_i, conds_i, dt)
    # repeat from the same start: bitwise the same (fixed slab order, fixed tile ownership)
    dev.set_factors("byUser", U)
    dev.set_factors("byItem", V)
    dev.step("byUser")
    assert np.array_equal(dev.get_factors("byUser"), U1)
    dev.destroy()
    # level 1: the same rows through the portion op
    from ycnr_als.data import csr_to_portion
    rows, pi, pv = csr_to_portion(bu, 0, users)
    s2 = U.copy()
    assert als.als_calc_portion(0.05, k, rows, pi, pv, V, s2) == bu.nnz
    check_rows(s2, want, conds, dt)
    assert np.array_equal(s2[0], U[0])


@pytest.mark.parametrize("k", [36, 37])  # 37 in float32: the kernels work on copies padded to 40 columns (pad in front of the graph, unpad inside it)
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_graph_replay_does_not_change_results(als, dt, k):
    """Uploads below 2 M ratings replay their half-step as a captured hipGraph from the third call on (first call:
    launch by launch, second: capture + launch) with the chunk Gramians -> reduce, the row kernel and the dual
    classes as parallel branches.  Three iterations with and without (YCNR_FLAG_NO_GRAPH) must agree bit for bit,
    numeric errors must still be reported through a replayed graph, and a new upload must drop the graph."""
    from ycnr_als import YcnrError, _lib
    users, items = 3000, 1200   # 360 K ratings: between the graph path's bounds (256 K ... 2 M per side)
    bu, bi, U, V = make_problem(users, items, k, density=0.1, seed=77, dtype=dt, empty_rows=(4,))
    assert 256 * 1024 <= bu.nnz < 2 * 1024 * 1024
    res = {}
    for name, flags in (("graph", 0), ("launches", _lib.FLAG_NO_GRAPH)):
        dev = als.AlsDevice(k, users, items, useDoublePrecision=(dt == np.float64), flags=flags, chunkRatings=32)  # split rows too
        dev.set_ratings("byUser", bu.rowPtr, bu.indx, bu.vals)
        dev.set_ratings("byItem", bi.rowPtr, bi.indx, bi.vals)
        dev.set_factors("byUser", U)
        dev.set_factors("byItem", V)
        infos = []
        for _ in range(3):
            infos.append((dev.step("byUser"), dev.step("byItem")))
        res[name] = (dev.get_factors("byUser"), dev.get_factors("byItem"))
        assert all(i.numericErrors == 0 and i.totalMs > 0 for pair in infos for i in pair)
        if name == "graph":
            # NaN in the fixed matrix through the replayed graph: reported, not fatal; then a clean step again
            bad = res[name][1].copy()
            bad[7, 3] = np.nan
            dev.set_factors("byItem", bad)
            with pytest.raises(YcnrError) as e:
                dev.step("byUser")
            assert e.value.code == _lib.ERR_NUMERIC
            dev.set_factors("byItem", res[name][1])
            assert dev.step("byUser").numericErrors == 0
            # a new upload drops the captured graph: half as many ratings, results of a fresh handle
            half = Csr(users, items, np.minimum(bu.rowPtr, bu.rowPtr[users // 2]), bu.indx, bu.vals)
            dev.set_ratings("byUser", half.rowPtr, half.indx, half.vals)
            dev.set_factors("byUser", U)
            i2 = dev.step("byUser")
            assert i2.ratings == int(bu.rowPtr[users // 2])
            assert np.array_equal(dev.get_factors("byUser")[users // 2:], U[users // 2:])
        dev.destroy()
    assert np.array_equal(res["graph"][0], res["launches"][0]) and np.array_equal(res["graph"][1], res["launches"][1])


@pytest.mark.parametrize("users,items,density", [(3000, 1200, 0.1), (400, 300, 0.2)])
def test_iteration_in_flight_equals_two_awaited_steps(als, users, items, density):
    """AlsDevice.iteration() -- both half-steps of EmfLord.alsTrainIter (lib/emf/EmfLord.js:954-958) enqueued before the host
    waits, ycnr_als_step_info_of for the infos -- against step('byUser'); step('byItem'): the same factors bit for bit over
    four iterations (first launch by launch, then captured, then replayed as graphs at the larger size; plain launches at the
    smaller), the infos of the two sides kept apart, numeric errors of a half-step in flight reported by the sync."""
    from ycnr_als import YcnrError, _lib
    k = 36
    bu, bi, U, V = make_problem(users, items, k, density=density, seed=91, dtype=np.float32, empty_rows=(4,))
    res = {}
    for name in ("awaited", "in flight"):
        dev = als.AlsDevice(k, users, items, chunkRatings=32)
        dev.set_ratings("byUser", bu.rowPtr, bu.indx, bu.vals)
        dev.set_ratings("byItem", bi.rowPtr, bi.indx, bi.vals)
        dev.set_factors("byUser", U)
        dev.set_factors("byItem", V)
        for _ in range(4):
            iu, ii = (dev.step("byUser"), dev.step("byItem")) if name == "awaited" else dev.iteration()
            assert (iu.side, ii.side) == (_lib.BY_USER, _lib.BY_ITEM)
            assert iu.ratings == bu.nnz and ii.ratings == bi.nnz and iu.rows == users - 1
            assert iu.numericErrors == 0 and ii.numericErrors == 0 and iu.totalMs > 0 and ii.totalMs > 0
        res[name] = (dev.get_factors("byUser"), dev.get_factors("byItem"))
        if name == "in flight":
            bad = res[name][1].copy()
            bad[7, 3] = np.nan
            dev.set_factors("byItem", bad)
            with pytest.raises(YcnrError) as e:
                dev.iteration()
            assert e.value.code == _lib.ERR_NUMERIC
            dev.set_factors("byUser", U)
            dev.set_factors("byItem", V)
            iu, ii = dev.iteration()
            assert iu.numericErrors == 0 and ii.numericErrors == 0
        dev.destroy()
    assert np.array_equal(res["awaited"][0], res["in flight"][0]) and np.array_equal(res["awaited"][1], res["in flight"][1])


@pytest.mark.parametrize("k", [4, 8, 12, 16, 24, 32, 48, 52, 80, 96, 108, 112, 116, 120, 124, 128])
def test_lds_dma_gramian_every_block_count(als, k):
    """The LDS-DMA staged bf16x6 Gramian (k % 4 == 0, k <= 128; eight blocks at one wave per SIMD) at every block count, with the
    right-hand side in the padded column (k % 16 != 0) and on the VALU (k % 16 == 0): whole rows of
    1..4 steps including exact multiples of 32 ratings (fused row kernel; the dual form is switched
    off so that short rows take it too) and split rows (chunk kernel + reduce)."""
    from ycnr_als import _lib
    items = 600
    lens = [1, 2, 31, 32, 33, 63, 64, 65, 95, 96, 97, 127, 128, 129, 200, 256, 300, 511, 520]
    rng = np.random.default_rng(1000 + k)
    rowPtr = np.zeros(len(lens) + 1, np.int64)
    rowPtr[1:] = np.cumsum(lens)
    indx = np.concatenate([np.sort(rng.choice(items, n, replace=False)) for n in lens]).astype(np.int32)
    vals = rng.integers(1, 11, rowPtr[-1]).astype(np.float32)
    bu = Csr(len(lens), items, rowPtr, indx, vals)
    U = np.zeros((len(lens), k), np.float32)
    V = (rng.standard_normal((items, k)) / np.sqrt(k)).astype(np.float32)
    want, conds = numpy_step(0.05, k, bu, V, U)
    res = {}
    for name, flags, chunk in (("fused", _lib.FLAG_NO_DUAL, 0), ("chunks", _lib.FLAG_NO_DUAL, 96),
                               ("f32mfma", _lib.FLAG_NO_DUAL | _lib.FLAG_NO_BF16X6, 0)):
        dev = als.AlsDevice(k, len(lens), items, flags=flags, chunkRatings=chunk)
        dev.set_ratings("byUser", bu.rowPtr, bu.indx, bu.vals)
        dev.set_factors("byUser", U)
        dev.set_factors("byItem", V)
        info = dev.step("byUser")
        assert info.dualRows == 0 and info.numericErrors == 0
        if name == "chunks":
            assert info.splitRows == sum(1 for n in lens if n > 96)
        res[name] = dev.get_factors("byUser")
        check_rows(res[name], want, conds, np.float32)
        dev.destroy()
    for other in ("chunks", "f32mfma"):
        err = row_rel_err(res["fused"], res[other])
        assert (err <= np.maximum(16 * conds * EPS32, 2e-6)).all()


def test_band_major_chunks(als, monkeypatch):
    """Split rows cut at common column-id boundaries (bands of the fixed matrix) instead of every
    chunkRatings ratings: same results as the plain chunks within float32 rounding, both against
    float64.  Bands of 1 MB so that a small matrix has many; rows that miss whole bands, rows with
    a few ratings in a band (merged into the next one) and rows heavy enough to hit the
    64-chunks-per-row limit."""
    from ycnr_als import _lib
    k, users, items = 64, 12000, 400     # fixed matrix 12000 x 64 x 4 B = 3 MB = 3 bands
    monkeypatch.setenv("YCNR_BAND_MB", "1")
    rng = np.random.default_rng(77)
    lens = [11000, 9000, 5000, 3000, 2500, 1500, 1100, 1030, 700, 64, 3] + [int(x) for x in rng.integers(1, 2500, items - 11)]
    rows = []
    for i, n in enumerate(lens):
        if i == 3:      # only the last band
            cols = 9000 + rng.choice(3000, n, replace=False)
        elif i == 4:    # 5 ratings in the first band, the rest in the second
            cols = np.concatenate([rng.choice(4000, 5, replace=False), 4200 + rng.choice(3000, n - 5, replace=False)])
        else:
            cols = rng.choice(users, n, replace=False)
        rows.append(np.sort(cols))
    rowPtr = np.zeros(items + 1, np.int64)
    rowPtr[1:] = np.cumsum(lens)
    indx = np.concatenate(rows).astype(np.int32)
    vals = rng.integers(1, 11, rowPtr[-1]).astype(np.float32)
    bi = Csr(items, users, rowPtr, indx, vals)
    U = (rng.standard_normal((users, k)) / np.sqrt(k)).astype(np.float32)
    V = np.zeros((items, k), np.float32)
    want, conds = numpy_step(0.05, k, bi, U, V)
    got = {}
    for name, flags, chunk in (("bands", 0, 0), ("plain", _lib.FLAG_NO_BANDS, 0), ("bands256", 0, 256)):
        dev = als.AlsDevice(k, users, items, flags=flags, chunkRatings=chunk)
        dev.set_ratings("byItem", bi.rowPtr, bi.indx, bi.vals)
        dev.set_factors("byUser", U)
        dev.set_factors("byItem", V)
        info = dev.step("byItem")
        got[name] = (dev.get_factors("byItem"), info.units)
        check_rows(got[name][0], want, conds, np.float32)
        dev.destroy()
    assert got["bands"][1] != got["plain"][1]   # the schedules differ ...
    err = row_rel_err(got["bands"][0], got["plain"][0])
    assert (err <= np.maximum(16 * conds * EPS32, 2e-6)).all()   # ... the results do not


@pytest.mark.parametrize("k", [64, 100])
def test_many_rows_at_full_occupancy(als, k):
    """Tens of thousands of rows in one launch, so that every CU holds its full set of workgroups
    while rows start and finish: the LDS-DMA staged Gramian (k % 16 == 0: right-hand side on the
    VALU; otherwise out of the padded Gramian) showed errors only under that load during
    development, none with a handful of workgroups.  Every row against float64."""
    users, items = 30000, 20000
    rng = np.random.default_rng(100 + k)
    lens = np.clip(rng.lognormal(np.log(110), 0.8, users).astype(np.int64), 1, 3000)
    rowPtr = np.zeros(users + 1, np.int64)
    np.cumsum(lens, out=rowPtr[1:])
    # distinct columns per row: a random start and stride walk of the item range
    start = rng.integers(0, items, users)
    indx = np.empty(rowPtr[-1], np.int32)
    for u in range(users):
        indx[rowPtr[u]:rowPtr[u + 1]] = np.sort((start[u] + 7 * np.arange(lens[u])) % items)
    vals = rng.integers(1, 11, rowPtr[-1]).astype(np.float32)
    bu = Csr(users, items, rowPtr, indx, vals)
    U = np.zeros((users, k), np.float32)
    V = (rng.standard_normal((items, k)) / np.sqrt(k)).astype(np.float32)
    dev = als.AlsDevice(k, users, items)
    dev.set_ratings("byUser", bu.rowPtr, bu.indx, bu.vals)
    dev.set_factors("byUser", U)
    dev.set_factors("byItem", V)
    info = dev.step("byUser")
    got = dev.get_factors("byUser")
    dev.destroy()
    assert info.fusedRows > 10000 and info.dualRows > 2000
    want, conds = numpy_step(0.05, k, bu, V, U)
    check_rows(got, want, conds, np.float32)


def test_side_streams_do_not_change_results(als):
    """The dual-form kernels run on the handle's two side streams next to the row kernel
    (YCNR_FLAG_NO_OVERLAP: everything in stream order).  Same kernels, same rows: three
    alternating half-steps on one handle must give the same bits either way, and the step info
    must say which way it went."""
    from ycnr_als import _lib
    users, items, k = 20000, 3000, 64
    rng = np.random.default_rng(77)
    lens = np.clip(rng.lognormal(np.log(40), 1.0, users).astype(np.int64), 1, 2000)
    rowPtr = np.zeros(users + 1, np.int64)
    np.cumsum(lens, out=rowPtr[1:])
    start = rng.integers(0, items, users)
    indx = np.empty(rowPtr[-1], np.int32)
    for u in range(users):
        indx[rowPtr[u]:rowPtr[u + 1]] = np.sort((start[u] + 1 + np.arange(lens[u])) % items)
    vals = rng.integers(1, 11, rowPtr[-1]).astype(np.float32)
    bu = Csr(users, items, rowPtr, indx, vals)
    import torch
    bi = transpose_csr(Csr(users, items, torch.from_numpy(rowPtr), torch.from_numpy(indx), torch.from_numpy(vals))).numpy()
    U0 = (rng.standard_normal((users, k)) / np.sqrt(k)).astype(np.float32)
    V0 = (rng.standard_normal((items, k)) / np.sqrt(k)).astype(np.float32)
    got = {}
    for name, flags in (("side", 0), ("serial", _lib.FLAG_NO_OVERLAP)):
        dev = als.AlsDevice(k, users, items, flags=flags)
        dev.set_ratings("byUser", bu.rowPtr, bu.indx, bu.vals)
        dev.set_ratings("byItem", bi.rowPtr, bi.indx, bi.vals)
        dev.set_factors("byUser", U0)
        dev.set_factors("byItem", V0)
        info = dev.step("byUser")
        assert info.dualRows > 5000 and info.fusedRows > info.dualRows
        assert info.dualOverlapped == (1 if flags == 0 else 0)
        dev.step("byItem")
        dev.step("byUser")
        got[name] = (dev.get_factors("byUser"), dev.get_factors("byItem"))
        dev.destroy()
    assert np.array_equal(got["side"][0], got["serial"][0])
    assert np.array_equal(got["side"][1], got["serial"][1])


@pytest.mark.parametrize("k", [20, 100, 128])
def test_float64_solvers_agree(als, k):
    """useDoublePrecision: the register-resident f64 MFMA Cholesky (default) and the plain LDS
    Cholesky (YCNR_FLAG_LDS_SOLVER) against float64 numpy and each other, rows of every length
    class including split rows."""
    from ycnr_als import _lib
    items = 500
    lens = [1, 2, 5, 15, 16, 17, 33, 64, 99, 100, 101, 130, 300, 0, 450]
    rng = np.random.default_rng(k + 1)
    rowPtr = np.zeros(len(lens) + 1, np.int64)
    rowPtr[1:] = np.cumsum(lens)
    indx = np.concatenate([np.sort(rng.choice(items, n, replace=False)) for n in lens]).astype(np.int32)
    vals = rng.integers(1, 11, rowPtr[-1]).astype(np.float64)
    bu = Csr(len(lens), items, rowPtr, indx, vals)
    U = rng.standard_normal((len(lens), k)) / k
    V = rng.standard_normal((items, k)) / np.sqrt(k)
    want, conds = numpy_step(0.05, k, bu, V, U)
    got = {}
    for name, flags in (("mfma", 0), ("lds", _lib.FLAG_LDS_SOLVER)):
        dev = als.AlsDevice(k, len(lens), items, useDoublePrecision=True, flags=flags, chunkRatings=128)
        dev.set_ratings("byUser", bu.rowPtr, bu.indx, bu.vals)
        dev.set_factors("byUser", U)
        dev.set_factors("byItem", V)
        info = dev.step("byUser")
        assert info.splitRows == 3 and info.numericErrors == 0
        got[name] = dev.get_factors("byUser")
        assert row_rel_err(got[name], want).max() < 1e-9
        assert np.array_equal(got[name][13], U[13])
        dev.destroy()
    assert row_rel_err(got["mfma"], got["lds"]).max() < 1e-9


def test_big_k_split_rows_and_batches(als):
    """k = 256 through the workgroup-per-row kernels: rows cut into many chunks (slabs + reduce), a row
    longer than 64 chunks, whole rows; checked against float64."""
    k, users, items = 256, 40, 3000
    rng = np.random.default_rng(3)
    lens = [0, 1, 50, 96, 97, 130, 500, 1024, 1025, 2500, 2999] + list(rng.integers(100, 400, 29))
    rowPtr = np.zeros(users + 1, np.int64)
    rowPtr[1:] = np.cumsum(lens)
    indx = np.concatenate([np.sort(rng.choice(items, n, replace=False)) for n in lens]).astype(np.int32)
    vals = rng.integers(1, 11, rowPtr[-1]).astype(np.float32)
    bu = Csr(users, items, rowPtr, indx, vals)
    U = (rng.standard_normal((users, k)) / k).astype(np.float32)
    V = (rng.standard_normal((items, k)) / np.sqrt(k)).astype(np.float32)
    want, conds = numpy_step(0.05, k, bu, V, U)
    for chunk in (32, 0):   # 32: 2999 ratings -> 64 chunks of 48; 0: the library's own (whole rows here)
        dev = als.AlsDevice(k, users, items, chunkRatings=chunk)
        dev.set_ratings("byUser", bu.rowPtr, bu.indx, bu.vals)
        dev.set_factors("byUser", U)
        dev.set_factors("byItem", V)
        info = dev.step("byUser")
        assert info.numericErrors == 0 and info.rows == users - 1
        assert (info.splitRows > 20) == (chunk == 32)
        got = dev.get_factors("byUser")
        check_rows(got, want, conds, np.float32)
        assert np.array_equal(got[0], U[0])
        dev.destroy()


@pytest.mark.parametrize("k", [132, 144, 160, 176, 192, 208, 224, 240, 256])
def test_workgroup_path_every_block_count(als, k):
    """128 < k <= 256 (als_wg_kernels.hip.h) at every tile count NB = 9..16: more rows than CUs (the
    persistent loop and its prefetch of the next row), rows of 1..70 steps of 32 ratings including exact
    multiples, split rows, both against float64; a second run must reproduce the first bit for bit."""
    from ycnr_als import _lib
    items = 2600
    rng = np.random.default_rng(k)
    lens = [161, 162, 191, 192, 193, 223, 224, 225, 256, 320, 321, 1000, 2240, 2241] + list(rng.integers(161, 420, 700)) + [0, 3, 160]
    rowPtr = np.zeros(len(lens) + 1, np.int64)
    rowPtr[1:] = np.cumsum(lens)
    indx = np.concatenate([np.sort(rng.choice(items, n, replace=False)) for n in lens]).astype(np.int32)
    vals = rng.integers(1, 11, rowPtr[-1]).astype(np.float32)
    bu = Csr(len(lens), items, rowPtr, indx, vals)
    U = (rng.standard_normal((len(lens), k)) / k).astype(np.float32)
    V = (rng.standard_normal((items, k)) / np.sqrt(k)).astype(np.float32)
    rows = np.r_[0:40, len(lens) - 60:len(lens)]      # float64 check on a subset (numpy at k = 256 is slow)
    sub = Csr(len(rows), items, np.concatenate([[0], np.cumsum(np.asarray(lens)[rows])]),
              np.concatenate([indx[rowPtr[r]:rowPtr[r + 1]] for r in rows]), np.concatenate([vals[rowPtr[r]:rowPtr[r + 1]] for r in rows]))
    want, conds = numpy_step(0.05, k, sub, V, U[rows])
    res = []
    for chunk, flags in ((0, _lib.FLAG_NO_DUAL), (0, _lib.FLAG_NO_DUAL), (96, 0)):
        dev = als.AlsDevice(k, len(lens), items, chunkRatings=chunk, flags=flags)
        dev.set_ratings("byUser", bu.rowPtr, bu.indx, bu.vals)
        dev.set_factors("byUser", U)
        dev.set_factors("byItem", V)
        info = dev.step("byUser")
        assert info.numericErrors == 0
        got = dev.get_factors("byUser")
        check_rows(got[rows], want, conds, np.float32)
        assert np.array_equal(got[len(lens) - 3], U[len(lens) - 3])
        res.append(got)
        dev.destroy()
    assert np.array_equal(res[0], res[1])
    err = row_rel_err(res[0], res[2])
    assert err.max() < 2e-4, err.max()   # whole rows vs chunks + reduce: same error class


def test_sharded_rows_equal_unsharded(als):
    """Solving a row shard [rowBegin, rowEnd) gives bit-identical rows to the full solve:
    the per-row schedule does not depend on which GPU owns the row (determinism)."""
    k, users, items = 24, 90, 60
    bu, bi, U, V = make_problem(users, items, k, density=0.4, seed=77, dtype=np.float32)
    full = als.AlsDevice(k, users, items)
    full.set_ratings("byUser", bu.rowPtr, bu.indx, bu.vals)
    full.set_factors("byUser", U)
    full.set_factors("byItem", V)
    full.step("byUser")
    ref = full.get_factors("byUser")
    got = U.copy()
    for lo, hi in ((0, 31), (31, 32), (32, 90)):
        d = als.AlsDevice(k, users, items)
        d.set_ratings("byUser", bu.rowPtr, bu.indx, bu.vals, lo, hi)
        d.set_factors("byUser", U)
        d.set_factors("byItem", V)
        info = d.step("byUser")
        assert info.ratings == bu.rowPtr[hi] - bu.rowPtr[lo]
        part = d.get_factors("byUser")
        assert np.array_equal(part[:lo], U[:lo]) and np.array_equal(part[hi:], U[hi:])
        got[lo:hi] = part[lo:hi]
        d.destroy()
    assert np.array_equal(got, ref)
    full.destroy()


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_rmse_portions(als, oracle, dt):
    k, users, items = 20, 64, 40
    bu, bi, U, V = make_problem(users, items, k, density=0.3, seed=9, dtype=dt, empty_rows=(5,))
    dev = als.AlsDevice(k, users, items, useDoublePrecision=(dt == np.float64))
    dev.set_factors("byUser", U)
    dev.set_factors("byItem", V)
    dev.set_rmse_ratings("rmseTest", bu.rowPtr, bu.indx, bu.vals)
    ends = np.array([10, 11, 40, 64], np.int64)
    got = dev.rmse("rmseTest", 0.5, ends)
    lo = 0
    for p, e in enumerate(ends):
        want = oracle.rmse_csr(k, bu.rowPtr, bu.indx, bu.vals, U, V, 0.5, lo, int(e))
        assert got[p, 1] == want[1]
        assert np.allclose(got[p], want, rtol=1e-6 if dt == np.float32 else 1e-12)
        lo = int(e)
    tot = dev.rmse("rmseTest", 0.5)
    assert np.allclose(tot[0], got.sum(0), rtol=1e-12)
    rm_gpu = np.sqrt(tot[0, 0] / tot[0, 1])
    o = oracle.rmse_csr(k, bu.rowPtr, bu.indx, bu.vals, U, V, 0.5)
    assert abs(rm_gpu - np.sqrt(o[0] / o[1])) <= 1e-6
    dev.destroy()


def test_errors_are_reported_not_fatal(als):
    from ycnr_als import YcnrError, _lib
    with pytest.raises(YcnrError) as e:
        als.AlsDevice(4097, 10, 10)  # any factorsCount up to 4096 (test_any_factors_count), in either precision
    assert e.value.code == _lib.ERR_UNSUPPORTED
    dev = als.AlsDevice(8, 4, 5)
    with pytest.raises(YcnrError) as e:
        dev.step("byUser")  # no ratings yet
    assert e.value.code == _lib.ERR_STATE
    rowPtr = np.array([0, 1, 2, 2, 3], np.int64)
    with pytest.raises(YcnrError) as e:  # column id 5 >= 5 items
        dev.set_ratings("byUser", rowPtr, np.array([0, 5, 1], np.int32), np.ones(3, np.float32))
    assert e.value.code == _lib.ERR_INVALID
    with pytest.raises(TypeError, match="invalid type!"):  # cpp_utils.js:12
        dev.set_ratings("byUser", rowPtr, np.array([0, 4, 1], np.int32), np.ones(3, np.float64))
    # NaN input -> the normal matrix is not positive definite -> ERR_NUMERIC, handle stays usable
    dev.set_ratings("byUser", rowPtr, np.array([0, 4, 1], np.int32), np.ones(3, np.float32))
    V = np.ones((5, 8), np.float32)
    V[4, 2] = np.nan
    dev.set_factors("byItem", V)
    with pytest.raises(YcnrError) as e:
        dev.step("byUser")
    assert e.value.code == _lib.ERR_NUMERIC
    dev.set_factors("byItem", np.ones((5, 8), np.float32))
    assert dev.step("byUser").numericErrors == 0
    dev.destroy()


@pytest.mark.parametrize("double", [True, False], ids=["f64", "f32"])
def test_full_iterations_ml100k_shape(als, oracle, double):
    """C1 shape (943 x 1682, ~100k ratings, k = 20): 3 full ALS iterations through the host
    mirror vs the oracle run the same way; factors and RMSE must track each other.  float64 is the
    strict gate; float32 (the reference's default, lib/emf/EmfBase.js:112) is held to the flat 1e-5 /
    1e-6 of the north star on this small, well-conditioned problem."""
    import torch
    from ycnr_als.data import select_csr, split_to_sets, synth_ratings
    from ycnr_als.emf import Dataset, EmfLord
    from helpers import OracleBackend
    by_user, _ = synth_ratings(943, 1682, 100_000, max_rating=5, seed=20260001, degree_sigma=0.9, zipf_a=0.8)
    t = split_to_sets(by_user, (85, 10, 5), seed=3)
    from ycnr_als.data import transpose_csr
    tr = select_csr(by_user, t <= 2)
    ds = Dataset(tr, transpose_csr(tr), select_csr(by_user, t == 2), select_csr(by_user, t == 3),
                 float(by_user.vals.double().mean()))
    res = {}
    for name, factory in (("hip", None), ("oracle", lambda o, u, i, d: OracleBackend(o, u, i, d))):
        lord = EmfLord(options={"factorsCount": 20, "trainIters": 3, "useDoublePrecision": double,
                                "dataDir": "/tmp/ycnr_test_" + name + ("64" if double else "32")}, backend_factory=factory)
        lord.prepareToTrain(ds, seed=7)
        hist = lord.train()
        res[name] = (hist, lord.backend.get_factors(0), lord.backend.get_factors(1), lord.getCalcInfo())
        lord.destroy()
    (h1, U1, V1, c1), (h2, U2, V2, c2) = res["hip"], res["oracle"]
    # (float32: three iterations of two float32 implementations; measured 3e-6, the gate leaves room for the
    # conditioning of a few rows -- see test_gpu_configs.py for what ten iterations do to the flat bound)
    tol = 1e-5 if double else 5e-5
    assert row_rel_err(U1, U2).max() < tol and row_rel_err(V1, V2).max() < tol, (row_rel_err(U1, U2).max(), row_rel_err(V1, V2).max())
    for a, b in zip(h1, h2):
        for key in ("rmseValidate", "rmseTest", "rmseTestShifted"):
            assert abs(a[key] - b[key]) <= 1e-6, (key, a[key], b[key])
    # (the shift is totalRatingsAvg - predAvg of the LAST portion, a float32 mean over a few hundred predictions)
    assert abs(c1["globalAvgShift"] - c2["globalAvgShift"]) <= (1e-6 if double else 1e-5)
    assert h1[-1]["rmseValidate"] < h1[0]["rmseValidate"]  # it learns


@pytest.mark.parametrize("double", [True, False])
def test_warm_start_and_extend_on_the_hip_path_against_the_oracle(als, oracle, tmp_path, double):
    """N4 on the product backend: train one iteration, save, warm-start from the result files with 7 more
    users and 4 more items (prepareSharedFactors over _loadSharedFactorsForTrain, lib/emf/EmfMaster.js:347-358,
    EmfManager.js:405-457; new rows drawn by initSharedFactorsRandom(oldUsersCnt, oldItemsCnt), EmfBase.js:457-513), train
    again -- the same sequence on HipBackend and on the CPU oracle.  Old rows must come back from the files bit for bit, the
    new rows must be the seeded draw, factors and calc_info.json must agree between the two backends (float64 1e-9;
    float32 within the conditioning bound of every row's solve, as in test_half_steps_all_k)."""
    import copy
    import json
    from ycnr_als.data import Csr, init_factors
    from ycnr_als.emf import Dataset, EmfLord
    from helpers import OracleBackend, make_problem
    dt = np.float64 if double else np.float32
    users, items, k = 300, 180, 36
    bu, bi, U, V = make_problem(users, items, k, density=0.12, seed=21, dtype=dt, min_per_row=2)
    sel = np.arange(bu.nnz)

    def pick(c, mask):
        rows_of = np.repeat(np.arange(c.rows), c.counts())
        rp = np.zeros(c.rows + 1, np.int64)
        rp[1:] = np.cumsum(np.bincount(rows_of[mask], minlength=c.rows))
        return Csr(c.rows, c.cols, rp, c.indx[mask].copy(), c.vals[mask].copy())
    ds = Dataset(bu, bi, pick(bu, sel % 4 == 0), pick(bu, sel % 9 == 1), float(bu.vals.mean()))
    # the grown data set: 7 more users and 4 more items, the new users with ratings (also of the new items)
    rng = np.random.default_rng(5)
    U2n, I2n = users + 7, items + 4
    extra_cols = [np.sort(rng.choice(I2n, 9, replace=False)).astype(np.int32) for _ in range(7)]
    rp2 = np.concatenate([bu.rowPtr, bu.rowPtr[-1] + 9 * np.arange(1, 8)]).astype(np.int64)
    big_u = Csr(U2n, I2n, rp2, np.concatenate([bu.indx] + extra_cols), np.concatenate([bu.vals, rng.integers(1, 6, 63).astype(dt)]))
    import torch
    from ycnr_als.data import transpose_csr
    big_i = transpose_csr(big_u.to("cpu")).numpy()
    sel2 = np.arange(big_u.nnz)
    big = Dataset(big_u, big_i, pick(big_u, sel2 % 4 == 0), pick(big_u, sel2 % 9 == 1), float(big_u.vals.mean()))
    res = {}
    for name, factory in (("hip", None), ("oracle", lambda o, u, i, d: OracleBackend(o, u, i, d))):
        d = tmp_path / name
        opts = {"factorsCount": k, "trainIters": 1, "useDoublePrecision": double, "dataDir": str(d), "dbType": "ml",
                "ratingsInPortionForRmse": 400}
        first = EmfLord(options=opts, backend_factory=factory)
        first.prepareToTrain(ds, U.copy(), V.copy())
        first.train()
        U1, V1 = first.backend.get_factors(0), first.backend.get_factors(1)
        first.destroy()
        ready = d / "ml_factors_ready"
        assert np.array_equal(np.fromfile(ready / "user_factors", dt).reshape(-1, k), U1)   # the files ARE the matrices
        ext = EmfLord(options=dict(opts, warmStart=True), backend_factory=factory)
        ext.prepareToTrain(big, seed=5)
        assert (ext.recreated, ext.extended) == (False, True) and ext.calcCnt == 1
        Ue, Ve = ext.backend.get_factors(0), ext.backend.get_factors(1)
        assert np.array_equal(Ue[:users], U1) and np.array_equal(Ve[:items], V1)                         # old rows bit-kept
        assert np.array_equal(Ue[users:], init_factors(U2n, k, 10, dt)[users:]) and np.abs(Ue[users:]).max() > 0  # seeded N(0, 1/k)
        assert np.array_equal(Ve[items:], init_factors(I2n, k, 11, dt)[items:])
        hist = ext.train()
        info = json.loads((ready / "calc_info.json").read_text())
        res[name] = (U1, V1, ext.backend.get_factors(0), ext.backend.get_factors(1), info, hist)
        ext.destroy()
    h, o = res["hip"], res["oracle"]
    # float64: 1e-9 everywhere.  float32: the first half-step (same fixed matrix on both backends) within the conditioning bound
    # of every row's solve -- two float32 implementations, each within 8 cond(A) kappa_b eps32 of float64 --, what follows within a
    # flat 1e-4 (every later half-step starts from matrices that already differ by those bounds)
    _, conds = numpy_step(0.05, k, bu, V.astype(np.float64), np.zeros((users, k)))
    for idx, what in enumerate(("U after the first train", "V after the first train", "U after the warm-started train",
                                "V after the warm-started train")):
        e = row_rel_err(h[idx], o[idx])
        if double:
            assert e.max() <= 1e-9, (what, float(e.max()))
        elif idx == 0:
            assert (e <= np.maximum(16 * conds * EPS32, 2e-6)).all(), (what, float(e.max()))
        else:
            assert e.max() <= 1e-4, (what, float(e.max()))
    for key in ("alg", "algOptions", "useDoublePrecision", "factorsCount", "dataSetDistr", "totalUsersCount", "totalItemsCount", "dbType",
                "calcCnt", "globalBias"):
        assert h[4][key] == o[4][key], key
    assert h[4]["calcCnt"] == 2 and h[4]["totalUsersCount"] == U2n and h[4]["totalItemsCount"] == I2n
    assert list(h[4].keys()) == list(o[4].keys())
    assert abs(h[4]["globalAvgShift"] - o[4]["globalAvgShift"]) <= (1e-9 if double else 1e-5)
    for a, b in zip(h[5], o[5]):
        for key in ("rmseValidate", "rmseTest", "rmseTestShifted"):
            # (float32, shifted: the shift is totalRatingsAvg - predAvg of the LAST portion, a float32 mean over a few hundred
            # predictions of matrices that differ by the bounds above after two trains)
            tol = 1e-9 if double else (5e-6 if key == "rmseTestShifted" else 1e-6)
            assert abs(a[key] - b[key]) <= tol, (key, a[key], b[key])


def test_bench_starts_its_own_ranks(tmp_path):
    """`python3 bench.py --gpus 2 ...` exactly as the driver's scaling job calls it -- plain python, no launcher: bench.py must
    start its ranks itself, as the reference's master forks its workers (lib/emf/EmfMaster.js:44-98), relay rank 0's line
    and exit 0.  Two ranks share cuda:0 over the device-to-device `ipc` transport (RCCL refuses duplicate devices)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--same-device", "--backend", "gloo",
                        "--transport", "ipc", "--workload", "ml1m", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2
    ex = out["exchange"]
    assert ex["path"] == "libycnr_als:ipc" and ex["replicas_consistent"] is True
    assert ex["comm"]["world"] == 2 and ex["comm"]["transport"] == "ipc" and ex["comm"]["rccl_ranks"] is None
    assert out["rmse_ms"] > 0


def test_bench_falls_back_to_the_other_transport(tmp_path):
    """A transport that cannot be set up on every rank must not cost the run: two ranks on ONE device ask for `rccl` (which refuses
    duplicate devices), the host moves on to `ipc`, the line says over which path it ran and why; with --strict-transport the
    same command exits non-zero (a measurement that must not run over another path than the one it names)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--same-device", "--backend", "gloo", "--transport", "rccl",
           "--workload", "ml100k", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    ex = out["exchange"]
    assert ex["path"] == "libycnr_als:ipc" and ex["requested_transport"] == "rccl" and ex["replicas_consistent"] is True
    assert ex["fallback"] and ex["fallback"][0].startswith("rccl:")
    r = subprocess.run(cmd + ["--strict-transport"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode != 0 and "unavailable" in (r.stderr + r.stdout)


@pytest.mark.parametrize("world,transport,sharding", [(2, "shm", "rows"), (4, "shm", "rows"), (3, "ipc", "rows"), (2, "ipc", "bands"), (4, "ipc", "bands")])
def test_ranks_on_one_gpu_equal_one_rank(tmp_path, world, transport, sharding):
    """The sharded HIP path end to end: 2 / 4 gloo ranks sharing cuda:0 (functional stand-in for
    that many GPUs over RCCL; the GPU boxes allow at most 6 processes on the card, this one included) must reproduce the single-process factors bit for bit -- shard
    ranges, per-shard CSR upload, the chunked pipelined exchange, the padded all-gather and
    the RMSE all-reduce included."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (sharding = 'bands': EmfLord's itemStepSharding -- users in 8 bands, the items' Gramians reduce-scattered; one rank runs the
    # same bands, so the order of every sum is the same)
    common = ["--workload", "ml100k", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--item-sharding", sharding]
    one = str(tmp_path / "one.npz")
    two = str(tmp_path / "two.npz")
    subprocess.check_call([sys.executable, os.path.join(root, "bench.py")] + common + ["--dump-factors", one], timeout=600)
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                           "--master-addr", "127.0.0.1", "--master-port", str(29541 + world), os.path.join(root, "bench.py"),
                           "--gpus", str(world), "--backend", "gloo", "--same-device", "--transport", transport, "--dump-factors", two] + common,
                          timeout=900)
    a, b = np.load(one), np.load(two)
    assert np.array_equal(a["U"], b["U"]) and np.array_equal(a["V"], b["V"])
    assert abs(float(a["rmse"]) - float(b["rmse"])) < 1e-12

@pytest.mark.gpu
def test_four_rows_per_wave_kernel_against_the_one_row_kernel(tmp_path):
    """Rows of at most 16 ratings take als_dual_quad_kernel (four rows per wave, Gaussian elimination in 16-lane
    groups); YCNR_NO_DUAL_QUAD=1 sends them through als_dual_solve_kernel<1> (block Cholesky) instead.  The
    toggle is read once per process, so each build of the result is a child process of bench.py; the two must
    agree to float32 rounding on every row, and the half of the users that have at most 16 ratings must
    actually differ somewhere (else the toggle did nothing)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = [sys.executable, os.path.join(root, "bench.py"), "--workload", "ml100k", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    quad, one = str(tmp_path / "quad.npz"), str(tmp_path / "one.npz")
    env = dict(os.environ)
    env.pop("YCNR_NO_DUAL_QUAD", None)
    subprocess.check_call(common + ["--dump-factors", quad], timeout=600, env=env, stdout=subprocess.DEVNULL)
    env["YCNR_NO_DUAL_QUAD"] = "1"
    subprocess.check_call(common + ["--dump-factors", one], timeout=600, env=env, stdout=subprocess.DEVNULL)
    a, b = np.load(quad), np.load(one)
    for name in ("U", "V"):
        d = np.abs(a[name].astype(np.float64) - b[name]).max(axis=1) / np.maximum(np.abs(b[name]).max(axis=1), 1e-30)
        assert d.max() <= 2e-5, (name, float(d.max()))
    assert not np.array_equal(a["U"], b["U"])


@pytest.mark.gpu
def test_launch_order_of_the_rows_does_not_change_a_bit(tmp_path):
    """Large launches take their primal rows in a fixed pseudo-random order instead of longest-first (build_part: waves that share
    a SIMD then differ in length and phase); YCNR_ROW_ORDER=0 keeps the sorted order.  Every row is solved by its own wave, so
    both orders must give every factor bit for bit (200 K x 20 K shape at k = 100: ~100 K primal rows on the user side).
    The toggle is read once per process: each result is a child process of bench.py."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = [sys.executable, os.path.join(root, "bench.py"), "--workload", "c3", "--factors", "100", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    rnd, srt = str(tmp_path / "rnd.npz"), str(tmp_path / "srt.npz")
    env = dict(os.environ)
    env.pop("YCNR_ROW_ORDER", None)
    subprocess.check_call(common + ["--dump-factors", rnd], timeout=600, env=env, stdout=subprocess.DEVNULL)
    env["YCNR_ROW_ORDER"] = "0"
    subprocess.check_call(common + ["--dump-factors", srt], timeout=600, env=env, stdout=subprocess.DEVNULL)
    a, b = np.load(rnd), np.load(srt)
    assert np.array_equal(a["U"], b["U"]) and np.array_equal(a["V"], b["V"])
    assert float(a["rmse"]) == float(b["rmse"])


@pytest.mark.gpu
def test_gram32_kernels_against_the_shipped_gramian(tmp_path):
    """k = 256: YCNR_G32=1 sends the Gramians of whole rows and of the chunks of heavy rows through
    als_gram32_kernels.hip.h (32 x 32 x 16 bf16 MFMAs, one wave per SIMD) instead of WgGram (16 x 16 x 32, two waves per
    SIMD).  Same exact products, a different order of the sums over a row's ratings: the two must agree to float32
    rounding on every row and differ somewhere (else the toggle did nothing).  The toggle is read once per process, so
    each result comes from a child process of bench.py."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # round 4: the kernel left the product build (it measured equal to WgGram); it lives in csrc/devtest and is only in
    # a library built with `make EXTRA=-DYCNR_WITH_G32 OUT=devtest/ablibs/libycnr_g32.so`
    lib = os.path.join(root, "you-can-not-recommend_amd", "csrc", "devtest", "ablibs", "libycnr_g32.so")
    if not os.path.exists(lib):
        pytest.skip("no -DYCNR_WITH_G32 build of the library (devtest kernel, not part of the product)")
    common = [sys.executable, os.path.join(root, "bench.py"), "--workload", "c3k256", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    wg, g32 = str(tmp_path / "wg.npz"), str(tmp_path / "g32.npz")
    env = dict(os.environ)
    env["YCNR_ALS_LIB"] = lib
    env.pop("YCNR_G32", None)
    subprocess.check_call(common + ["--dump-factors", wg], timeout=900, env=env, stdout=subprocess.DEVNULL)
    env["YCNR_G32"] = "1"
    subprocess.check_call(common + ["--dump-factors", g32], timeout=900, env=env, stdout=subprocess.DEVNULL)
    a, b = np.load(wg), np.load(g32)
    for name in ("U", "V"):
        d = np.abs(a[name].astype(np.float64) - b[name]).max(axis=1) / np.maximum(np.abs(b[name]).max(axis=1), 1e-30)
        assert d.max() <= 2e-5, (name, float(d.max()))
    assert not np.array_equal(a["V"], b["V"])
    assert abs(float(a["rmse"]) - float(b["rmse"])) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("k", [4, 20, 36, 64, 100, 112, 128, 256])
def test_signed_fractional_ratings(als, k):
    """Ratings that are neither positive nor exact in bf16 (standard normal x 3, a few exact zeros and tiny values): the
    right-hand side then needs all three bf16 planes of the ratings, and a negative rating must not leak a sign bit into
    a padded lane (at k = 16 m + 4 the planes of the last block share one MFMA operand: a "-0" in a padded lane of the high
    plane flipped the sign of another column's middle term -- found by the full-size linearity test, reproduced here at
    test size).  Whole rows in primal form, rows cut into chunks, and the dual classes, against float64."""
    from ycnr_als import _lib
    items = 600
    lens = [1, 5, 16, 17, 40, 80, 81, 97, 130, 177, 200, 333, 500, 0, 64] * 3
    rng = np.random.default_rng(1000 + k)
    rowPtr = np.zeros(len(lens) + 1, np.int64)
    rowPtr[1:] = np.cumsum(lens)
    indx = np.concatenate([np.sort(rng.choice(items, n, replace=False)) for n in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
    vals = (rng.standard_normal(rowPtr[-1]) * 3.0).astype(np.float32)
    vals[::37] = 0.0
    vals[5::41] *= 1e-6
    bu = Csr(len(lens), items, rowPtr, indx, vals)
    U = np.zeros((len(lens), k), np.float32)
    V = (rng.standard_normal((items, k)) / np.sqrt(k)).astype(np.float32)
    want, conds = numpy_step(0.05, k, bu, V, U)
    for name, flags, chunk in (("default", 0, 0), ("primal", _lib.FLAG_NO_DUAL, 0), ("chunks", _lib.FLAG_NO_DUAL, 64)):
        dev = als.AlsDevice(k, len(lens), items, flags=flags, chunkRatings=chunk)
        dev.set_ratings("byUser", bu.rowPtr, bu.indx, bu.vals)
        dev.set_factors("byUser", U)
        dev.set_factors("byItem", V)
        info = dev.step("byUser")
        assert info.numericErrors == 0, name
        if name == "chunks":
            assert info.splitRows > 0
        check_rows(dev.get_factors("byUser"), want, conds, np.float32)
        dev.destroy()


@pytest.mark.gpu
def test_planes_row_kernel_against_the_float_gather():
    """Round 4: the user half-step's fused row kernel gathers bf16 PLANES of the item matrix, split once per half-step
    (GramX6P), instead of floats that every wave splits again (GramX6D).  Same truncations, same products in the same order:
    every row bit-identical for k = 4 ... 108, except the 5 x 5 corner of the packed last block at k = 16 m + 4 (another
    order of nine float32 sums: rows within 2e-5).  tests/tools/x6p_check.py runs both forms in child processes (the toggle
    YCNR_NO_X6P is read once per process) over whole rows and over rows short enough for the dual classes."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "x6p_check.py")], capture_output=True, text=True, timeout=900)
    print(r.stdout[-1500:])
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]

