"""Host logic of the product (options, partitioners, train loop, result files) exercised on
CPU with the oracle standing in for the GPU backend."""
import json
import os

import numpy as np
import pytest

from helpers import OracleBackend, make_problem
from ycnr_als.data import Csr, csr_to_portion, init_factors
from ycnr_als.emf import Dataset, EmfLord, deepmerge, default_options, shard_ranges, split_to_portions


def oracle_factory(o, u, i, d):
    return OracleBackend(o, u, i, d)


def small_dataset(seed=0, users=50, items=35, dt=np.float32):
    bu, bi, U, V = make_problem(users, items, 8, density=0.3, seed=seed, dtype=dt, min_per_row=2)
    # validate = every 4th rating, test = a disjoint every 9th
    sel = np.arange(bu.nnz)
    val = pick(bu, sel % 4 == 0)
    tst = pick(bu, sel % 9 == 1)
    return Dataset(bu, bi, val, tst, float(bu.vals.mean())), U, V


def pick(c, mask):
    rows_of = np.repeat(np.arange(c.rows), c.counts())
    cnt = np.bincount(rows_of[mask], minlength=c.rows)
    rp = np.zeros(c.rows + 1, np.int64)
    rp[1:] = np.cumsum(cnt)
    return Csr(c.rows, c.cols, rp, c.indx[mask].copy(), c.vals[mask].copy())


def test_option_names_follow_the_reference():
    o = default_options()
    for key in ("factorsCount", "trainIters", "useDoublePrecision", "dataSetDistr", "ratingsInPortionForAls",
                "ratingsInPortionForRmse", "numThreadsForTrain", "alg", "dbType"):
        assert key in o
    assert o["als"]["userFactReg"] == 0.05 and o["als"]["itemFactReg"] == 0.05  # EmfBase.js:67-69
    assert o["factorsCount"] == 100 and o["trainIters"] == 10 and o["useDoublePrecision"] is False
    m = deepmerge(o, {"als": {"userFactReg": 0.1}}, {"factorsCount": 20})
    assert m["als"] == {"userFactReg": 0.1, "itemFactReg": 0.05, "initFirstFactorAsAvgRating": False}
    assert m["factorsCount"] == 20 and o["factorsCount"] == 100


def test_shard_ranges_are_contiguous_and_balanced():
    rng = np.random.default_rng(0)
    cnt = rng.integers(0, 100, 1000)
    for w in (1, 2, 3, 8):
        b = shard_ranges(cnt, w)
        assert b[0] == 0 and b[-1] == 1000 and (np.diff(b) >= 0).all() and len(b) == w + 1
        loads = [cnt[b[i]:b[i + 1]].sum() for i in range(w)]
        assert max(loads) - min(loads) <= 2 * cnt.max()
    assert list(shard_ranges(np.array([5, 0, 0, 0]), 2)) == [0, 1, 4]  # one heavy row, rest empty
    assert list(shard_ranges(np.zeros(4, np.int64), 2)) [0] == 0


def test_portion_views_of_a_csr():
    bu, _, _, _ = make_problem(20, 15, 4, density=0.3, seed=3, empty_rows=(0, 7, 19))
    rows, indx, vals = csr_to_portion(bu, 0, 20)
    assert rows[0] == 17 and rows[2::2].sum() == bu.nnz
    assert 0 not in rows[1::2] and 7 not in rows[1::2]
    assert np.array_equal(indx, bu.indx) and np.array_equal(vals, bu.vals)


def test_train_loop_and_result_files(tmp_path):
    ds, U, V = small_dataset()
    lord = EmfLord(options={"factorsCount": 8, "trainIters": 4, "dataDir": str(tmp_path), "dbType": "ml",
                            "ratingsInPortionForRmse": 40}, backend_factory=oracle_factory)
    with pytest.raises(RuntimeError, match="Not ready to train"):
        lord.train()
    lord.prepareToTrain(ds, U, V)
    assert lord.status == "ready"
    hist = lord.train()
    assert len(hist) == 4 and [h["iter"] for h in hist] == [0, 1, 2, 3]
    assert hist[-1]["rmseValidate"] < hist[0]["rmseValidate"]
    for h in hist:
        assert set(h) >= {"rmseValidate", "rmseTest", "rmseTestShifted", "globalAvgShift"}
    ready = tmp_path / "ml_factors_ready"
    assert sorted(os.listdir(ready)) == ["calc_info.json", "item_factors", "user_factors"]
    assert not (tmp_path / "ml_factors_tmp").exists()  # renamed, EmfManager.js:557
    assert os.path.getsize(ready / "user_factors") == 50 * 8 * 4  # headerless raw dump, EmfBase.js:384
    assert os.path.getsize(ready / "item_factors") == 35 * 8 * 4
    text = (ready / "calc_info.json").read_text()
    ci = json.loads(text)
    assert text == json.dumps(ci, indent=2)  # JSON.stringify(calcInfo, null, 2), EmfManager.js:549
    assert list(ci) == ["alg", "algOptions", "useDoublePrecision", "factorsCount", "dataSetDistr", "totalUsersCount",
                        "totalItemsCount", "dbType", "calcDate", "calcCnt", "globalAvgShift", "globalBias"]
    assert ci["calcCnt"] == 1 and ci["totalUsersCount"] == 50 and ci["totalItemsCount"] == 35
    got = np.fromfile(ready / "user_factors", np.float32).reshape(50, 8)
    assert np.array_equal(got, lord.backend.get_factors(0))
    # warm start is possible only with matching alg / dbType / factorsCount / precision (EmfManager.js:179-191)
    assert lord.loadCalcResults() is not None
    other = EmfLord(options={"factorsCount": 9, "dataDir": str(tmp_path), "dbType": "ml"}, backend_factory=oracle_factory)
    assert other.loadCalcResults() is None


def test_warm_start_extends_factors_for_new_users_and_items(tmp_path):
    """N4: prepareSharedFactors over _loadSharedFactorsForTrain (EmfMaster.js:347-358,
    EmfManager.js:405-457): a compatible previous result is kept, rows are drawn only for users and
    items added since (initSharedFactorsRandom(oldUsersCnt, oldItemsCnt), EmfBase.js:457-513);
    incompatible or larger previous results are discarded."""
    ds, U, V = small_dataset()
    opts = {"factorsCount": 8, "trainIters": 1, "dataDir": str(tmp_path), "dbType": "ml", "ratingsInPortionForRmse": 40}
    first = EmfLord(options=opts, backend_factory=oracle_factory)
    first.prepareToTrain(ds, U, V)
    first.train()
    U1, V1 = first.backend.get_factors(0).copy(), first.backend.get_factors(1).copy()
    # same sizes: plain warm start
    again = EmfLord(options=dict(opts, warmStart=True), backend_factory=oracle_factory)
    again.prepareToTrain(ds)
    assert (again.recreated, again.extended) == (False, False) and again.calcCnt == 1
    assert np.array_equal(again.backend.get_factors(0), U1) and np.array_equal(again.backend.get_factors(1), V1)
    # 7 more users and 4 more items
    import copy
    big = copy.copy(ds)
    big.totalUsersCount, big.totalItemsCount = ds.totalUsersCount + 7, ds.totalItemsCount + 4

    def grow(a, rows, cols):
        from ycnr_als.data import Csr
        rp = np.concatenate([_np(a.rowPtr), np.full(rows - a.rows, _np(a.rowPtr)[-1])])
        return Csr(rows, cols, rp, a.indx, a.vals)
    _np = lambda x: x if isinstance(x, np.ndarray) else x.cpu().numpy()
    big.train_by_user = grow(ds.train_by_user, big.totalUsersCount, big.totalItemsCount)
    big.train_by_item = grow(ds.train_by_item, big.totalItemsCount, big.totalUsersCount)
    big.validate = grow(ds.validate, big.totalUsersCount, big.totalItemsCount)
    big.test = grow(ds.test, big.totalUsersCount, big.totalItemsCount)
    ext = EmfLord(options=dict(opts, warmStart=True), backend_factory=oracle_factory)
    ext.prepareToTrain(big, seed=5)
    assert (ext.recreated, ext.extended) == (False, True)
    Ue, Ve = ext.backend.get_factors(0), ext.backend.get_factors(1)
    assert Ue.shape == (57, 8) and Ve.shape == (39, 8)
    assert np.array_equal(Ue[:50], U1) and np.array_equal(Ve[:35], V1)          # old rows kept
    assert np.array_equal(Ue[50:], init_factors(57, 8, 10)[50:]) and np.abs(Ue[50:]).max() > 0   # new rows N(0, 1/k), seeded
    ext.train()
    assert json.loads((tmp_path / "ml_factors_ready" / "calc_info.json").read_text())["calcCnt"] == 2
    # a previous result with MORE users than now cannot be reused: everything is drawn again
    shrunk = EmfLord(options=dict(opts, warmStart=True), backend_factory=oracle_factory)
    shrunk.prepareToTrain(ds, seed=5)
    assert (shrunk.recreated, shrunk.extended) == (True, False)
    assert np.array_equal(shrunk.backend.get_factors(0), init_factors(50, 8, 10))


def test_reference_packer_quirk_is_an_opt_in_flag(oracle, tmp_path):
    """N2: csr_to_portion / Dataset reproduce the end-of-data branch of the reference's packer
    (lib/emf/EmfMaster.js:594-603) when asked to: the row tables equal the oracle's literal restatement (PackPortion, compat),
    and a train with options.dropLastRatingPerPortion consumes exactly the ratings those tables name."""
    from ycnr_als.data import Csr, drop_last_rating_per_portion
    rng = np.random.default_rng(12)
    for trial in range(60):
        rows_n, cols_n = int(rng.integers(1, 12)), 9
        cnt = rng.integers(0, 5, rows_n)
        if trial % 5 == 0:
            cnt[-1] = 1                       # a trailing row of one rating: never recorded
        if trial % 7 == 0:
            cnt[:] = 0
            cnt[rng.integers(0, rows_n)] = 1  # a portion of ONE rating: recorded with cols = 0
        rp = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        indx = np.concatenate([np.sort(rng.choice(cols_n, c, replace=False)) for c in cnt] + [np.zeros(0, np.int64)]).astype(np.int32)
        vals = rng.integers(1, 6, len(indx)).astype(np.float32)
        c = Csr(rows_n, cols_n, rp, indx, vals)
        r1 = np.repeat(np.arange(rows_n), cnt).astype(np.int32) + 1
        for compat in (True, False):
            want_rows, want_indx, want_vals = oracle.pack_portion(r1, indx + 1, vals, compat=compat)
            rows, gi, gv = csr_to_portion(c, 0, rows_n, dropLastRatingPerPortion=compat)
            n = int(want_rows[0])
            assert rows[0] == n and np.array_equal(rows[1:], want_rows[1:1 + 2 * n]), (trial, compat)
            assert np.array_equal(gi, want_indx[:len(gi)]) and np.array_equal(gv, want_vals[:len(gv)])
    # whole data set: every portion of a pass without its last rating
    ds, U, V = small_dataset()
    tu = ds.train_by_user
    cu = np.diff(tu.rowPtr)
    ends, _, _ = split_to_portions(cu, int((cu > 0).sum()), 40, 2)
    dropped = drop_last_rating_per_portion(tu, ends)
    b = 0
    for e in ends:
        rows, _, _ = csr_to_portion(tu, b, int(e), dropLastRatingPerPortion=True)
        got = np.diff(dropped.rowPtr)[b:int(e)]
        want = np.zeros(int(e) - b, np.int64)
        want[rows[1::2] - b] = rows[2::2]
        assert np.array_equal(got, want)
        b = int(e)
    assert tu.nnz - dropped.nnz == len(ends)
    # the option: a train on the full data with the flag == a train without it on the data the flag leaves
    opts = {"factorsCount": 8, "trainIters": 1, "dataDir": str(tmp_path), "dbType": "ml", "ratingsInPortionForRmse": 40,
            "ratingsInPortionForAls": {"byUser": 40, "byItem": 40}, "numThreadsForTrain": {"als": 2}}
    a = EmfLord(options=dict(opts, dropLastRatingPerPortion=True), backend_factory=oracle_factory)
    a.prepareToTrain(ds, U, V)
    ha = a.train()
    cut = a.dataset
    assert cut.train_by_user.nnz < ds.train_by_user.nnz and cut.train_by_item.nnz == ds.train_by_item.nnz - len(
        split_to_portions(np.diff(ds.train_by_item.rowPtr), int((np.diff(ds.train_by_item.rowPtr) > 0).sum()), 40, 2)[0])
    b2 = EmfLord(options=opts, backend_factory=oracle_factory)
    b2.prepareToTrain(cut, U, V)
    # (portions of the RMSE passes are cut from the stats of the data a Lord is given; keep those of the full data)
    b2.portionsRowIdTo = a.portionsRowIdTo
    hb = b2.train()
    assert np.array_equal(a.backend.get_factors(0), b2.backend.get_factors(0))
    assert np.array_equal(a.backend.get_factors(1), b2.backend.get_factors(1))
    assert ha[-1]["rmseValidate"] == hb[-1]["rmseValidate"]
    plain = EmfLord(options=opts, backend_factory=oracle_factory)
    plain.prepareToTrain(ds, U, V)
    plain.train()
    assert not np.array_equal(plain.backend.get_factors(0), a.backend.get_factors(0))  # the flag is not a no-op


def test_item_sharding_by_user_bands_is_refused_where_it_cannot_run():
    """options.itemStepSharding = 'bands' (DESIGN.md 6): 8 user bands, so 1 / 2 / 4 / 8 ranks only, and only on a backend that has
    the banded half-step (the HIP library) -- never a silent fall back to row shards."""
    ds, U, V = small_dataset()
    lord = EmfLord(options={"factorsCount": 8, "itemStepSharding": "bands"}, backend_factory=oracle_factory)
    with pytest.raises(RuntimeError, match="set_ratings_banded"):
        lord.prepareToTrain(ds, U, V)

    class ThreeRanks:
        def get_rank(self):
            return 0

        def get_world_size(self):
            return 3
    with pytest.raises(ValueError, match="1, 2, 4 or 8"):
        EmfLord(options={"factorsCount": 8, "itemStepSharding": "bands"}, backend_factory=oracle_factory, dist=ThreeRanks()).prepareToTrain(ds, U, V)
    assert default_options()["itemStepSharding"] == "rows"


def test_first_factor_as_average_rating():
    """als.initFirstFactorAsAvgRating (EmfBase.js:74, :493-511): rows drawn at prepareToTrain get
    their average train rating as first factor; rows without ratings keep the random draw."""
    ds, _, _ = small_dataset()
    lord = EmfLord(options={"factorsCount": 8, "als": {"initFirstFactorAsAvgRating": True}}, backend_factory=oracle_factory)
    lord.prepareToTrain(ds, seed=4)
    U, V = lord.backend.get_factors(0), lord.backend.get_factors(1)
    plain_u, plain_v = init_factors(50, 8, 8), init_factors(35, 8, 9)
    for fac, plain, csr in ((U, plain_u, ds.train_by_user), (V, plain_v, ds.train_by_item)):
        rp, vals = np.asarray(csr.rowPtr), np.asarray(csr.vals)
        assert np.array_equal(fac[:, 1:], plain[:, 1:])
        for r in range(len(rp) - 1):
            if rp[r + 1] > rp[r]:
                assert abs(fac[r, 0] - vals[rp[r]:rp[r + 1]].astype(np.float64).mean()) < 1e-6
            else:
                assert fac[r, 0] == plain[r, 0]


def test_checkpoint_after_every_iteration(tmp_path):
    """N4: the reference's open todo 'saveCalcResults every iter' (lib/YcnrController.js:288)."""
    ds, U, V = small_dataset()
    seen = []

    class Spy(EmfLord):
        def saveCalcResults(self, calcInfo):
            super().saveCalcResults(calcInfo)
            seen.append((self.trainIter, calcInfo["calcCnt"], np.fromfile(os.path.join(self.factorsReadyPath, "user_factors"), np.float32)))
    lord = Spy(options={"factorsCount": 8, "trainIters": 3, "dataDir": str(tmp_path), "dbType": "ml", "ratingsInPortionForRmse": 40,
                        "saveCalcResultsEveryIter": True}, backend_factory=oracle_factory)
    lord.prepareToTrain(ds, U, V)
    lord.train()
    assert [(i, c) for i, c, _ in seen] == [(1, 0), (2, 0), (3, 1)]   # two checkpoints, then the final save
    assert not np.array_equal(seen[0][2], seen[1][2]) and np.array_equal(seen[2][2], lord.backend.get_factors(0).ravel())


def test_rmse_reduce_and_shift_quirk(oracle):
    """rmse = sqrt(sum / cnt) over all portions, but predAvg (hence globalAvgShift) uses the LAST
    portion's sums only (EmfMaster.js:778-782)."""
    ds, U, V = small_dataset(seed=2)
    lord = EmfLord(options={"factorsCount": 8, "ratingsInPortionForRmse": 25, "numThreadsForTrain": {"als": 1}},
                   backend_factory=oracle_factory)
    lord.prepareToTrain(ds, U, V)
    ends = lord.portionsRowIdTo["rmseTest"]
    assert len(ends) > 2
    r = lord.calcRmse("rmseTest", False)
    t = ds.test
    tot = oracle.rmse_csr(8, t.rowPtr, t.indx, t.vals, U, V, 0.0)
    assert abs(r - np.sqrt(tot[0] / tot[1])) < 1e-12
    last = oracle.rmse_csr(8, t.rowPtr, t.indx, t.vals, U, V, 0.0, int(ends[-2]), 50)
    assert abs(lord.predAvg - last[2] / last[1]) < 1e-12
    assert abs(lord.globalAvgShift - (ds.totalRatingsAvg - lord.predAvg)) < 1e-12
    shift = lord.globalAvgShift
    r2 = lord.calcRmse("rmseTest", True)  # applies, does not recompute
    assert lord.globalAvgShift == shift
    tot2 = oracle.rmse_csr(8, t.rowPtr, t.indx, t.vals, U, V, shift)
    assert abs(r2 - np.sqrt(tot2[0] / tot2[1])) < 1e-12
    lord.options["dataSetDistr"] = [90, 10, 0]
    assert lord.calcRmse("rmseTest", False) is None  # EmfLord.js:1048


def test_init_factors_are_seeded_and_scaled():
    a = init_factors(1000, 50, 3)
    b = init_factors(1000, 50, 3)
    assert np.array_equal(a, b) and a.dtype == np.float32
    assert abs(a.std() - 1 / 50) < 2e-3  # randomNormal(1 / factorsCount), EmfBase.js:486


def test_rebalanced_ranges_equalise_measured_times():
    """The feedback that stands in for the reference's work-stealing portion dispenser (lib/emf/EmfLord.js:996-1006):
    shards re-cut from measured times converge to equal times when the true cost differs from the model, tile the
    rows, and stay put when a measurement is missing."""
    from ycnr_als.emf import rebalanced_ranges, row_cost, shard_ranges
    rng = np.random.default_rng(3)
    cnt = rng.integers(0, 300, 50_000)
    true = row_cost(cnt, 100) * (1.0 + np.arange(len(cnt)) / len(cnt))   # the later rows cost up to twice the model
    ms = lambda b: np.array([true[b[r]:b[r + 1]].sum() for r in range(len(b) - 1)])
    b = shard_ranges(cnt, 8, 100)
    assert ms(b).max() / ms(b).mean() > 1.2
    for _ in range(3):
        b = rebalanced_ranges(cnt, b, ms(b), 100)
        assert b[0] == 0 and b[-1] == len(cnt) and (np.diff(b) >= 0).all()
    assert ms(b).max() / ms(b).mean() < 1.01
    assert np.array_equal(rebalanced_ranges(cnt, b, [1.0] * 7 + [0.0], 100), b)     # a rank without a time: no re-cut
    assert np.array_equal(rebalanced_ranges(cnt, b, [1.0] * 7 + [float("nan")], 100), b)
