"""N3: top-N recommend.

CPU: the oracle restates the loop of YcnrController.recommendItemsForUser
(lib/YcnrController.js:255-274) literally; it is pinned here against what that loop does by
construction -- at most limit - 1 items, best first, the threshold, the skip list -- and against a
plain numpy argsort.  GPU: ycnr_recommend_items against the oracle: exact in float64 on inputs
without near-ties, ids equal wherever predicts are separated by more than the dot's rounding in
float32, predicts within tolerance.
"""
import numpy as np
import pytest

from oracle import oracle as orc


@pytest.fixture(scope="module")
def als():
    import ycnr_als
    L = ycnr_als._lib.load()  # raises if libycnr_als.so is missing: no fallback
    assert L.ycnr_device_count() >= 1, L.ycnr_last_error()
    return ycnr_als


def problem(users, items, k, seed, dt=np.float32, skip_frac=0.1):
    rng = np.random.default_rng(seed)
    U = (rng.standard_normal((users, k)) * 0.8).astype(dt)
    V = (rng.standard_normal((items, k)) * 0.8).astype(dt)
    skips = [np.sort(rng.choice(items, rng.integers(0, max(2, int(items * skip_frac))), replace=False)).astype(np.int32) for _ in range(users)]
    ptr = np.zeros(users + 1, np.int64)
    ptr[1:] = np.cumsum([len(x) for x in skips])
    return U, V, ptr, (np.concatenate(skips) if users else np.zeros(0, np.int32)), skips


def numpy_top(u, V, skip, shift, min_rating, limit):
    pred = (V.astype(np.float64) @ u.astype(np.float64)) + shift
    ok = np.ones(len(V), bool)
    ok[skip] = False
    ok &= pred >= min_rating
    ids = np.flatnonzero(ok)
    order = ids[np.lexsort((ids, -pred[ids]))][:max(0, limit - 1)]
    return order.astype(np.int32), pred[order]


def test_oracle_is_the_reference_loop():
    U, V, _, _, skips = problem(6, 300, 10, 1, np.float64)
    for u in range(6):
        for limit, thr in ((20, 1.0), (5, -100.0), (1, -100.0), (2, 0.0), (400, -100.0)):
            ids, pr = orc.recommend(U[u], V, skips[u], 0.25, thr, limit)
            wids, wpr = numpy_top(U[u], V, skips[u], 0.25, thr, limit)
            assert len(ids) <= max(0, limit - 1)                 # the reference never returns `limit` items
            assert np.array_equal(ids, wids) and np.allclose(pr, wpr, rtol=1e-12)
            assert not np.isin(ids, skips[u]).any() and (pr >= thr).all() and (np.diff(pr) <= 0).all()
    # equal predicts: ascending item id (push order, stable sort)
    V2 = np.tile(V[:1], (8, 1))
    ids, pr = orc.recommend(U[0], V2, np.array([2], np.int32), 0.0, -100.0, 6)
    assert ids.tolist() == [0, 1, 3, 4, 5]


@pytest.mark.gpu
def test_gpu_float64_exact(als):
    U, V, ptr, sk, skips = problem(40, 1234, 100, 3, np.float64)
    ids, pred, cnt, ms = als.recommend_items(U, V, ptr, sk, globalAvgShift=0.3, minRecommendRating=2.0, limit=20)
    assert ms > 0
    for u in range(40):
        oid, opr = orc.recommend(U[u], V, skips[u], 0.3, 2.0, 20)
        assert cnt[u] == len(oid) <= 19
        assert np.array_equal(ids[u, :cnt[u]], oid) and np.allclose(pred[u, :cnt[u]], opr, rtol=1e-12)
        assert (ids[u, cnt[u]:] == -1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("k", [20, 100, 7, 1000])
def test_gpu_float32_against_oracle(als, k):
    # k = 20, 100, 7: the scores on the matrix cores (k padded to 32, 112, 16 in LDS); k = 1000: sixteen users' factors no
    # longer fit the default LDS limit, the 16-lane-group kernel takes over
    U, V, ptr, sk, skips = problem(64, 5000 if k <= 100 else 700, k, 7 + k)
    ids, pred, cnt, _ = als.recommend_items(U, V, ptr, sk, globalAvgShift=-0.1, minRecommendRating=1.5, limit=20)
    for u in range(64):
        oid, opr = orc.recommend(U[u], V, skips[u], -0.1, 1.5, 20)
        # the two float32 dots round differently: compare as sets unless predicts are well separated
        tol = 1e-5 * max(1.0, float(np.abs(opr).max()) if len(opr) else 1.0)
        assert abs(int(cnt[u]) - len(oid)) <= 1
        n = min(int(cnt[u]), len(oid))
        assert np.allclose(pred[u, :n], opr[:n], atol=tol)
        sep = np.ones(n, bool)
        if n > 1:
            gap = np.abs(np.diff(opr[:n]))
            sep[1:] &= gap > 4 * tol
            sep[:-1] &= gap > 4 * tol
        sep &= np.abs(opr[:n] - 1.5) > 4 * tol
        if n == len(oid) == int(cnt[u]):
            assert np.array_equal(ids[u, :n][sep], oid[:n][sep])
        assert not np.isin(ids[u, :cnt[u]], skips[u]).any()


@pytest.mark.gpu
def test_gpu_edge_cases(als):
    U, V, ptr, sk, skips = problem(3, 50, 8, 11)
    # everything skipped for user 0; threshold above every predict for all
    ptr2 = np.array([0, 50, 50, 50], np.int64)
    sk2 = np.arange(50, dtype=np.int32)
    ids, pred, cnt, _ = als.recommend_items(U, V, ptr2, sk2, 0.0, -1e9, 10)
    assert cnt[0] == 0 and cnt[1] == 9 and (ids[0] == -1).all()
    ids, pred, cnt, _ = als.recommend_items(U, V, ptr, sk, 0.0, 1e9, 10)
    assert (cnt == 0).all()
    ids, pred, cnt, _ = als.recommend_items(U, V, ptr, sk, 0.0, -1e9, 1)       # limit 1: the reference returns nothing
    assert (cnt == 0).all()
    ids, pred, cnt, _ = als.recommend_items(U[:0], V, np.zeros(1, np.int64), np.zeros(0, np.int32))
    assert ids.shape == (0, 20)
    with pytest.raises(als.YcnrError):
        als.recommend_items(U, V, ptr2, sk2[::-1].copy(), 0.0, 0.0, 10)         # skip ids not ascending
