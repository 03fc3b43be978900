"""N1: train / validate / test split and per-row statistics.

CPU: the oracle's restatement of EmfLord.doSplitToSets (lib/emf/EmfLord.js:402-505) against the
count formula of lines 447-457 worked by hand, and against an independent numpy statement of
the keyed order.  GPU: ycnr_split_to_sets / ycnr_rating_stats bit-exact against the oracle
(integer work), including rows longer than the kernel's LDS key cache and the incremental
("split more") form with ratings that are already assigned.
"""
import math

import numpy as np
import pytest

from oracle import oracle as orc


@pytest.fixture(scope="module")
def als():
    import ycnr_als
    L = ycnr_als._lib.load()  # raises if libycnr_als.so is missing: no fallback
    assert L.ycnr_device_count() >= 1, L.ycnr_last_error()
    return ycnr_als


def fmix32(h):
    h = np.asarray(h, np.uint64) & 0xFFFFFFFF
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & 0xFFFFFFFF
    h ^= h >> 16
    return h


def numpy_split(rowPtr, types, pcts, seed):
    """Independent statement: targets by the reference's formula in Python floats + math.ceil
    (as the JS does), order by numpy lexsort of (key, j)."""
    t = np.array(types, np.int8, copy=True)
    for r in range(len(rowPtr) - 1):
        b, e = int(rowPtr[r]), int(rowPtr[r + 1])
        row = t[b:e]
        free = np.flatnonzero(row == 0)
        if len(free) == 0:
            continue
        c = [int((row == s).sum()) for s in (1, 2, 3)]
        total = len(free) + sum(c)
        tg0 = math.ceil(total * pcts[0] / 100)
        tg1 = math.ceil(total * (pcts[0] + pcts[1]) / 100) - tg0
        tg2 = total - tg0 - tg1
        nw = [max(0, tg0 - c[0]), max(0, tg1 - c[1]), max(0, tg2 - c[2])]
        if sum(nw) < len(free):
            nw[0] += len(free) - sum(nw)
        key = fmix32(fmix32((seed + 0x9E3779B9 * r) & 0xFFFFFFFF) ^ free.astype(np.uint64))
        order = free[np.lexsort((free, key))]
        offs = 0
        for s in range(3):
            row[order[offs:offs + nw[s]]] = s + 1
            offs += nw[s]
    return t


def problem(rows, seed, max_len=300, preassigned=False, long_row=0):
    rng = np.random.default_rng(seed)
    lens = np.clip(rng.lognormal(np.log(20), 1.0, rows).astype(np.int64), 0, max_len)
    lens[rng.integers(0, rows, max(1, rows // 20))] = 0
    if long_row:
        lens[rows // 2] = long_row
    rowPtr = np.zeros(rows + 1, np.int64)
    np.cumsum(lens, out=rowPtr[1:])
    types = np.zeros(rowPtr[-1], np.int8)
    if preassigned:  # "split more": some ratings already carry a set, some are excluded (4)
        types[:] = rng.choice(np.array([0, 0, 0, 1, 1, 2, 3, 4], np.int8), rowPtr[-1])
    vals = rng.integers(1, 11, rowPtr[-1]).astype(np.float32)
    return rowPtr, types, vals


def test_counts_follow_the_reference_formula():
    # one user with 7 ratings, 85/10/5: ceil(5.95) = 6 train, ceil(6.65) - 6 = 1 validate, 0 test
    t = orc.split_to_sets([0, 7], np.zeros(7, np.int8), (85, 10, 5), seed=3)
    assert sorted(t.tolist()) == [1, 1, 1, 1, 1, 1, 2]
    # 100 ratings: 85 / 10 / 5 exactly; 1 rating: train
    t = orc.split_to_sets([0, 100, 101], np.zeros(101, np.int8), (85, 10, 5), seed=3)
    assert [(t[:100] == s).sum() for s in (1, 2, 3)] == [85, 10, 5] and t[100] == 1
    # already assigned ratings count towards the targets; excluded ones (4) are left alone;
    # a set that is over its target takes nothing and the left-overs go to train (EmfLord.js:455-457)
    types = np.array([2, 2, 2, 0, 0, 0, 0, 4, 0, 0], np.int8)   # total 9: targets 8 / 1 / 0
    t = orc.split_to_sets([0, 10], types, (85, 10, 5), seed=9)
    assert t[7] == 4 and (t[:3] == 2).all() and (t[[3, 4, 5, 6, 8, 9]] == 1).all()


@pytest.mark.parametrize("pre", [False, True])
def test_oracle_equals_independent_numpy_statement(pre):
    rowPtr, types, _ = problem(400, 5 + pre, preassigned=pre)
    for pcts, seed in (((85, 10, 5), 1), ((60, 25, 15), 2026), ((100, 0, 0), 7)):
        got = orc.split_to_sets(rowPtr, types, pcts, seed)
        want = numpy_split(rowPtr, types, pcts, seed)
        assert np.array_equal(got, want)
        assert not (got == 0).any()
        assert np.array_equal(got[types != 0], types[types != 0])  # nothing reassigned
    # a different seed gives a different split with the same counts
    a, b = orc.split_to_sets(rowPtr, types, (85, 10, 5), 1), orc.split_to_sets(rowPtr, types, (85, 10, 5), 2)
    assert not np.array_equal(a, b)
    assert [np.bincount(a, minlength=5).tolist()] == [np.bincount(b, minlength=5).tolist()]


def test_oracle_stats():
    rowPtr, types, vals = problem(300, 11, preassigned=True)
    cnt, sm = orc.rating_stats(rowPtr, vals, types)
    for r in (0, 17, 150, 299):
        sl = slice(rowPtr[r], rowPtr[r + 1])
        m = (types[sl] >= 1) & (types[sl] <= 3)
        assert cnt[r] == m.sum() and sm[r] == vals[sl][m].astype(np.float64).sum()
    cnt_all, sm_all = orc.rating_stats(rowPtr, vals.astype(np.float64))
    assert np.array_equal(cnt_all, np.diff(rowPtr)) and np.isclose(sm_all.sum(), vals.astype(np.float64).sum())


@pytest.mark.gpu
@pytest.mark.parametrize("pre", [False, True])
def test_gpu_split_bit_exact(als, pre):
    # 13000 > the kernel's 12288-key LDS cache: that row takes the recompute path
    rowPtr, types, _ = problem(3000, 21 + pre, max_len=2000, preassigned=pre, long_row=13000)
    for pcts, seed in (((85, 10, 5), 1), ((70, 20, 10), 123456789)):
        got, ms = als.split_to_sets(rowPtr, types, pcts, seed)
        want = orc.split_to_sets(rowPtr, types, pcts, seed)
        assert np.array_equal(got, want)
        assert ms > 0


@pytest.mark.gpu
@pytest.mark.parametrize("pcts", [(85, 10, 5), (100, 0, 0), (0, 0, 100), (0, 100, 0), (1, 1, 98), (50, 50, 0), (33, 33, 34)])
def test_gpu_split_class_boundaries_and_degenerate_percentages(als, pcts):
    """Rows on both sides of every kernel class boundary (ranking up to 40 ratings, one-wave bisection up to 1024, a
    1024-thread workgroup beyond, keys recomputed past 12 288), fresh and with some ratings already assigned (types 1..3,
    and 4 = excluded), with percentages that put a threshold at rank 0 or past the last free rating: bit-exact against
    the oracle."""
    lens = [0, 1, 2, 3, 16, 39, 40, 41, 42, 63, 64, 65, 100, 191, 192, 193, 500, 1023, 1024, 1025, 1026, 2000, 4097, 12288, 12289, 13001]
    rng = np.random.default_rng(sum(pcts) * 7 + pcts[0])
    rowPtr = np.zeros(len(lens) + 1, np.int64)
    rowPtr[1:] = np.cumsum(lens)
    for pre in (False, True):
        types = np.zeros(rowPtr[-1], np.int8)
        if pre:
            types[:] = rng.choice(np.array([0, 0, 0, 1, 2, 3, 4], np.int8), rowPtr[-1])
            types[rowPtr[5]:rowPtr[6]] = 1      # a row with nothing left to assign
            types[rowPtr[12]:rowPtr[13]] = 4    # a row of excluded ratings only
        for seed in (1, 0xFFFFFFFF):
            got, _ = als.split_to_sets(rowPtr, types, pcts, seed)
            want = orc.split_to_sets(rowPtr, types, pcts, seed)
            assert np.array_equal(got, want), (pcts, pre, seed, int(np.flatnonzero(got != want)[0]))


@pytest.mark.gpu
def test_gpu_split_edge_cases(als):
    empty = np.zeros(0, np.int8)
    got, _ = als.split_to_sets(np.zeros(5, np.int64), empty)          # rows without ratings
    assert got.size == 0
    got, _ = als.split_to_sets([0, 1, 1, 3], np.zeros(3, np.int8))    # single ratings go to train
    assert np.array_equal(got, orc.split_to_sets([0, 1, 1, 3], np.zeros(3, np.int8)))
    with pytest.raises(als.YcnrError):
        als.split_to_sets([0, 3], np.zeros(3, np.int8), (80, 10, 5))  # does not sum to 100
    with pytest.raises(als.YcnrError):
        als.split_to_sets([0, 3, 2], np.zeros(3, np.int8))            # rowPtr decreases


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_gpu_stats(als, dt):
    rowPtr, types, vals = problem(5000, 31, preassigned=True)
    vals = (vals + np.random.default_rng(1).random(vals.size)).astype(dt)   # not integers
    for t in (types, None):
        cnt, sm, ms = als.rating_stats(rowPtr, vals, t)
        ocnt, osm = orc.rating_stats(rowPtr, vals, t)
        assert np.array_equal(cnt, ocnt)
        assert np.allclose(sm, osm, rtol=1e-13, atol=0)   # same addends, different summation tree
