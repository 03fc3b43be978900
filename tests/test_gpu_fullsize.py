"""BASELINE.json's full size on the GPU: size-independent properties of the row solve.

At MAL scale (1.75 M x 12.7 K, 121 M ratings, k = 100: the configuration bench.py measures) the
CPU oracle needs minutes per half-step, so the HIP path is checked there through properties
that hold at any size:

  * the solve is linear in the ratings of a row (x = (Y^T Y + lam n I)^-1 Y^T r, Y and n fixed):
    x(r1 + 2 r2) = x(r1) + 2 x(r2) for every one of the 1.75 M user rows, with r2 arbitrary
    floats (the integer ratings 1..10 of the workload never exercise the low bits of r);
  * the normal equations hold: for sampled rows of every length class (and the longest rows,
    which go through chunks + reduce) the float64 residual of the returned x is at the level
    of float32 rounding, on both sides;
  * rows without ratings are left untouched.

Tolerances: 1e-5 of the norms involved for linearity (observed on MI355X: worst 5.9e-7, median
6.4e-8 over the 1.75 M rows), normwise backward error <= 2e-6 for the residuals (observed worst
1.8e-7 on 3048 user rows, 2.6e-7 on 415 item rows, medians 2e-8 and 4e-8).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

USERS, ITEMS, NNZ, K = 1_750_000, 12_700, 121_000_000, 100
LAM = 0.05


@pytest.fixture(scope="module")
def mal():
    import torch
    import ycnr_als
    from ycnr_als.data import synth_ratings
    L = ycnr_als._lib.load()  # raises if libycnr_als.so is missing: no fallback
    assert L.ycnr_device_count() >= 1, L.ycnr_last_error()
    dev = torch.device("cuda", 0)
    by_user, by_item = synth_ratings(USERS, ITEMS, NNZ, max_rating=10, seed=20260004, device=dev,
                                     degree_sigma=1.2, zipf_a=0.6)
    return ycnr_als, torch, dev, by_user, by_item


def backward_errors(torch, csr, vals, fixed, solved, rows):
    """float64 normwise backward error ||A x - b|| / (||A|| ||x|| + ||b||) of the sampled rows"""
    out = []
    for r in rows.tolist():
        b0, e0 = int(csr.rowPtr[r]), int(csr.rowPtr[r + 1])
        n = e0 - b0
        Y = fixed[csr.indx[b0:e0].long()].double()
        A = Y.T @ Y + LAM * n * torch.eye(K, dtype=torch.float64, device=Y.device)
        b = Y.T @ vals[b0:e0].double()
        x = solved[r].double()
        out.append(float(torch.linalg.norm(A @ x - b) / (torch.linalg.norm(A) * torch.linalg.norm(x) + torch.linalg.norm(b))))
    return np.array(out)


def sample_rows(torch, csr, count, longest, seed):
    lens = csr.rowPtr[1:] - csr.rowPtr[:-1]
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    nz = torch.nonzero(lens > 0).flatten().cpu()
    pick = nz[torch.randperm(nz.numel(), generator=g)[:count]]
    top = torch.topk(lens, longest).indices.cpu()
    return torch.unique(torch.cat([pick, top]))


def test_user_half_step_is_linear_in_the_ratings_and_solves_the_normal_equations(mal):
    als, torch, dev, bu, bi = mal
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    V = torch.randn(ITEMS, K, generator=g, device=dev) / K ** 0.5
    r1 = bu.vals
    r2 = torch.randn(bu.nnz, generator=g, device=dev) * 3.0
    xs = []
    for vals in (r1, r2, r1 + 2.0 * r2):
        U = torch.full((USERS, K), 7.0, device=dev)  # rows without ratings must keep this
        h = als.AlsDevice(K, USERS, ITEMS, userFactReg=LAM, itemFactReg=LAM)
        h.bind_factors("byUser", U)
        h.bind_factors("byItem", V)
        h.set_ratings("byUser", bu.rowPtr, bu.indx, vals.contiguous())
        info = h.step("byUser")
        torch.cuda.synchronize()
        assert info.numericErrors == 0
        h.destroy()
        xs.append(U)
    lens = bu.rowPtr[1:] - bu.rowPtr[:-1]
    assert int((lens > 0).sum()) == info.rows
    empty = lens == 0
    if bool(empty.any()):
        assert bool((xs[0][empty] == 7.0).all())
    x1, x2, x3 = xs
    diff = torch.linalg.norm((x3 - x1 - 2.0 * x2).double(), dim=1)
    scale = (torch.linalg.norm(x1.double(), dim=1) + 2.0 * torch.linalg.norm(x2.double(), dim=1)
             + torch.linalg.norm(x3.double(), dim=1))
    ratio = torch.where(empty, torch.zeros_like(diff), diff / scale.clamp_min(1e-30))
    assert bool(torch.isfinite(ratio).all())
    worst = float(ratio.max())
    print(f"\nlinearity over {info.rows} rows: worst ratio {worst:.3g}, median {float(ratio.median()):.3g}")
    assert worst <= 1e-5, f"linearity violated: worst ratio {worst:.3g} at row {int(ratio.argmax())}"
    rows = sample_rows(torch, bu, 3000, 48, seed=3)
    for vals, x in ((r1, x1), (r2, x2)):
        eta = backward_errors(torch, bu, vals, V, x, rows)
        print(f"user rows sampled {len(eta)}: backward error worst {eta.max():.3g}, median {np.median(eta):.3g}")
        assert eta.max() <= 2e-6, f"normal equations: worst backward error {eta.max():.3g}"


def test_item_half_step_solves_the_normal_equations(mal):
    als, torch, dev, bu, bi = mal
    g = torch.Generator(device=dev)
    g.manual_seed(12)
    U = torch.randn(USERS, K, generator=g, device=dev) / K ** 0.5
    V = torch.zeros(ITEMS, K, device=dev)
    h = als.AlsDevice(K, USERS, ITEMS, userFactReg=LAM, itemFactReg=LAM)
    h.bind_factors("byUser", U)
    h.bind_factors("byItem", V)
    h.set_ratings("byItem", bi.rowPtr, bi.indx, bi.vals)
    info = h.step("byItem")
    torch.cuda.synchronize()
    assert info.numericErrors == 0 and info.splitRows > 0
    h.destroy()
    rows = sample_rows(torch, bi, 400, 16, seed=4)
    eta = backward_errors(torch, bi, bi.vals, U, V, rows)
    print(f"\nitem rows sampled {len(eta)} (split rows {info.splitRows}): backward error worst {eta.max():.3g}, median {np.median(eta):.3g}")
    assert eta.max() <= 2e-6, f"normal equations: worst backward error {eta.max():.3g}"


def test_rmse_sums_at_full_size(mal):
    """The RMSE pass over all 121 M ratings against torch float64 on the same device: the sum of
    squared differences, the count and the sum of the predictions of the whole set and of 7 portions."""
    als, torch, dev, bu, bi = mal
    g = torch.Generator(device=dev)
    g.manual_seed(13)
    U = torch.randn(USERS, K, generator=g, device=dev) / K ** 0.25
    V = torch.randn(ITEMS, K, generator=g, device=dev) / K ** 0.25
    h = als.AlsDevice(K, USERS, ITEMS)
    h.bind_factors("byUser", U)
    h.bind_factors("byItem", V)
    h.set_rmse_ratings("rmseValidate", bu.rowPtr, bu.indx, bu.vals)
    shift = 0.37
    ends = np.linspace(0, USERS, 8).astype(np.int64)[1:]
    whole = h.rmse("rmseValidate", shift)[0]
    parts = h.rmse("rmseValidate", shift, ends)
    h.destroy()
    rows = torch.repeat_interleave(torch.arange(USERS, device=dev), bu.rowPtr[1:] - bu.rowPtr[:-1])
    d2 = torch.empty(bu.nnz, dtype=torch.float64, device=dev)
    psum = pabs = 0.0
    step = 1 << 23
    for s in range(0, bu.nnz, step):
        e = min(bu.nnz, s + step)
        pred = (U[rows[s:e]].double() * V[bu.indx[s:e].long()].double()).sum(1) + shift
        d2[s:e] = (bu.vals[s:e].double() - pred) ** 2
        psum += float(pred.sum())
        pabs += float(pred.abs().sum())
    want = float(d2.sum())
    assert whole[1] == bu.nnz                                   # rCnt: exact
    assert abs(whole[0] - want) <= 1e-6 * want                  # rSumDiff2: float32 dot products inside
    assert abs(whole[2] - psum) <= 1e-6 * pabs                  # rSum: the sum of the predictions
    tot = parts.sum(0)                                          # portions add up to the whole
    assert tot[1] == whole[1] and abs(tot[0] - whole[0]) <= 1e-12 * whole[0] and abs(tot[2] - whole[2]) <= 1e-12 * pabs
    lo = 0
    for p, hi in enumerate(ends.tolist()):
        b0, e0 = int(bu.rowPtr[lo]), int(bu.rowPtr[hi])
        assert parts[p, 1] == e0 - b0
        assert abs(parts[p, 0] - float(d2[b0:e0].sum())) <= 1e-6 * float(d2[b0:e0].sum())
        lo = hi


def test_ingest_and_split_round_trips_at_full_size(mal):
    """The steps either side of the path on the whole matrix (host buffers, as their C entry points
    take them): shuffled triplets come back as the CSR they were cut from, transposing twice is
    the identity and the transpose is the generator's by-item matrix, bit for bit; the split into
    sets assigns every rating once, is idempotent, follows dataSetDistr, and the per-row
    statistics of the sets add up to the whole."""
    als, torch, dev, bu, bi = mal
    from ycnr_als import csrfile
    from ycnr_als.data import Csr
    a, t = bu.numpy(), bi.numpy()
    rng = np.random.default_rng(5)
    row = np.repeat(np.arange(USERS, dtype=np.int32), a.rowPtr[1:] - a.rowPtr[:-1])
    perm = rng.permutation(a.nnz)
    got, _ = csrfile.csr_from_triplets(row[perm], a.indx[perm], a.vals[perm], USERS, ITEMS)
    del perm, row
    assert np.array_equal(got.rowPtr, a.rowPtr) and np.array_equal(got.indx, a.indx) and np.array_equal(got.vals, a.vals)
    tr, _ = csrfile.transpose(got)
    assert np.array_equal(tr.rowPtr, t.rowPtr) and np.array_equal(tr.indx, t.indx) and np.array_equal(tr.vals, t.vals)
    back, _ = csrfile.transpose(tr)
    assert np.array_equal(back.rowPtr, a.rowPtr) and np.array_equal(back.indx, a.indx) and np.array_equal(back.vals, a.vals)
    del got, tr, back
    types, _ = als.split_to_sets(a.rowPtr, np.zeros(a.nnz, np.int8), (85, 10, 5), seed=9)
    assert types.min() >= 1 and types.max() <= 3
    again, _ = als.split_to_sets(a.rowPtr, types, (85, 10, 5), seed=10)  # nothing left to assign
    assert np.array_equal(again, types)
    share = np.bincount(types, minlength=4)[1:] / a.nnz
    assert np.abs(share - np.array([0.85, 0.10, 0.05])).max() < 0.02
    cnt_all, sum_all, _ = als.rating_stats(a.rowPtr, a.vals)
    cnt_sets, sum_sets, _ = als.rating_stats(a.rowPtr, a.vals, types)
    assert np.array_equal(cnt_all, (a.rowPtr[1:] - a.rowPtr[:-1]).astype(np.int32))
    assert np.array_equal(cnt_sets, cnt_all) and np.array_equal(sum_sets, sum_all)  # integer ratings: exact sums
    assert sum_all.sum() == float(a.vals.astype(np.float64).sum())
