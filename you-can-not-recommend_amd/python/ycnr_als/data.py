"""Ratings containers and synthetic inputs for the ALS path.

The reference pulls ratings from PostgreSQL one portion at a time
(EmfMaster.m_fetchPortionTrainAlsOrRmse, lib/emf/EmfMaster.js:501-541).  Here a data set is
a pair of CSR structures of the same matrix -- by user and by item -- built once.
Ids are 0-based (db id - 1, EmfMaster.js:584-586); within a row column ids ascend
(ORDER BY user_list_id, item_id, EmfMaster.js:528).
"""
import numpy as np
import torch


class Csr:
    """rows x cols ratings in CSR: rowPtr int64[rows+1], indx int32[nnz], vals float[nnz]."""

    def __init__(self, rows, cols, rowPtr, indx, vals):
        self.rows, self.cols = int(rows), int(cols)
        self.rowPtr, self.indx, self.vals = rowPtr, indx, vals

    @property
    def nnz(self):
        return int(self.rowPtr[-1])

    def counts(self):
        return (self.rowPtr[1:] - self.rowPtr[:-1])

    def numpy(self):
        if isinstance(self.rowPtr, np.ndarray):
            return self
        return Csr(self.rows, self.cols, self.rowPtr.cpu().numpy(), self.indx.cpu().numpy(),
                   self.vals.cpu().numpy())

    def to(self, device):
        t = lambda a: (torch.from_numpy(a) if isinstance(a, np.ndarray) else a).to(device)
        return Csr(self.rows, self.cols, t(self.rowPtr), t(self.indx), t(self.vals))

    def astype(self, dtype):
        if isinstance(self.vals, np.ndarray):
            return Csr(self.rows, self.cols, self.rowPtr, self.indx, self.vals.astype(dtype))
        tdt = torch.float64 if np.dtype(dtype) == np.float64 else torch.float32
        return Csr(self.rows, self.cols, self.rowPtr, self.indx, self.vals.to(tdt))


def csr_from_coo(rows, cols, r, c, v):
    """Sort (r, c, v) triplets by (row, col) and build a Csr. torch tensors on any device."""
    key = r.to(torch.int64) * int(cols) + c.to(torch.int64)
    order = torch.argsort(key)
    r, c, v = r[order], c[order], v[order]
    counts = torch.bincount(r.to(torch.int64), minlength=int(rows))
    rowPtr = torch.zeros(int(rows) + 1, dtype=torch.int64, device=r.device)
    rowPtr[1:] = torch.cumsum(counts, 0)
    return Csr(rows, cols, rowPtr, c.to(torch.int32).contiguous(), v.contiguous())


def transpose_csr(a):
    """CSR by user -> CSR by item of the same ratings (torch tensors)."""
    rows_of = torch.repeat_interleave(torch.arange(a.rows, device=a.indx.device, dtype=torch.int64), a.counts())
    return csr_from_coo(a.cols, a.rows, a.indx.to(torch.int64), rows_of, a.vals)


def synth_ratings(users, items, nnz, max_rating=10, seed=20260001, device="cpu", dtype=torch.float32,
                  degree_sigma=1.2, zipf_a=1.0, rank=16, noise=0.7, hit_target=True):
    """synth_ratings_once, re-drawn once with an inflated request when dropping duplicate
    (user, item) pairs left the matrix more than 2 % short of the requested nnz."""
    a = synth_ratings_once(users, items, nnz, max_rating, seed, device, dtype, degree_sigma, zipf_a, rank, noise)
    got = a[0].nnz
    if hit_target and got < 0.98 * nnz:
        del a
        a = synth_ratings_once(users, items, int(nnz * (nnz / got) ** 1.15), max_rating, seed, device, dtype,
                               degree_sigma, zipf_a, rank, noise)
    return a


def synth_ratings_once(users, items, nnz, max_rating=10, seed=20260001, device="cpu", dtype=torch.float32,
                       degree_sigma=1.2, zipf_a=1.0, rank=16, noise=0.7):
    """Synthetic explicit-feedback matrix of the shape SURVEY.md 8(d) prescribes.

    * user degrees ~ log-normal, clipped to [1, items], scaled to the target nnz (MAL-like:
      long tail of heavy users);
    * item popularity ~ Zipf(zipf_a) over a random permutation of the item ids;
    * duplicates within a user are dropped (so nnz comes out slightly below the target),
      columns ascend within a row;
    * rating = clip(round(mu + p_u . q_i + eps), 1, max_rating) from a planted rank-`rank`
      model, stored as float (the db column is smallint, data/db-schema.sql:887-893).
    Returns (byUser: Csr, byItem: Csr) as torch tensors on `device`."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    dev = torch.device(device)
    deg = torch.exp(torch.randn(users, generator=g, device=dev) * degree_sigma)
    deg = deg * (nnz / float(deg.sum()))
    deg = deg.clamp(1, items).round().to(torch.int64)
    # a second scaling pass after clipping keeps the total near the target
    scale = nnz / float(deg.sum())
    deg = (deg.to(torch.float64) * scale).round().clamp(1, items).to(torch.int64)
    total = int(deg.sum())
    u = torch.repeat_interleave(torch.arange(users, device=dev, dtype=torch.int64), deg)
    w = 1.0 / torch.arange(1, items + 1, device=dev, dtype=torch.float64) ** zipf_a
    cdf = torch.cumsum(w / w.sum(), 0)
    perm = torch.randperm(items, generator=g, device=dev)
    pick = torch.searchsorted(cdf, torch.rand(total, generator=g, device=dev, dtype=torch.float64))
    i = perm[pick.clamp_(max=items - 1)]
    key = torch.unique(u * items + i)  # sorted, duplicates dropped
    u = key // items
    i = key % items
    del key, pick
    p = torch.randn(users, rank, generator=g, device=dev) / rank ** 0.25
    q = torch.randn(items, rank, generator=g, device=dev) / rank ** 0.25
    mu = (1 + max_rating) / 2.0 + 0.1 * max_rating
    val = torch.empty(u.numel(), device=dev, dtype=torch.float32)
    step = 1 << 24
    for s in range(0, u.numel(), step):  # chunked: p[u] * q[i] of 1e8 rows would not fit
        e = min(u.numel(), s + step)
        dot = (p[u[s:e]] * q[i[s:e]]).sum(1)
        eps = torch.randn(e - s, generator=g, device=dev) * noise
        val[s:e] = (mu + dot * (max_rating / 5.0) + eps * (max_rating / 5.0)).round().clamp(1, max_rating)
    val = val.to(dtype)
    counts = torch.bincount(u, minlength=users)
    rowPtr = torch.zeros(users + 1, dtype=torch.int64, device=dev)
    rowPtr[1:] = torch.cumsum(counts, 0)
    by_user = Csr(users, items, rowPtr, i.to(torch.int32).contiguous(), val)
    by_item = csr_from_coo(items, users, i, u, val)
    return by_user, by_item


def split_to_sets(by_user, distr=(85, 10, 5), seed=1):
    """Per-user split of the ratings into train / validate / test (dataset_type 1 / 2 / 3).

    Mirrors the intent of EmfLord.doSplitToSets (lib/emf/EmfLord.js:402-505): within each
    user a seeded shuffle, then the first distr[0]% train, next distr[1]% validate, rest test.
    Returns an int8 tensor of dataset types aligned with by_user.indx."""
    a = by_user
    dev = a.indx.device
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    nnz = a.indx.numel()
    rows_of = torch.repeat_interleave(torch.arange(a.rows, device=dev, dtype=torch.int64), a.counts())
    rnd = torch.rand(nnz, generator=g, device=dev, dtype=torch.float64)
    order = torch.argsort(rows_of.to(torch.float64) + rnd)  # shuffle inside each row
    rank_in_row = torch.empty(nnz, dtype=torch.int64, device=dev)
    rank_in_row[order] = torch.arange(nnz, device=dev) - a.rowPtr[:-1][rows_of[order]]
    cnt = a.counts()[rows_of].to(torch.float64)
    frac = (rank_in_row.to(torch.float64) + 0.5) / cnt
    t = torch.full((nnz,), 3, dtype=torch.int8, device=dev)
    t[frac < (distr[0] + distr[1]) / 100.0] = 2
    t[frac < distr[0] / 100.0] = 1
    return t


def select_csr(a, mask):
    """Sub-matrix of a Csr keeping the ratings where mask is True (same shape, fewer nnz)."""
    rows_of = torch.repeat_interleave(torch.arange(a.rows, device=a.indx.device, dtype=torch.int64), a.counts())
    counts = torch.bincount(rows_of[mask], minlength=a.rows)
    rowPtr = torch.zeros(a.rows + 1, dtype=torch.int64, device=a.indx.device)
    rowPtr[1:] = torch.cumsum(counts, 0)
    return Csr(a.rows, a.cols, rowPtr, a.indx[mask].contiguous(), a.vals[mask].contiguous())


def init_factors(rows, k, seed, dtype=np.float32):
    """N(0, sigma = 1/k) initial factors (Matrix.randomNormal(1 / factorsCount),
    lib/emf/EmfBase.js:486-493), seeded so CPU and GPU runs start from identical bytes."""
    rng = np.random.Generator(np.random.Philox(int(seed)))
    return (rng.standard_normal((int(rows), int(k))) / k).astype(dtype)


def csr_to_portion(csr, row_begin, row_end, dropLastRatingPerPortion=False):
    """Rows [row_begin, row_end) of a numpy Csr in the reference's portion-buffer format
    (alsRows / alsIndx / alsVals, lib/emf/EmfMaster.js:589-609), rows without ratings omitted.

    dropLastRatingPerPortion (opt-in): the row table the REFERENCE's packer writes for these
    ratings, bug for bug (its end-of-data branch, lib/emf/EmfMaster.js:594-603, records the row that is open when
    the last rating arrives BEFORE counting that rating): the last row's count is one short; a last row of a
    single rating is not recorded at all -- unless it is the portion's only rating, then it is recorded with
    cols = 0 (which libycnr_als skips, include/ycnr_als.h).  alsIndx / alsVals hold every rating either way, as
    the reference's buffers do."""
    c = csr
    b, e = int(c.rowPtr[row_begin]), int(c.rowPtr[row_end])
    cnt = (c.rowPtr[row_begin + 1:row_end + 1] - c.rowPtr[row_begin:row_end]).astype(np.int64)
    ids = np.nonzero(cnt)[0]
    cols = cnt[ids]
    if dropLastRatingPerPortion and len(ids):
        if cols[-1] > 1 or e - b == 1:
            cols = cols.copy()
            cols[-1] -= 1
        else:
            ids, cols = ids[:-1], cols[:-1]
    rows = np.empty(1 + 2 * len(ids), np.int32)
    rows[0] = len(ids)
    rows[1::2] = ids + row_begin
    rows[2::2] = cols
    return rows, np.ascontiguousarray(c.indx[b:e]), np.ascontiguousarray(c.vals[b:e])


def drop_last_rating_per_portion(csr, portion_row_ends):
    """The ratings a pass of the REFERENCE really consumes when `csr` is fed to it in the portions that end at the
    0-based exclusive row ids `portion_row_ends` (= its 1-based inclusive portionsRowIdTo, lib/emf/EmfLord.js:571-592):
    every portion loses its last rating (csr_to_portion(..., dropLastRatingPerPortion=True); the packer of
    lib/emf/EmfMaster.js:594-603).  Returns a Csr of the same shape (numpy or torch, like the input); a row left
    without ratings is not solved, as the rows the reference never records or records with cols = 0.
    Opt-in compatibility for bug-for-bug replays: the resident trainer consumes every rating by default."""
    is_np = isinstance(csr.rowPtr, np.ndarray)
    rp = csr.rowPtr if is_np else csr.rowPtr.cpu().numpy()
    ends = np.asarray(portion_row_ends, np.int64)
    ends = ends[(ends > 0) & (ends <= csr.rows)]
    last = rp[ends] - 1                                   # position of the last rating of every portion ...
    begin = rp[np.concatenate([[0], ends[:-1]])]
    last = np.unique(last[rp[ends] > begin])              # ... that holds a rating at all
    keep = np.ones(int(rp[-1]), bool)
    keep[last] = False
    cnt = np.diff(rp).copy()
    if len(last):
        cnt[np.searchsorted(rp, last, side="right") - 1] -= 1
    new_rp = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    if is_np:
        return Csr(csr.rows, csr.cols, new_rp, np.ascontiguousarray(csr.indx[keep]), np.ascontiguousarray(csr.vals[keep]))
    dev = csr.indx.device
    k = torch.from_numpy(keep).to(dev)
    return Csr(csr.rows, csr.cols, torch.from_numpy(new_rp).to(dev), csr.indx[k].contiguous(), csr.vals[k].contiguous())
