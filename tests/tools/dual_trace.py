"""Where and when the rows of one dual class ran, and which of them came out wrong (devtest/dual7).

Needs a library built with -DYCNR_DUAL_TRACE=<rows> (the kernel writes HW_ID, XCC_ID and the 100 MHz clock at its phase
boundaries into the rows of the solved matrix from row <rows> on; this probe leaves those rows without ratings):
  make -C you-can-not-recommend_amd/csrc OUT=../../ablibs/lib_d7w2_trace.so EXTRA="-DYCNR_DUAL7_WAVES=2 -DYCNR_DUAL_TRACE=4800"
  YCNR_ALS_LIB=$PWD/ablibs/lib_d7w2_trace.so python tests/tools/dual_trace.py [blocks=7] [k=256] [rows_per_n=300] [out.npz]
Prints, for wrong and right rows: how many shared their SIMD with another wave of the launch while they were in their
Gramian / solve / x = Y^T w phase, and which phase the other wave was in."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "you-can-not-recommend_amd", "python"))
import ycnr_als  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 7
k = int(sys.argv[2]) if len(sys.argv) > 2 else 256
per = int(sys.argv[3]) if len(sys.argv) > 3 else 300
out = sys.argv[4] if len(sys.argv) > 4 else None
items = 3000
rng = np.random.default_rng(5)
lens = np.repeat(np.arange(16 * (m - 1) + 1, 16 * m + 1), per).astype(np.int64)
rng.shuffle(lens)
users = len(lens)
extra = (users * 2048 + 4 * k - 1) // (4 * k) + 1
rowPtr = np.zeros(users + extra + 1, np.int64)
np.cumsum(lens, out=rowPtr[1:users + 1])
rowPtr[users + 1:] = rowPtr[users]
indx = np.concatenate([np.sort(rng.choice(items, n, replace=False)) for n in lens]).astype(np.int32)
vals = (rng.standard_normal(rowPtr[-1]) * 2.0 + 5.0).astype(np.float32)
V = (rng.standard_normal((items, k)) / np.sqrt(k)).astype(np.float32)
V64 = V.astype(np.float64)
want = np.zeros((users, k))
for u in range(users):
    Y = V64[indx[rowPtr[u]:rowPtr[u + 1]]]
    want[u] = np.linalg.solve(Y.T @ Y + 0.05 * len(Y) * np.eye(k), Y.T @ vals[rowPtr[u]:rowPtr[u + 1]].astype(np.float64))

PH = ("gram", "solve", "xpass")
for rep in range(2):
    dev = ycnr_als.AlsDevice(k, users + extra, items)
    dev.set_ratings("byUser", rowPtr, indx, vals)
    dev.set_factors("byUser", np.zeros((users + extra, k), np.float32))
    dev.set_factors("byItem", V)
    dev.step("byUser")
    got32 = dev.get_factors("byUser")
    dev.destroy()
    got = got32[:users].astype(np.float64)
    err = np.linalg.norm(got - want, axis=1) / np.linalg.norm(want, axis=1)
    wrong = err > 1e-4
    rec = got32[users:].view(np.uint32).reshape(-1)[:users * 512].reshape(users, 512)
    tr = rec[:, :8]
    if not (tr[:, 7] == 0x7ACE7ACE).all():
        print("no trace records: the library was not built with -DYCNR_DUAL_TRACE=%d" % users)
        sys.exit(1)
    hw, xcc = tr[:, 0], tr[:, 1] & 0xF
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
    place = ((xcc.astype(np.int64) * 8 + se) * 2 + sh) * 16 + cu
    slot = place * 4 + simd
    t = tr[:, 2:6].astype(np.int64)
    t -= t[:, 0].min()
    t[t < 0] += 1 << 32
    print(f"run {rep}: wrong rows {int(wrong.sum())} of {users}; launch spans {t.max() / 100.0:.1f} us; distinct SIMDs used {len(np.unique(slot))}, CUs {len(np.unique(place))}")
    print("  phase lengths (us, median): gram %.1f solve %.1f xpass %.1f" % tuple(np.median(t[:, i + 1] - t[:, i]) / 100.0 for i in range(3)))
    # overlaps: for every row, the other rows on its SIMD whose lifetime intersects each of its phases
    order = np.argsort(slot, kind="stable")
    stats = {w: {p: {q: 0 for q in PH + ("none",)} for p in PH} for w in (True, False)}
    alone = {True: 0, False: 0}
    bounds = np.flatnonzero(np.diff(slot[order])) + 1
    for grp in np.split(order, bounds):
        for i in grp:
            others = [j for j in grp if j != i and t[j, 0] < t[i, 3] and t[j, 3] > t[i, 0]]
            if not others:
                alone[bool(wrong[i])] += 1
            for pi, p in enumerate(PH):
                a0, a1 = t[i, pi], t[i, pi + 1]
                hit = set()
                for j in others:
                    for qi, q in enumerate(PH):
                        if t[j, qi] < a1 and t[j, qi + 1] > a0:
                            hit.add(q)
                if not hit:
                    stats[bool(wrong[i])][p]["none"] += 1
                for q in hit:
                    stats[bool(wrong[i])][p][q] += 1
    for w in (True, False):
        nrows = int((wrong == w).sum())
        print(f"  {'WRONG' if w else 'right'} rows: {nrows}; never shared their SIMD: {alone[w]}")
        for p in PH:
            print(f"    while in {p:5s}: partner in " + ", ".join(f"{q} {stats[w][p][q]}" for q in PH + ("none",)))
    # position in the launch (blockIdx) of the wrong rows
    pos = tr[:, 6].astype(np.int64)
    hist = np.histogram(pos[wrong], bins=np.arange(0, users + 1, 300))[0]
    print("  wrong rows per 300 launch positions:", hist.tolist())
    print("  wrong rows per SIMD id:", np.bincount(simd[wrong], minlength=4).tolist(), " per XCC:", np.bincount(xcc[wrong], minlength=8).tolist())
    # per wrong row: z = L^-1 r as the forward elimination left it and w = (Y Y^T + lam n I)^-1 r, block by block against
    # float64, and where in the victim's block steps its SIMD partner went from its Gramian to its solve
    zg = rec[:, 16:128].view(np.float32).astype(np.float64)
    wg = rec[:, 128:240].view(np.float32).astype(np.float64)
    bg = rec[:, 256:368].view(np.float32).astype(np.float64)   # b block J as step J found it (all four lane groups summed)
    bp1 = rec[:, 384:496].view(np.float32).astype(np.float64)  # lane group 1's share of it
    tj = rec[:, 8:16].astype(np.int64) - int(rec[:, 2].astype(np.int64).min())
    order_by_slot = {}
    for i in range(users):
        order_by_slot.setdefault(int(slot[i]), []).append(i)
    def blocks_off(u):
        Y = V64[indx[rowPtr[u]:rowPtr[u + 1]]]
        n = len(Y)
        G = Y @ Y.T + 0.05 * n * np.eye(n)
        r = vals[rowPtr[u]:rowPtr[u + 1]].astype(np.float64)
        L = np.linalg.cholesky(G)
        z = np.linalg.solve(L, r)
        w = np.linalg.solve(G, r)
        ez = np.abs(zg[u, :n] - z) / np.abs(z).max()
        ew = np.abs(wg[u, :n] - w) / np.abs(w).max()
        nb = (n + 15) // 16
        # what block step J should find as its right-hand side: L_JJ z_J
        bt = np.zeros(16 * nb)
        for j in range(nb):
            sl = slice(16 * j, min(n, 16 * j + 16))
            bt[sl] = L[sl, sl] @ z[sl]
        eb = np.abs(np.pad(bg[u, :n], (0, 16 * nb - n)) - bt) / np.abs(bt).max()
        return ([float(ez[16 * j:16 * j + 16].max()) for j in range(nb)], [float(ew[16 * j:16 * j + 16].max()) for j in range(nb)],
                [float(eb[16 * j:16 * j + 16].max()) for j in range(nb)])
    for u in np.flatnonzero(wrong)[:16]:
        ez, ew, eb = blocks_off(u)
        partners = [j for j in order_by_slot[int(slot[u])] if j != u and t[u, 1] <= t[j, 1] <= t[u, 2]]
        when = []
        for j in partners:
            steps = tj[u, :8]
            when.append(int(np.searchsorted(steps[:7], t[j, 1], side="right") - 1) if t[j, 1] < steps[7] else 7)
        print(f"  row {u} n {int(lens[u])} err {err[u]:.1e}  z off by block " + " ".join(f"{e:.0e}" for e in ez) + "  | w off by block " +
              " ".join(f"{e:.0e}" for e in ew) + "  | b as found by block " + " ".join(f"{e:.0e}" for e in eb) + f"  | partner's Gramian ended during block step {when} (7 = back substitution)")
    sample = np.flatnonzero(~wrong)[:4]
    for u in sample:
        ez, ew, eb = blocks_off(u)
        print(f"  (right row {u}: z " + " ".join(f"{e:.0e}" for e in ez) + " | w " + " ".join(f"{e:.0e}" for e in ew) + " | b " + " ".join(f"{e:.0e}" for e in eb) + ")")
    if out:
        np.savez(out if rep == 0 else out.replace(".npz", "_1.npz"), trace=tr, err=err, lens=lens)
