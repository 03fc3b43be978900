# One byUser step on a c3-like problem with the float32-MFMA fused kernel and with the bf16x6 /
# LDS-DMA one, both compared with a float64 numpy solve of the rows where they disagree most.
import os, sys, numpy as np
sys.path.insert(0, 'you-can-not-recommend_amd/python'); sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import ycnr_als
from ycnr_als.data import Csr
k = int(sys.argv[1]) if len(sys.argv) > 1 else 64
users = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
items = 20000
rng = np.random.default_rng(3)
lens = np.clip((rng.lognormal(np.log(100), 0.9, users)).astype(np.int64), 1, 4000)
rowPtr = np.zeros(users + 1, np.int64); np.cumsum(lens, out=rowPtr[1:])
indx = np.empty(rowPtr[-1], np.int32)
for u in range(users):
    indx[rowPtr[u]:rowPtr[u + 1]] = np.sort(rng.choice(items, lens[u], replace=False))
vals = rng.integers(1, 11, rowPtr[-1]).astype(np.float32)
bu = Csr(users, items, rowPtr, indx, vals)
V = (rng.standard_normal((items, k)) / np.sqrt(k)).astype(np.float32)
U = np.zeros((users, k), np.float32)
out = {}
for name, env in (('f32', '1'), ('x6d', '')):
    if env: os.environ['YCNR_NO_FUSED_X6D'] = env
    else: os.environ.pop('YCNR_NO_FUSED_X6D', None)
    d = ycnr_als.AlsDevice(k, users, items, flags=0)
    d.set_ratings('byUser', bu.rowPtr, bu.indx, bu.vals); d.set_factors('byUser', U); d.set_factors('byItem', V)
    info = d.step('byUser'); out[name] = d.get_factors('byUser'); d.destroy()
    print(name, 'fusedRows', info.fusedRows, 'dualRows', info.dualRows, 'splitRows', info.splitRows)
diff = np.abs(out['f32'].astype(np.float64) - out['x6d']).max(axis=1)
worst = np.argsort(-diff)[:12]
print('rows differing > 1e-3:', int((diff > 1e-3).sum()), 'of', users)
V64 = V.astype(np.float64)
for u in worst:
    sl = slice(rowPtr[u], rowPtr[u + 1]); Y = V64[indx[sl]]; r = vals[sl].astype(np.float64)
    A = Y.T @ Y + 0.05 * len(r) * np.eye(k); x = np.linalg.solve(A, Y.T @ r)
    e = lambda z: np.abs(z - x).max() / np.abs(x).max()
    print('row', int(u), 'n', int(lens[u]), 'n%32', int(lens[u] % 32), 'diff', float(diff[u]), 'err f32', e(out['f32'][u]), 'err x6d', e(out['x6d'][u]), 'cond', np.linalg.cond(A))
bad = diff > 1e-3
if bad.any():
    print('n of differing rows: min', lens[bad].min(), 'max', lens[bad].max(), 'n%32 histogram', np.bincount(lens[bad] % 32, minlength=32).tolist())
