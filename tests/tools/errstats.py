# Row-wise forward error of one byUser step against float64, for the library named in YCNR_ALS_LIB:
# distribution of err / (cond * kappa_b * eps32), the quantity the parity tests bound by 8.
import sys, numpy as np
sys.path.insert(0, 'you-can-not-recommend_amd/python'); sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import ycnr_als
from ycnr_als.data import Csr
from helpers import numpy_step, row_rel_err
k = int(sys.argv[1]) if len(sys.argv) > 1 else 100
users, items = 6000, 3000
rng = np.random.default_rng(5)
lens = np.clip(rng.lognormal(np.log(90), 0.9, users).astype(np.int64), 1, 2500)
rowPtr = np.zeros(users + 1, np.int64); np.cumsum(lens, out=rowPtr[1:])
indx = np.concatenate([np.sort(rng.choice(items, n, replace=False)) for n in lens]).astype(np.int32)
vals = rng.integers(1, 11, rowPtr[-1]).astype(np.float32)
bu = Csr(users, items, rowPtr, indx, vals)
V = (rng.standard_normal((items, k)) / np.sqrt(k)).astype(np.float32); U = np.zeros((users, k), np.float32)
d = ycnr_als.AlsDevice(k, users, items); d.set_ratings('byUser', bu.rowPtr, bu.indx, bu.vals); d.set_factors('byUser', U); d.set_factors('byItem', V)
d.step('byUser'); got = d.get_factors('byUser'); d.destroy()
want, amp = numpy_step(0.05, k, bu, V, U)
err = row_rel_err(got, want) / (amp * np.finfo(np.float32).eps)
print('k', k, 'rows', users, 'err / (cond kappa eps32): median %.3f  p99 %.3f  max %.3f   (tests allow 8)' % (np.median(err), np.quantile(err, 0.99), err.max()))
print('plain relative error: median %.3g max %.3g' % (np.median(row_rel_err(got, want)), row_rel_err(got, want).max()))
