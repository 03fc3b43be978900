import sys, numpy as np
sys.path.insert(0,'you-can-not-recommend_amd/python'); sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import ycnr_als
from ycnr_als.data import Csr
from helpers import numpy_step, row_rel_err
for k in (16, 20, 32):
  for n in (64, 96, 100):
    items=200; rng=np.random.default_rng(1)
    rowPtr=np.array([0,n],np.int64); indx=np.sort(rng.choice(items,n,replace=False)).astype(np.int32); vals=rng.integers(1,6,n).astype(np.float32)
    bu=Csr(1,items,rowPtr,indx,vals); U=np.zeros((1,k),np.float32); V=(rng.standard_normal((items,k))/np.sqrt(k)).astype(np.float32)
    out={}
    for name,flags in (('x6',0),('f32',16)):
        d=ycnr_als.AlsDevice(k,1,items,chunkRatings=32,flags=flags); d.set_ratings('byUser',bu.rowPtr,bu.indx,bu.vals); d.set_factors('byUser',U); d.set_factors('byItem',V)
        info=d.step('byUser'); out[name]=d.get_factors('byUser'); d.destroy()
    want,c=numpy_step(0.05,k,bu,V,U)
    print(k,n,'split rows',info.splitRows,'err x6',row_rel_err(out['x6'],want)[0],'err f32',row_rel_err(out['f32'],want)[0])
