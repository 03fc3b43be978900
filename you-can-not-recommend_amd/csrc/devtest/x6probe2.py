import sys, numpy as np
sys.path.insert(0,'you-can-not-recommend_amd/python'); sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import ycnr_als
from ycnr_als.data import Csr
from helpers import numpy_step, row_rel_err
k=32; items=200
for case in ('full','blk0only','blk1only','n32'):
    rng=np.random.default_rng(1); n=32 if case=='n32' else 64
    rowPtr=np.array([0,n],np.int64); indx=np.sort(rng.choice(items,n,replace=False)).astype(np.int32); vals=rng.integers(1,6,n).astype(np.float32)
    bu=Csr(1,items,rowPtr,indx,vals); U=np.zeros((1,k),np.float32); V=(rng.standard_normal((items,k))/np.sqrt(k)).astype(np.float32)
    if case=='blk0only': V[:,16:]=0
    if case=='blk1only': V[:,:16]=0
    for rep in range(2):
        d=ycnr_als.AlsDevice(k,1,items,chunkRatings=32,flags=0); d.set_ratings('byUser',bu.rowPtr,bu.indx,bu.vals); d.set_factors('byUser',U); d.set_factors('byItem',V)
        d.step('byUser'); x=d.get_factors('byUser'); d.destroy()
        want,c=numpy_step(0.05,k,bu,V,U)
        print(case, rep, 'err', row_rel_err(x,want)[0], 'err blk0', np.abs(x[0,:16]-want[0,:16]).max(), 'blk1', np.abs(x[0,16:]-want[0,16:]).max())
