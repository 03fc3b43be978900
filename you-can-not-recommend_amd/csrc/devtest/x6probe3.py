import sys, numpy as np
sys.path.insert(0,'you-can-not-recommend_amd/python'); sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import ycnr_als
from ycnr_als.data import Csr
from helpers import numpy_step, row_rel_err
k=32; items=200
for n,chunk in ((64,32),(65,64),(128,64),(64,16),(33,32),(40,32),(200,64)):
    rng=np.random.default_rng(1)
    rowPtr=np.array([0,n],np.int64); indx=np.sort(rng.choice(items,n,replace=False)).astype(np.int32); vals=rng.integers(1,6,n).astype(np.float32)
    bu=Csr(1,items,rowPtr,indx,vals); U=np.zeros((1,k),np.float32); V=(rng.standard_normal((items,k))/np.sqrt(k)).astype(np.float32)
    errs=[]
    for rep in range(3):
        d=ycnr_als.AlsDevice(k,1,items,chunkRatings=chunk,flags=0); d.set_ratings('byUser',bu.rowPtr,bu.indx,bu.vals); d.set_factors('byUser',U); d.set_factors('byItem',V)
        info=d.step('byUser'); x=d.get_factors('byUser'); d.destroy()
        want,c=numpy_step(0.05,k,bu,V,U); errs.append(float(row_rel_err(x,want)[0]))
    print('n',n,'chunk',chunk,'units',info.units,'errs',errs)
