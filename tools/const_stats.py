import sys, numpy as np
sys.path.insert(0,'.')
from halo2_vectordb_amd import api
from oracle import oracle as O
api.init()
rng=np.random.default_rng(20260004)
a=api.quantize(rng.integers(0,219,(4,128)).astype(float)); b=api.quantize(rng.integers(0,219,(4,128)).astype(float))
g=api.wit_distance('euclidean',a,b,L=15,selectors=True)
R=O.R_MOD
vals=O.limbs_to_ints(O.fr_to_canonical(g['stream']))
m=g['const_mask'].astype(bool)
fold=np.array([min(v,R-v) for v in vals],dtype=object)
def digits(v,c=11):
    n=0;carry=0
    while v or carry:
        d=(v&((1<<c)-1))+carry; v>>=c
        if d>(1<<(c-1)): carry=1; d=(1<<c)-d
        else: carry=0
        if d: n+=1
    return n
dg=np.array([digits(int(v)) for v in fold])
nz=np.array([v!=0 for v in fold])
print('cells',len(vals),'const frac',m.mean(),'nonzero frac',nz.mean(),'const&nonzero',(m&nz).mean())
print('digits total',dg.sum(),'const digits',dg[m].sum(), 'frac',dg[m].sum()/dg.sum())
from collections import Counter
c=Counter(int(v) for v,mm in zip(fold,m) if not mm and v!=0)
print('top non-const values',[(hex(k),n) for k,n in c.most_common(12)])
