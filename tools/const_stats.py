#!/usr/bin/env python3
"""Keygen statistic: share of constant cells and of their MSM digits in a distance witness."""
import sys, numpy as np
sys.path.insert(0,'.')
from halo2_vectordb_amd import api
api.init()
rng=np.random.default_rng(20260004)
a=api.quantize(rng.integers(0,219,(4,128)).astype(float)); b=api.quantize(rng.integers(0,219,(4,128)).astype(float))
g=api.wit_distance('euclidean',a,b,L=15,selectors=True)
R=0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
canon=api.fr_to_canonical(g['stream'])
vals=[sum(int(r[i])<<(64*i) for i in range(4)) for r in canon]
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
