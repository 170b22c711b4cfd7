#!/usr/bin/env python3
"""CPU simulation (no GPU): padding of the sliced-ELL id stream of pass 1 (64 documents per slice, 30 word bands, 4 entries per super-round)
under different document orders — by length (what the build does), by planted topic, by heaviest band(s), random.  Result of round 3:
profiles/r03_padding_simulation.txt."""
import numpy as np, sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.synth import make_B
V,D,k=100_000,256_000,1000
B=make_B(V,D,k,31337)
offs=B["offs"]; rows=B["rows"]; planted=B["planted"].astype(np.int64)
Dn=B["D"]; nnz=B["nnz"]; RB=3412; NB=(V+RB-1)//RB
print("D",Dn,"nnz",nnz,"bands",NB)
band=(rows//RB).astype(np.int64)
doc=np.repeat(np.arange(Dn),np.diff(offs))
cnt=np.zeros((Dn,NB),np.int32)
np.add.at(cnt,(doc,band),1)
lens=cnt.sum(1)
def padding(order):
    c=cnt[order]
    n=(len(order)//64)*64
    c=c[:n].reshape(-1,64,NB)
    mx=c.max(1)               # slices x NB
    sr=(mx+3)//4
    return sr.sum()*256.0/c.sum()
o_len=np.argsort(-lens,kind='stable')
print("by length (current):", padding(o_len))
o_topic=np.lexsort((-lens,planted))
print("by planted topic, then length:", padding(o_topic))
# cheap library-side proxies
am=cnt.argmax(1)
print("by heaviest band, then length:", padding(np.lexsort((-lens,am))))
top2=np.argsort(-cnt,axis=1)[:,:2]
print("by two heaviest bands, then length:", padding(np.lexsort((-lens,top2[:,1],top2[:,0]))))
# random
print("random order:", padding(np.random.default_rng(0).permutation(Dn)))
# granularity 1 instead of 4 with current order
c=cnt[o_len][:(Dn//64)*64].reshape(-1,64,NB); print("current order, granularity 1:", c.max(1).sum()*64.0/c.sum())
c=cnt[o_topic][:(Dn//64)*64].reshape(-1,64,NB); print("topic order, granularity 1:", c.max(1).sum()*64.0/c.sum())
