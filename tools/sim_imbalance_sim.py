#!/usr/bin/env python3
"""CPU simulation (no GPU): how level the waves of a pass-1 workgroup are per band — sum over bands of the largest super-round count among
the 16 waves against the sum of the means (1.11 - 1.14: what the barrier per band costs), and the waves' totals (the systematic part).
DESIGN.md section 12."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.synth import make_B
V,D,k=100_000,160_000,1000
B=make_B(V,D,k,31337)
offs=B["offs"]; rows=B["rows"].astype(np.int64); Dn=B["D"]
RB=4078; NB=(V+RB-1)//RB
doc=np.repeat(np.arange(Dn),np.diff(offs))
cnt=np.zeros((Dn,NB),np.int32); np.add.at(cnt,(doc,rows//RB),1)
lens=cnt.sum(1); order=np.argsort(-lens,kind='stable')
nslice=Dn//64
c=cnt[order][:nslice*64].reshape(nslice,64,NB)
sr=(c.max(1)+3)//4                     # slices x bands: super-rounds
def report(G, deal):
    n=nslice//G          # waves
    wpw=16; nwg=n//wpw
    so=np.zeros((n,G),int)
    for g in range(G):
        for wv in range(n):
            so[wv,g]=deal(g,wv,n)
    wave_band=sr[so].sum(1)            # waves x bands
    tot=0; mx=0; wtot_ratio=[]
    for j in range(nwg):
        ws=[j+w*nwg for w in range(wpw)]
        wb=wave_band[ws]               # 16 x NB
        tot+=wb.mean(0).sum(); mx+=wb.max(0).sum()
        t=wb.sum(1); wtot_ratio.append(t.max()/t.mean())
    return mx/tot, np.mean(wtot_ratio)
serp=lambda g,wv,n: ((g+1)*n-1-wv) if (g&1) else (g*n+wv)
for G in (4,5,7):
    r,wt=report(G,serp)
    print("G=%d current dealing: sum_b max_w / sum_b mean_w = %.3f ; wave totals max/mean within a WG = %.3f"%(G,r,wt))
