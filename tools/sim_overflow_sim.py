#!/usr/bin/env python3
"""CPU simulation (no GPU): padded slots of pass 1 if a slice were cut at less than its longest lane and the excess entries taken another
way (cost ratio = price of such an entry in lane-slots), for bands of 3397 and 4078 rows.  Not built: at the price a gather from global
memory has (ratio ~32) it saves a fifth of the slots for 0.7 % of the entries, and the epilogue that gathers them costs half of that."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.synth import make_B
V,D,k=100_000,128_000,1000
B=make_B(V,D,k,31337)
offs=B["offs"]; rows=B["rows"].astype(np.int64); Dn=B["D"]
doc=np.repeat(np.arange(Dn),np.diff(offs))
lens=np.diff(offs)
order=np.argsort(-lens,kind='stable')
for RB in (3397,4080):
    NB=(V+RB-1)//RB
    cnt=np.zeros((Dn,NB),np.int32)
    np.add.at(cnt,(doc,rows//RB),1)
    c=cnt[order][:(Dn//64)*64].reshape(-1,64,NB)      # slices x lanes x bands
    nreal=c.sum()
    mx=c.max(1)
    n_full=(mx+3)//4
    print("RB",RB,"bands",NB,"padding full: %.3f"%(n_full.sum()*256/nreal))
    for ratio in (8,16,32,64):   # cost of an overflow entry in units of one lane-slot... c_ov / c_slot
        best_n=n_full.copy(); best_cost=n_full*256.0; best_ov=np.zeros_like(n_full)
        for dec in range(1,8):
            n=np.maximum(n_full-dec,0)
            ov=np.maximum(c-4*n[:,None,:],0).sum(1)
            cost=n*256.0+ov*ratio
            better=cost<best_cost
            best_n=np.where(better,n,best_n); best_ov=np.where(better,ov,best_ov); best_cost=np.where(better,cost,best_cost)
        print("  ratio %3d: slots %.3f x nnz (%.1f%% of full), overflow %.2f%% of entries; per-lane overflow per doc: mean %.2f max-of-64 mean %.1f"%(
            ratio,best_n.sum()*256/nreal,100*best_n.sum()/n_full.sum(),100*best_ov.sum()/nreal,
            0,0))
