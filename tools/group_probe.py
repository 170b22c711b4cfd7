"""Timing probe (not a test): do the Yinyang groups (8 centres) of Lloyd on B and the bound tiles (32 centres) of the projected Lloyd
prune better when centres that lie close together share a group?  The centres are handed over in a permuted order (labels permute with
them); the grouping is a balanced k-means on the leading coordinates.  usage: group_probe.py c2|c3shard"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isle_amd import HotPath
from tools.synth import Corpus

def group_perm(C, gsize, q=32, iters=5):
    """C: k x dims (centre = row).  Returns perm with perm[new] = old; consecutive runs of gsize new indices are spatial groups."""
    k = C.shape[0]
    X = C[:, :min(q, C.shape[1])].astype(np.float64)
    G = (k + gsize - 1) // gsize
    cen = X[(np.arange(G) * k) // G].copy()
    for _ in range(iters):
        d = ((X[:, None, :] - cen[None, :, :]) ** 2).sum(-1)
        a = d.argmin(1)
        for g in range(G):
            if (a == g).any(): cen[g] = X[a == g].mean(0)
    d = ((X[:, None, :] - cen[None, :, :]) ** 2).sum(-1)
    cap = np.full(G, gsize); cap[-1] = k - (G - 1) * gsize
    order = np.argsort(d, axis=None, kind="stable")
    grp = np.full(k, -1)
    left = k
    for f in order:
        i, g = divmod(int(f), G)
        if grp[i] < 0 and cap[g] > 0:
            grp[i] = g; cap[g] -= 1; left -= 1
            if left == 0: break
    return np.concatenate([np.flatnonzero(grp == g) for g in range(G)])

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
V, D, k, seed = {"c2": (50_000, 1_000_000, 200, 2024), "c3shard": (100_000, 1_250_000, 1000, 31337)}[wl]
c = Corpus(V, D, k, seed); B = c.threshold(k, free_A=True)
hp = HotPath(0); hp.upload_csc(V, B["vals"], B["rows"], B["offs"])
r = hp.compute_block_ks(k, allow_noconv=True)
g = hp.kmeans_init_on_projected_space(k, rng_seed=1)
ident = np.arange(k)
t0 = time.time(); p32 = group_perm(g["C_lowd"], 32); t32 = time.time() - t0
for label, perm in [("projected Lloyd, centres in seeding order", ident), ("projected Lloyd, tiles of 32 close centres", p32)]:
    for rep in range(2):
        hp.timing_enable(1); hp.timing_reset()
        lp = hp.run_lloyds_on_projected_space(k, g["C_lowd"][perm])
        tm = hp.timing_get(); hp.timing_enable(0)
    sizes = np.sort(np.bincount(lp["assign"], minlength=k))
    print("%-52s device lloyd_proj %.1f ms, iters %d, sizes hash %d" % (label, tm["lloyd_proj"][0], lp["iters"], int((sizes * np.arange(k)).sum())), flush=True)
lp = hp.run_lloyds_on_projected_space(k, g["C_lowd"])
t0 = time.time(); p8 = group_perm(lp["C_lowd"], 8); t8 = time.time() - t0
print("grouping on the host (numpy): tiles of 32 %.3f s, groups of 8 %.3f s" % (t32, t8))
for label, perm in [("Lloyd on B, centres in seeding order", ident), ("Lloyd on B, groups of 8 close centres", p8)]:
    for rep in range(2):
        hp.left_multiply_by_U(lp["C_lowd"][perm], fetch=False)
        hp.timing_enable(1); hp.timing_reset()
        ls = hp.run_lloyds(k, fetch_centers=False)
        tm = hp.timing_get(); hp.timing_enable(0)
    sizes = np.sort(np.bincount(ls["assign"], minlength=k))
    print("%-52s device sparse_assign %.1f ms, iters %d, sizes hash %d" % (label, tm["sparse_assign"][0], ls["iters"], int((sizes * np.arange(k)).sum())), flush=True)
