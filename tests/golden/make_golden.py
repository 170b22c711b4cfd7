#!/usr/bin/env python3
"""Generates tests/golden/tiny_fp64.npz — ground truth for the hot path on a tiny corpus, computed in
fp64 NumPy WITHOUT the oracle or the HIP library (dense eigvalsh of B B^T, brute-force k-means steps).

The reference ships no golden vectors for this path and cannot be compiled here (needs <mkl.h>), so
these vectors pin the oracle against independent mathematics, not against the reference binary
("parity unpinned").  Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools.synth import make_B  # noqa: E402


def main():
    V, D, k, seed = 600, 1500, 10, 3
    B = make_B(V, D, k, seed, L0=60.0)
    S = sp.csc_matrix((B["vals"].astype(np.float64), B["rows"], B["offs"]), shape=(V, B["D"]))
    G = (S @ S.T).toarray()
    lam, vec = np.linalg.eigh(G)
    lam, vec = lam[::-1], vec[:, ::-1]
    U = vec[:, :k]
    P = (S.T @ U)  # D x k
    rng = np.random.default_rng(0)
    X = rng.standard_normal((V, 10))
    Z = S @ (S.T @ X)
    seeds = np.sort(rng.choice(B["D"], size=k, replace=False)).astype(np.uint64)
    # min squared distance of every doc to the seed set, in the projected space
    Cs = P[seeds.astype(np.int64)]
    d2 = np.maximum(((P[:, None, :] - Cs[None, :, :]) ** 2).sum(-1), 0).min(1)
    # 3 Lloyd iterations in the projected space with isamin semantics (first index of min |dist|)
    C = Cs.copy()
    for _ in range(3):
        dist = (P ** 2).sum(1)[:, None] + (C ** 2).sum(1)[None, :] - 2 * P @ C.T
        a = np.abs(dist).argmin(1)
        for c in range(k):
            C[c] = P[a == c].mean(0) if (a == c).any() else 0.0
    # 2 Lloyd iterations in word space from the lifted centres
    Sd = S.toarray()
    Cw = U @ C.T  # V x k
    for _ in range(2):
        dist = (Sd ** 2).sum(0)[:, None] + (Cw ** 2).sum(0)[None, :] - 2 * Sd.T @ Cw
        aw = np.abs(dist).argmin(1)
        for c in range(k):
            Cw[:, c] = Sd[:, aw == c].mean(1) if (aw == c).any() else 0.0
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tiny_fp64.npz")
    np.savez_compressed(out, V=V, D=B["D"], k=k, vals=B["vals"], rows=B["rows"], offs=B["offs"],
                        lam=lam[:2 * k], U=U.astype(np.float32), X=X.astype(np.float32), Z=Z,
                        seeds=seeds, min_d2=d2, C_proj3=C, assign_proj3=a.astype(np.uint32),
                        C_word2=Cw, assign_word2=aw.astype(np.uint32), frob=float((B["vals"].astype(np.float64) ** 2).sum()))
    print("wrote", out, os.path.getsize(out), "bytes; nnz", B["nnz"])


if __name__ == "__main__":
    main()
