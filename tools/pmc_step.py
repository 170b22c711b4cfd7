#!/usr/bin/env python3
"""One hot-path pass of config 3 (10 M documents, k = 1000) for the counter passes of tools/r05_steppmc.sh: upload, block Krylov-Schur,
k-means++, both Lloyd loops; nothing else (no accuracy leg, no timing)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isle_amd import HotPath  # noqa: E402
from tools.synth import Corpus  # noqa: E402

V, D, k, seed = 100_000, 10_000_000, 1000, 31337
B = Corpus(V, D, k, seed).threshold(k, free_A=True)
hp = HotPath(0)
hp.upload_csc(V, B["vals"], B["rows"], B["offs"])
hp.compute_block_ks(k, seed=1, allow_noconv=True)
g = hp.kmeans_init_on_projected_space(k, rng_seed=1)
lp = hp.run_lloyds_on_projected_space(k, g["C_lowd"])
hp.left_multiply_by_U(lp["C_lowd"], fetch=False)
ls = hp.run_lloyds(k, fetch_centers=False)
print("iterations", lp["iters"], ls["iters"], flush=True)
