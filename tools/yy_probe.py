#!/usr/bin/env python3
"""Diagnostic probe for Lloyd on B at k = 1000 (C3-shard shape): runs the hot path once with ISLE_DEBUG_HAMERLY=1 so that the library
prints, per iteration, the active documents, the Yinyang group scans and the nonzeros they gather; then times the sparse Lloyd loop.
usage: yy_probe.py [c3shard|c2]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isle_amd import HotPath  # noqa: E402
from tools.synth import Corpus  # noqa: E402

SHAPES = {"c2": (50_000, 1_000_000, 200, 2024), "c3shard": (100_000, 1_250_000, 1000, 31337)}
V, D, k, seed = SHAPES[sys.argv[1] if len(sys.argv) > 1 else "c3shard"]
B = Corpus(V, D, k, seed).threshold(k, free_A=True)
hp = HotPath(0)
hp.upload_csc(V, B["vals"], B["rows"], B["offs"])
hp.compute_block_ks(k, seed=1, allow_noconv=True)
g = hp.kmeans_init_on_projected_space(k, rng_seed=1)
lp = hp.run_lloyds_on_projected_space(k, g["C_lowd"])
for rep in range(2):
    if rep == 1:
        os.environ["ISLE_DEBUG_HAMERLY"] = "1"
    hp.left_multiply_by_U(lp["C_lowd"], fetch=False)
    hp.timing_enable(True)
    hp.timing_reset()
    t0 = time.perf_counter()
    ls = hp.run_lloyds(k, fetch_centers=False)
    dt = time.perf_counter() - t0
    t = hp.timing_get()
    print("run_lloyds: %.1f ms wall, %d iterations; device ms: sparse_assign %.1f, sparse_update %.1f" %
          (dt * 1e3, ls["iters"], t["sparse_assign"][0], t["sparse_update"][0]), flush=True)
    hp.timing_enable(False)
