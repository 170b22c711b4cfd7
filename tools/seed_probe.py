#!/usr/bin/env python3
"""Diagnostic: one hot-path pass per given seed at config 3 on one GPU (the seeds bench.py's timed steps use are 1 + step), with the
per-phase wall times and, under ISLE_DEBUG_HAMERLY=1, the library's per-iteration counts.  usage: seed_probe.py [c3full|c3shard] seed ..."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isle_amd import HotPath  # noqa: E402
from tools.synth import Corpus  # noqa: E402

SHAPES = {"c3shard": (100_000, 1_250_000, 1000, 31337), "c3full": (100_000, 10_000_000, 1000, 31337)}
V, D, k, seed = SHAPES[sys.argv[1]]
B = Corpus(V, D, k, seed).threshold(k, free_A=True)
hp = HotPath(0)
hp.upload_csc(V, B["vals"], B["rows"], B["offs"])
for s in [int(x) for x in sys.argv[2:]]:
    t = [time.perf_counter()]
    hp.compute_block_ks(k, seed=s, allow_noconv=True)
    t.append(time.perf_counter())
    g = hp.kmeans_init_on_projected_space(k, rng_seed=s)
    t.append(time.perf_counter())
    lp = hp.run_lloyds_on_projected_space(k, g["C_lowd"])
    t.append(time.perf_counter())
    hp.left_multiply_by_U(lp["C_lowd"], fetch=False)
    ls = hp.run_lloyds(k, fetch_centers=False)
    t.append(time.perf_counter())
    print("seed %d: block_ks %.0f ms, k-means++ %.0f ms, Lloyd projected %.0f ms (%d it), lift + Lloyd on B %.0f ms (%d it)" %
          (s, 1e3 * (t[1] - t[0]), 1e3 * (t[2] - t[1]), 1e3 * (t[3] - t[2]), lp["iters"], 1e3 * (t[4] - t[3]), ls["iters"]), flush=True)
