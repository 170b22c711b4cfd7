#!/usr/bin/env python3
"""Diagnostic probe for Lloyd in span(U) at k = 1000 (config 3 on one GPU by default): times run_lloyds_on_projected_space for a list of
environment settings, checks that they return the same partition, and (third repetition, ISLE_DEBUG_HAMERLY=1) prints the active documents and
tiles per iteration.  usage: proj_probe.py [c3shard|c2|c3full] ['ENV=VAL,ENV=VAL' ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isle_amd import HotPath  # noqa: E402
from tools.synth import Corpus  # noqa: E402

SHAPES = {"c2": (50_000, 1_000_000, 200, 2024), "c3shard": (100_000, 1_250_000, 1000, 31337), "c3full": (100_000, 10_000_000, 1000, 31337)}
V, D, k, seed = SHAPES[sys.argv[1] if len(sys.argv) > 1 else "c3full"]
B = Corpus(V, D, k, seed).threshold(k, free_A=True)
hp = HotPath(0)
hp.upload_csc(V, B["vals"], B["rows"], B["offs"])
hp.compute_block_ks(k, seed=1, allow_noconv=True)
ref = None
for setting in (sys.argv[2:] or [""]):
    kv = [x.split("=") for x in setting.split(",") if x]
    for a_, b_ in kv:
        os.environ[a_] = b_
    for rep in range(3):
        if rep == 2:
            os.environ["ISLE_DEBUG_HAMERLY"] = "1"
        g = hp.kmeans_init_on_projected_space(k, rng_seed=1)
        perm = np.arange(k)
        if os.environ.get("PROBE_PERM") == "norm":  # the seeds handed over in the order of their squared norms (a relabelling: tiles = 32 consecutive labels)
            perm = np.argsort((g["C_lowd"].astype(np.float64) ** 2).sum(1), kind="stable")
            g["C_lowd"] = np.ascontiguousarray(g["C_lowd"][perm])
        hp.timing_enable(True)
        hp.timing_reset()
        t0 = time.perf_counter()
        lp = hp.run_lloyds_on_projected_space(k, g["C_lowd"])
        dt = time.perf_counter() - t0
        t = hp.timing_get()
        hp.timing_enable(False)
        lp["assign"] = perm[lp["assign"]].astype(np.uint32)
        if ref is None:
            ref = lp["assign"].copy()
        print("%-40s run_lloyds_on_projected_space: %.1f ms wall, %d iterations; device ms lloyd_proj %.1f; partition agreement with the first run %.7f" %
              (setting or "(default)", dt * 1e3, lp["iters"], t["lloyd_proj"][0], float((ref == lp["assign"]).mean())), flush=True)
    os.environ.pop("ISLE_DEBUG_HAMERLY", None)
    for a_, b_ in kv:
        os.environ.pop(a_, None)
