#!/usr/bin/env python3
"""Diagnostic probe for Lloyd on B at k = 1000 (C3-shard shape by default): times the sparse Lloyd loop under the three forms of the
Yinyang iteration (ISLE_YY_MODE = doc | docg | group), checks that they return the same partition, and with ISLE_DEBUG_HAMERLY=1 prints
per iteration the active documents and the group scans.  usage: yy_probe.py [c3shard|c2|c3full]"""
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isle_amd import HotPath  # noqa: E402
from tools.synth import Corpus  # noqa: E402

SHAPES = {"c2": (50_000, 1_000_000, 200, 2024), "c3shard": (100_000, 1_250_000, 1000, 31337), "c3full": (100_000, 10_000_000, 1000, 31337)}
V, D, k, seed = SHAPES[sys.argv[1] if len(sys.argv) > 1 else "c3shard"]
k = int(os.environ.get("PROBE_K", k))  # e.g. 1024: rows of the group bounds are then whole cache lines
B = Corpus(V, D, k, seed).threshold(k, free_A=True)
hp = HotPath(0)
hp.upload_csc(V, B["vals"], B["rows"], B["offs"])
hp.compute_block_ks(k, seed=1, allow_noconv=True)
g = hp.kmeans_init_on_projected_space(k, rng_seed=1)
for rep in range(2):
    g = hp.kmeans_init_on_projected_space(k, rng_seed=1)
    hp.timing_enable(True)
    hp.timing_reset()
    lp = hp.run_lloyds_on_projected_space(k, g["C_lowd"])
    t = hp.timing_get()
    hp.timing_enable(False)
    print("run_lloyds_on_projected_space: %d iterations, device ms %.1f" % (lp["iters"], t["lloyd_proj"][0]), flush=True)
ref = None
# PROBE_PERM=norm | normdesc | random: the centres are handed to the lift in another ORDER (a relabelling: the Yinyang groups are eight
# consecutive labels) — the partition is the same up to that relabelling; what changes is which centres share a group
C_l = lp["C_lowd"]
perm = np.arange(k)
pm = os.environ.get("PROBE_PERM")
if pm in ("norm", "normdesc"):
    perm = np.argsort((C_l.astype(np.float64) ** 2).sum(1), kind="stable")
    if pm == "normdesc":
        perm = perm[::-1].copy()
elif pm == "random":
    perm = np.random.default_rng(5).permutation(k)
elif pm == "size":
    perm = np.argsort(-np.bincount(lp["assign"], minlength=k), kind="stable")
C_l = np.ascontiguousarray(C_l[perm])
# settings: comma-separated ENV=VAL lists; default: the by-group form with the fused filter / tightening launch and without, then by document
settings = sys.argv[2:] or ["ISLE_YY_MODE=group", "ISLE_YY_MODE=group,ISLE_YY_FUSED=0", "ISLE_YY_MODE=doc"]
for setting in settings:
    kv = [x.split("=") for x in setting.split(",") if x]
    for a_, b_ in kv:
        os.environ[a_] = b_
    for rep in range(3):
        if rep == 2:
            os.environ["ISLE_DEBUG_HAMERLY"] = "1"
        hp.left_multiply_by_U(C_l, fetch=False)
        hp.timing_enable(True)
        hp.timing_reset()
        t0 = time.perf_counter()
        ls = hp.run_lloyds(k, fetch_centers=False)
        dt = time.perf_counter() - t0
        t = hp.timing_get()
        hp.timing_enable(False)
        ls["assign"] = perm[ls["assign"]].astype(np.uint32)  # back to the labels of Lloyd in span(U)
        if ref is None:
            ref = ls["assign"].copy()
        print("%-40s run_lloyds: %.1f ms wall, %d iterations; device ms: sparse_assign %.1f, sparse_update %.1f; partition equal to the first run's: %s (%d documents differ; crc %08x; Lloyd in span(U): %d iterations, crc %08x)" %
              (setting, dt * 1e3, ls["iters"], t["sparse_assign"][0], t["sparse_update"][0], bool(np.array_equal(ref, ls["assign"])), int((ref != ls["assign"]).sum()),
               zlib.crc32(ls["assign"].tobytes()), lp["iters"], zlib.crc32(lp["assign"].tobytes())), flush=True)
    os.environ.pop("ISLE_DEBUG_HAMERLY", None)
    for a_, b_ in kv:
        os.environ.pop(a_, None)
