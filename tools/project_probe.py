#!/usr/bin/env python3
"""Timing probe for the projection P = U^T B (100 panel passes of the pass-1 stream at k = 1000) and the k-means++ rounds behind it:
device ms of the families project / kmpp.  usage: project_probe.py [c3full|c3shard|c2]"""
import os
import sys

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
U = np.linalg.qr(np.random.default_rng(0).standard_normal((V, k)).astype(np.float32))[0].astype(np.float32)
for rep in range(3):
    hp.set_U(U)  # invalidates the projection
    hp.gram_apply(np.zeros((V, 10), np.float32)) if rep == 0 else None  # operator build outside the timed part
    hp.timing_enable(True)
    hp.timing_reset()
    rounds = -1
    try:
        rounds = hp.kmeans_init_on_projected_space(k, rng_seed=1)["rounds"]
    except Exception as e:  # (timing-only builds leave a projection the seeding cannot work with)
        print("k-means++ failed:", str(e)[:100])
    t = hp.timing_get()
    hp.timing_enable(False)
    print("projection %.1f ms, k-means++ %.1f ms (%d rounds)" % (t["project"][0], t["kmpp"][0], rounds), flush=True)
