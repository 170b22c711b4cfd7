#!/usr/bin/env python3
"""Short probe for rocprofv3 --pmc passes: one Frobenius reduction (known byte count: 4*nnz, used to calibrate FETCH_SIZE for
4-B/lane streaming loads) and three Gram applies with b = 10.  usage: pmc_probe.py [c2|c3shard|c3full]  (default c2)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isle_amd import HotPath  # noqa: E402
from tools.synth import Corpus  # noqa: E402

SHAPES = {"c2": (50_000, 1_000_000, 200, 2024), "c3shard": (100_000, 1_250_000, 1000, 31337), "c3full": (100_000, 10_000_000, 1000, 31337)}
V, D, k, seed = SHAPES[sys.argv[1] if len(sys.argv) > 1 else "c2"]
B = Corpus(V, D, k, seed).threshold(k, free_A=True)
hp = HotPath(0)
hp.upload_csc(V, B["vals"], B["rows"], B["offs"])
print("nnz", B["nnz"], "D", B["D"], flush=True)
hp.frobenius()
X = np.random.default_rng(0).standard_normal((V, 10)).astype(np.float32)
for _ in range(3):
    hp.gram_apply(X)
