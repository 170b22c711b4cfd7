#!/usr/bin/env python3
"""Timing probe for ISLEInfer on the device: synthetic peaked model (V x k), D documents of ~100 distinct words."""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from isle_amd import HotPath  # noqa: E402

V, k, D = 50_000, 200, int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
rng = np.random.default_rng(0)
M = rng.random((V, k), dtype=np.float32) ** 8
M /= M.sum(0, keepdims=True)
lens = np.clip(rng.lognormal(np.log(100), 0.4, size=D).astype(np.int64), 20, 400)
offs = np.zeros(D + 1, np.int64)
offs[1:] = np.cumsum(lens)
rows = np.empty(offs[-1], np.uint32)
for d in range(D):
    rows[offs[d]:offs[d + 1]] = np.sort(rng.choice(V, size=lens[d], replace=False))
counts = rng.integers(1, 4, size=rows.shape[0]).astype(np.float32)
hp = HotPath(0)
hp.timing_enable(True)
for rep in range(2):
    hp.timing_reset()
    t0 = time.perf_counter()
    r = hp.infer(M, offs, rows, counts, want_weights=False)
    t1 = time.perf_counter()
    tm = hp.timing_get()["infer"][0]
    print("docs %d nnz %d: device %.1f ms (%.0f docs/s), wall %.1f ms, converged %d" % (D, rows.shape[0], tm, D / tm * 1e3, (t1 - t0) * 1e3, r["nconverged"]), flush=True)
