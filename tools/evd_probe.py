#!/usr/bin/env python3
"""Wall time and accuracy of the small symmetric eigensolver (isle_hip_eig_sym) at the sizes the Krylov-Schur restarts use.
usage: evd_probe.py [n ...]   (environment: ISLE_TD_CHAIN=1 selects the launch-chain tridiagonalisation)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import isle_amd  # noqa: E402

hp = isle_amd.HotPath()
rng = np.random.default_rng(0)
for n in [int(a) for a in sys.argv[1:]] or [110, 400, 512]:
    A = rng.standard_normal((n, n)).astype(np.float32)
    S = (A + A.T) / 2
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        ev, vec = hp.eig_sym(S)
        ts.append((time.perf_counter() - t) * 1e3)
    w = np.linalg.eigvalsh(S.astype(np.float64))[::-1]
    V = vec.astype(np.float64)
    print("n=%d  evd ms (5 calls): %s  eigenvalue err %.2e  orthogonality %.2e  residual %.2e" % (
        n, " ".join("%.2f" % x for x in ts), np.abs(ev - w).max() / np.abs(w).max(), np.abs(V.T @ V - np.eye(n)).max(),
        np.abs(S.astype(np.float64) @ V - V * ev).max() / np.abs(w).max()), flush=True)
