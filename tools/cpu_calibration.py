#!/usr/bin/env python3
"""Calibration of bench.py's `cpu_baseline` ("port" = oracle/) against the reference's own CPU path ("report the ratio").

The reference cannot be run here (every TU of its hot path needs <mkl.h>); what exists are the unit costs the survey
measured with the compiled reference on THIS container's 8 cores (mid shape: V = 50 000, D = 200 000, 19.9 M nonzeros in
B, k = 200): 0.22 s per operator application (b = 10), 6.51 s / 27 = 0.241 s per k-means++ round, 0.65 s per projected-Lloyd iteration,
0.29 s per sparse-Lloyd iteration.  This script times the oracle's same four units on the same shape (same generator, seed 2024) with
8 threads and writes port / reference per unit to profiles/cpu_port_calibration.json, which bench.py quotes in `cpu_baseline`.

    OMP_NUM_THREADS=8 python tools/cpu_calibration.py
"""
import json
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

from oracle.oracle import OracleCsc, lift
from tools.synth import Corpus, effective_cpus

REFERENCE_UNIT_SECONDS = {  # BASELINE.md §2, mid column; 8 cores, MKL 2021.4
    "gram_apply_b10": 0.22,
    "kmeanspp_round": 6.51 / 27,
    "lloyd_projected_iteration": 0.65,
    "lloyd_sparse_iteration": 0.29,
}


def main():
    V, D, k, seed = 50_000, 200_000, 200, 2024
    corp = Corpus(V, D, k, seed)
    nnz_A = corp.nnz_A
    B = corp.threshold(k, free_A=True)
    o = OracleCsc(V, B["D"], B["vals"], B["rows"], B["offs"])
    rng = np.random.default_rng(0)
    X = rng.standard_normal((V, 10)).astype(np.float32)
    o.gram_apply(X)
    t = time.time()
    reps = 5
    for _ in range(reps):
        o.gram_apply(X)
    t_apply = (time.time() - t) / reps
    U, _ = np.linalg.qr(rng.standard_normal((V, k)))
    U = np.asfortranarray(U.astype(np.float32))
    t = time.time()
    kr = o.kmeanspp(U, k, seed=1, max_rounds=4)
    t_round = (time.time() - t) / max(kr["rounds"], 1)
    C0 = np.ascontiguousarray(o.project(U)[0][:k])
    t = time.time()
    o.lloyds_projected(U, C0, max_reps=1)
    ta = time.time() - t
    t = time.time()
    o.lloyds_projected(U, C0, max_reps=3)
    t_lp = max(time.time() - t - ta, 1e-9) / 2
    cen = lift(U, C0)
    t = time.time()
    o.lloyds_sparse(cen, max_reps=1)
    ta = time.time() - t
    t = time.time()
    o.lloyds_sparse(cen, max_reps=3)
    t_ls = max(time.time() - t - ta, 1e-9) / 2
    port = {"gram_apply_b10": t_apply, "kmeanspp_round": t_round, "lloyd_projected_iteration": t_lp, "lloyd_sparse_iteration": t_ls}
    out = {
        "shape": {"V": V, "D": D, "k": k, "nnz_A": int(nnz_A), "nnz_B": int(B["nnz"]), "docs_B": int(B["D"]), "generator_seed": seed},
        "threads": int(os.environ["OMP_NUM_THREADS"]), "cpus_available": effective_cpus(),
        "reference_unit_seconds": {u: round(v, 4) for u, v in REFERENCE_UNIT_SECONDS.items()},
        "reference_source": "BASELINE.md §2 (survey container, reference compiled against MKL 2021.4, 8 cores; nnz(B) = 19.9 M)",
        "port_unit_seconds": {u: round(v, 4) for u, v in port.items()},
        "port_over_reference": {u: round(port[u] / REFERENCE_UNIT_SECONDS[u], 3) for u in port},
        "note": "ratio < 1: the oracle (bench.py's cpu_baseline, kind 'port') is FASTER per unit than the reference's MKL path, i.e. the "
                "reported CPU docs/s over-states what the reference would reach; ratio > 1: slower (the baseline would be sand-bagged). "
                "The k-means++ figure of the reference includes its sequential prefix sum and draws.",
    }
    path = os.path.join(ROOT, "profiles", "cpu_port_calibration.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
