#!/usr/bin/env python3
"""Timing probe for the Gram apply at C2 size: device ms of pass 1 / pass 2 / operator build for a list of environment
settings (each setting re-uploads B, which rebuilds the operator).  usage: gram_probe.py ['ENV=VAL,ENV=VAL' ...]
(GRAM_PROBE_WORKLOAD=c3shard: vocab 100k, 1.25M documents instead)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isle_amd import HotPath  # noqa: E402
from tools.synth import Corpus  # noqa: E402

V, D, k, seed = 50_000, 1_000_000, 200, 2024
if os.environ.get("GRAM_PROBE_WORKLOAD") == "c3shard":  # one GPU's share of BASELINE config 3
    V, D, k, seed = 100_000, 1_250_000, 1000, 31337
if os.environ.get("GRAM_PROBE_WORKLOAD") == "c3full":  # all of BASELINE config 3
    V, D, k, seed = 100_000, 10_000_000, 1000, 31337
B = Corpus(V, D, k, seed).threshold(k, free_A=True)
hp = HotPath(0)
X = np.random.default_rng(0).standard_normal((V, 10)).astype(np.float32)
ref = None
for setting in (sys.argv[1:] or [""]):
    kv = [s.split("=") for s in setting.split(",") if s]
    for a, b in kv:
        os.environ[a] = b
    hp.timing_enable(True)
    hp.timing_reset()
    hp.upload_csc(V, B["vals"], B["rows"], B["offs"])
    Z = hp.gram_apply(X)
    t = hp.timing_get()
    op_build = t["op_build"][0] if "op_build" in t else -1.0
    if ref is None:
        ref = Z
    err = float(np.linalg.norm(Z - ref) / np.linalg.norm(ref))
    reps = int(os.environ.get("GRAM_PROBE_REPS", "20"))
    for rnd in range(int(os.environ.get("GRAM_PROBE_ROUNDS", "1"))):  # several rounds show the run-to-run spread inside one process
        hp.timing_reset()
        for _ in range(reps):
            hp.gram_apply(X)
        t = hp.timing_get()
        print("%-40s form=%d pass1 %.4f ms  pass2 %.4f ms  op_build %.1f ms (relerr vs first %.1e)" % (setting or "(default)", hp.operator_form(),
              t["gram_pass1"][0] / reps, t["gram_pass2"][0] / reps, op_build, err), flush=True)
    for a, b in kv:
        del os.environ[a]
