"""Timing probe: isle_hip_upload_csc_u32 of all of BASELINE config 3 (8.1 GB from pageable host memory, validation on the device included).
Measured in round 4: 0.155 s = 52.9 GB/s — the link's rate; a pinned staging pipeline has nothing to add.  usage: python tools/upload_probe.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from isle_amd import HotPath
from tools.synth import Corpus
V, D, k, seed = 100_000, 10_000_000, 1000, 31337
B = Corpus(V, D, k, seed).threshold(k, free_A=True)
hp = HotPath(0)
for rep in range(3):
    t0 = time.perf_counter()
    hp.upload_csc(V, B["vals"], B["rows"], B["offs"])
    dt = time.perf_counter() - t0
    nbytes = B["vals"].nbytes + B["rows"].nbytes + B["offs"].nbytes
    print("upload %.3f s  %.1f GB  %.1f GB/s" % (dt, nbytes / 1e9, nbytes / 1e9 / dt), flush=True)
