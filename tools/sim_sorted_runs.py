#!/usr/bin/env python3
"""CPU simulation (no GPU): padded slots of the two id streams of the LDS-banded Gram apply if every lane visited its G items of a band
in the order of their entry counts in THAT band (largest first) instead of in a fixed order — run j of a (wave, band) then holds every
lane's j-th largest cell and is padded to the largest of those, not to the largest cell of a fixed group of 64 items.  A lane's sums of
a run go to the accumulator of whichever item it was (one masked add per accumulator at the end of the run).

    python tools/sim_sorted_runs.py [docs]     (vocab 100 000, k = 1000: the geometry of config 3; bands of 4078 rows)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.synth import make_B

V, D, k = 100_000, int(sys.argv[1]) if len(sys.argv) > 1 else 512_000, 1000
RB = 4078
B = make_B(V, D, k, 31337)
offs, rows = B["offs"], B["rows"].astype(np.int64)
Dn, nnz = B["D"], B["nnz"]
doc = np.repeat(np.arange(Dn), np.diff(offs))
print("docs", Dn, "nnz", nnz)


def slots(c, G, scope, gran):
    """c: items x bands counts in the order the build visits them (64 consecutive items = a slice; a wave owns G slices: here simply G
    consecutive slices).  scope: items of a lane sorted together (1 = today's fixed order, G = all of a lane's items).  Returns padded slots."""
    n = (c.shape[0] // (64 * G)) * 64 * G
    NBn = c.shape[1]
    x = c[:n].reshape(-1, G, 64, NBn)  # wave, group, lane, band
    if scope > 1:
        x = x.reshape(-1, G // scope, scope, 64, NBn)
        x = -np.sort(-x, axis=2)  # per lane and band: the items of a scope by decreasing count
        x = x.reshape(-1, G, 64, NBn)
    mx = x.max(2)  # wave, run, band
    sr = (mx + gran - 1) // gran
    return float(sr.sum()) * gran * 64, float(x.sum())


# pass 1: items = documents by decreasing length, bands of words
NB1 = (V + RB - 1) // RB
c1 = np.zeros((Dn, NB1), np.int32)
np.add.at(c1, (doc, rows // RB), 1)
c1 = c1[np.argsort(-np.diff(offs), kind="stable")]
# pass 2: items = words by decreasing row length, bands of documents (positions in the length order)
dpos = np.empty(Dn, np.int64)
dpos[np.argsort(-np.diff(offs), kind="stable")] = np.arange(Dn)
NB2 = (Dn + RB - 1) // RB
c2 = np.zeros((V, NB2), np.int32)
np.add.at(c2, (rows, dpos[doc] // RB), 1)
c2 = c2[np.argsort(-c2.sum(1), kind="stable")]
for name, c, Gs in (("pass 1", c1, (8, 7, 4)), ("pass 2", c2, (8, 6, 4))):
    for G in Gs:
        for gran in (4, 2):
            base, real = slots(c, G, 1, gran)
            line = "%s G=%d granularity %d: fixed order %.3f x" % (name, G, gran, base / real)
            for scope in (2, 4, 8):
                if G % scope == 0:
                    s, _ = slots(c, G, scope, gran)
                    line += " | sorted by %d: %.3f x (%.0f %% of the slots)" % (scope, s / real, 100 * s / base)
            if G not in (2, 4, 8):
                s, _ = slots(c, G, G, gran)
                line += " | sorted by %d: %.3f x (%.0f %%)" % (G, s / real, 100 * s / base)
            print(line)
