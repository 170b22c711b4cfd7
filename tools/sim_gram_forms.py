#!/usr/bin/env python3
"""Round 6, VERDICT item 1 — "price first": CPU models (no GPU) of other formulations of the Gram-apply family, on a sample of the
config-3 corpus (same generator and seed, vocab 100k, k = 1000), against the LDS-banded sliced-ELL form that runs (gl_apply_k).

Unit prices are the MEASURED ones of this kernel (profiles/r03_lds_exec_mask_microbench.log, r04_gl_apply_ablations_and_stamps.txt):
  * an LDS gather instruction costs its full cycles whatever the exec mask: 40 bytes x 64 lanes = 10 array cycles (+ conflicts: 10.8), the LDS
    serves 256 B per clock and CU; 16 waves share it;
  * one VALU instruction of a wave64 occupies its SIMD for 4 cycles, 4 SIMDs per CU;
  * a pass = walk (63-67 %) + band staging (15-18 %) + band barriers (14-19 %), the walk at ~86 % of the LDS issue bound.

 (a) nonzero-parallel tiles: a workgroup owns a document range as LDS accumulators AND a word band of X; lanes take consecutive nonzeros sorted
     by document, gather the 40-byte row, reduce runs of one document over the lanes (segmented DPP scan), the run's head adds into the
     document's LDS row.
 (b) jointly co-clustered word + document order (label-free: alternating assignment passes over B), padding of the present form under it.
 (c) the id stream read once for 20 columns in the wide products (projection, k-means++).
 (d) RANK-SORTED SLICES: a lane's G cells of a band walked in the order of their sizes (slice r = every lane's r-th largest cell), the
     accumulators permuted between bands by a network of exec-masked register swaps.

usage: sim_gram_forms.py [docs (default 128000)]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.synth import make_B  # noqa: E402

V, k = 100_000, 1000
D = int(sys.argv[1]) if len(sys.argv) > 1 else 128_000
RB = 4078
t0 = time.time()
B = make_B(V, D, k, 31337)
offs, rows = B["offs"], B["rows"].astype(np.int64)
Dn, nnz = B["D"], B["nnz"]
doc = np.repeat(np.arange(Dn), np.diff(offs))
print("corpus sample: V=%d D=%d nnz=%d (%.1f per document), generated in %.0f s" % (V, Dn, nnz, nnz / Dn, time.time() - t0))

# word order of the build: by decreasing row length (wperm); document order: by decreasing length (dperm)
wlen = np.bincount(rows, minlength=V)
wpos = np.empty(V, np.int64)
wpos[np.argsort(-wlen, kind="stable")] = np.arange(V)
NB = (V + RB - 1) // RB


def cells(word_pos, doc_order):
    """cnt[position of document][band] under a word placement (word -> position) and a document order"""
    band = word_pos[rows] // RB
    c = np.zeros((Dn, NB), np.int32)
    np.add.at(c, (doc, band), 1)
    return c[doc_order]


def slots_sliced(c, G, deal="block"):
    """padded slots / nnz of the present form: slices of 64 consecutive documents, a slice's width in a band = max over its lanes,
    in super-rounds of 4 (the last one may be a half: 2)"""
    n = (c.shape[0] // 64) * 64
    cc = c[:n].reshape(-1, 64, NB)
    mx = cc.max(1)
    sl = np.where(mx % 4 == 0, mx, np.where(mx % 4 <= 2, mx - mx % 4 + 2, mx - mx % 4 + 4))
    return sl.sum() * 64.0 / cc.sum()


def slots_rank_sorted(c, G, deal):
    """(d): a wave owns G slices (deal: which); per band every lane's G cells are sorted by size, slice r = the r-th largest of every lane"""
    n = (c.shape[0] // (64 * G)) * 64 * G
    nsl = n // 64
    cc = c[:n].reshape(nsl, 64, NB)
    if deal == "block":       # G consecutive slices per wave
        w = cc.reshape(nsl // G, G, 64, NB)
    else:                     # serpentine over quantile ranges: slice i of quantile q goes to wave i (what the build's pass 1 does)
        nw = nsl // G
        w = np.stack([cc[q * nw:(q + 1) * nw] if q % 2 == 0 else cc[q * nw:(q + 1) * nw][::-1] for q in range(G)], axis=1)
    s = -np.sort(-w, axis=1)              # per (wave, lane, band): cells in decreasing size
    mx = s.max(2)                         # (wave, rank, band)
    sl = np.where(mx % 4 == 0, mx, np.where(mx % 4 <= 2, mx - mx % 4 + 2, mx - mx % 4 + 4))
    mx0 = w.max(2)
    sl0 = np.where(mx0 % 4 == 0, mx0, np.where(mx0 % 4 <= 2, mx0 - mx0 % 4 + 2, mx0 - mx0 % 4 + 4))
    return sl.sum() * 64.0 / w.sum(), sl0.sum() * 64.0 / w.sum()


dlen = np.diff(offs)
o_len = np.argsort(-dlen, kind="stable")
c_now = cells(wpos, o_len)
pad_now = slots_sliced(c_now, 7)
print("\n== present form, pass 1 (documents by length, words by frequency, %d bands): padded slots %.3f x nnz (config 3 at 10 M documents: 2.63)" % (NB, pad_now))

# ---------------------------------------------------------------------------------------------------------------------------------
print("\n== (d) rank-sorted slices, pass 1")
for G in (4, 6, 7, 8):
    for deal in ("block", "serpentine"):
        r, r0 = slots_rank_sorted(c_now, G, deal)
        print("   G = %d, %-10s dealing: rank-sorted %.3f x nnz   (same waves, document order: %.3f)   slots x %.2f" % (G, deal, r, r0, r / r0))
# pass 2: words own lanes, document bands of 4078 documents.  A cell (word, band of 4078 documents) holds nnz_w x 4078 / D entries whatever D is,
# so the sample's bands have config 3's cell sizes (there are just fewer of them)
NB2 = (Dn + RB - 1) // RB
dpos = np.empty(Dn, np.int64)
dpos[o_len] = np.arange(Dn)
c2 = np.zeros((V, NB2), np.int32)
np.add.at(c2, (wpos[rows], dpos[doc] // RB), 1)
n2 = (V // 64) * 64
rnd = lambda m: np.where(m % 4 == 0, m, np.where(m % 4 <= 2, m - m % 4 + 2, m - m % 4 + 4))  # noqa: E731
pad2 = rnd(c2[:n2].reshape(-1, 64, NB2).max(1)).sum() * 64.0 / c2[:n2].sum()
print("   pass 2 (words by frequency own lanes, %d document bands of %d documents): present form %.3f x nnz (config 3: 3.10)" % (NB2, RB, pad2))
pad2_ratio = {}
for G in (4, 6):
    nsl = n2 // 64
    nw = nsl // G
    w = c2[:nw * G * 64].reshape(nw, G, 64, NB2)
    s = -np.sort(-w, axis=1)
    a_, b_ = rnd(s.max(2)).sum() * 64.0 / w.sum(), rnd(w.max(2)).sum() * 64.0 / w.sum()
    pad2_ratio[G] = a_ / b_
    print("      G = %d (consecutive slices): rank-sorted %.3f x nnz (word order %.3f): slots x %.2f" % (G, a_, b_, a_ / b_))

# price of (d) per (wave, band): walk + routing
for G, pad_old, pad_new in ((7, 2.63, None),):
    r, r0 = slots_rank_sorted(c_now, G, "serpentine")
    pad_new = pad_old * r / r0
    real = nnz / Dn * 64 * G / NB      # real entries per (wave, band)
    sr_old, sr_new = pad_old * real / 256, pad_new * real / 256   # super-rounds (4 slots x 64 lanes)
    lds_old, lds_new = sr_old * 43, sr_new * 43                   # LDS cycles (12 reads per super-round, 43 cycles with conflicts)
    sw = {4: 5, 5: 9, 6: 12, 7: 16, 8: 19}[G]                     # comparators of the optimal sorting network = exec-masked swap stages
    valu_route = sw * (2 + 10)                                    # v_and + v_cmp, ten v_swap_b32 (40 bytes of accumulators) per stage
    valu_old, valu_new = sr_old * 30, sr_new * 30 + valu_route
    print("   price per (wave, band) at G = %d, config 3: super-rounds %.1f -> %.1f; LDS cycles %.0f -> %.0f; VALU instructions %.0f -> %.0f (routing %d)"
          % (G, sr_old, sr_new, lds_old, lds_new, valu_old, valu_new, valu_route))
    cu_old = max(16 * lds_old, 16 * valu_old * 4 / 4)
    cu_new = max(16 * lds_new, 16 * valu_new * 4 / 4)
    print("   per CU and band (16 waves; LDS shared, 4 SIMDs x 4 cycles per instruction): max(LDS, VALU) %.0f -> %.0f cycles: walk x %.2f; the routing can run"
          " behind the band's LDS-DMA (15-18 %% of a pass waits there); with the walk at 65 %% of a pass: pass x %.2f" % (cu_old, cu_new, cu_new / cu_old, 0.35 + 0.65 * cu_new / cu_old))
    print("   id stream: %.2f -> %.2f B per nonzero of HBM traffic (+ %.2f B of routing words)" % (2 * pad_old, 2 * pad_new, 2.0 * NB * 64 / (nnz / Dn * 64 * G)))

# ---------------------------------------------------------------------------------------------------------------------------------
print("\n== (a) nonzero-parallel tiles")
lam = c_now[c_now > 0].mean()
cells_per_nnz = (c_now > 0).sum() / c_now.sum()
print("   cells (document, band) that hold entries: %.3f per nonzero (mean run of one document inside a band: %.2f nonzeros)" % (cells_per_nnz, 1 / cells_per_nnz))
lds_B = 40 + 80 * cells_per_nnz
print("   LDS bytes per real nonzero: 40 (row of X) + 80 x %.3f (read-modify-write of the document's accumulator per run) = %.1f   (present form: 40 x 2.63 = 105)" % (cells_per_nnz, lds_B))
valu = (6 * 10 * 2 + 14) / 64.0
print("   VALU per real nonzero: segmented scan of 10 floats over 64 lanes = 6 steps x 10 x (DPP move + predicated add) + 14 (ids, head flags, addresses) = %.2f instructions" % valu)
print("   (present form: 30 per super-round of 256 slots = %.2f per real nonzero)" % (30 * 2.63 / 256))
t_valu = valu * 4 / 4 * 1.006e9 / 256 / 2.4e9 * 1e3
t_lds = lds_B / 256 * 1.006e9 / 256 / 2.4e9 * 1e3
print("   config 3, per pass: VALU-bound %.2f ms, LDS-bound %.2f ms (present: pass 1 1.47 ms, pass 2 1.86 ms measured)" % (t_valu, t_lds))
docs_tile = 2048
print("   LDS split: %d accumulator rows (80 KB) leave %d rows of X per band -> %d bands; every tile stages every band: %.1f GB of staging per pass (present: 6.3 GB)"
      % (docs_tile, (160 * 1024 - docs_tile * 40) // 40, (V * 40 + 160 * 1024 - docs_tile * 40 - 1) // (160 * 1024 - docs_tile * 40),
         10_000_000 / docs_tile * V * 40 / 1e9))
print("   -> slower than the present form on the VALU alone: NOT BUILT")

# ---------------------------------------------------------------------------------------------------------------------------------
print("\n== (b) co-clustered word + document order (label-free)")
# alternate: documents to their heaviest band (then length) <-> words to the band most of their documents sit in (capacity RB per band)
word_pos = wpos.copy()
order = o_len
for it in range(3):
    c_doc = cells(word_pos, np.arange(Dn))
    heavy = c_doc.argmax(1)
    order = np.lexsort((-dlen, heavy))
    c_it = c_doc[order]
    print("   pass %d: documents by heaviest band, then length: padded slots %.3f x nnz" % (it + 1, slots_sliced(c_it, 7)))
    # words: histogram of the heavy band of their documents
    H = np.zeros((V, NB), np.int32)
    np.add.at(H, (rows, heavy[doc]), 1)
    pref = np.argsort(-H, axis=1)
    strength = H.max(1) / np.maximum(H.sum(1), 1)
    cap = np.full(NB, RB)
    cap[-1] = V - RB * (NB - 1)
    band_of = np.full(V, -1)
    for w in np.argsort(-strength, kind="stable"):
        for b in pref[w]:
            if cap[b] > 0:
                band_of[w] = b
                cap[b] -= 1
                break
    # position inside the band: by frequency
    word_pos = np.empty(V, np.int64)
    for b in range(NB):
        ws = np.flatnonzero(band_of == b)
        ws = ws[np.argsort(-wlen[ws], kind="stable")]
        word_pos[ws] = b * RB + np.arange(len(ws))
c_doc = cells(word_pos, np.arange(Dn))
heavy = c_doc.argmax(1)
order = np.lexsort((-dlen, heavy))
pad_b = slots_sliced(c_doc[order], 7)
print("   after 3 alternations: %.3f x nnz against %.3f (x %.2f): the planted topics' word sets overlap (every topic is the same Zipf law over its own"
      " order of ALL words), a document's entries outside its topic's head stay spread over all bands" % (pad_b, pad_now, pad_b / pad_now))
r, r0 = slots_rank_sorted(c_doc[order], 7, "block")
print("   (and rank-sorted slices on top of that order: %.3f)" % r)

# ---------------------------------------------------------------------------------------------------------------------------------
print("\n== (c) the id stream read once for 20 columns (wide products: projection, k-means++)")
print("   80-byte rows halve the band height (2038 rows, 50 bands): measured padding 3.68x (profiles/r05_gl_apply_item3_measurements.txt: half16 2.075 ms per 10-column pass)")
walk10, rest10 = 2.075 * 0.66, 2.075 * 0.34
print("   20 columns per pass at that height: the walk doubles its LDS reads per slot (%.2f -> %.2f ms), staging + barriers per pass stay (%.2f ms) but the"
      " accumulators double (G 7 -> 3: twice the workgroups, twice the staging): %.2f ms per 20 columns = %.2f ms per 10 against 1.47 ms today"
      % (walk10, 2 * walk10, rest10, 2 * walk10 + 2 * rest10, walk10 + rest10))
print("   HBM: ids 2 x 3.68 / 2 = 3.7 B per nonzero and 10 columns against 5.3: -30 % of a stream that is not the binding resource -> NOT BUILT")
print("   with (d): the ids of a 10-column pass fall to %.1f B per nonzero without widening the rows" % (2 * 2.63 * slots_rank_sorted(c_now, 7, 'serpentine')[0] / slots_rank_sorted(c_now, 7, 'serpentine')[1]))
