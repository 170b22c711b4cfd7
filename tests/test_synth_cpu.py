"""The synthetic-input tool restates ISLE's thresholding pre-stage (src/sparseMatrix.cpp:357-485, :1285-1361);
check it against a direct NumPy restatement of the same rule."""
import numpy as np

from tools.synth import Corpus


def numpy_threshold(V, D, counts, rows, offs, k):
    doc = np.repeat(np.arange(D), np.diff(offs))
    tokens = int(counts.sum())
    nz = int((np.diff(offs) > 0).sum())
    avg = np.float32(tokens // nz)
    sums = np.add.reduceat(counts, offs[:-1][np.diff(offs) > 0]) if nz else np.zeros(0)
    s_full = np.zeros(D, np.float32)
    s_full[np.diff(offs) > 0] = sums
    rnd = np.floor(avg * (counts / s_full[doc]) + np.float32(0.5)).astype(np.float32)  # std::round: half away from zero
    count_gr = max(1, int(np.float32(nz) / (2.0 * np.float32(k))))
    count_eq = max(1, int(np.ceil(3.0 * (1.0 / 60.0) * np.float32(nz) / np.float32(k))))
    zetas = np.ones(V, np.float32)
    for w in range(V):
        f = np.sort(rnd[(rows == w) & (rnd > 0)])[::-1]
        if len(f) == 0 or count_gr > len(f):
            continue
        zeta = f[count_gr - 1]
        while True:
            n_eq = int((f == zeta).sum())
            if n_eq < count_eq:
                zetas[w] = zeta
                break
            lower = f[f < zeta]
            if len(lower) == 0 or zeta == 1:
                zetas[w] = 1.0
                break
            zeta = lower[0]
    keep = rnd >= zetas[rows]
    return zetas, keep, doc


def test_threshold_matches_numpy_restatement():
    V, D, k = 300, 2000, 5
    c = Corpus(V, D, k, seed=4, L0=40.0)
    counts, rows, offs = c.A()
    B = c.threshold(k)
    zetas, keep, doc = numpy_threshold(V, D, counts, rows, offs, k)
    assert np.array_equal(B["zetas"], zetas)
    assert B["nnz"] == int(keep.sum())
    kept_docs = np.unique(doc[keep])
    assert np.array_equal(B["original_cols"].astype(np.int64), kept_docs)
    assert np.array_equal(B["rows"], rows[keep])
    assert np.allclose(B["vals"], np.sqrt(zetas[rows[keep]]))
    assert len(np.unique(zetas)) > 1  # the rule actually bites on this corpus


def test_generator_is_deterministic_and_shardable():
    a = Corpus(500, 400, 4, seed=9).A()
    b = Corpus(500, 400, 4, seed=9).A()
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    lo = Corpus(500, 200, 4, seed=9, doc_base=0).A()
    hi = Corpus(500, 200, 4, seed=9, doc_base=200).A()
    assert np.array_equal(np.concatenate([lo[1], hi[1]]), a[1])
    assert np.array_equal(np.concatenate([lo[0], hi[0]]), a[0])


def test_importance_sampling_keeps_rate_times_docs_and_prefers_heavy_docs():
    """sampled_threshold_and_copy (src/sparseMatrix.cpp:1365-1435): floor(rate*D)+1 docs survive (keys >= pivot),
    weighted toward documents with large sum of zeta."""
    from tools.synth import make_B
    full = make_B(1500, 4000, 8, 5)
    samp = make_B(1500, 4000, 8, 5, sample_rate=0.25)
    assert samp["D"] == int(np.float32(0.25) * np.float32(4000)) + 1
    assert np.array_equal(samp["zetas"], full["zetas"])  # thresholds come from the whole corpus
    # the sampled columns are a subset, copied verbatim
    pos = np.searchsorted(full["original_cols"], samp["original_cols"])
    assert np.array_equal(full["original_cols"][pos], samp["original_cols"])
    lens_full, lens_s = np.diff(full["offs"]), np.diff(samp["offs"])
    assert np.array_equal(lens_full[pos], lens_s)
    w_full = np.add.reduceat(full["vals"] ** 2, full["offs"][:-1])  # sum of zeta per doc
    assert w_full[pos].mean() > 1.05 * w_full.mean()
