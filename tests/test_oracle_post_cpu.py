"""CPU: the downstream-stage oracle (oracle/isle_post_oracle.cpp) against an independent brute-force NumPy statement of
the same rules (src/sparseMatrix.cpp:491-524, :573-595, :597-838; src/trainer.cpp:1116-1167) on a tiny corpus."""
import numpy as np

from oracle import oracle as O


def _tiny(seed=0, V=40, D=300, k=4):
    rng = np.random.default_rng(seed)
    cols, cl = [], rng.integers(0, k, size=D).astype(np.int32)
    cl[rng.random(D) < 0.1] = -1                     # documents that dropped out of B
    for d in range(D):
        t = cl[d] if cl[d] >= 0 else int(rng.integers(0, k))
        n = int(rng.integers(3, 12))
        p = np.full(V, 1.0)
        p[t * (V // k):(t + 1) * (V // k)] = 8.0     # topic-specific words
        r = np.sort(rng.choice(V, size=n, replace=False, p=p / p.sum())).astype(np.uint32)
        cols.append((r, rng.integers(1, 6, size=n).astype(np.float32)))
    offs = np.zeros(D + 1, np.int64)
    offs[1:] = np.cumsum([len(r) for r, _ in cols])
    rows = np.concatenate([r for r, _ in cols])
    cnt = np.concatenate([c for _, c in cols])
    return V, D, k, offs, rows, cnt, cl


def test_post_oracle_matches_bruteforce():
    V, D, k, offs, rows, cnt, cl = _tiny()
    avg = float(int(cnt.sum()) // D)
    nv = O.post_normalize(offs, cnt, avg)
    # normalisation
    for d in (0, 7, D - 1):
        s, e = offs[d], offs[d + 1]
        np.testing.assert_array_equal(nv[s:e], np.float32(avg) * (cnt[s:e] / np.float32(cnt[s:e].sum(dtype=np.float32))))
    dense = np.zeros((V, D), np.float32)
    for d in range(D):
        dense[rows[offs[d]:offs[d + 1]], d] = nv[offs[d]:offs[d + 1]]
    r = 5
    thr = O.post_catch_thresholds(V, offs, rows, nv, cl, k, r)
    for t in range(k):
        docs = np.nonzero(cl == t)[0]
        for w in range(V):
            f = np.sort(dense[w, docs][dense[w, docs] > 0])[::-1]
            if len(docs) == 0:
                want = 0.0
            elif len(f) > r:
                want = f[r - 1]
            elif r >= len(docs) and len(f) == len(docs):
                want = f.min()
            else:
                want = 0.0
            assert thr[w, t] == np.float32(want), (w, t)
    ct = O.post_find_catchwords(thr)
    for w in range(V):
        t = int(np.argmax(thr[w]))
        others = np.delete(thr[w], t)
        want = t if (float(thr[w, t]) > 1.1 * others.astype(np.float64)).all() else -1
        assert ct[w] == want
    rank = 10
    tm = O.post_topic_model(V, offs, rows, nv, cl, ct, k, rank)
    # document-topic sums, top-two topics
    sums = np.zeros((D, k), np.float32)
    for d in range(D):
        for i in range(offs[d], offs[d + 1]):
            if ct[rows[i]] >= 0:
                sums[d, ct[rows[i]]] = np.float32(sums[d, ct[rows[i]]] + nv[i])
    dd, tt = np.nonzero(sums)
    np.testing.assert_array_equal(tm["dts_doc"], dd.astype(np.uint64))
    np.testing.assert_array_equal(tm["dts_topic"], tt.astype(np.uint32))
    np.testing.assert_array_equal(tm["dts_val"], sums[dd, tt])
    mthr = np.zeros(k, np.float32)
    for t in range(k):
        v = np.sort(sums[:, t][sums[:, t] != 0])[::-1]
        if (ct == t).any() and len(v) >= rank:
            mthr[t] = v[rank - 1]
    np.testing.assert_array_equal(tm["model_threshold"], mthr)
    M = np.zeros((V, k), np.float64)
    for d in range(D):
        for t in range(k):
            if sums[d, t] != 0 and sums[d, t] > mthr[t]:
                M[:, t] += dense[:, d]
        if cl[d] >= 0:
            M[:, cl[d]] += dense[:, d]
    M /= np.abs(M).sum(0, keepdims=True)
    np.testing.assert_allclose(tm["model"], M, rtol=1e-5, atol=1e-9)
    for d in range(D):
        nzt = np.nonzero(sums[d])[0]
        if len(nzt) >= 2:
            order = sorted(nzt, key=lambda t: (-sums[d, t], t))
            assert (tm["top1"][d], tm["top2"][d]) == (order[0], order[1])
        else:
            assert tm["top1"][d] == -1 and tm["top2"][d] == -1
    # edge topics: most frequent (top1, top2) pairs, 0.7 / 0.3 mix
    pairs, edge = O.post_edge_topics(tm["model"], tm["top1"], tm["top2"], 3)
    from collections import Counter
    cntr = Counter((int(a), int(b)) for a, b in zip(tm["top1"], tm["top2"]) if a >= 0)
    best = sorted(cntr.items(), key=lambda kv: (-kv[1], kv[0]))[:3]
    assert [(int(p), int(q), int(n)) for p, q, n in pairs] == [(p, q, n) for (p, q), n in best]
    for e, (p, q, _) in enumerate(pairs):
        np.testing.assert_allclose(edge[:, e], 0.7 * tm["model"][:, p] + 0.3 * tm["model"][:, q], rtol=1e-6)


def test_rank_formulas():
    # src/trainer.cpp:579-583 and src/sparseMatrix.cpp:720 at BASELINE config 2
    assert O.catchword_rank(1_000_000, 200) == 833
    assert O.model_rank_threshold(1_000_000, 200) == 12500
    assert O.catchword_rank(1_000_000, 200, sample_rate=0.1) == 83
