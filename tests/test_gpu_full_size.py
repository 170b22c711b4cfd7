"""GPU parity at FULL size: BASELINE configs[1] as bench.py runs it (vocab 50k, 1M documents, 99M nonzeros, k = 200) and a
one eighth of configs[2] (vocab 100k = 30 word bands, 1.25M documents, 126M nonzeros, k = 1000, ncv = 2010).

The fixtures of test_gpu_big_k.py hold k = 200 / 1000 on reduced shapes (V <= 8000, D <= 60000).  Shapes of this size select code they
never reach: several items per lane and several rounds of workgroups in pass 1 of the Gram apply, document-band columns of more than
12 bands in pass 2, the thin sparse route of the k-means++ rounds, the D x k x k first assignment through the 16-wave GEMM, the
group-ordered Yinyang iteration with its pair lists, 30-band panels of the k-wide products.

No fixture can hold a result of this size, so the checks are the ones bench.py's untimed accuracy leg makes, under asserts:
  * sigma: |sigma - sigma_true| / sigma <= ||A u - lambda u|| / (2 lambda) with A u computed by the CPU ORACLE's operator on Ritz pairs
    spread over the spectrum — a rigorous bound a self-consistent but wrong HIP operator cannot pass (contract: 1e-4);
  * k-means: on a RANDOM sample of the documents the HIP path and the oracle run k-means++ (oracle's seeds injected), both Lloyd loops
    and the lift from the same U: partitions >= 99.9 % equal, iteration counts equal;
  * the whole-corpus run: every cluster non-empty, agreement with the planted topics inside the range the reference itself reaches on
    this kind of corpus (0.81 - 0.87).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SIGMA_TOL = 1e-4


@pytest.fixture
def hp():
    """A context of its own per full-size test (shadows the session's): a context keeps the largest buffers it ever needed — config 3 leaves
    ~200 GB of projections, products and scratch behind, and config 5 on top of that does not fit the 288 GB (first seen in round 6: out of memory
    in the topic model's scratch when the whole suite ran in one process) — and isle_hip_destroy is what releases them."""
    from isle_amd import HotPath
    h = HotPath(0)
    yield h
    h.close()


def _columns(B, cols):
    offs = B["offs"]
    lens = (offs[cols + 1] - offs[cols]).astype(np.int64)
    so = np.zeros(len(cols) + 1, np.int64)
    np.cumsum(lens, out=so[1:])
    idx = np.repeat(offs[cols] - so[:-1], lens) + np.arange(so[-1], dtype=np.int64)
    return dict(vals=B["vals"][idx], rows=B["rows"][idx], offs=so)


def _default_route_against_the_oracle(hp, B, V, k, U, g, lp):
    """The DEFAULT route at full size against the oracle, step by step (a sample uploaded on its own selects the small-shape routes instead):
    whatever the library assigns in iteration n + 1 of a loop must be the oracle's arg-min against the centres the library itself returned
    after n updates — iteration 1 of Lloyd in span(U) is the hand-over from the k-means++ rounds (ISLE_KMPP_TRACK left on), iteration 5 a
    bounded one (tile bounds, movers, sums kept up to date); iteration 1 of Lloyd on B is the two-term product through the projection (read by
    LDS-DMA, Yinyang groups by norm).  hp holds B with U installed; g = the k-means++ result, lp = the finished Lloyd in span(U)."""
    from oracle.oracle import OracleCsc, lift
    cols_a = np.sort(np.random.default_rng(9).choice(B["D"], min(100_000, B["D"]), replace=False))
    Ba = _columns(B, cols_a)
    oa = OracleCsc(V, len(cols_a), Ba["vals"], Ba["rows"], Ba["offs"])

    def projected(n):
        g_ = hp.kmeans_init_on_projected_space(k, inject_seeds=g["seeds"])  # the rounds' tracked state again
        assert np.array_equal(g_["C_lowd"].view(np.uint32), g["C_lowd"].view(np.uint32))
        return hp.run_lloyds_on_projected_space(k, g_["C_lowd"], max_reps=n)

    U64 = None

    def only_near_ties(got, want, C_lowd):
        """Where the two assignments differ, the two centres must be equidistant to the document up to fp32 dot-product noise (distances in fp64
        from the lifted centres U C^T); at most 3 documents in 10 000 may be such near-ties (measured: 0 - 1.3)."""
        nonlocal U64
        bad = np.flatnonzero(got != want)
        assert len(bad) <= 3e-4 * len(want), len(bad)
        if U64 is None:
            U64 = U.astype(np.float64)
        cmax = float((C_lowd.astype(np.float64) ** 2).sum(1).max())
        for i in bad[:40]:
            lo, hi = Ba["offs"][i], Ba["offs"][i + 1]
            b = np.zeros(V)
            b[Ba["rows"][lo:hi]] = Ba["vals"][lo:hi]
            d = [float(((b - U64 @ C_lowd[c].astype(np.float64)) ** 2).sum()) for c in (int(got[i]), int(want[i]))]
            # E = 1e-4 (|b|^2 + max |c|^2): the error either side's fp32 evaluation of a squared distance is allowed (isle_amd/csrc/hamerly.h).
            # Measured: up to 2.5e-3 at config 3 with the library's centre the closer one in fp64 — the oracle sums a centre's squared norm
            # sequentially over 100 000 words in fp32, which shifts all its distances to that centre by up to ~1e-3.
            tol = 1e-4 * (float((b ** 2).sum()) + cmax)
            assert abs(d[0] - d[1]) <= tol, (int(cols_a[i]), d, tol)

    l1, l4, l5 = projected(1), projected(4), projected(5)
    for got, cen in ((l1, g["C_lowd"]), (l5, l4["C_lowd"])):
        want = oa.lloyds_projected(U, cen, max_reps=1)["assign"]
        only_near_ties(got["assign"][cols_a], want, cen)
    hp.left_multiply_by_U(lp["C_lowd"], fetch=False)
    s1 = hp.run_lloyds(k, max_reps=1, fetch_centers=False)
    want = oa.lloyds_sparse(lift(U, lp["C_lowd"]), max_reps=1)["assign"]
    only_near_ties(s1["assign"][cols_a], want, lp["C_lowd"])


def _run(hp, V, D, k, seed, n_pairs, n_sample):
    from tools.synth import make_B
    B = make_B(V, D, k, seed)
    hp.upload_csc(V, B["vals"], B["rows"], B["offs"])
    return _checks(hp, B, V, k, n_pairs, n_sample)


def _checks(hp, B, V, k, n_pairs, n_sample):
    """hp holds B (uploaded, or built on the device); B also on the host with B["planted"]."""
    from oracle.oracle import OracleCsc, lift
    r = hp.compute_block_ks(k, seed=1, allow_noconv=True)
    assert r["rc"] == 0 and r["nconv"] == k and hp.operator_form() == 1
    U = hp.get_U(k)
    ev = r["evals"].astype(np.float64)
    # sigma bound through the oracle's operator
    o = OracleCsc(V, B["D"], B["vals"], B["rows"], B["offs"])
    pick = np.unique(np.concatenate([np.arange(12), np.arange(k - 8, k), np.linspace(0, k - 1, n_pairs - 20).astype(np.int64)]))
    AU = o.gram_apply(np.asfortranarray(U[:, pick])).astype(np.float64)
    resid = np.linalg.norm(AU - U[:, pick].astype(np.float64) * ev[pick], axis=0) / ev[pick]
    assert resid.max() / 2.0 <= SIGMA_TOL, "sigma bound %.2e" % (resid.max() / 2.0)
    assert np.abs(U[:, :64].astype(np.float64).T @ U[:, :64] - np.eye(64)).max() <= 1e-5
    del o, AU
    # the whole-corpus k-means chain
    g = hp.kmeans_init_on_projected_space(k, rng_seed=1)
    lp = hp.run_lloyds_on_projected_space(k, g["C_lowd"])
    hp.left_multiply_by_U(lp["C_lowd"], fetch=False)
    ls = hp.run_lloyds(k, fetch_centers=False)
    sizes = np.bincount(ls["assign"], minlength=k)
    assert (sizes > 0).all()
    maj = np.zeros((k, k), np.int64)
    np.add.at(maj, (ls["assign"].astype(np.int64), B["planted"].astype(np.int64) % k), 1)
    purity = maj.max(1).sum() / B["D"]
    assert 0.78 <= purity <= 0.95, purity
    _default_route_against_the_oracle(hp, B, V, k, U, g, lp)
    # same-input parity with the oracle on a random sample of the documents
    cols = np.sort(np.random.default_rng(7).choice(B["D"], n_sample, replace=False))
    Bs = _columns(B, cols)
    os_ = OracleCsc(V, n_sample, Bs["vals"], Bs["rows"], Bs["offs"])
    ko = os_.kmeanspp(U, k, seed=11)
    lo = os_.lloyds_projected(U, ko["C_lowd"])
    so = os_.lloyds_sparse(lift(U, lo["C_lowd"]))
    hp.upload_csc(V, Bs["vals"], Bs["rows"], Bs["offs"])
    hp.set_U(U)
    g2 = hp.kmeans_init_on_projected_space(k, inject_seeds=ko["seeds"])
    lp2 = hp.run_lloyds_on_projected_space(k, g2["C_lowd"])
    hp.left_multiply_by_U(lp2["C_lowd"], fetch=False)
    ls2 = hp.run_lloyds(k)
    assert (lp2["assign"] == lo["assign"]).mean() >= 0.999 and (ls2["assign"] == so["assign"]).mean() >= 0.999
    assert lp2["iters"] == lo["iters"] and ls2["iters"] == so["iters"]
    return r


def test_config2_at_full_size(hp):
    """vocab 50k x 1M documents x 99M nonzeros, k = 200 (ncv = 410): what bench.py --workload c2 times."""
    r = _run(hp, 50_000, 1_000_000, 200, 2024, n_pairs=32, n_sample=50_000)
    assert r["napplies"] == 40 + 20 * r["restarts"] and r["restarts"] <= 2  # 1 + (2k/b - 1) + R k/b operator applications


def test_one_eighth_of_config3_at_k1000(hp):
    """vocab 100k (30 word bands) x 1.25M documents x 126M nonzeros, k = 1000 (ncv = 2010): one GPU's share of configs[2] on eight,
    what bench.py --workload c3shard times."""
    r = _run(hp, 100_000, 1_250_000, 1000, 31337, n_pairs=32, n_sample=15_000)
    assert r["napplies"] == 200 + 100 * r["restarts"] and r["restarts"] <= 3


def test_config3_at_its_own_size(hp, monkeypatch):
    """BASELINE configs[2] itself: vocab 100k x 10M documents x 1.006 B nonzeros, k = 1000, all of it on one GPU — what bench.py times by
    default.  Index spaces here reach 10^10 elements (the D x k projection and products: 40 GB each); round 3's silent 2^32-thread launch
    (collapsed partitions for some seeds) existed only at this size.  Under asserts, one seed:
      * sigma bound <= 1e-4 through the CPU oracle's operator on Ritz pairs spread over the spectrum;
      * Lloyd in span(U) (src/sparseMatrix.cpp:1921-2072): tile bounds and no bounds (ISLE_NO_HAMERLY=1) give the same partition, bit for bit,
        where both evaluate distances by the same kernel; the default route (GEMM full passes) agrees up to near-ties (<= 2 documents in a million);
      * Lloyd on B (:1587-1746): Yinyang by group (default) and no bounds (ISLE_KMEANS_BOUNDS=none) give the same partition up to near-ties, same iterations;
      * on 100 000 documents drawn at random the oracle's arg-min against the FETCHED centres of iteration 3 equals the library's next
        assignment (:1553-1572), and every cluster is non-empty, none holds more than a fifth of the corpus."""
    import psutil
    if psutil.virtual_memory().available < 70e9:
        pytest.skip("needs ~60 GB of host memory for the 10 M-document corpus and the oracle's copy")
    from oracle.oracle import OracleCsc
    from tools.synth import Corpus
    V, D, k, seed = 100_000, 10_000_000, 1000, 31337
    B = Corpus(V, D, k, seed).threshold(k, free_A=True)
    Dn = B["D"]
    hp.upload_csc(V, B["vals"], B["rows"], B["offs"])
    r = hp.compute_block_ks(k, seed=1, allow_noconv=True)
    assert r["rc"] == 0 and r["nconv"] == k and hp.operator_form() == 1
    U = hp.get_U(k)
    ev = r["evals"].astype(np.float64)
    o = OracleCsc(V, Dn, B["vals"], B["rows"], B["offs"])
    pick = np.unique(np.concatenate([np.arange(6), np.arange(k - 4, k), np.linspace(0, k - 1, 8).astype(np.int64)]))
    AU = o.gram_apply(np.asfortranarray(U[:, pick])).astype(np.float64)
    resid = np.linalg.norm(AU - U[:, pick].astype(np.float64) * ev[pick], axis=0) / ev[pick]
    assert resid.max() / 2.0 <= SIGMA_TOL, "sigma bound %.2e" % (resid.max() / 2.0)
    del o, AU
    g = hp.kmeans_init_on_projected_space(k, rng_seed=1)
    # Lloyd in span(U): bounds on / off.  Bit for bit where both runs evaluate a distance by the same kernel (every pass by the register
    # kernel proj_assign_reg_k: ISLE_PROJ_FULL=fused, ISLE_PROJ_ACTIVE=tiles, no hand-over from the k-means++ rounds); the default run takes its full passes through
    # the GEMM and starts from what k-means++ kept — other summation orders, so a near-tie between two centres may fall the other way there
    # (first measured here: 1 document of 10 M; 13 since the active documents go through the product as well)
    lp = hp.run_lloyds_on_projected_space(k, g["C_lowd"])
    monkeypatch.setenv("ISLE_PROJ_FULL", "fused")
    monkeypatch.setenv("ISLE_PROJ_ACTIVE", "tiles")
    monkeypatch.setenv("ISLE_KMPP_TRACK", "0")
    lp1 = hp.run_lloyds_on_projected_space(k, g["C_lowd"])
    monkeypatch.setenv("ISLE_NO_HAMERLY", "1")
    lp0 = hp.run_lloyds_on_projected_space(k, g["C_lowd"])
    for name in ("ISLE_NO_HAMERLY", "ISLE_PROJ_FULL", "ISLE_PROJ_ACTIVE", "ISLE_KMPP_TRACK"):
        monkeypatch.delenv(name)
    assert lp1["iters"] == lp0["iters"]
    assert np.array_equal(lp1["assign"], lp0["assign"]), float((lp1["assign"] == lp0["assign"]).mean())
    assert np.array_equal(lp1["C_lowd"].view(np.uint32), lp0["C_lowd"].view(np.uint32))
    assert lp["iters"] == lp0["iters"] and (lp["assign"] == lp0["assign"]).mean() >= 1.0 - 5e-6, float((lp["assign"] == lp0["assign"]).mean())
    del lp0, lp1
    # Lloyd on B: bounds on / off
    hp.left_multiply_by_U(lp["C_lowd"], fetch=False)
    ls = hp.run_lloyds(k, fetch_centers=False)
    monkeypatch.setenv("ISLE_KMEANS_BOUNDS", "none")
    hp.left_multiply_by_U(lp["C_lowd"], fetch=False)
    ls0 = hp.run_lloyds(k, fetch_centers=False)
    monkeypatch.delenv("ISLE_KMEANS_BOUNDS")
    # (the unbounded loop forms its distances by the LDS-banded k-wide product, the bounded one by per-document gathers: equal up to near-ties)
    assert ls["iters"] == ls0["iters"]
    assert (ls["assign"] == ls0["assign"]).mean() >= 1.0 - 1e-5, float((ls["assign"] == ls0["assign"]).mean())  # measured: 26 documents of 10 M
    sizes = np.bincount(ls["assign"], minlength=k)
    assert (sizes > 0).all() and sizes.max() <= 0.2 * Dn, (int((sizes == 0).sum()), int(sizes.max()))
    del ls0
    # the library's assignment against fetched centres, checked by the oracle on a sample
    hp.left_multiply_by_U(lp["C_lowd"], fetch=False)
    l3 = hp.run_lloyds(k, max_reps=3)  # centres after three updates (V x k, fetched)
    l4 = hp.run_lloyds(k, centers=l3["centers"], max_reps=1, fetch_centers=False)  # one assignment against them
    cols = np.sort(np.random.default_rng(5).choice(Dn, 100_000, replace=False))
    Bs = _columns(B, cols)
    os_ = OracleCsc(V, len(cols), Bs["vals"], Bs["rows"], Bs["offs"])
    so = os_.lloyds_sparse(l3["centers"], max_reps=1)
    agree = float((so["assign"] == l4["assign"][cols]).mean())
    assert agree >= 0.9999, agree  # near-ties between two centres are all that may differ (other summation order)
    # and the bounded iteration 4 of the uninterrupted run is that full scan (the two sum a document's dot products in different orders:
    # a near-tie between two centres may fall either way, nothing else)
    hp.left_multiply_by_U(lp["C_lowd"], fetch=False)
    l4b = hp.run_lloyds(k, max_reps=4, fetch_centers=False)
    agree = float((l4b["assign"] == l4["assign"]).mean())
    assert agree >= 0.99999, agree
    # ... and the default route, step by step, against the oracle's arg-min (the k-means++ hand-over, a bounded iteration, the product through the projection)
    _default_route_against_the_oracle(hp, B, V, k, U, g, lp)


def _need_host_memory(gb):
    import psutil
    if psutil.virtual_memory().available < gb * 1e9:
        pytest.skip("needs ~%d GB of host memory" % gb)


def test_config4_at_its_own_size(hp):
    """BASELINE configs[3]: sample = 1, sample_rate = 0.1 on the 10 M-document corpus of configs[2].  The count matrix A (1.1 B entries) goes
    to the device, normalize_docs + compute_thresholds + sampled_threshold_and_copy (src/trainer.cpp:430-485, src/sparseMatrix.cpp:1365-1435)
    run there: the weights' prefix, 10 M keys rand_fraction()^(1/w), the pivot = the floor(0.1 D)-th largest key, original_cols of the kept
    documents.  Under asserts: B, the kept-document map and the thresholds equal the CPU port's (tools/synth_corpus.cpp) bit for bit; then the
    hot path on that B (about 1 M heavy documents, k = 1000) through every check of _checks — sigma bound through the oracle's operator, the
    default route step by step against the oracle's arg-min, sample parity from injected seeds."""
    _need_host_memory(50)
    from tools.synth import Corpus
    V, D, k, seed = 100_000, 10_000_000, 1000, 31337
    corp = Corpus(V, D, k, seed)
    cnt, rows, offs = corp.A_views()
    nnz_A = int(cnt.shape[0])
    assert nnz_A > 1_000_000_000  # beyond 2^30 entries
    hp.upload_counts(V, cnt, rows, offs)
    info = hp.threshold(k, sample_rate=0.1, sample_seed=seed)
    B = hp.get_B()
    del cnt, rows, offs
    planted_all = corp.planted()
    Bc = corp.threshold(k, free_A=True, sample_rate=0.1, sample_seed=seed)
    assert info["docs_kept"] == Bc["D"] == B["D"] and info["nnz_kept"] == Bc["nnz"] == B["nnz"]
    # every document whose key reaches the pivot is kept (src/sparseMatrix.cpp:1409-1421): floor(0.1 D) + 1 of them, plus the documents whose
    # keys TIE with the pivot — rand() has 2^31 values and the keys are rounded to fp32, so a few dozen of 10 M do (first seen: 1 000 038)
    assert D // 10 + 1 <= B["D"] <= D // 10 + 200, B["D"]
    for x in ("original_cols", "offs", "rows", "vals", "zetas"):
        assert np.array_equal(B[x], Bc[x]), x
    oc = B["original_cols"].astype(np.int64)
    assert (np.diff(oc) > 0).all() and oc[-1] > 0.99 * D  # the kept documents in ascending order, drawn from the whole corpus
    # importance sampling favours the heavy documents: the kept documents hold more entries each than the corpus's documents do before
    # thresholding (at 200 000 documents: 106 against 105, and 96 in the unsampled B)
    assert B["nnz"] / B["D"] > nnz_A / D, (B["nnz"] / B["D"], nnz_A / D)
    del Bc, corp
    B["planted"] = planted_all[oc]
    r = _checks(hp, B, V, k, n_pairs=32, n_sample=15_000)
    assert r["napplies"] == 200 + 100 * r["restarts"] and r["restarts"] <= 3


def test_config5_at_its_own_size(hp):
    """BASELINE configs[4]: edge_topics = 1, max_edge_topics = 5000 behind a config-3 hot path (src/trainer.cpp:673-685, :1116-1167;
    top_topic_pairs from construct_topic_model, src/sparseMatrix.cpp:687-708) — at its own size: A on the device, B built there (equal to the
    CPU port's bit for bit), one hot-path step at 10 M documents, then catchwords, the topic model (top-two topics of 10 M documents), the
    pair selection over up to 10 M (top1, top2, doc) triples and the V x 5000 edge model (2 GB).  Checker oracle/isle_post_oracle.cpp:
    catchword thresholds and catchwords bit for bit at full size, the pair selection (all pairs), the edge columns of the 64 most frequent
    pairs, and the topic model with the top-two topics at full size."""
    _need_host_memory(90)
    from isle_amd.hot_path import select_edge_pairs
    from oracle import oracle as O
    from tools.synth import Corpus
    V, D, k, seed = 100_000, 10_000_000, 1000, 31337
    corp = Corpus(V, D, k, seed)
    cnt, rows, offs = corp.A()  # copies: the CPU restatement of the downstream stage reads A after the port's thresholding has freed its own
    hp.upload_counts(V, cnt, rows, offs)
    info = hp.threshold(k)
    B = hp.get_B()
    Bc = corp.threshold(k, free_A=True)
    for x in ("original_cols", "offs", "rows", "vals", "zetas"):
        assert np.array_equal(B[x], Bc[x]), x
    del Bc, corp
    r = hp.compute_block_ks(k, seed=1, allow_noconv=True)
    assert r["rc"] == 0 and r["nconv"] == k
    g = hp.kmeans_init_on_projected_space(k, rng_seed=1)
    lp = hp.run_lloyds_on_projected_space(k, g["C_lowd"])
    hp.left_multiply_by_U(lp["C_lowd"], fetch=False)
    ls = hp.run_lloyds(k, fetch_centers=False)
    rk, rank_thr = O.catchword_rank(D, k), O.model_rank_threshold(D, k)
    assert rk >= 1 and rank_thr >= 1
    got = hp.find_catchwords(k, rk)  # the resident partition
    tm = hp.construct_topic_model(k, rank_thr, D, fetch_sums=False)
    pairs = select_edge_pairs(tm["top1"], tm["top2"], 5000)
    assert pairs.shape[0] == 5000 and (pairs[:-1, 2] >= pairs[1:, 2]).all()
    E = hp.edge_topics(pairs[:, :2])
    assert E.shape == (V, 5000)
    # --- the CPU restatement
    cl = np.full(D, -1, np.int32)
    cl[B["original_cols"].astype(np.int64)] = ls["assign"].astype(np.int32)
    del B
    nv = O.post_normalize(offs, cnt, info["avg_doc_sz"])
    thr = O.post_catch_thresholds(V, offs, rows, nv, cl, k, rk)
    np.testing.assert_array_equal(got["thresholds"], thr)
    ct = O.post_find_catchwords(thr)
    np.testing.assert_array_equal(got["catch_topic"], ct)
    assert got["num_catchwords"] == int((ct >= 0).sum()) and got["num_catchwords"] > 5 * k
    del thr, got
    ref_pairs, ref_edge = O.post_edge_topics(tm["model"], tm["top1"], tm["top2"], 5000, want_edge=False)
    np.testing.assert_array_equal(pairs, ref_pairs)
    head_pairs, head_edge = O.post_edge_topics(tm["model"], tm["top1"], tm["top2"], 64)
    np.testing.assert_array_equal(head_pairs, pairs[:64])
    ok = np.isfinite(head_edge)
    assert np.array_equal(np.isfinite(E[:, :64]), ok)
    np.testing.assert_allclose(E[:, :64][ok], head_edge[ok], rtol=3e-7, atol=1e-12)
    for t_ in (100, 2500, 4999):  # and columns beyond the oracle's head, by the formula of src/trainer.cpp:1152-1159
        want = np.float32(0.7) * tm["model"][:, pairs[t_, 0]] + np.float32(1.0 - 0.7) * tm["model"][:, pairs[t_, 1]]
        okc = np.isfinite(want)
        np.testing.assert_allclose(E[okc, t_], want[okc], rtol=3e-7, atol=1e-12)
    del E, head_edge
    ref = O.post_topic_model(V, offs, rows, nv, cl, ct, k, rank_thr)
    np.testing.assert_array_equal(tm["model_threshold"], ref["model_threshold"])
    np.testing.assert_array_equal(tm["top1"], ref["top1"])
    np.testing.assert_array_equal(tm["top2"], ref["top2"])
    okm = np.isfinite(ref["model"])
    assert np.array_equal(np.isfinite(tm["model"]), okm)
    # fp32 sums in another order (atomics): 2e-5 holds on the reduced shapes of tests/test_gpu_post.py; a topic vector of config 3 adds ~10 000
    # documents, where 287 of the 10^8 entries were seen beyond it (the largest at 2.4e-5)
    np.testing.assert_allclose(tm["model"][okm], ref["model"][okm], rtol=6e-5, atol=1e-9)
