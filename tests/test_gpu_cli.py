"""ISLETrain end to end on the GPU: tdf text in, reference-format log directory and model files out."""
import os
import subprocess

import numpy as np
import pytest

from tools.synth import Corpus
from test_cli_cpu import write_tdf

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("sample", [0, 1])
def test_isletrain_cli(tmp_path, sample):
    from oracle.oracle import OracleCsc
    V, D, k = 1500, 4000, 20
    c = Corpus(V, D, k, seed=6)
    counts, rows, offs = c.A()
    tdf = str(tmp_path / "corpus.tdf")
    n = write_tdf(tdf, counts, rows, offs)
    vocab = str(tmp_path / "vocab.txt")
    open(vocab, "w").write("\n".join("w%d" % i for i in range(V)))
    out = str(tmp_path / "out")
    os.mkdir(out)
    args = [os.path.join(ROOT, "isle_amd", "host", "ISLETrain"), tdf, vocab, out, str(V), str(D), str(n), str(k), "0", str(sample), "0.5", "1", "30"]
    r = subprocess.run(args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ISLE Trainer failed" not in r.stderr, r.stderr[-2000:]
    sub = "log_t_%d_eps1_0.016667_eps2_0.333333_eps3_5.000000_kMppReps_1_kMLowDReps_10_kMReps_10_sample_%d_tfidf_0" % (k, sample)
    if sample:
        sub += "_Rate_0.500000"
    logdir = os.path.join(out, sub)  # src/utils.cpp:28-48
    assert os.path.isdir(logdir), os.listdir(out)
    diag = open(os.path.join(logdir, "diagnosticLog.txt")).read()
    timer = open(os.path.join(logdir, "timerLog.txt")).read()
    assert "Number of entries above threshold: " in diag and "Frob(B_fl_CSC): " in diag and "Eigvals:  (0): " in diag
    for label in ("Reading file Entries", "Spectra eigen solve", "K-means seeds initialization", "Converging LLoyds k-means on B_k", "k-means on B"):
        assert ("Time for " + label) in timer
    assert "Total time for TVSD" in timer
    # singular values against the oracle on the same B (thresholding restated independently in tools/)
    B = c.threshold(k) if not sample else None
    sv = np.loadtxt(os.path.join(logdir, "HotPathSingularValues.txt"))
    if B is not None:
        o = OracleCsc(V, B["D"], B["vals"], B["rows"], B["offs"]).block_ks(k)
        assert np.max(np.abs(sv - np.sqrt(o["evals"])) / np.sqrt(o["evals"])) <= 1e-4
        assert "Number of entries above threshold: %d" % B["nnz"] in diag
    cl = np.loadtxt(os.path.join(logdir, "HotPathClusters.tsv"), dtype=np.int64)
    docs = cl[:, 1] - 1
    assert len(np.unique(docs)) == len(docs) and cl[:, 0].min() >= 1 and cl[:, 0].max() <= k
    if B is not None:
        assert np.array_equal(np.sort(docs), B["original_cols"].astype(np.int64))  # mapped through original_cols (trainer.cpp:573-575)
    else:
        assert len(docs) == int(np.float32(0.5) * np.float32(D)) + 1

    # ---- downstream stage: the model file against the oracle driven by the SAME partition (read back from the CLI) -----
    from oracle import oracle as O
    avg = float(int(counts.sum()) // int((np.diff(offs) > 0).sum()))
    nv = O.post_normalize(offs, counts, avg)
    cluster_of = np.full(D, -1, np.int32)
    cluster_of[docs] = (cl[:, 0] - 1).astype(np.int32)
    r = O.catchword_rank(D, k, sample_rate=0.5 if sample else None)
    thr = O.post_catch_thresholds(V, offs, rows, nv, cluster_of, k, r)
    ct = O.post_find_catchwords(thr)
    ref = O.post_topic_model(V, offs, rows, nv, cluster_of, ct, k, O.model_rank_threshold(D, k))
    M = np.zeros((V, k), np.float64)
    lines = open(os.path.join(logdir, "M_hat_catch_sparse")).read().splitlines()
    for ln in lines:
        t, w, x = ln.split("\t")
        assert len(x.split(".")[1]) == 6                      # six truncated decimals (include/utils.h:421-478)
        M[int(w) - 1, int(t) - 1] = float(x)
    want = ref["model"].astype(np.float64)
    keep = want > 1e-8
    assert np.array_equal(M > 0, keep & (np.floor(want * 1e6) > 0)) or np.abs(M - np.floor(want * 1e6) / 1e6).max() <= 1.01e-6
    assert np.abs(M - want * keep).max() <= 2e-6                # truncation (1e-6) + fp32 summation-order noise
    top = open(os.path.join(logdir, "TopWordsPerTopic_catch.txt")).read().splitlines()
    assert len(top) == k and all(len(t.split("\t")) >= 10 for t in top)
    for t in range(k):                                        # heaviest word of every topic agrees with the oracle's model
        w0 = int(top[t].split("\t")[0][1:])
        assert want[w0, t] >= want[:, t].max() * (1 - 1e-5)
    assert ("---------- Topic: 0, Cluster_size: ") in diag and "Catchwords:" in diag and "#Topics with no catchwords: " in diag
    for label in ("Collecting word freqs in clusters", "Finding catchwords for clusters", "Constructing topic vectors", "Output model",
                  "Constructing edge topic model", "Output edge model"):
        assert ("Time for " + label) in timer
    # edge topics: at most 30, each the 0.7 / 0.3 mix of two basic topics
    pairs, edge = O.post_edge_topics(ref["model"], ref["top1"], ref["top2"], 30)
    E = np.zeros((V, 30), np.float64)
    for ln in open(os.path.join(logdir, "EdgeModel_sparse")).read().splitlines():
        t, w, x = ln.split("\t")
        E[int(w) - 1, int(t) - 1] = float(x)
    assert pairs.shape[0] == 30 and np.abs(E - edge * (edge > 1e-8)).max() <= 2e-6


def test_trainer_class_fed_document_by_document_equals_the_file_load(tmp_path):
    """ISLE::ISLETrainer (isle_amd/host/trainer_hip.h) in ITERATIVE_DATA_LOAD mode — feed_data per document (shuffled documents, words in
    reverse order), finalize_data, train, get_basic_model: the call sequence of the reference's export layer
    (drivers/trainer_export.cpp:31-98) — must leave the model the file-loading CLI leaves for the same corpus: same A, same B, and the
    path is deterministic."""
    V, D, k = 1500, 4000, 20
    c = Corpus(V, D, k, seed=6)
    counts, rows, offs = c.A()
    tdf = str(tmp_path / "corpus.tdf")
    n = write_tdf(tdf, counts, rows, offs)
    vocab = str(tmp_path / "vocab.txt")
    open(vocab, "w").write("\n".join("w%d" % i for i in range(V)))
    out_a, out_b = str(tmp_path / "a"), str(tmp_path / "b")
    os.mkdir(out_a)
    os.mkdir(out_b)
    r = subprocess.run([os.path.join(ROOT, "isle_amd", "host", "ISLETrain"), tdf, vocab, out_a, str(V), str(D), str(n), str(k), "0", "0", "0", "0", "0"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    model_b = str(tmp_path / "model_b.txt")
    r = subprocess.run([os.path.join(ROOT, "isle_amd", "host", "trainer_feed_main"), tdf, out_b, str(V), str(D), str(k), model_b],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    sub = "log_t_%d_eps1_0.016667_eps2_0.333333_eps3_5.000000_kMppReps_1_kMLowDReps_10_kMReps_10_sample_0_tfidf_0" % k
    A = np.zeros((V, k))
    for ln in open(os.path.join(out_a, sub, "M_hat_catch_sparse")).read().splitlines():
        t, w, x = ln.split("\t")
        A[int(w) - 1, int(t) - 1] = float(x)
    Bm = np.zeros((V, k))
    for ln in open(model_b).read().splitlines():
        w, t, x = ln.split()
        Bm[int(w), int(t)] = float(x)
    assert np.array_equal(np.loadtxt(os.path.join(out_a, sub, "HotPathClusters.tsv"), dtype=np.int64),
                          np.loadtxt(os.path.join(out_b, sub, "HotPathClusters.tsv"), dtype=np.int64))
    assert np.abs(A - np.floor(Bm * 1e6) / 1e6 * (Bm > 1e-8)).max() <= 1.01e-6  # the file holds six truncated decimals
