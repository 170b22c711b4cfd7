"""ISLETrain end to end on the GPU: tdf text in, reference-format log directory out (hot path only)."""
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
    args = [os.path.join(ROOT, "isle_amd", "host", "ISLETrain"), tdf, vocab, out, str(V), str(D), str(n), str(k), "0", str(sample), "0.5", "0", "0"]
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
