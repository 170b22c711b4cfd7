"""GPU parity for the stage upstream of the hot path: A -> B thresholding on the device.

Checker: tools/synth_corpus.cpp's CPU thresholding (a restatement of normalize_docs, compute_thresholds and
(sampled_)threshold_and_copy, src/sparseMatrix.cpp:136-167, :357-485, :1285-1435).  Integer / index work: bit-exact.
"""
import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu


def _check_same(got, want):
    assert got["D"] == want["D"] and got["nnz"] == want["nnz"]
    np.testing.assert_array_equal(got["zetas"], want["zetas"])
    np.testing.assert_array_equal(got["offs"], want["offs"])
    np.testing.assert_array_equal(got["original_cols"], want["original_cols"])
    np.testing.assert_array_equal(got["rows"], want["rows"])
    np.testing.assert_array_equal(got["vals"], want["vals"])


def _run(hp, c, k, **kw):
    cnt, rows, offs = c.A()
    hp.upload_counts(c.V, cnt, rows, offs)
    info = hp.threshold(k, **kw)
    got = hp.get_B()
    want = c.threshold(k, **kw)
    _check_same(got, want)
    return info, got, want


@pytest.mark.parametrize("V,D,k,seed", [(2000, 3000, 10, 3), (5000, 20000, 20, 4), (20000, 60000, 50, 5)])
def test_threshold_matches_cpu(hp, V, D, k, seed):
    from tools.synth import Corpus
    c = Corpus(V, D, k, seed)
    info, got, want = _run(hp, c, k)
    assert info["docs_kept"] == want["D"] and info["nnz_kept"] == want["nnz"]
    assert info["entries_above_threshold"] == want["nnz"]
    # values are sqrt(zeta) of the row (src/sparseMatrix.cpp:1347)
    np.testing.assert_array_equal(got["vals"], np.sqrt(got["zetas"][got["rows"]]))


def test_threshold_sampled_matches_cpu(hp):
    from tools.synth import Corpus
    c = Corpus(5000, 30000, 20, 7)
    info, got, want = _run(hp, c, 20, sample_rate=0.25, sample_seed=11)
    assert 0 < got["D"] < 30000 * 0.3
    assert info["entries_above_threshold"] > got["nnz"]


def test_threshold_edge_cases(hp):
    """Empty documents, a word that never occurs, a document whose entries all fall below the threshold."""
    from tools.synth import Corpus
    rng = np.random.default_rng(0)
    V, D = 64, 500
    cols = []
    for d in range(D):
        if d % 7 == 0:
            cols.append((np.zeros(0, np.uint32), np.zeros(0, np.float32)))  # empty document
            continue
        n = int(rng.integers(1, 20))
        r = np.sort(rng.choice(V - 1, size=n, replace=False)).astype(np.uint32)  # word V-1 never occurs
        cnt = rng.integers(1, 9, size=n).astype(np.float32)
        cols.append((r, cnt))
    offs = np.zeros(D + 1, np.int64)
    offs[1:] = np.cumsum([len(r) for r, _ in cols])
    rows = np.concatenate([r for r, _ in cols])
    cnt = np.concatenate([c_ for _, c_ in cols])
    c = Corpus.from_csc(V, D, cnt, rows, offs)
    info, got, want = _run(hp, c, 5)
    assert got["D"] < D  # empty columns removed
    assert got["zetas"][V - 1] == 1.0


def test_threshold_then_hot_path_equals_upload(hp):
    """B built on the device drives the eigensolver exactly like the same B uploaded from the host."""
    from tools.synth import Corpus
    c = Corpus(5000, 20000, 20, 9)
    cnt, rows, offs = c.A()
    hp.upload_counts(c.V, cnt, rows, offs)
    hp.threshold(20)
    B = hp.get_B()
    fro_dev = hp.frobenius()
    r1 = hp.compute_block_ks(20, seed=5)
    hp.upload_csc(B["V"], B["vals"], B["rows"], B["offs"])
    assert abs(hp.frobenius() - fro_dev) <= 1e-6 * fro_dev
    r2 = hp.compute_block_ks(20, seed=5)
    assert relerr(r1["evals"], r2["evals"]) < 1e-6


def test_threshold_requires_counts(hp):
    import isle_amd
    h2 = isle_amd.HotPath()
    with pytest.raises(Exception):
        h2.threshold(10)
