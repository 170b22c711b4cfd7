"""GPU parity for tdf ingest on the device: text -> count matrix A in HBM.

Checker: an independent NumPy statement of the reference's rules (include/utils.h:158-228 parser, src/trainer.cpp:236-247
sort + de-duplication, src/sparseMatrix.cpp:58-87 CSC build).  Integer / index work: bit-exact.
"""
import numpy as np
import pytest

from tools.synth import Corpus
from test_cli_cpu import write_tdf

pytestmark = pytest.mark.gpu


def numpy_ingest(text, V, D):
    trip = np.array([[int(x) for x in ln.split()] for ln in text.decode().replace("\r", "").split("\n") if ln.strip()], np.int64).reshape(-1, 3)
    doc, word, cnt = trip[:, 0] - 1, trip[:, 1] - 1, trip[:, 2]
    order = np.lexsort((np.arange(len(doc)), word, doc))       # stable: first occurrence in the file wins
    doc, word, cnt = doc[order], word[order], cnt[order]
    first = np.ones(len(doc), bool)
    first[1:] = (doc[1:] != doc[:-1]) | (word[1:] != word[:-1])
    doc, word, cnt = doc[first], word[first], cnt[first]
    offs = np.zeros(D + 1, np.int64)
    np.add.at(offs, doc + 1, 1)
    return cnt.astype(np.float32), word.astype(np.uint32), np.cumsum(offs), len(trip)


@pytest.mark.parametrize("style,shuffle", [("plain", None), ("messy", 7)])
def test_ingest_matches_numpy(hp, tmp_path, style, shuffle):
    V, D, k = 3000, 5000, 10
    c = Corpus(V, D, k, seed=11)
    counts, rows, offs = c.A()
    path = str(tmp_path / "c.tdf")
    n = write_tdf(path, counts, rows, offs, shuffle_seed=shuffle, style=style)
    text = open(path, "rb").read()
    info = hp.ingest_tdf(text, V, D, max_entries=n)
    assert info["entries_read"] == n and info["nnz"] == len(counts)
    gc, gr, go = hp.get_A()
    wc, wr, wo, _ = numpy_ingest(text, V, D)
    np.testing.assert_array_equal(go, wo)
    np.testing.assert_array_equal(gr, wr)
    np.testing.assert_array_equal(gc, wc)
    np.testing.assert_array_equal(go, offs)       # and equal to the generator's own CSC
    np.testing.assert_array_equal(gr, rows)
    np.testing.assert_array_equal(gc, counts)


def test_ingest_duplicates_empty_docs_and_blank_lines(hp):
    text = b"3 2 5\n\n1 4 1\n3 2 9\r\n  \n3 1 2\n7 7 7\n1 4 8\n6 1 3"   # (3,2) and (1,4) repeated; docs 2,4,5 empty; no final newline
    V, D = 8, 9
    info = hp.ingest_tdf(text, V, D)
    assert info["entries_read"] == 7 and info["nnz"] == 5
    gc, gr, go = hp.get_A()
    wc, wr, wo, _ = numpy_ingest(text, V, D)
    np.testing.assert_array_equal(go, wo)
    np.testing.assert_array_equal(gr, wr)
    np.testing.assert_array_equal(gc, wc)
    assert list(gc) == [1.0, 2.0, 5.0, 3.0, 7.0]  # first occurrences: (1,4)->1, (3,1)->2, (3,2)->5


def test_ingest_errors(hp):
    with pytest.raises(Exception, match="bad character"):
        hp.ingest_tdf(b"1 2 3\n1 x 3\n", 5, 5)
    with pytest.raises(Exception, match="more than three"):
        hp.ingest_tdf(b"1 2 3 4\n", 5, 5)
    with pytest.raises(Exception, match="fewer than three"):
        hp.ingest_tdf(b"1 2\n", 5, 5)
    with pytest.raises(Exception, match="exceeds"):
        hp.ingest_tdf(b"6 1 1\n", 5, 5)
    with pytest.raises(Exception, match="max_entries"):
        hp.ingest_tdf(b"1 1 1\n2 2 2\n", 5, 5, max_entries=3)
    with pytest.raises(Exception, match="count is 0 on line 2"):  # a document of zero counts would normalise to 0 / 0
        hp.ingest_tdf(b"1 1 1\n2 2 0\n", 5, 5)


def test_ingest_then_threshold_equals_upload(hp, tmp_path):
    """text -> A -> B entirely on the device equals upload_counts -> threshold."""
    V, D, k = 3000, 8000, 10
    c = Corpus(V, D, k, seed=12)
    counts, rows, offs = c.A()
    path = str(tmp_path / "c.tdf")
    write_tdf(path, counts, rows, offs, shuffle_seed=3)
    hp.ingest_tdf(open(path, "rb").read(), V, D)
    hp.threshold(k)
    got = hp.get_B()
    want = c.threshold(k)
    for f in ("vals", "rows", "offs", "original_cols", "zetas"):
        np.testing.assert_array_equal(got[f], want[f])
