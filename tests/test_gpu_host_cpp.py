"""The C++ host side (isle_amd/host: FPSparseMatrixHip + the trainer.cpp:490-571 slice) run as a real process on
the GPU, checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest

from conftest import corpus

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_trainer_slice(tmp_path):
    B = corpus(2000, 5000, 20, 0)
    k = 20
    exe = os.path.join(ROOT, "isle_amd", "host", "hot_path_main")
    assert os.path.exists(exe), "build with make -C isle_amd/csrc"
    fin, fout = str(tmp_path / "B.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        np.array([B["V"], B["D"], B["nnz"]], np.uint64).tofile(f)
        B["vals"].astype(np.float32).tofile(f)
        B["rows"].astype(np.uint64).tofile(f)  # the reference's 8-byte word_id_t
        B["offs"].astype(np.int64).tofile(f)
    r = subprocess.run([exe, fin, str(k), fout], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Frob(B_fl_CSC): " in r.stdout and "Eigvals:  (0): " in r.stdout and "k-means init residual" in r.stdout
    raw = open(fout, "rb").read()
    kk = int(np.frombuffer(raw, np.uint64, 1)[0])
    assert kk == k
    off = 8
    ev = np.frombuffer(raw, np.float32, k, off); off += 4 * k
    seeds = np.frombuffer(raw, np.uint64, k, off); off += 8 * k
    centers = np.frombuffer(raw, np.float32, B["V"] * k, off).reshape((B["V"], k), order="F"); off += 4 * B["V"] * k
    sizes = np.frombuffer(raw, np.uint64, k, off); off += 8 * k
    part = np.frombuffer(raw, np.uint64, int(sizes.sum()), off)
    o = B["oracle"].block_ks(k)
    assert np.max(np.abs(np.sqrt(ev) - np.sqrt(o["evals"])) / np.sqrt(o["evals"])) <= 1e-4
    # the printed singular values are the same numbers at ostream precision (6 significant digits)
    first = float(r.stdout.split("Eigvals:  (0): ")[1].split("\t")[0])
    assert abs(first - np.sqrt(ev[0])) <= 1e-5 * first
    assert sizes.sum() == B["D"] and len(set(seeds.tolist())) == k
    assert np.array_equal(np.sort(part), np.arange(B["D"], dtype=np.uint64))  # a partition of all documents
    # every centre is the mean of its members (one more Lloyd step would not move an assignment-consistent centre far)
    start = 0
    import scipy.sparse as sp
    S = sp.csc_matrix((B["vals"], B["rows"], B["offs"]), shape=(B["V"], B["D"]))
    for c in range(k):
        mem = part[start:start + int(sizes[c])].astype(np.int64)
        start += int(sizes[c])
        assert (np.diff(mem) > 0).all()  # ascending ids inside a cluster
        if len(mem):
            mean = np.asarray(S[:, mem].mean(axis=1)).ravel()
            assert np.abs(mean - centers[:, c]).max() <= 1e-4 * max(1.0, np.abs(mean).max())
