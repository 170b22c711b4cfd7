"""The oracle against the REFERENCE's own eigensolver.

tests/golden/ref_spectra.npz holds eigenvalues (and, for the tiny cases, eigenvectors) of B B^T computed in the build container
by the reference's vendored Spectra::SymEigsSolver + Eigen, called as FPSparseMatrix::compute_Spectra calls it
(/root/reference/src/sparseMatrix.cpp:1161-1190; recipe oracle/Makefile -> oracle/_ref/spectra_eigs, generator
tests/golden/make_golden_ref.py).  This is the pin of the oracle's eigensolver by a run of reference code; the MKL-bound
parts of the reference (its operator, BlockKs, k-means) cannot be built in this image."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, corpus, subspace_cosines

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
GOLD = os.path.join(ROOT, "tests", "golden", "ref_spectra.npz")
CASES = ["tiny10", "tiny20", "small50", "mid30"]
SIGMA_TOL = 1e-4  # BASELINE.json: top-k singular values within 1e-4 relative


def load_case(g, name):
    V, D, k, seed = (int(x) for x in g[name + "_params"])
    B = corpus(V, D, k, seed)
    sig = np.array([B["V"], B["D"], B["nnz"], int(B["rows"].astype(np.int64).sum())], np.int64)
    assert np.array_equal(sig, g[name + "_sig"]), "the corpus generator no longer reproduces the matrix the fixture was made from"
    return B, k


@pytest.mark.parametrize("name", CASES)
def test_oracle_sigma_matches_reference_solver(name):
    g = np.load(GOLD)
    B, k = load_case(g, name)
    ev_ref = g[name + "_evalues"].astype(np.float64)
    assert (np.diff(ev_ref) <= 0).all() and ev_ref[-1] > 0  # descending, positive (asserts of compute_Spectra :1179)
    o = B["oracle"].block_ks(k)
    s, s_ref = np.sqrt(o["evals"].astype(np.float64)), np.sqrt(ev_ref)
    assert np.max(np.abs(s - s_ref) / s_ref) <= SIGMA_TOL
    if name + "_U" in g:
        U_ref = g[name + "_U"]
        assert np.abs(U_ref.astype(np.float64).T @ U_ref - np.eye(k)).max() <= 1e-4
        # same invariant subspace away from the edge of the wanted cluster (sigma_k is not separated from sigma_{k+1})
        assert subspace_cosines(o["U"][:, : k - 2], U_ref).min() >= 1 - 1e-3
        # and the reference's vectors are eigenvectors of the oracle's operator
        AU = B["oracle"].gram_apply(U_ref)
        assert np.abs(AU - U_ref * g[name + "_evalues"]).max() <= 2e-3 * ev_ref[0]


def test_fixture_is_what_the_reference_build_produces_today():
    """Container only: re-run oracle/_ref/spectra_eigs and compare with the committed fixture (bit for bit: the run is
    deterministic — Spectra's own SimpleRandom start vector, single-threaded operator)."""
    binary = os.path.join(ROOT, "oracle", "_ref", "spectra_eigs")
    if not os.path.exists(binary):
        pytest.skip("oracle/_ref is built only where /root/reference exists")
    from make_golden_ref import run_reference
    g = np.load(GOLD)
    B, k = load_case(g, "tiny20")
    nconv, info, ev, U = run_reference(B, k)
    assert nconv == k and info == 0
    assert np.array_equal(ev, g["tiny20_evalues"]) and np.array_equal(U, g["tiny20_U"])
