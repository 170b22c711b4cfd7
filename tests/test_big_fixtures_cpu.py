"""The oracle (fp32 CPU restatement) against tests/golden/big_<case>.npz: the reference's Spectra solver run in the build
container plus fp64 brute-force mathematics (tests/golden/make_golden_big.py).  This is the independent pin for the half of the
path whose reference code cannot be built here (operator, BlockKs, k-means — all behind <mkl.h>): eigenvalues, the spanned
subspace, k-means++ distances and both Lloyd loops at k = 50 and k = 200 are recomputed by the oracle and held to the fixture;
the k = 1000 cases (two minutes of CPU eigensolve each) are checked through the figures the generator recorded."""
import os

import numpy as np
import pytest

from conftest import corpus

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name):
    f = np.load(os.path.join(GOLD, "big_%s.npz" % name))
    V, D, k, seed, sr = (int(x) for x in f["params"])
    return f, V, D, k, seed, sr / 1000.0


@pytest.mark.parametrize("name", ["c1k50", "c2k200"])
def test_oracle_against_fp64_brute_force(name):
    from oracle.oracle import lift
    f, V, D, k, seed, sr = load(name)
    B = corpus(V, D, k, seed, sample_rate=sr)
    assert np.array_equal(np.array([B["V"], B["D"], B["nnz"], int(B["rows"].astype(np.int64).sum())], np.int64), f["sig"])
    o = B["oracle"]
    r = o.block_ks(k)
    truth = np.sqrt(f["truth_evalues"][:k])
    assert r["nconv"] == k
    assert np.max(np.abs(np.sqrt(r["evals"].astype(np.float64)) - truth) / truth) <= 1e-5
    ref = np.sqrt(f["spectra_evalues"].astype(np.float64))
    assert np.max(np.abs(ref - truth) / truth) <= 1e-4        # the reference's own solver against fp64 (2e-5 at k = 200)
    U = r["U"].astype(np.float64)
    R = np.random.default_rng(12345).standard_normal((V, 8))
    assert np.linalg.norm(U @ (U.T @ R) - f["sketch"]) / np.linalg.norm(f["sketch"]) <= 1e-3
    ko = o.kmeanspp(r["U"], k, inject=f["seeds"])
    assert np.abs(ko["min_dist"][f["min_d2_idx"]] - f["min_d2_val"]).max() <= 5e-4 * f["min_d2_val"].max()
    lo = o.lloyds_projected(r["U"], ko["C_lowd"])
    assert lo["iters"] == int(f["lp_iters"])
    assert (lo["assign"] == f["lp_assign"]).mean() >= 0.999
    so = o.lloyds_sparse(lift(r["U"], lo["C_lowd"]))
    assert so["iters"] == int(f["ls_iters"])
    assert (so["assign"] == f["ls_assign"]).mean() >= 0.999
    cn = np.sqrt((so["centers"].astype(np.float64) ** 2).sum(0))
    big = f["ls_cnorm"] > 1e-3 * f["ls_cnorm"].max()
    assert np.median(np.abs(cn[big] - f["ls_cnorm"][big]) / f["ls_cnorm"][big]) <= 1e-4


@pytest.mark.parametrize("name", ["c3k1000", "c4k1000s"])
def test_recorded_figures_at_k1000(name):
    f, V, D, k, seed, sr = load(name)
    assert k == 1000
    assert float(f["oracle_sigma_err"]) <= 1e-5                 # restated block Krylov-Schur against fp64
    assert float(f["spectra_sigma_err_vs_truth"]) <= 3e-4       # the reference's Spectra solver, fp32 with ncv = 2k + 1
    assert f["oracle_agreement"].min() >= 0.999                 # both Lloyd loops against fp64 brute force
    assert float(f["oracle_min_d2_err"]) <= 1e-4
    assert float(f["oracle_sketch_err"]) <= 1e-2


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "spectra_eigs")), reason="needs the build container's oracle/_ref")
def test_fixture_is_what_the_reference_solver_returns_today():
    """Re-runs the reference's Spectra solver (k = 200, ~10 s) and requires the committed eigenvalues bit for bit."""
    import sys
    sys.path.insert(0, GOLD)
    from make_golden_ref import run_reference
    f, V, D, k, seed, sr = load("c2k200")
    B = corpus(V, D, k, seed, sample_rate=sr)
    nconv, info, ev, _ = run_reference(B, k)
    assert nconv == k and info == 0
    assert np.array_equal(ev, f["spectra_evalues"])
