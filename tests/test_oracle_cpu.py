"""CPU tests of the oracle (the checker): against the committed fp64 golden vectors, fp64 NumPy ground truth,
and the reference's own known-spectrum recipe utils::get_seed_eigs (block-ks/ks_utils.h:136-165) run through
a dense operator like utils::ArmaMatProdOp (:167-182).  No GPU."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import relerr, subspace_cosines
from oracle import oracle as orc

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_fp64.npz"))


@pytest.fixture(scope="module")
def gold():
    g = GOLD
    return g, orc.OracleCsc(int(g["V"]), int(g["D"]), g["vals"], g["rows"], g["offs"])


def test_golden_gram_apply_and_frobenius(gold):
    g, m = gold
    assert relerr(m.gram_apply(g["X"]), g["Z"]) <= 2e-6
    assert abs(m.frobenius() - float(g["frob"])) <= 1e-5 * float(g["frob"])


@pytest.mark.parametrize("k,blk", [(10, 10), (5, 10), (10, 5)])
def test_golden_block_ks_sigma(gold, k, blk):
    g, m = gold
    r = m.block_ks(k, blk=blk, ncv=2 * k + 10)
    assert r["rc"] == 0 and r["nconv"] == k
    sig, truth = np.sqrt(r["evals"].astype(np.float64)), np.sqrt(g["lam"][:k])
    assert np.max(np.abs(sig - truth) / truth) <= 1e-5
    U = r["U"].astype(np.float64)
    assert np.abs(U.T @ U - np.eye(k)).max() <= 1e-5
    if k == 10:
        assert subspace_cosines(r["U"][:, :8], g["U"]).min() >= 1 - 1e-4


def test_golden_kmeans_chain(gold):
    g, m = gold
    k = int(g["k"])
    U = g["U"]
    P, n2 = m.project(U)
    km = m.kmeanspp(U, k, inject=g["seeds"])
    assert (km["seeds"] == g["seeds"]).all()
    # min distances: all seeds but those of the last round are folded in
    Cs = P[g["seeds"].astype(np.int64)].astype(np.float64)
    assert relerr(km["C_lowd"], Cs) <= 1e-6
    lo = m.lloyds_projected(U, Cs.astype(np.float32), max_reps=3)
    assert lo["iters"] == 3
    assert (lo["assign"] == g["assign_proj3"]).mean() >= 0.995
    assert relerr(lo["C_lowd"], g["C_proj3"]) <= 1e-4
    cen = orc.lift(U, g["C_proj3"].astype(np.float32))
    so = m.lloyds_sparse(cen, max_reps=2)
    assert (so["assign"] == g["assign_word2"]).mean() >= 0.995
    assert relerr(so["centers"], g["C_word2"]) <= 1e-4


def test_kmeanspp_min_dist_matches_bruteforce(gold):
    g, m = gold
    k = int(g["k"])
    # inject seeds; after the final round every seed except the last round's has been folded into min_dist
    km = m.kmeanspp(g["U"], k, inject=g["seeds"])
    P = m.project(g["U"])[0].astype(np.float64)
    n_last = 1  # with k = 10 every round adds one seed (1 + ceil(sqrt(max(s-5,0))) >= 1; only the last is pending)
    # reconstruct: number of seeds folded = k - (seeds added in last round)
    best = None
    for folded in range(k - 4, k):
        Cs = P[g["seeds"][:folded].astype(np.int64)]
        d2 = np.maximum(((P[:, None, :] - Cs[None]) ** 2).sum(-1), 0).min(1)
        err = np.abs(d2 - km["min_dist"]).max() / d2.max()
        best = err if best is None else min(best, err)
    assert best <= 1e-4


@pytest.mark.parametrize("n", [1, 2, 9, 40, 101])
def test_eig_sym_vs_lapack(n):
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n))
    S = (A + A.T).astype(np.float32)
    e, v = orc.eig_sym(S)
    er = np.linalg.eigvalsh(S.astype(np.float64))[::-1]
    assert np.abs(e - er).max() <= 1e-5 * max(1.0, np.abs(er).max())
    assert np.abs(S.astype(np.float64) @ v - v * e).max() <= 1e-4 * max(1.0, np.abs(er).max())


def test_compute_qr_rank_revealing():
    rng = np.random.default_rng(0)
    A = rng.integers(-8, 9, size=(500, 6)).astype(np.float32)  # integers: the dependency below is exact in fp32
    A[:, 3] = 2 * A[:, 1] - A[:, 0]  # dependent column is dropped (block-ks/ks_utils.h:69)
    A[:, 5] = 0
    Q, R, rk = orc.qr(A)
    assert rk == 4
    assert np.abs(Q.astype(np.float64).T @ Q - np.eye(rk)).max() <= 1e-6
    keep = [0, 1, 2, 4]
    assert relerr(Q @ R[:, keep], A[:, keep]) <= 1e-6


@pytest.mark.parametrize("kind", [1, 2, 3])
def test_known_spectrum_recipes(kind):
    """A = Q diag(evs) Q^T with the reference's seed spectra (Zipf 1/i, 1/sqrt(i), linear); top-k recovered."""
    n, k = 300, 20
    i = np.arange(1, n + 1, dtype=np.float64)
    evs = {1: 1.0 / i, 2: 1.0 / np.sqrt(i), 3: i / n}[kind]
    rng = np.random.default_rng(kind)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    A = (Q * evs) @ Q.T
    r = orc.block_ks_dense(A.astype(np.float32), k, blk=10, ncv=2 * k + 10)
    top = np.sort(evs)[::-1][:k]
    # kind 3 (evenly spaced, gap 1/n) converges slowly: the solver may exhaust maxit yet the values are accurate
    assert np.max(np.abs(r["evals"] - top) / top) <= 2e-4


def test_block_ks_matches_dense_truth_on_synthetic(tiny20):
    B = tiny20
    S = sp.csc_matrix((B["vals"].astype(np.float64), B["rows"], B["offs"]), shape=(B["V"], B["D"]))
    lam = np.linalg.eigvalsh((S @ S.T).toarray())[::-1]
    r = B["oracle"].block_ks(20)
    assert np.max(np.abs(np.sqrt(r["evals"]) - np.sqrt(lam[:20])) / np.sqrt(lam[:20])) <= 1e-5


def test_stop_rule_runs_one_extra_idempotent_iteration(gold):
    """converged only when sizes match AND the partition equals the last stored one."""
    g, m = gold
    lo10 = m.lloyds_projected(g["U"], g["C_proj3"].astype(np.float32), max_reps=50)
    lo = m.lloyds_projected(g["U"], lo10["C_lowd"], max_reps=50)
    assert lo["iters"] == 3  # iteration 1: sizes differ from zeros; 2: sizes equal, snapshot taken; 3: equal -> stop
    assert relerr(lo["C_lowd"], lo10["C_lowd"]) <= 1e-6
