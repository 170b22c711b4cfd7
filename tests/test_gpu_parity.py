"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Tolerances (fp32 path):
  Gram apply             rel-Frobenius <= 1e-5
  sigma_i                rel <= 1e-4 for all i < k            (north_star contract)
  k-means (same U, injected seeds): >= 99 % identical assignments, centres rel-Frobenius <= 1e-3
"""
import numpy as np
import pytest

from conftest import corpus, relerr, subspace_cosines, upload

pytestmark = pytest.mark.gpu


def dense_eigs(B):
    import scipy.sparse as sp
    S = sp.csc_matrix((B["vals"].astype(np.float64), B["rows"], B["offs"]), shape=(B["V"], B["D"]))
    return np.linalg.eigvalsh((S @ S.T).toarray())[::-1]


@pytest.mark.parametrize("b", [1, 3, 4, 10, 16, 17, 32])
def test_gram_apply_matches_oracle(hp, tiny10, b):
    B = tiny10
    upload(hp, B)
    X = np.random.default_rng(b).standard_normal((B["V"], b)).astype(np.float32)
    Z = hp.gram_apply(X)
    Zo = B["oracle"].gram_apply(X)
    assert relerr(Z, Zo) <= 1e-5


def test_gram_apply_medium_and_properties(hp, small50):
    B = small50
    upload(hp, B)
    rng = np.random.default_rng(3)
    X = rng.standard_normal((B["V"], 10)).astype(np.float32)
    Y = rng.standard_normal((B["V"], 10)).astype(np.float32)
    ZX, ZY = hp.gram_apply(X), hp.gram_apply(Y)
    assert relerr(ZX, B["oracle"].gram_apply(X)) <= 1e-5
    # linearity, symmetry and positive semi-definiteness of X -> B B^T X
    assert relerr(hp.gram_apply(X + 2 * Y), ZX + 2 * ZY) <= 1e-5
    a = np.sum(X.astype(np.float64) * ZY, axis=0)
    b = np.sum(Y.astype(np.float64) * ZX, axis=0)
    assert np.allclose(a, b, rtol=1e-4, atol=1e-3 * np.abs(a).max())
    assert (np.sum(X.astype(np.float64) * ZX, axis=0) > 0).all()
    assert abs(hp.frobenius() - B["oracle"].frobenius()) <= 1e-5 * B["oracle"].frobenius()


def test_gram_apply_ragged_and_empty_columns(hp):
    # hand-built matrix: empty columns, single-entry columns, a dense column, last band partially filled
    V, D = 4500, 300
    rng = np.random.default_rng(0)
    cols = []
    for d in range(D):
        n = [0, 1, 7, 64, 65, 700][d % 6]
        if d == 17:
            n = V
        cols.append(np.sort(rng.choice(V, size=n, replace=False)).astype(np.uint32))
    offs = np.zeros(D + 1, np.int64)
    offs[1:] = np.cumsum([len(c) for c in cols])
    rows = np.concatenate(cols)
    vals = rng.uniform(0.5, 3.0, size=rows.shape[0]).astype(np.float32)
    from oracle.oracle import OracleCsc
    o = OracleCsc(V, D, vals, rows, offs)
    hp.upload_csc(V, vals, rows, offs)
    X = rng.standard_normal((V, 10)).astype(np.float32)
    assert relerr(hp.gram_apply(X), o.gram_apply(X)) <= 1e-5


def _ragged(V, D, seed, row_constant):
    rng = np.random.default_rng(seed)
    cols = []
    for d in range(D):
        n = [0, 1, 7, 64, 65, 700][d % 6]
        if d == 17:
            n = V
        cols.append(np.sort(rng.choice(V, size=min(n, V), replace=False)).astype(np.uint32))
    offs = np.zeros(D + 1, np.int64)
    offs[1:] = np.cumsum([len(c) for c in cols])
    rows = np.concatenate(cols)
    if row_constant:
        vals = rng.uniform(0.5, 3.0, size=V).astype(np.float32)[rows]
    else:
        vals = rng.uniform(0.5, 3.0, size=rows.shape[0]).astype(np.float32)
    return vals, rows, offs


@pytest.mark.parametrize("V,D", [(4500, 300), (4078, 64), (4079, 65), (8157, 4077), (3412, 64), (7000, 9000), (100, 5), (90000, 300)])
@pytest.mark.parametrize("b", [1, 5, 10, 12, 13, 25])
def test_gram_apply_lds_form_ragged(hp, V, D, b):
    # B = diag(s) * pattern (what threshold_and_copy builds): the LDS-banded form must be chosen and agree with the oracle;
    # sizes straddle the band size (4078 rows), slices of 64 and workgroup blocks of 4096, and the 81920-word vocabulary
    # parts of the build's LDS histograms; empty / dense columns included
    from oracle.oracle import OracleCsc
    vals, rows, offs = _ragged(V, D, 11, True)
    o = OracleCsc(V, D, vals, rows, offs)
    hp.upload_csc(V, vals, rows, offs)
    X = np.random.default_rng(b).standard_normal((V, b)).astype(np.float32)
    Z = hp.gram_apply(X)
    assert hp.operator_form() == 1
    # a dense 90000-entry column puts sequential fp32 sums (the oracle's, the reference's) 1.1e-5 off the fp64 product: the HIP
    # result is held to the fp64 truth at 1e-5 or 1.5 x the oracle's own distance from it, and to the oracle within that distance
    import scipy.sparse as sp
    M = sp.csc_matrix((vals.astype(np.float64), rows, offs), shape=(V, D))
    truth = M @ (M.T @ X.astype(np.float64))
    Zo = o.gram_apply(X)
    eo = relerr(Zo, truth)
    assert relerr(Z, truth) <= max(1e-5, 1.5 * eo)
    assert relerr(Z, Zo) <= 1e-5 + eo


@pytest.mark.parametrize("cus", ["3", "7"])
def test_whole_rounds_of_workgroups_of_adjacent_waves(hp, small50, monkeypatch, cus):
    """Pass 1 of a matrix that needs more than one round of workgroups (all of config 3 on one GPU: 1395 on 256 CUs) fills WHOLE rounds:
    more waves, the last quantile range only partly dealt, workgroups of adjacent waves (k_gl_build).  ISLE_GL_TEST_CUS lays a small
    matrix out as for a device of 3 or 7 CUs; ISLE_GL_ROUNDS=0 is the strided form.  Same operator (the slices are the same, so the
    bits are), same k-wide products."""
    from oracle.oracle import OracleCsc
    monkeypatch.setenv("ISLE_GL_TEST_CUS", cus)
    for V, D in [(9000, 70000), (4079, 33000)]:
        rng = np.random.default_rng(D)
        lens = rng.integers(1, 40, size=D)
        offs = np.zeros(D + 1, np.int64)
        offs[1:] = np.cumsum(lens)
        rows = np.concatenate([np.sort(rng.choice(V, size=int(n), replace=False)) for n in lens]).astype(np.uint32)
        vals = rng.uniform(0.5, 3.0, size=V).astype(np.float32)[rows]
        o = OracleCsc(V, D, vals, rows, offs)
        X = rng.standard_normal((V, 10)).astype(np.float32)
        Zo = o.gram_apply(X)
        Z = {}
        for rounds in ("1", "0"):
            monkeypatch.setenv("ISLE_GL_ROUNDS", rounds)
            hp.upload_csc(V, vals, rows, offs)
            Z[rounds] = hp.gram_apply(X)
            assert hp.operator_form() == 1 and relerr(Z[rounds], Zo) <= 1e-5
        assert np.array_equal(Z["1"].view(np.uint32), Z["0"].view(np.uint32))
    B, k = small50, 50
    U = B["oracle"].block_ks(k)["U"]
    seeds = np.random.default_rng(3).choice(B["D"], size=k, replace=False).astype(np.uint64)
    res = {}
    for rounds in ("1", "0"):
        monkeypatch.setenv("ISLE_GL_ROUNDS", rounds)
        monkeypatch.setenv("ISLE_WIDE_LDS", "1")
        upload(hp, B)
        hp.set_U(U)
        g = hp.kmeans_init_on_projected_space(k, inject_seeds=seeds)
        lg = hp.run_lloyds_on_projected_space(k, g["C_lowd"])
        res[rounds] = (g["C_lowd"], lg["assign"])
    assert np.array_equal(res["1"][0], res["0"][0]) and np.array_equal(res["1"][1], res["0"][1])


def test_bank_aware_placement_changes_the_order_of_the_sums_only(hp, monkeypatch):
    """gl_place_k (gram_lds.hip) gives a lane's entries the slots in which the rows of a ds_read_b128 lane group lie on different LDS
    banks; ISLE_GL_PLACE=0 leaves them packed at the front of their slots in ascending order.  Same operator: both forms against the
    oracle and against each other (the sums differ in their order), on ragged matrices that straddle band and slice
    boundaries; the placed form is bitwise reproducible across builds."""
    from oracle.oracle import OracleCsc
    for V, D in [(4500, 300), (4079, 65), (8157, 4077), (9000, 9000)]:
        vals, rows, offs = _ragged(V, D, 5, True)
        o = OracleCsc(V, D, vals, rows, offs)
        X = np.random.default_rng(V).standard_normal((V, 10)).astype(np.float32)
        Zo = o.gram_apply(X)
        Z = {}
        for place in ("1", "0", "1"):
            monkeypatch.setenv("ISLE_GL_PLACE", place)
            hp.upload_csc(V, vals, rows, offs)
            Zp = hp.gram_apply(X)
            assert hp.operator_form() == 1 and relerr(Zp, Zo) <= 1e-5
            if place in Z:
                assert np.array_equal(Zp.view(np.uint32), Z[place].view(np.uint32))
            Z[place] = Zp
        assert relerr(Z["1"], Z["0"]) <= 2e-6
    monkeypatch.delenv("ISLE_GL_PLACE")


@pytest.mark.parametrize("items", [("6", "4"), ("8", "8"), ("5", "7")])
def test_items_per_lane_of_the_id_streams_do_not_change_the_operator(hp, small50, monkeypatch, items):
    """ISLE_GL_G1 / ISLE_GL_G2 (gram_lds.hip): 4 ... 8 output items per lane of a wave (the build picks 4 ... 7 for pass 1 by its
    makespan model, 4 for pass 2).  Same operator, same k-wide products: Gram apply on ragged matrices (band and block boundaries,
    dense and empty columns) against the oracle, and the k-means chain against the form with four items."""
    from oracle.oracle import OracleCsc
    monkeypatch.setenv("ISLE_GL_G1", items[0])
    monkeypatch.setenv("ISLE_GL_G2", items[1])
    for V, D in [(4500, 300), (4079, 65), (8157, 4077), (7000, 9000), (90000, 300)]:
        vals, rows, offs = _ragged(V, D, 11, True)
        o = OracleCsc(V, D, vals, rows, offs)
        hp.upload_csc(V, vals, rows, offs)
        for b in (1, 10, 13):
            X = np.random.default_rng(b).standard_normal((V, b)).astype(np.float32)
            Z = hp.gram_apply(X)
            assert hp.operator_form() == 1
            assert relerr(Z, o.gram_apply(X)) <= 3e-5  # 3e-5: the oracle's own sequential sums on the 90000-entry column
    B, k = small50, 50
    U = B["oracle"].block_ks(k)["U"]
    seeds = np.random.default_rng(3).choice(B["D"], size=k, replace=False).astype(np.uint64)
    res = {}
    for form in (items, ("4", "4")):
        monkeypatch.setenv("ISLE_GL_G1", form[0])
        monkeypatch.setenv("ISLE_GL_G2", form[1])
        upload(hp, B)
        r = hp.compute_block_ks(k, allow_noconv=True)
        hp.set_U(U)
        g = hp.kmeans_init_on_projected_space(k, inject_seeds=seeds)
        lg = hp.run_lloyds_on_projected_space(k, g["C_lowd"])
        hp.left_multiply_by_U(lg["C_lowd"], fetch=False)
        sg = hp.run_lloyds(k)
        res[form] = (r["evals"], g["C_lowd"], lg["assign"], sg["assign"])
    a, b = res[items], res[("4", "4")]
    assert np.max(np.abs(a[0] - b[0]) / b[0]) <= 1e-5 and relerr(a[1], b[1]) <= 1e-5
    assert (a[2] == b[2]).mean() >= 0.999 and (a[3] == b[3]).mean() >= 0.999


def test_gram_apply_forms_agree(hp, small50):
    # the two forms of the operator on the same thresholded matrix; a perturbed value switches to the gather form
    B = small50
    upload(hp, B)
    X = np.random.default_rng(5).standard_normal((B["V"], 10)).astype(np.float32)
    Z1 = hp.gram_apply(X)
    assert hp.operator_form() == 1
    vals = B["vals"].copy()
    vals[len(vals) // 2] *= 1.5
    hp.upload_csc(B["V"], vals, B["rows"], B["offs"])
    Z0 = hp.gram_apply(X)
    assert hp.operator_form() == 0
    vals[len(vals) // 2] = B["vals"][len(vals) // 2]
    assert relerr(Z1, B["oracle"].gram_apply(X)) <= 1e-5
    # Z0 is the product with the perturbed matrix: only the rows / columns touched by that entry differ
    from oracle.oracle import OracleCsc
    vals2 = B["vals"].copy()
    vals2[len(vals2) // 2] *= 1.5
    o2 = OracleCsc(B["V"], B["D"], vals2, B["rows"], B["offs"])
    assert relerr(Z0, o2.gram_apply(X)) <= 1e-5


@pytest.mark.parametrize("n", [1, 2, 7, 30, 57, 200])
def test_eig_sym_matches_lapack(hp, n):
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n))
    S = (A @ A.T + (A + A.T)).astype(np.float32)  # symmetric, indefinite in general
    e, v = hp.eig_sym(S)
    er = np.linalg.eigvalsh(S.astype(np.float64))[::-1]
    scale = max(np.abs(er).max(), 1e-30)
    assert np.abs(e - er).max() <= 2e-6 * scale
    assert np.abs(v.astype(np.float64).T @ v - np.eye(n)).max() <= 1e-5
    assert np.abs(S.astype(np.float64) @ v - v * e).max() <= 2e-5 * scale


@pytest.mark.parametrize("case", ["identity", "zero", "rank1", "pairs", "cluster", "graded", "n3", "n17"])
def test_eig_sym_hard_spectra(hp, case):
    """The tridiagonal solver (bisection + twisted factorisation) needs separated eigenvalues for orthogonal vectors; exact
    multiplicities must be caught by its orthogonality check and handed to the Jacobi solver.  Either way the result has to be
    an orthonormal eigenbasis with LAPACK's eigenvalues."""
    rng = np.random.default_rng(7)
    n = 64
    if case == "identity":
        S = np.eye(n)
    elif case == "zero":
        S = np.zeros((n, n))
    elif case == "rank1":
        x = rng.standard_normal(n)
        S = np.outer(x, x)
    else:
        if case == "n3":
            n = 3
        if case == "n17":
            n = 17
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        if case == "pairs":
            lam = np.repeat(np.arange(n // 2, 0, -1.0), 2)          # every eigenvalue twice
        elif case == "cluster":
            lam = np.concatenate([[1e3], 10 * (1 + 1e-5 * np.arange(n - 1))])  # relative gaps 1e-5
        elif case == "graded":
            lam = 10.0 ** (-np.arange(n) / 8.0)
        else:
            lam = np.arange(n, 0, -1.0)
        S = (Q * lam) @ Q.T
    S = S.astype(np.float32)
    e, v = hp.eig_sym(S)
    er = np.linalg.eigvalsh(S.astype(np.float64))[::-1]
    scale = max(np.abs(er).max(), 1e-30)
    assert np.abs(e - er).max() <= 2e-6 * scale
    assert np.abs(v.astype(np.float64).T @ v - np.eye(n)).max() <= 1e-5
    assert np.abs(S.astype(np.float64) @ v - v * e).max() <= 2e-5 * scale


@pytest.mark.parametrize("which", ["tiny10", "tiny20"])
def test_block_ks_sigma(hp, which, request):
    B = request.getfixturevalue(which)
    k = 10 if which == "tiny10" else 20
    upload(hp, B)
    r = hp.compute_block_ks(k, allow_noconv=True)
    o = B["oracle"].block_ks(k)
    truth = dense_eigs(B)[:k]
    sig, sig_o, sig_t = np.sqrt(r["evals"]), np.sqrt(o["evals"]), np.sqrt(truth)
    assert np.max(np.abs(sig - sig_o) / sig_o) <= 1e-4, (sig, sig_o)
    assert np.max(np.abs(sig - sig_t) / sig_t) <= 1e-4
    U = hp.get_U(k)
    assert np.abs(U.astype(np.float64).T @ U - np.eye(k)).max() <= 1e-4
    # invariant subspace: || A U - U diag(lambda) || small relative to lambda_1
    AU = B["oracle"].gram_apply(U)
    assert np.abs(AU - U * r["evals"]).max() <= 2e-3 * r["evals"][0]
    # subspace agreement with the oracle away from the edge of the cluster
    cos = subspace_cosines(U[:, : k - 2], o["U"])
    assert cos.min() >= 1 - 1e-3


@pytest.mark.parametrize("name", ["tiny10", "tiny20", "small50", "mid30"])
def test_block_ks_matches_reference_solver_goldens(hp, name):
    """sigma and U against the fixtures computed by the reference's own Spectra eigensolver (tests/golden/ref_spectra.npz,
    tests/golden/make_golden_ref.py; compute_Spectra, /root/reference/src/sparseMatrix.cpp:1161-1190)."""
    import os
    from conftest import ROOT, corpus
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_spectra.npz"))
    V, D, k, seed = (int(x) for x in g[name + "_params"])
    B = corpus(V, D, k, seed)
    assert np.array_equal(np.array([B["V"], B["D"], B["nnz"], int(B["rows"].astype(np.int64).sum())]), g[name + "_sig"])
    upload(hp, B)
    r = hp.compute_block_ks(k, allow_noconv=True)
    s, s_ref = np.sqrt(r["evals"].astype(np.float64)), np.sqrt(g[name + "_evalues"].astype(np.float64))
    assert np.max(np.abs(s - s_ref) / s_ref) <= 1e-4  # BASELINE.json tolerance
    if name + "_U" in g:
        assert subspace_cosines(hp.get_U(k)[:, : k - 2], g[name + "_U"]).min() >= 1 - 1e-3
        Z = hp.gram_apply(g[name + "_U"])  # the reference's eigenvectors under the HIP operator
        assert np.abs(Z - g[name + "_U"] * g[name + "_evalues"]).max() <= 2e-3 * g[name + "_evalues"][0]


def test_two_gram_schmidt_passes_equal_the_references_three(hp, small50, monkeypatch):
    """BlockKs::expand orthogonalises three times (CGS + 2 DGKS, restarted_block_ks.h:83-91); the HIP path makes two passes by default
    (the third block of coefficients is at rounding level).  Same sigma, same number of restarts and operator applications."""
    B, k = small50, 50
    res = {}
    for passes in ("2", "3"):
        monkeypatch.setenv("ISLE_KS_ORTHO_PASSES", passes)
        upload(hp, B)
        r = hp.compute_block_ks(k, seed=3)
        U = hp.get_U(k)
        res[passes] = (r["evals"], r["restarts"], r["napplies"], np.abs(U.astype(np.float64).T @ U - np.eye(k)).max())
    a, b = res["2"], res["3"]
    assert np.max(np.abs(a[0] - b[0]) / b[0]) <= 2e-6
    assert a[1] == b[1] and a[2] == b[2]
    assert a[3] <= 1e-4 and b[3] <= 1e-4


def test_block_ks_medium(hp, small50):
    B = small50
    upload(hp, B)
    r = hp.compute_block_ks(50)
    o = B["oracle"].block_ks(50)
    assert r["rc"] == 0 and r["nconv"] == 50
    sig, sig_o = np.sqrt(r["evals"]), np.sqrt(o["evals"])
    assert np.max(np.abs(sig - sig_o) / sig_o) <= 1e-4
    assert (np.diff(r["evals"]) <= 1e-3 * r["evals"][0]).all()  # descending


def test_block_ks_on_a_vocabulary_that_is_not_a_multiple_of_four(hp):
    """V = 3001: the basis columns are not 16-byte aligned, so the orthogonalisation's matrix-core kernels (vtf_mfma_k with its panel chunk
    staged in LDS, update_mfma_k) take their element-wise paths at every basis width, the last 1024-row chunk holds 953 rows, and the
    panel QR and the rotation see an odd leading dimension.  k = 40: ncv = 90, basis widths 10 ... 90 on both sides of the 64-column switch."""
    from conftest import corpus
    B, k = corpus(3001, 9000, 40, 11), 40
    upload(hp, B)
    r = hp.compute_block_ks(k, allow_noconv=True)
    o = B["oracle"].block_ks(k)
    sig, sig_o = np.sqrt(r["evals"]), np.sqrt(o["evals"])
    assert np.max(np.abs(sig - sig_o) / sig_o) <= 1e-4, (sig, sig_o)
    U = hp.get_U(k)
    assert np.abs(U.astype(np.float64).T @ U - np.eye(k)).max() <= 1e-4
    AU = B["oracle"].gram_apply(U)
    assert np.abs(AU - U * r["evals"]).max() <= 2e-3 * r["evals"][0]


def _kmeans_setup(hp, B, k):
    upload(hp, B)
    o = B["oracle"].block_ks(k)
    hp.set_U(o["U"])
    return o["U"]


def test_kmeanspp_injected_matches_oracle(hp, small50):
    B, k = small50, 50
    U = _kmeans_setup(hp, B, k)
    ko = B["oracle"].kmeanspp(U, k, seed=5)
    g = hp.kmeans_init_on_projected_space(k, inject_seeds=ko["seeds"])
    assert (g["seeds"] == ko["seeds"]).all()
    assert g["rounds"] == ko["rounds"]
    assert relerr(g["C_lowd"], ko["C_lowd"]) <= 1e-5
    md, mdo = hp.get_min_dist(), ko["min_dist"]
    assert np.abs(md - mdo).max() <= 1e-4 * mdo.max()
    assert abs(g["residual"] - ko["residual"]) <= 1e-3 * ko["residual"]


def test_kmeanspp_free_run_is_distributionally_sane(hp, small50):
    B, k = small50, 50
    U = _kmeans_setup(hp, B, k)
    g = hp.kmeans_init_on_projected_space(k, rng_seed=11)
    assert len(set(g["seeds"].tolist())) == k and g["seeds"].max() < B["D"]
    pots = [B["oracle"].kmeanspp(U, k, seed=s)["residual"] for s in range(1, 6)]
    assert 0.5 * min(pots) <= g["residual"] <= 2.0 * max(pots)
    # the seeds' own min-distance is ~0 (src/sparseMatrix.cpp:2175 assert); the seeds drawn in the last
    # round (at most 1 + ceil(sqrt(k)) of them) have not been folded into min_dist yet
    md = hp.get_min_dist()
    last = 1 + int(np.ceil(np.sqrt(k)))
    assert md[g["seeds"][: k - last].astype(np.int64)].max() <= 1e-3 * md.max()


def test_lloyds_projected_matches_oracle(hp, small50):
    B, k = small50, 50
    U = _kmeans_setup(hp, B, k)
    ko = B["oracle"].kmeanspp(U, k, seed=5)
    lo = B["oracle"].lloyds_projected(U, ko["C_lowd"])
    lg = hp.run_lloyds_on_projected_space(k, ko["C_lowd"])
    assert lg["iters"] == lo["iters"]
    assert (lg["assign"] == lo["assign"]).mean() >= 0.999
    assert relerr(lg["C_lowd"], lo["C_lowd"]) <= 1e-3


def test_lift_and_sparse_lloyds_match_oracle(hp, small50):
    from oracle.oracle import lift
    B, k = small50, 50
    U = _kmeans_setup(hp, B, k)
    ko = B["oracle"].kmeanspp(U, k, seed=5)
    lo = B["oracle"].lloyds_projected(U, ko["C_lowd"])
    cen_o = lift(U, lo["C_lowd"])
    cen_g = hp.left_multiply_by_U(lo["C_lowd"])
    assert relerr(cen_g, cen_o) <= 1e-5
    so = B["oracle"].lloyds_sparse(cen_o)
    sg = hp.run_lloyds(k)  # device-resident lifted centres
    assert sg["iters"] == so["iters"]
    assert (sg["assign"] == so["assign"]).mean() >= 0.999
    assert relerr(sg["centers"], so["centers"]) <= 1e-3
    assert np.bincount(sg["assign"], minlength=k).sum() == B["D"]  # partition complete (trainer.cpp:567-570)
    # host-provided centres take the same path
    sg2 = hp.run_lloyds(k, centers=cen_o)
    assert (sg2["assign"] == so["assign"]).mean() >= 0.999


@pytest.mark.parametrize("k", [50, 37])
def test_first_sparse_assignment_through_the_projection(hp, small50, monkeypatch, k):
    """Lloyd on B after lift: the centres are U C^T, so the first assignment's k-wide sparse product B^T C equals the dense product
    P C^T on the projection already on the device (api.cpp, isle_hip_lloyds_sparse).  Same partition as the sparse product
    (ISLE_FIRST_ASSIGN=sparse), same as the oracle, same iteration count; host-provided centres keep the sparse product."""
    from oracle.oracle import lift
    B = small50
    U = B["oracle"].block_ks(k)["U"]
    seeds = np.random.default_rng(k).choice(B["D"], size=k, replace=False).astype(np.uint64)
    res = {}
    for mode in ("projection", "sparse"):
        monkeypatch.setenv("ISLE_FIRST_ASSIGN", mode)
        upload(hp, B)
        hp.set_U(U)
        g = hp.kmeans_init_on_projected_space(k, inject_seeds=seeds)
        lg = hp.run_lloyds_on_projected_space(k, g["C_lowd"])
        hp.left_multiply_by_U(lg["C_lowd"], fetch=False)
        hp.timing_enable(True)
        hp.timing_reset()
        sg = hp.run_lloyds(k)
        res[mode] = (sg["assign"], sg["centers"], sg["iters"], lg["C_lowd"])
        hp.timing_enable(False)
    monkeypatch.delenv("ISLE_FIRST_ASSIGN")
    a, b = res["projection"], res["sparse"]
    assert a[2] == b[2]
    assert (a[0] == b[0]).mean() >= 0.999 and relerr(a[1], b[1]) <= 1e-4
    so = B["oracle"].lloyds_sparse(lift(U, a[3]))
    assert (a[0] == so["assign"]).mean() >= 0.999 and a[2] == so["iters"]
    assert relerr(a[1], so["centers"]) <= 1e-3


def test_gather_form_forced_by_env(hp, small50, monkeypatch):
    """ISLE_GRAM_LDS=0 keeps the gather kernels (and the chunk-major cells the centroid update then walks) covered on a
    thresholded matrix: same sigma, same sparse-Lloyd partition as the LDS-banded form."""
    from oracle.oracle import lift
    B, k = small50, 50
    res = {}
    for form in (1, 0):
        monkeypatch.setenv("ISLE_GRAM_LDS", str(form))
        upload(hp, B)
        r = hp.compute_block_ks(k)
        assert hp.operator_form() == form
        U = B["oracle"].block_ks(k)["U"] if "U_or" not in res else res["U_or"]
        res["U_or"] = U
        hp.set_U(U)
        ko = B["oracle"].kmeanspp(U, k, seed=5)
        lo = B["oracle"].lloyds_projected(U, ko["C_lowd"])
        hp.left_multiply_by_U(lo["C_lowd"], fetch=False)
        sg = hp.run_lloyds(k)
        res[form] = (np.sqrt(r["evals"]), sg["assign"], sg["centers"])
    monkeypatch.delenv("ISLE_GRAM_LDS")
    assert np.max(np.abs(res[1][0] - res[0][0]) / res[0][0]) <= 1e-5
    assert (res[1][1] == res[0][1]).mean() >= 0.999
    assert relerr(res[1][2], res[0][2]) <= 1e-4


@pytest.mark.parametrize("panel", ["10", "8"])
@pytest.mark.parametrize("k", [7, 13, 25, 30, 37])
def test_wide_products_odd_topic_counts(hp, small50, monkeypatch, k, panel):
    """k-wide products (projection, first full assignment) with topic counts that are not multiples of the panel (10 or 8 columns per
    pass of the pass-1 stream: ISLE_GL_PANEL; steps of 10 leave the panels 8-byte aligned inside the output rows) or of the padded
    row (ldk = 4 ceil(k/4)): the LDS-banded form, the row-gather form and the oracle agree."""
    from oracle.oracle import lift
    B = small50
    monkeypatch.setenv("ISLE_GL_PANEL", panel)
    rng = np.random.default_rng(k)
    U = B["oracle"].block_ks(k)["U"]
    seeds = rng.choice(B["D"], size=k, replace=False).astype(np.uint64)
    res = {}
    for form in ("ISLE_WIDE_LDS", "ISLE_WIDE_GATHER"):
        monkeypatch.setenv(form, "1")
        upload(hp, B)
        hp.set_U(U)
        g = hp.kmeans_init_on_projected_space(k, inject_seeds=seeds)
        lg = hp.run_lloyds_on_projected_space(k, g["C_lowd"])
        hp.left_multiply_by_U(lg["C_lowd"], fetch=False)
        sg = hp.run_lloyds(k)
        res[form] = (g["C_lowd"], hp.get_min_dist(), lg["assign"], sg["assign"], sg["centers"])
        monkeypatch.delenv(form)
    a, b = res["ISLE_WIDE_LDS"], res["ISLE_WIDE_GATHER"]
    assert relerr(a[0], b[0]) <= 1e-5 and np.abs(a[1] - b[1]).max() <= 1e-4 * b[1].max()
    assert (a[2] == b[2]).mean() >= 0.995 and (a[3] == b[3]).mean() >= 0.995
    lo = B["oracle"].lloyds_projected(U, a[0])
    assert (a[2] == lo["assign"]).mean() >= 0.999
    so = B["oracle"].lloyds_sparse(lift(U, lo["C_lowd"]))
    assert (a[3] == so["assign"]).mean() >= 0.999


def test_sparse_lloyd_wide_vocabulary(hp, monkeypatch):
    """Vocabulary beyond one 81920-word part of the LDS histograms (two parts in the operator build and in the counting centroid
    update): Lloyd on B from given centres, LDS-banded / counting forms against the gather / float-histogram forms."""
    V, D, k = 90_000, 3_000, 12
    rng = np.random.default_rng(21)
    cols = [np.sort(rng.choice(V, size=int(n), replace=False)).astype(np.uint32) for n in rng.integers(20, 120, size=D)]
    offs = np.zeros(D + 1, np.int64)
    offs[1:] = np.cumsum([len(c) for c in cols])
    rows = np.concatenate(cols)
    vals = rng.uniform(0.5, 2.0, size=V).astype(np.float32)[rows]
    seeds = rng.choice(D, size=k, replace=False)
    cen = np.zeros((V, k), np.float32, order="F")
    for j, d in enumerate(seeds):
        cen[rows[offs[d]:offs[d + 1]], j] = vals[offs[d]:offs[d + 1]]
    res = {}
    for form in (1, 0):
        monkeypatch.setenv("ISLE_GRAM_LDS", str(form))
        hp.upload_csc(V, vals, rows, offs)
        X = rng.standard_normal((V, 10)).astype(np.float32) if form == 1 else res["X"]
        res["X"] = X
        Z = hp.gram_apply(X)
        assert hp.operator_form() == form
        sg = hp.run_lloyds(k, centers=cen)
        res[form] = (Z, sg["assign"], sg["centers"], sg["iters"])
    monkeypatch.delenv("ISLE_GRAM_LDS")
    assert relerr(res[1][0], res[0][0]) <= 1e-5
    assert res[1][3] == res[0][3]
    assert (res[1][1] == res[0][1]).mean() >= 0.999
    assert relerr(res[1][2], res[0][2]) <= 1e-4


def test_full_hot_path_end_to_end(hp, small50):
    """src/trainer.cpp:490-571 call sequence on the GPU, checked against planted topics and the oracle's ranges."""
    B, k = small50, 50
    upload(hp, B)
    r = hp.compute_block_ks(k)
    sig_o = np.sqrt(B["oracle"].block_ks(k)["evals"])
    assert np.max(np.abs(np.sqrt(r["evals"]) - sig_o) / sig_o) <= 1e-4
    g = hp.kmeans_init_on_projected_space(k, rng_seed=3)
    lg = hp.run_lloyds_on_projected_space(k, g["C_lowd"])
    hp.left_multiply_by_U(lg["C_lowd"], fetch=False)
    sg = hp.run_lloyds(k)
    sizes = np.bincount(sg["assign"], minlength=k)
    assert sizes.sum() == B["D"]
    # agreement with planted dominant topics (majority label per cluster); the reference itself
    # reaches 0.81-0.87 on this kind of corpus
    planted = B["planted"]
    agree = sum(np.bincount(planted[sg["assign"] == c]).max() for c in range(k) if sizes[c] > 0) / B["D"]
    assert agree >= 0.75, agree
    # ... and no worse than the oracle's own partition from the same eigenvectors and the same seeds
    from oracle.oracle import lift
    o, U = B["oracle"], hp.get_U(k)
    lo = o.lloyds_projected(U, g["C_lowd"])
    so = o.lloyds_sparse(lift(U, lo["C_lowd"]))
    so_sizes = np.bincount(so["assign"], minlength=k)
    agree_o = sum(np.bincount(planted[so["assign"] == c]).max() for c in range(k) if so_sizes[c] > 0) / B["D"]
    assert agree >= agree_o - 0.005, (agree, agree_o)
    assert (sg["assign"] == so["assign"]).mean() >= 0.999


def test_config1_size_sigma_and_kmeans(hp):
    """BASELINE.json configs[0]: vocab 10k, docs 50k, ~5M nnz, k = 50 (the reference's own CPU-runnable case)."""
    B = corpus(10_000, 50_000, 50, 12345)
    k = 50
    upload(hp, B)
    r = hp.compute_block_ks(k)
    o = B["oracle"].block_ks(k)
    assert np.max(np.abs(np.sqrt(r["evals"]) - np.sqrt(o["evals"])) / np.sqrt(o["evals"])) <= 1e-4
    hp.set_U(o["U"])
    ko = B["oracle"].kmeanspp(o["U"], k, seed=2)
    g = hp.kmeans_init_on_projected_space(k, inject_seeds=ko["seeds"])
    lo = B["oracle"].lloyds_projected(o["U"], ko["C_lowd"])
    lg = hp.run_lloyds_on_projected_space(k, g["C_lowd"])
    assert lg["iters"] == lo["iters"] and (lg["assign"] == lo["assign"]).mean() >= 0.999
    from oracle.oracle import lift
    hp.left_multiply_by_U(lo["C_lowd"], fetch=False)
    so, sg = B["oracle"].lloyds_sparse(lift(o["U"], lo["C_lowd"])), hp.run_lloyds(k)
    assert sg["iters"] == so["iters"] and (sg["assign"] == so["assign"]).mean() >= 0.999


def test_config4_importance_sampled_matrix(hp):
    """BASELINE.json configs[3] shape of input: sample=1, sample_rate=0.1 -> B keeps floor(0.1*D)+1 heavy documents;
    the hot path then runs unchanged on the sampled B (src/trainer.cpp:476-484)."""
    B = corpus(6000, 40000, 20, 11, sample_rate=0.1)
    assert B["D"] == 4001
    k = 20
    upload(hp, B)
    r = hp.compute_block_ks(k, allow_noconv=True)
    o = B["oracle"].block_ks(k)
    assert np.max(np.abs(np.sqrt(r["evals"]) - np.sqrt(o["evals"])) / np.sqrt(o["evals"])) <= 1e-4
    X = np.random.default_rng(1).standard_normal((B["V"], 10)).astype(np.float32)
    assert relerr(hp.gram_apply(X), B["oracle"].gram_apply(X)) <= 1e-5
    hp.set_U(o["U"])
    ko = B["oracle"].kmeanspp(o["U"], k, seed=4)
    g = hp.kmeans_init_on_projected_space(k, inject_seeds=ko["seeds"])
    lo, lg = B["oracle"].lloyds_projected(o["U"], ko["C_lowd"]), hp.run_lloyds_on_projected_space(k, g["C_lowd"])
    assert (lg["assign"] == lo["assign"]).mean() >= 0.999


def test_operator_is_bitwise_reproducible_across_builds(hp, small50):
    """The reference's operator is bitwise reproducible.  The pass-2 stream is placed by LDS atomics whose arrival
    order changes from run to run; gl_sort2_k orders every (word, document band) cell afterwards, so Z — and sigma, and every
    borderline k-means assignment downstream — is a function of B alone: five rebuilds of the operator give identical bits."""
    B = small50
    X = np.random.default_rng(5).standard_normal((B["V"], 10)).astype(np.float32)
    ref = None
    for _ in range(5):
        upload(hp, B)  # drops the operator: the next apply rebuilds both streams
        Z = hp.gram_apply(X)
        assert hp.operator_form() == 1
        if ref is None:
            ref = Z
        assert np.array_equal(Z.view(np.uint32), ref.view(np.uint32))
    r1 = hp.compute_block_ks(50, seed=3)
    r2 = hp.compute_block_ks(50, seed=3)
    assert np.array_equal(r1["evals"].view(np.uint32), r2["evals"].view(np.uint32))


def test_kmeanspp_dice_thrown_on_the_device_pick_the_hosts_seeds(hp, small50, monkeypatch):
    """One rank: a round's dice are total x (host-drawn fraction); search_frac_k forms that product on the device, where the total
    lives, and searches it in the same launch (one host round trip per round).  The IEEE double product is the same on both sides,
    so the seeds, the round count and the residual equal those of the two-trip form (which several ranks still use)."""
    B, k = small50, 50
    _kmeans_setup(hp, B, k)
    g1 = hp.kmeans_init_on_projected_space(k, rng_seed=17)
    monkeypatch.setenv("ISLE_KMPP_HOST_DICE", "1")
    g0 = hp.kmeans_init_on_projected_space(k, rng_seed=17)
    assert (g0["seeds"] == g1["seeds"]).all() and g0["rounds"] == g1["rounds"]
    assert g0["residual"] == g1["residual"]
    assert np.array_equal(g0["C_lowd"], g1["C_lowd"])


def test_roctx_family_markers_leave_the_results_alone(hp, tiny10, monkeypatch):
    """ISLE_ROCTX=1 wraps every kernel family's launches in a roctx range (SURVEY 5.1's build hook; `rocprofv3 --marker-trace` shows them,
    profiles/r05_roctx_marker_trace_sample.txt).  The markers are host-side API calls: same bits with and without."""
    B = tiny10
    upload(hp, B)
    X = np.random.default_rng(3).standard_normal((B["V"], 10)).astype(np.float32)
    Z0 = hp.gram_apply(X)
    monkeypatch.setenv("ISLE_ROCTX", "1")
    Z1 = hp.gram_apply(X)
    r = hp.compute_block_ks(10, allow_noconv=True)
    monkeypatch.delenv("ISLE_ROCTX")
    assert np.array_equal(Z0.view(np.uint32), Z1.view(np.uint32))
    assert r["nconv"] == 10


@pytest.mark.parametrize("k", [163, 250, 321])
def test_grouped_projection_at_topic_counts_that_end_inside_a_group(hp, small50, monkeypatch, k):
    """The grouped form of P = U^T B (k_gl_wide: sixteen 10-column panels per group, rows assembled by gl_wide_assemble_k) where the last
    group is short, ends in an 8-column panel and carries the row's padding columns (ldk = 4 ceil(k / 4)): rows of P against fp64 products,
    and against the direct form (ISLE_GL_WIDE_GROUPED=0) bit for bit; the norms (summed in another order) through the k-means++ distances."""
    B = small50
    rng = np.random.default_rng(k)
    U = np.asfortranarray(np.linalg.qr(rng.standard_normal((B["V"], k)))[0].astype(np.float32))
    seeds = np.sort(rng.choice(B["D"], size=k, replace=False)).astype(np.uint64)
    res = {}
    for name, env in (("grouped", None), ("direct", "0")):
        if env:
            monkeypatch.setenv("ISLE_GL_WIDE_GROUPED", env)
        upload(hp, B)
        hp.set_U(U)
        g = hp.kmeans_init_on_projected_space(k, inject_seeds=seeds)
        res[name] = (g["C_lowd"], hp.get_min_dist())
        if env:
            monkeypatch.delenv("ISLE_GL_WIDE_GROUPED")
    assert np.array_equal(res["grouped"][0].view(np.uint32), res["direct"][0].view(np.uint32))
    assert np.abs(res["grouped"][1] - res["direct"][1]).max() <= 1e-5 * res["direct"][1].max()
    U64 = U.astype(np.float64)
    for i in range(0, k, 7):
        d = int(seeds[i])
        lo, hi = B["offs"][d], B["offs"][d + 1]
        ref = (B["vals"][lo:hi].astype(np.float64)[:, None] * U64[B["rows"][lo:hi]]).sum(0)
        assert np.abs(res["grouped"][0][i] - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


def test_pipelined_expand_equals_the_synchronous_form(hp, small50, monkeypatch):
    """The expand loop (api_ks.cpp) enqueues step i + 1 behind the panel QR of step i and fetches step i's mailbox on a copy stream of its own
    (round 6: double-buffered device mailbox); ISLE_KS_SYNC=1 runs one step at a time with a host synchronisation each.  Same kernels on the same
    data in the same order: eigenvalues, U and the work counts must agree bit for bit — a mailbox read before its step had finished writing it,
    or a coefficient block landing in a mailbox whose copy was still in flight, would show here."""
    upload(hp, small50)
    r0 = hp.compute_block_ks(50, seed=11)
    U0 = hp.get_U(50)
    monkeypatch.setenv("ISLE_KS_SYNC", "1")
    r1 = hp.compute_block_ks(50, seed=11)
    U1 = hp.get_U(50)
    monkeypatch.delenv("ISLE_KS_SYNC")
    r2 = hp.compute_block_ks(50, seed=11)
    assert r0["napplies"] == r1["napplies"] == r2["napplies"] and r0["restarts"] == r1["restarts"]
    assert np.array_equal(r0["evals"].view(np.uint32), r1["evals"].view(np.uint32))
    assert np.array_equal(r0["evals"].view(np.uint32), r2["evals"].view(np.uint32))
    assert np.array_equal(U0.view(np.uint32), U1.view(np.uint32))
