"""Topic counts BETWEEN the BASELINE configs' (k = 200 and k = 1000 are held by test_gpu_big_k.py, k = 2048 by the edge test): the kernels of
the two Lloyd loops are instantiated by k — rows of `pt_tighten_ahead_k` of 1 … 4 x 256 coordinates (and the plain kernel beyond), group-bound
rows of `yy_lower_block` of 1 … 4 x 64 groups — and every instantiation must give the partition of the unbounded loop
(src/sparseMatrix.cpp:1494-1585 projected, :1587-1746 sparse) and, for Lloyd on B, the oracle's from the same centres."""
import numpy as np
import pytest

from conftest import corpus, upload

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k", [300, 600, 1200])
def test_both_lloyd_loops_at_other_topic_counts(hp, monkeypatch, k):
    B = corpus(4000, 20000, 40, 11)
    upload(hp, B)
    hp.compute_block_ks(k, seed=1, allow_noconv=True)
    g = hp.kmeans_init_on_projected_space(k, rng_seed=5)
    res = {}
    for mode in ("bounds", "none"):
        if mode == "none":
            monkeypatch.setenv("ISLE_NO_HAMERLY", "1")
        lp = hp.run_lloyds_on_projected_space(k, g["C_lowd"])
        cen = hp.left_multiply_by_U(lp["C_lowd"], fetch=True)
        ls = hp.run_lloyds(k, fetch_centers=False)
        res[mode] = (lp["assign"], lp["iters"], ls["assign"], ls["iters"], cen)
    monkeypatch.delenv("ISLE_NO_HAMERLY")
    assert res["none"][1] == res["bounds"][1] and res["none"][3] == res["bounds"][3], (res["none"][1], res["bounds"][1], res["none"][3], res["bounds"][3])
    assert np.array_equal(res["none"][0], res["bounds"][0]), float((res["none"][0] == res["bounds"][0]).mean())
    assert np.array_equal(res["none"][2], res["bounds"][2]), float((res["none"][2] == res["bounds"][2]).mean())
    # the by-document form of the Yinyang iteration (the by-group form is the default from 32 groups on)
    lp = hp.run_lloyds_on_projected_space(k, g["C_lowd"])
    hp.left_multiply_by_U(lp["C_lowd"], fetch=False)
    monkeypatch.setenv("ISLE_YY_MODE", "doc")
    ld = hp.run_lloyds(k, fetch_centers=False)
    monkeypatch.delenv("ISLE_YY_MODE")
    assert ld["iters"] == res["bounds"][3] and np.array_equal(ld["assign"], res["bounds"][2])
    # the oracle from the same lifted centres
    so = B["oracle"].lloyds_sparse(np.asfortranarray(res["bounds"][4]))
    same = float((so["assign"] == res["bounds"][2]).mean())
    assert same >= 0.999 and so["iters"] == res["bounds"][3], (same, so["iters"], res["bounds"][3])


@pytest.mark.parametrize("k", [20, 300])
def test_orthogonalisation_update_on_the_matrix_cores_agrees_with_the_fma_chains(hp, monkeypatch, k):
    """`F -= V H` (block-ks/ks_utils.h:129-175, the Gram-Schmidt passes of an expand step) runs on the matrix cores by default
    (`update_mfma_k`); `ISLE_UPDATE_MFMA=0` keeps the FMA chains of `update_k`.  Same sums in another order: the singular values agree to
    1e-5 (the contract is 1e-4 against the truth), the subspaces to 1e-3 in every principal angle, and both bases are orthonormal."""
    B = corpus(4000, 20000, 40, 11)
    upload(hp, B)
    r1 = hp.compute_block_ks(k, seed=1, allow_noconv=True)
    U1 = hp.get_U(k).astype(np.float64)
    monkeypatch.setenv("ISLE_UPDATE_MFMA", "0")
    r0 = hp.compute_block_ks(k, seed=1, allow_noconv=True)
    U0 = hp.get_U(k).astype(np.float64)
    monkeypatch.delenv("ISLE_UPDATE_MFMA")
    s1, s0 = np.asarray(r1["evals"], np.float64), np.asarray(r0["evals"], np.float64)  # eigenvalues of B B^T, descending
    assert np.max(np.abs(s1 - s0) / s0) < 2e-5, float(np.max(np.abs(s1 - s0) / s0))
    for U in (U1, U0):
        assert np.max(np.abs(U.T @ U - np.eye(k))) < 1e-5
    # clusters of nearly equal singular values rotate freely: compare the leading subspace below a gap
    gaps = (s0[:-1] - s0[1:]) / s0[:-1]
    cut = int(np.argmax(gaps[: max(k // 2, 2)])) + 1
    sv = np.linalg.svd(U1[:, :cut].T @ U0[:, :cut], compute_uv=False)
    assert sv.min() > 1.0 - 1e-3, (cut, float(sv.min()))


def test_pass2_stream_filled_by_buckets_is_the_stream_of_the_direct_scatter(hp, monkeypatch):
    """The pass-2 id stream of the LDS-banded Gram apply (stands for the MKL_SpSpTrProd constructor, include/matUtils.h:52-273) is filled by
    buckets of word positions with whole lines assembled in LDS (round 5); `ISLE_GL_FILL_BUCKETS=0` keeps the direct 2-byte scatter.  Both put
    the same ids into the same cells and every cell is sorted afterwards: `Z = B (B^T X)` must be bit-identical, for panels of 1, 10 and 25
    columns, on a corpus whose vocabulary is not a multiple of the bucket width."""
    B = corpus(4000, 20000, 40, 11)
    rng = np.random.default_rng(2)
    Xs = [rng.standard_normal((B["V"], b)).astype(np.float32) for b in (1, 10, 25)]
    upload(hp, B)
    assert hp.operator_form() in (-1, 1)
    Z1 = [hp.gram_apply(X) for X in Xs]
    assert hp.operator_form() == 1
    monkeypatch.setenv("ISLE_GL_FILL_BUCKETS", "0")
    upload(hp, B)  # a new upload: the operator is built again
    Z0 = [hp.gram_apply(X) for X in Xs]
    monkeypatch.delenv("ISLE_GL_FILL_BUCKETS")
    for a, b in zip(Z1, Z0):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    Zo = B["oracle"].gram_apply(Xs[1])
    assert np.linalg.norm(Z1[1] - Zo) / np.linalg.norm(Zo) <= 1e-5
