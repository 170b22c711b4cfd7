"""isle_hip_upload_csc checks what the reference's operator constructor asserts (include/matUtils.h:138-148: row ids below the vocabulary
size, strictly ascending inside a column) and the offsets — the entries on the device, behind the copy (a kernel over 1 B nonzeros takes
milliseconds where the host loop took seconds)."""
import numpy as np
import pytest

from isle_amd._lib import IsleHipError

pytestmark = pytest.mark.gpu


def _csc(V=50, D=40, seed=0):
    rng = np.random.default_rng(seed)
    cols = [np.sort(rng.choice(V, size=int(n), replace=False)).astype(np.uint32) for n in rng.integers(1, 9, size=D)]
    offs = np.zeros(D + 1, np.int64)
    offs[1:] = np.cumsum([len(c) for c in cols])
    rows = np.concatenate(cols)
    s = rng.uniform(0.5, 2.0, size=V).astype(np.float32)
    return V, s[rows], rows, offs


def test_upload_refuses_bad_matrices_and_then_takes_a_good_one(hp):
    V, vals, rows, offs = _csc()
    bad = rows.copy()
    bad[offs[7]] = V + 3  # out of range, column 7
    with pytest.raises(IsleHipError, match="out of range in column 7"):
        hp.upload_csc(V, vals, bad, offs)
    bad = rows.copy()
    c = next(d for d in range(len(offs) - 1) if offs[d + 1] - offs[d] >= 2)
    bad[offs[c]], bad[offs[c] + 1] = rows[offs[c] + 1], rows[offs[c]]  # two entries of a column exchanged
    with pytest.raises(IsleHipError, match="not strictly ascending in column %d" % c):
        hp.upload_csc(V, vals, bad, offs)
    bad = rows.copy()
    bad[offs[c] + 1] = bad[offs[c]]  # a duplicate
    with pytest.raises(IsleHipError, match="not strictly ascending in column %d" % c):
        hp.upload_csc(V, vals, bad, offs)
    boffs = offs.copy()
    boffs[5] = boffs[6] + 1
    with pytest.raises(IsleHipError, match="offsets not monotone"):
        hp.upload_csc(V, vals, rows, boffs)
    hp.upload_csc(V, vals, rows, offs)  # the context is still usable
    X = np.random.default_rng(1).standard_normal((V, 10)).astype(np.float32)
    Z = hp.gram_apply(X)
    import scipy.sparse as sp
    Bm = sp.csc_matrix((vals.astype(np.float64), rows, offs), shape=(V, len(offs) - 1))
    ref = Bm @ (Bm.T @ X.astype(np.float64))
    assert np.linalg.norm(Z - ref) <= 1e-5 * np.linalg.norm(ref)


def test_a_smaller_matrix_after_a_large_one_releases_and_recomputes():
    """Uploading a far smaller matrix releases the derived buffers sized for the large one (isle_trim_derived, api.cpp: the projection and
    its copies, product and build scratch — here 250 000 x 12 floats of projection against the 2048 x 1000 a 1000-document matrix could ever
    need); nothing computed afterwards may see stale contents: the whole chain on the small matrix equals a fresh context's bit for bit, and
    going back to the large matrix reproduces its first run."""
    import isle_amd
    from tools.synth import make_B
    big, small = make_B(2000, 250_000, 10, 3), make_B(2000, 1000, 10, 4)

    def chain(h, B, k=10):
        h.upload_csc(B["V"], B["vals"], B["rows"], B["offs"])
        r = h.compute_block_ks(k, seed=2)
        g = h.kmeans_init_on_projected_space(k, rng_seed=5)
        lp = h.run_lloyds_on_projected_space(k, g["C_lowd"])
        h.left_multiply_by_U(lp["C_lowd"], fetch=False)
        ls = h.run_lloyds(k)
        return r["evals"], g["seeds"], g["C_lowd"], lp["assign"], ls["assign"], ls["centers"]

    h1, h2 = isle_amd.HotPath(0), isle_amd.HotPath(0)
    try:
        big1 = chain(h1, big)
        small_after_big = chain(h1, small)
        small_fresh = chain(h2, small)
        for a, b in zip(small_after_big, small_fresh):
            assert np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))
        big2 = chain(h1, big)
        for a, b in zip(big1, big2):
            assert np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))
    finally:
        h1.close()
        h2.close()
