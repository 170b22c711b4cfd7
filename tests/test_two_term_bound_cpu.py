"""The error bound the two-term assignment products rely on (isle_amd/csrc/dense.hip GA_ETA, gemm_bf16x3.h Cfg::NP = 2), checked in NumPy.
bf16 keeps 8 significand bits: |x - bf16(x)| <= 2^-8 |x|.  With x0 = bf16(x), x1 = bf16(x - x0): |x1| <= 2^-8 (1 + 2^-8) |x| and the remainder
|x - x0 - x1| <= 2^-16 |x|, so the three partial products a0 b0 + a0 b1 + a1 b0 are within 3 * 2^-16 (1 + 2^-7) |a b| of a b, a dot product within
4.6e-5 sum |a_k b_k| <= 4.6e-5 |a| |b|, and a squared distance |a|^2 + |b|^2 - 2 a.b within 4.6e-5 (|a|^2 + |b|^2) <= GA_ETA_TRUNC = 4.65e-5; the
library adds the worst-case f32 accumulation of both routes on top (ga_eta(K), 8.2e-5 at K = 1000).  No GPU, no library.  (This test caught the
first version of the constant, 2.5e-5, derived with one significand bit too many.)"""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bf16(x):
    """round-to-nearest-even to bfloat16, returned as float32 (what the (__bf16) cast of gemm_bf16x3.h does)"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32)


def split2(x):
    x = np.asarray(x, np.float32)
    x0 = bf16(x)
    x1 = bf16(x - x0)  # the subtraction is exact in f32
    return x0, x1


def ga_eta_trunc():
    src = open(os.path.join(ROOT, "isle_amd", "csrc", "dense.hip")).read()
    return float(re.search(r"constexpr float GA_ETA_TRUNC = ([0-9.e+-]+)f;", src).group(1))


def test_remainder_of_two_terms_is_at_most_2_to_minus_16():
    rng = np.random.default_rng(0)
    edge = np.float32(1.0) + np.arange(1, 4097, dtype=np.float32) * np.float32(2.0 ** -12)  # every 12-bit significand pattern behind the leading bit
    x = np.concatenate([rng.standard_normal(200_000), rng.standard_normal(50_000) * 1e-6, rng.standard_normal(50_000) * 1e6, edge,
                        edge + np.float32(2.0 ** -20), -edge * np.float32(3.0)]).astype(np.float32)
    X = x.astype(np.float64)
    x0, x1 = split2(x)
    assert np.all(np.abs(X - x0) <= 2.0 ** -8 * np.abs(X))
    assert np.all(np.abs(x1.astype(np.float64)) <= 2.0 ** -8 * (1 + 2.0 ** -8) * np.abs(X))
    assert np.all(np.abs(X - x0.astype(np.float64) - x1.astype(np.float64)) <= 2.0 ** -16 * np.abs(X))


def test_three_partial_products_are_within_the_bound_the_epilogues_widen_by():
    rng = np.random.default_rng(1)
    K = 1000
    worst_dot, worst_dist = 0.0, 0.0
    for scale_a, scale_b in ((1.0, 1.0), (30.0, 0.01), (1e-3, 1e3)):
        a = (rng.standard_normal((256, K)) * scale_a).astype(np.float32)
        b = (rng.standard_normal((K, 64)) * scale_b).astype(np.float32)
        # adversarial rows / columns: every entry just below the midpoint of two bf16 neighbours, all of one sign (errors add up)
        a[0] = np.float32(scale_a) * (np.float32(1.0) + np.float32(2.0 ** -8) - np.float32(2.0 ** -20))
        b[:, 0] = np.float32(scale_b) * (np.float32(1.0) + np.float32(2.0 ** -8) - np.float32(2.0 ** -20))
        a0, a1 = (t.astype(np.float64) for t in split2(a))
        b0, b1 = (t.astype(np.float64) for t in split2(b))
        A, B = a.astype(np.float64), b.astype(np.float64)
        exact = A @ B
        two = a0 @ b0 + a0 @ b1 + a1 @ b0
        mag = np.abs(A) @ np.abs(B)
        worst_dot = max(worst_dot, float(np.max(np.abs(two - exact) / mag)))
        na, nb = (A * A).sum(1)[:, None], (B * B).sum(0)[None, :]
        worst_dist = max(worst_dist, float(np.max(2 * np.abs(two - exact) / (na + nb))))
    assert worst_dot <= 3 * 2.0 ** -16 * (1 + 2.0 ** -7)  # 4.6e-5
    assert worst_dist <= ga_eta_trunc() <= 4.7e-5        # the library's truncation constant covers it (the accumulation term comes on top)
