import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_CACHE = {}


def corpus(V, D, k, seed, K=None, sample_rate=0.0):
    """Thresholded synthetic corpus B (dict) + oracle handle, cached per session."""
    key = (V, D, k, seed, K, sample_rate)
    if key not in _CACHE:
        from tools.synth import make_B
        from oracle.oracle import OracleCsc
        B = make_B(V, D, k, seed, K=K, sample_rate=sample_rate)
        B["oracle"] = OracleCsc(B["V"], B["D"], B["vals"], B["rows"], B["offs"])
        _CACHE[key] = B
    return _CACHE[key]


@pytest.fixture(scope="session")
def tiny10():
    return corpus(2000, 5000, 10, 0)


@pytest.fixture(scope="session")
def tiny20():
    return corpus(2000, 5000, 20, 0)


@pytest.fixture(scope="session")
def small50():
    return corpus(6000, 20000, 50, 7)


@pytest.fixture(scope="session")
def hp():
    """One HotPath context (GPU 0) for the whole session."""
    from isle_amd import HotPath
    h = HotPath(0)
    yield h
    h.close()


def upload(hp, B):
    hp.upload_csc(B["V"], B["vals"], B["rows"], B["offs"])


def relerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def subspace_cosines(U1, U2):
    """cosines of the principal angles between span(U1) and span(U2) (orthonormal columns)."""
    s = np.linalg.svd(U1.astype(np.float64).T @ U2.astype(np.float64), compute_uv=False)
    return np.clip(s, 0, 1)
