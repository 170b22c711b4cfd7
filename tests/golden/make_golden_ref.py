"""Generates tests/golden/ref_spectra.npz with the REFERENCE's own eigensolver (oracle/_ref/spectra_eigs = the reference's
vendored Spectra::SymEigsSolver<float, LARGEST_ALGE, Op>(op, k, 2k + 1) + Eigen, called as
FPSparseMatrix::compute_Spectra does, /root/reference/src/sparseMatrix.cpp:1161-1190) on the thresholded synthetic corpora
the tests use.  Run in the build container (needs /root/reference for `make -C oracle`):

    python tests/golden/make_golden_ref.py

Fixture contents per case `<name>`: <name>_params (V, D, k, seed of tools.synth.make_B), <name>_sig (V, D, nnz, sum of rows:
guards against generator drift), <name>_evalues f32[k] (descending), <name>_U f32[V, k] (tiny cases only).
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, "oracle", "_ref", "spectra_eigs")
CASES = {"tiny10": (2000, 5000, 10, 0, True), "tiny20": (2000, 5000, 20, 0, True), "small50": (6000, 20000, 50, 7, False),
         "mid30": (10000, 50000, 30, 11, False)}


def run_reference(B, k):
    """-> (nconv, info, evalues f32[k], U f32[V, k] F-order) from oracle/_ref/spectra_eigs."""
    with tempfile.TemporaryDirectory() as tmp:
        src, dst = os.path.join(tmp, "B.bin"), os.path.join(tmp, "out.bin")
        with open(src, "wb") as f:
            np.array([B["V"], B["D"], B["nnz"]], np.uint64).tofile(f)
            np.ascontiguousarray(B["vals"], np.float32).tofile(f)
            np.ascontiguousarray(B["rows"], np.uint32).tofile(f)
            np.ascontiguousarray(B["offs"], np.int64).tofile(f)
        subprocess.check_call([BIN, src, str(k), dst])
        raw = open(dst, "rb").read()
    nconv, info = np.frombuffer(raw[:8], np.int32)
    ev = np.frombuffer(raw[8:8 + 4 * k], np.float32).copy()
    U = np.frombuffer(raw[8 + 4 * k:], np.float32).reshape(k, B["V"]).T.copy(order="F")
    return int(nconv), int(info), ev, U


def signature(B):
    return np.array([B["V"], B["D"], B["nnz"], int(B["rows"].astype(np.int64).sum())], np.int64)


def main():
    from tools.synth import make_B
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    out = {}
    for name, (V, D, k, seed, keep_U) in CASES.items():
        B = make_B(V, D, k, seed)
        nconv, info, ev, U = run_reference(B, k)
        assert nconv == k and info == 0, (name, nconv, info)  # the asserts of compute_Spectra (:1176-1177)
        out[name + "_params"] = np.array([V, D, k, seed], np.int64)
        out[name + "_sig"] = signature(B)
        out[name + "_evalues"] = ev
        if keep_U:
            out[name + "_U"] = U
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_spectra.npz"), **out)
    print("wrote tests/golden/ref_spectra.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
