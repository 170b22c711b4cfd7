"""Synthetic planted-topic Zipf corpus + ISLE thresholding pre-stage.

Bench/test input generator — not product code, not the oracle.  Wraps tools/libisle_synth.so.
"""
import ctypes as C
import os
import subprocess


def effective_cpus():
    """CPUs this process may really use: min(affinity, cgroup quota).  The GPU boxes expose 256
    logical CPUs behind a 16-CPU cgroup quota; OpenMP's default of 256 threads is catastrophic there."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, n)


os.environ.setdefault("OMP_NUM_THREADS", str(effective_cpus()))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libisle_synth.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.synth_generate.restype = C.c_void_p
        L.synth_generate.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_uint64, C.c_uint64]
        L.synth_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.synth_hist.restype = C.c_void_p
        L.synth_hist.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        L.synth_apply.restype = C.c_uint64
        L.synth_apply.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_double, C.c_uint64]
        L.synth_from_csc.restype = C.c_void_p
        L.synth_from_csc.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.synth_threshold.restype = C.c_uint64
        L.synth_threshold.argtypes = [C.c_void_p, C.c_uint32]
        for f in ("synth_nnz_A", "synth_docs_B"):
            getattr(L, f).restype = C.c_uint64
            getattr(L, f).argtypes = [C.c_void_p]
        for f in ("synth_A_counts", "synth_A_rows", "synth_A_offs", "synth_dom", "synth_B_vals", "synth_B_rows",
                  "synth_B_offs", "synth_B_original_cols", "synth_zetas"):
            getattr(L, f).restype = C.c_void_p
            getattr(L, f).argtypes = [C.c_void_p]
        L.synth_destroy.argtypes = [C.c_void_p]
        L.synth_free_A.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


class Corpus:
    """A = counts (V x D CSC) and, after threshold(k), B (V x D_B CSC) as ISLE's trainer would build it."""

    def __init__(self, V, D, K, seed, zipf_s=1.05, L0=130.0, dom_w=0.8, doc_base=0, _handle=None):
        self.V, self.D, self.K = int(V), int(D), int(K)
        self._h = _handle if _handle is not None else C.c_void_p(
            _lib().synth_generate(self.V, self.D, self.K, zipf_s, L0, dom_w, seed, doc_base))
        self.nnz_A = int(_lib().synth_nnz_A(self._h))

    @classmethod
    def from_csc(cls, V, D, counts, rows, offs):
        counts = np.ascontiguousarray(counts, np.float32)
        rows = np.ascontiguousarray(rows, np.uint32)
        offs = np.ascontiguousarray(offs, np.int64)
        h = C.c_void_p(_lib().synth_from_csc(V, D, counts.ctypes.data, rows.ctypes.data, offs.ctypes.data))
        return cls(V, D, 1, 0, _handle=h)

    def __del__(self):
        try:
            _lib().synth_destroy(self._h)
        except Exception:
            pass

    def A(self):
        L = _lib()
        return (_arr(L.synth_A_counts(self._h), self.nnz_A, np.float32), _arr(L.synth_A_rows(self._h), self.nnz_A, np.uint32),
                _arr(L.synth_A_offs(self._h), self.D + 1, np.int64))

    def A_views(self):
        """The same three arrays WITHOUT copies: views onto the generator's buffers, valid while this object lives and A has not been freed
        (threshold(free_A=True)).  For handing a 10 M-document A to the device without a second 9 GB host copy."""
        L = _lib()

        def view(ptr, n, dtype):
            if n == 0:
                return np.zeros(0, dtype)
            return np.frombuffer((C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr), dtype=dtype, count=n)
        return (view(L.synth_A_counts(self._h), self.nnz_A, np.float32), view(L.synth_A_rows(self._h), self.nnz_A, np.uint32),
                view(L.synth_A_offs(self._h), self.D + 1, np.int64))

    def tdf_bytes(self):
        """The corpus as tdf text ("<doc> <word> <count>\\n", 1-based ids) in a uint8 array."""
        L = _lib()
        L.synth_tdf_bytes.restype = C.c_uint64
        L.synth_tdf_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        buf = np.empty(self.nnz_A * 26 + 16, np.uint8)
        n = int(L.synth_tdf_bytes(self._h, buf.ctypes.data, buf.size))
        assert n > 0
        return buf[:n]

    def planted(self):
        return _arr(_lib().synth_dom(self._h), self.D, np.uint32)

    def threshold(self, k, free_A=False, allreduce=None, sample_rate=0.0, sample_seed=0):
        """Returns dict(V, D, nnz, vals, rows, offs, original_cols, zetas) for B.

        allreduce: optional callable(np.ndarray int64/uint32) -> summed-in-place over all shards; lets a
        column-sharded corpus use GLOBAL avg_doc_sz / nz_docs / per-word histograms (the thresholds depend
        on the whole corpus: src/sparseMatrix.cpp:357-485)."""
        L = _lib()
        if allreduce is None and not sample_rate:
            nnz = int(L.synth_threshold(self._h, k))
        elif allreduce is None:
            st = np.zeros(2, np.uint64)
            L.synth_stats(self._h, st.ctypes.data, st.ctypes.data + 8)
            mv = C.c_uint32()
            L.synth_hist(self._h, int(st[0]), int(st[1]), C.byref(mv))
            nnz = int(L.synth_apply(self._h, k, int(st[1]), float(sample_rate), int(sample_seed)))
        else:
            st = np.zeros(2, np.uint64)
            L.synth_stats(self._h, st.ctypes.data, st.ctypes.data + 8)
            st64 = st.astype(np.int64)
            allreduce(st64)
            mv = C.c_uint32()
            hp_ = L.synth_hist(self._h, int(st64[0]), int(st64[1]), C.byref(mv))
            n = self.V * (mv.value + 1)
            buf = (C.c_char * (n * 4)).from_address(hp_)
            hist = np.frombuffer(buf, dtype=np.int32, count=n)  # view onto the C++ buffer, reduced in place
            allreduce(hist)
            nnz = int(L.synth_apply(self._h, k, int(st64[1]), 0.0, 0))
        Db = int(L.synth_docs_B(self._h))
        out = dict(V=self.V, D=Db, nnz=nnz,
                   vals=_arr(L.synth_B_vals(self._h), nnz, np.float32),
                   rows=_arr(L.synth_B_rows(self._h), nnz, np.uint32),
                   offs=_arr(L.synth_B_offs(self._h), Db + 1, np.int64),
                   original_cols=_arr(L.synth_B_original_cols(self._h), Db, np.uint64),
                   zetas=_arr(L.synth_zetas(self._h), self.V, np.float32))
        if free_A:
            L.synth_free_A(self._h)
        return out


def make_B(V, D, k, seed, K=None, sample_rate=0.0, **kw):
    """Convenience: planted-topic corpus with K (=k by default) topics, thresholded for k topics
    (sample_rate > 0: ISLE's importance sampling of documents, BASELINE config 4)."""
    c = Corpus(V, D, k if K is None else K, seed, **kw)
    B = c.threshold(k, sample_rate=sample_rate, sample_seed=seed)
    B["planted"] = c.planted()[B["original_cols"].astype(np.int64)]
    B["nnz_A"] = c.nnz_A
    return B
