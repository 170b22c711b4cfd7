"""ctypes loader for oracle/libisle_oracle.so.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg — never from isle_amd/ (the product must not route through the oracle).
PARITY UNPINNED (see isle_oracle.cpp header and DESIGN.md §Oracle).
"""
import ctypes as C
import os
import subprocess

import numpy as np


def effective_cpus():
    """min(affinity, cgroup CPU quota): the GPU boxes show 256 logical CPUs behind a 16-CPU quota."""
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

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libisle_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_csc_create.restype = C.c_void_p
        L.orc_csc_create.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_csc_destroy.argtypes = [C.c_void_p]
        L.orc_frobenius.restype = C.c_float
        L.orc_frobenius.argtypes = [C.c_void_p]
        L.orc_num_threads.restype = C.c_int
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleCsc:
    """Host CSC matrix B (V x D): vals f32[nnz], rows u32[nnz] ascending per column, offs i64[D+1]."""

    def __init__(self, V, D, vals, rows, offs):
        self.V, self.D = int(V), int(D)
        self.vals = np.ascontiguousarray(vals, dtype=np.float32)
        self.rows = np.ascontiguousarray(rows, dtype=np.uint32)
        self.offs = np.ascontiguousarray(offs, dtype=np.int64)
        self.nnz = int(self.offs[-1])
        assert self.offs.shape[0] == self.D + 1 and self.vals.shape[0] == self.nnz
        self.h = C.c_void_p(lib().orc_csc_create(self.V, self.D, self.nnz, _p(self.vals), _p(self.rows), _p(self.offs)))

    def __del__(self):
        try:
            lib().orc_csc_destroy(self.h)
        except Exception:
            pass

    def frobenius(self):
        return float(lib().orc_frobenius(self.h))

    def gram_apply(self, X):
        """X: (V, b) array (any layout) -> Z = B (B^T X), (V, b)."""
        X = np.asfortranarray(X, dtype=np.float32)
        b = X.shape[1]
        Z = np.empty_like(X, order="F")
        rc = lib().orc_gram_apply(self.h, _p(X), C.c_int(b), _p(Z))
        assert rc == 0
        return Z

    def block_ks(self, nev, blk=10, ncv=None, maxit=100, tol=1e-4, seed=1):
        ncv = 2 * nev + 10 if ncv is None else ncv
        ev = np.empty(nev, np.float32)
        U = np.empty((self.V, nev), np.float32, order="F")
        nconv, rst, nap = C.c_int(), C.c_int(), C.c_int()
        rc = lib().orc_block_ks(self.h, nev, ncv, maxit, blk, C.c_float(tol), C.c_uint64(seed), _p(ev), _p(U),
                                C.byref(nconv), C.byref(rst), C.byref(nap))
        return dict(rc=rc, evals=ev, U=U, nconv=nconv.value, restarts=rst.value, napplies=nap.value)

    def project(self, U):
        U = np.asfortranarray(U, dtype=np.float32)
        k = U.shape[1]
        P = np.empty((self.D, k), np.float32)
        n2 = np.empty(self.D, np.float32)
        lib().orc_project(self.h, _p(U), k, _p(P), _p(n2))
        return P, n2

    def kmeanspp(self, U, k, inject=None, seed=1, max_rounds=0):
        U = np.asfortranarray(U, dtype=np.float32)
        seeds = np.empty(k, np.uint64)
        Cl = np.empty((k, k), np.float32)
        res, rounds = C.c_float(), C.c_int()
        md = np.empty(self.D, np.float32)
        inj = None if inject is None else np.ascontiguousarray(inject, dtype=np.uint64)
        rc = lib().orc_kmeanspp(self.h, _p(U), k, _p(inj), C.c_uint64(seed), _p(seeds), _p(Cl), C.byref(res),
                                C.byref(rounds), _p(md), C.c_int(max_rounds))
        return dict(rc=rc, seeds=seeds, C_lowd=Cl, residual=res.value, rounds=rounds.value, min_dist=md)

    def lloyds_projected(self, U, C_lowd, max_reps=10):
        U = np.asfortranarray(U, dtype=np.float32)
        k = U.shape[1]
        Cl = np.array(C_lowd, dtype=np.float32, order="C", copy=True)
        it = C.c_int()
        assign = np.empty(self.D, np.uint32)
        lib().orc_lloyds_projected(self.h, _p(U), k, _p(Cl), max_reps, C.byref(it), _p(assign))
        return dict(C_lowd=Cl, iters=it.value, assign=assign)

    def lloyds_sparse(self, centers, max_reps=10):
        """centers: (V, k) Fortran-order (centre c = column c), updated copy returned."""
        Cn = np.array(centers, dtype=np.float32, order="F", copy=True)
        k = Cn.shape[1]
        it = C.c_int()
        assign = np.empty(self.D, np.uint32)
        lib().orc_lloyds_sparse(self.h, k, _p(Cn), _p(assign), max_reps, C.byref(it))
        return dict(centers=Cn, iters=it.value, assign=assign)


def lift(U, C_lowd):
    """centers (V x k, F-order) = U (V x k) * C_lowd^T-as-columns (centre c = row c of C_lowd)."""
    U = np.asfortranarray(U, dtype=np.float32)
    V, k = U.shape
    Cl = np.ascontiguousarray(C_lowd, dtype=np.float32)  # row c = centre c  == col-major k x n with ld = k
    n = Cl.shape[0]
    out = np.empty((V, n), np.float32, order="F")
    lib().orc_lift(_p(U), C.c_uint64(V), k, _p(Cl), k, n, _p(out))
    return out


def block_ks_dense(A, nev, blk=10, ncv=None, maxit=100, tol=1e-4, seed=1):
    A = np.asfortranarray(A, dtype=np.float32)
    n = A.shape[0]
    ncv = 2 * nev + 10 if ncv is None else ncv
    ev = np.empty(nev, np.float32)
    U = np.empty((n, nev), np.float32, order="F")
    nconv, rst, nap = C.c_int(), C.c_int(), C.c_int()
    rc = lib().orc_block_ks_dense(_p(A), C.c_uint64(n), nev, ncv, maxit, blk, C.c_float(tol), C.c_uint64(seed), _p(ev),
                                  _p(U), C.byref(nconv), C.byref(rst), C.byref(nap))
    return dict(rc=rc, evals=ev, U=U, nconv=nconv.value, restarts=rst.value, napplies=nap.value)


def eig_sym(S):
    S = np.asfortranarray(S, dtype=np.float32)
    n = S.shape[0]
    e = np.empty(n, np.float32)
    v = np.empty((n, n), np.float32, order="F")
    rc = lib().orc_eig_sym(_p(S), C.c_uint64(n), _p(e), _p(v))
    assert rc == 0
    return e, v


def qr(A):
    A = np.asfortranarray(A, dtype=np.float32)
    n, c = A.shape
    Q = np.zeros((n, c), np.float32, order="F")
    R = np.zeros(c * c, np.float32)
    rk = C.c_int()
    lib().orc_qr(_p(A), C.c_uint64(n), C.c_uint64(c), _p(Q), _p(R), C.byref(rk))
    r = rk.value
    return Q[:, :r], R[: r * c].reshape((c, r)).T.copy(), r


# ---- stage downstream of the hot path and edge topics (8a a19): isle_post_oracle.cpp -----------
W0_C, EPS2_C, EPS3_C, RHO_C = 1.0, 1.0 / 3.0, 5.0, 1.1       # include/hyperparams.h:8-12
EDGE_TOPIC_MIN_DOCS, EDGE_TOPIC_PRIMARY_RATIO = 1, 0.7       # include/hyperparams.h:77-79


def catchword_rank(num_docs, num_topics, sample_rate=None):
    """r of src/trainer.cpp:579-583 (float operands, double arithmetic)."""
    x = EPS2_C * W0_C * float(np.float32(num_docs))
    if sample_rate is not None:
        x = x * float(np.float32(sample_rate))
    return int(np.floor(x / float(np.float32(2.0 * num_topics))))


def model_rank_threshold(num_docs, num_topics):
    """rank_threshold of src/sparseMatrix.cpp:720."""
    return int(EPS3_C * W0_C * float(np.float32(num_docs)) / (float(np.float32(num_topics)) * 2.0))


def post_normalize(offs, counts, avg_doc_sz):
    offs = np.ascontiguousarray(offs, np.int64)
    counts = np.ascontiguousarray(counts, np.float32)
    nv = np.empty_like(counts)
    L = lib()
    L.orc_post_normalize.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    L.orc_post_normalize(offs.shape[0] - 1, _p(offs), _p(counts), float(avg_doc_sz), _p(nv))
    return nv


def post_catch_thresholds(V, offs, rows, nv, cluster_of, k, r):
    """-> thresholds (V, k) Fortran-ordered (catchword_thresholds of src/trainer.cpp:586-590)."""
    offs = np.ascontiguousarray(offs, np.int64)
    rows = np.ascontiguousarray(rows, np.uint32)
    nv = np.ascontiguousarray(nv, np.float32)
    cl = np.ascontiguousarray(cluster_of, np.int32)
    thr = np.empty((V, k), np.float32, order="F")
    L = lib()
    L.orc_post_catch_thresholds.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_uint64, C.c_void_p]
    L.orc_post_catch_thresholds(V, offs.shape[0] - 1, k, _p(offs), _p(rows), _p(nv), _p(cl), int(r), _p(thr))
    return thr


def post_find_catchwords(thr, rho=RHO_C):
    thr = np.asfortranarray(thr, np.float32)
    V, k = thr.shape
    ct = np.empty(V, np.int32)
    L = lib()
    L.orc_post_find_catchwords.restype = C.c_int64
    L.orc_post_find_catchwords.argtypes = [C.c_uint64, C.c_uint32, C.c_void_p, C.c_double, C.c_void_p]
    n = L.orc_post_find_catchwords(V, k, _p(thr), float(rho), _p(ct))
    assert n >= 0, "a word qualified as catchword of two topics"
    return ct


def post_topic_model(V, offs, rows, nv, cluster_of, catch_topic, k, rank_threshold):
    """-> dict(model (V,k) F-order, dts_doc, dts_topic, dts_val, model_threshold, top1, top2)."""
    offs = np.ascontiguousarray(offs, np.int64)
    rows = np.ascontiguousarray(rows, np.uint32)
    nv = np.ascontiguousarray(nv, np.float32)
    cl = np.ascontiguousarray(cluster_of, np.int32)
    ct = np.ascontiguousarray(catch_topic, np.int32)
    D = offs.shape[0] - 1
    M = np.empty((V, k), np.float32, order="F")
    L = lib()
    L.orc_post_topic_model.restype = C.c_void_p
    L.orc_post_topic_model.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32] + [C.c_void_p] * 5 + [C.c_uint64, C.c_void_p]
    L.orc_post_dts_size.restype = C.c_uint64
    L.orc_post_dts_size.argtypes = [C.c_void_p]
    L.orc_post_dts_get.argtypes = [C.c_void_p] * 7
    L.orc_post_free.argtypes = [C.c_void_p]
    h = C.c_void_p(L.orc_post_topic_model(V, D, k, _p(offs), _p(rows), _p(nv), _p(cl), _p(ct), int(rank_threshold), _p(M)))
    n = int(L.orc_post_dts_size(h))
    out = dict(model=M, dts_doc=np.empty(n, np.uint64), dts_topic=np.empty(n, np.uint32), dts_val=np.empty(n, np.float32),
               model_threshold=np.empty(k, np.float32), top1=np.empty(D, np.int32), top2=np.empty(D, np.int32))
    L.orc_post_dts_get(h, _p(out["dts_doc"]), _p(out["dts_topic"]), _p(out["dts_val"]), _p(out["model_threshold"]),
                       _p(out["top1"]), _p(out["top2"]))
    L.orc_post_free(h)
    return out


def post_edge_topics(model, top1, top2, max_edge_topics, want_edge=True):
    model = np.asfortranarray(model, np.float32)
    V, k = model.shape
    t1 = np.ascontiguousarray(top1, np.int32)
    t2 = np.ascontiguousarray(top2, np.int32)
    cap = int(max_edge_topics)
    pairs = np.zeros((cap, 3), np.int64)
    L = lib()
    L.orc_post_edge_topics.restype = C.c_uint64
    L.orc_post_edge_topics.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_float,
                                       C.c_void_p, C.c_void_p]
    # the count is not known before the call: size Edge for the worst case the caller allows
    ncap = min(cap, k * k)
    edge = np.zeros((V, ncap), np.float32, order="F") if want_edge else None
    pairs = np.zeros((ncap, 3), np.int64)
    n = int(L.orc_post_edge_topics(V, t1.shape[0], _p(t1), _p(t2), ncap, EDGE_TOPIC_MIN_DOCS, _p(model),
                                   EDGE_TOPIC_PRIMARY_RATIO, _p(pairs), _p(edge)))
    return pairs[:n], (edge[:, :n] if want_edge else None)


# ---- inference: isle_infer_oracle.cpp ----------------------------------------------------------
INFER_ITERS_DEFAULT = 15   # include/hyperparams.h:81
INFER_LF_DEFAULT = 10.0    # include/hyperparams.h:82


def infer(model_by_word, offs, rows, counts, iters=INFER_ITERS_DEFAULT, Lf=INFER_LF_DEFAULT, avg_doc_sz=None):
    """ISLEInfer over a count matrix (CSC) — drivers/ISLEInfer.cpp:60-100, src/infer.cpp:361-492.
    model_by_word: V x k row-major.  Returns weights (D x k), llh (D x 2), number of converged documents."""
    M = np.ascontiguousarray(model_by_word, np.float32)
    V, k = M.shape
    offs = np.ascontiguousarray(offs, np.int64)
    rows = np.ascontiguousarray(rows, np.uint32)
    counts = np.ascontiguousarray(counts, np.float32)
    D = offs.shape[0] - 1
    if avg_doc_sz is None:  # populate_CSC, src/sparseMatrix.cpp:87-98: integer division of the token total by the non-empty documents
        nz = int((np.diff(offs) > 0).sum())
        avg_doc_sz = float(int(counts.astype(np.float64).sum()) // max(nz, 1))
    W = np.empty((D, k), np.float32)
    llh = np.empty((D, 2), np.float32)
    L = lib()
    L.orc_infer.restype = C.c_uint64
    L.orc_infer.argtypes = [C.c_uint64, C.c_int] + [C.c_void_p] + [C.c_uint64] + [C.c_void_p] * 3 + [C.c_int, C.c_float, C.c_float,
                                                                                                   C.c_void_p, C.c_void_p]
    n = int(L.orc_infer(V, k, _p(M), D, _p(offs), _p(rows), _p(counts), int(iters), float(Lf), float(avg_doc_sz), _p(W), _p(llh)))
    return dict(weights=W, llh=llh, nconverged=n, avg_doc_sz=avg_doc_sz)
