"""Python binding of the C ABI (one object = one isle_ctx = one GPU).

Method names follow the reference methods they replace (ISLE::FPSparseMatrix<float>,
/root/reference include/sparseMatrix.h:242-466); see include/isle_hip.h for the contract.
numpy arrays in, numpy arrays out; all device work happens inside libisle_hip.so.
"""
import ctypes as C

import numpy as np

from ._lib import IsleHipError, load_library

TIMING_FAMILIES = ["gram_pass1", "gram_pass2", "ortho", "qr", "evd", "rotate", "project", "kmpp", "lloyd_proj",
                   "sparse_assign", "sparse_update", "op_build", "comm", "threshold", "post", "ingest", "infer", "lift"]

BLOCK_KS_MAX_ITERS = 100      # include/hyperparams.h:38
BLOCK_KS_BLOCK_SIZE = 10      # include/hyperparams.h:39
BLOCK_KS_TOLERANCE = 1e-4     # include/hyperparams.h:40
MAX_KMEANS_LOWD_REPS = 10     # include/hyperparams.h:60
MAX_KMEANS_REPS = 10          # include/hyperparams.h:68


W0_C, EPS2_C, EPS3_C, RHO_C = 1.0, 1.0 / 3.0, 5.0, 1.1   # include/hyperparams.h:8-12


def catchword_rank(num_docs, num_topics, sample_rate=None):
    """The rank r that ISLETrainer::train passes to rth_highest_element (src/trainer.cpp:579-583): eps2_c * w0_c * num_docs
    (* sample_rate) / (2 num_topics), float operands in double arithmetic, floored."""
    x = EPS2_C * W0_C * float(np.float32(num_docs))
    if sample_rate is not None:
        x = x * float(np.float32(sample_rate))
    return int(np.floor(x / float(np.float32(2.0 * num_topics))))


def model_rank_threshold(num_docs, num_topics):
    """rank_threshold of SparseMatrix::construct_topic_model (src/sparseMatrix.cpp:720)."""
    return int(EPS3_C * W0_C * float(np.float32(num_docs)) / (float(np.float32(num_topics)) * 2.0))


EDGE_TOPIC_MIN_DOCS = 1            # include/hyperparams.h:77
EDGE_TOPIC_PRIMARY_RATIO = 0.7     # include/hyperparams.h:79


def select_edge_pairs(top1, top2, max_edge_topics, min_docs=EDGE_TOPIC_MIN_DOCS):
    """Pair selection of ISLETrainer::construct_edge_topics_v2 (src/trainer.cpp:1116-1145), the host half of the edge-topic stage as
    isle_amd/host/fpsparse_hip.h runs it: the documents' (top topic, second topic) pairs are counted, pairs with >= min_docs documents
    are candidates, the max_edge_topics most frequent are kept (ties in the count by (primary, secondary) ascending — the reference's
    sort is unstable there).  top1 / top2: int32 per document of A, -1 where the document has no such topic (construct_topic_model,
    src/sparseMatrix.cpp:687-708).  -> int64 (n, 3): primary, secondary, documents."""
    t1 = np.asarray(top1, np.int64)
    t2 = np.asarray(top2, np.int64)
    ok = (t1 >= 0) & (t2 >= 0)
    if not ok.any():
        return np.zeros((0, 3), np.int64)
    base = int(max(t1.max(), t2.max())) + 1
    key, cnt = np.unique(t1[ok] * base + t2[ok], return_counts=True)  # ascending (primary, secondary)
    keep = cnt >= min_docs
    key, cnt = key[keep], cnt[keep]
    order = np.argsort(-cnt, kind="stable")[:max(int(max_edge_topics), 0)]
    return np.stack([key[order] // base, key[order] % base, cnt[order]], axis=1).astype(np.int64)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class HotPath:
    def __init__(self, device=0):
        self._lib = load_library()
        self._h = self._lib.isle_hip_create(device)
        if not self._h:
            raise IsleHipError("isle_hip_create(%d) failed: no usable HIP device (no CPU fallback exists)" % device)
        self.V = self.D = self.nnz = 0
        self.doc_offset = 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.isle_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, allow=()):
        if rc != 0 and rc not in allow:
            raise IsleHipError("isle_hip error %d: %s" % (rc, self._lib.isle_hip_last_error(self._h).decode()))
        return rc

    # ---- multi-GPU ------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        buf = np.zeros(128, np.uint8)
        rc = load_library().isle_hip_comm_unique_id(_p(buf))
        if rc != 0:
            raise IsleHipError("ncclGetUniqueId failed")
        return buf

    def comm_init(self, world, rank, uid):
        uid = np.ascontiguousarray(uid, np.uint8)
        self._chk(self._lib.isle_hip_comm_init(self._h, world, rank, _p(uid)))

    def comm_init_host(self, world, rank, exchange):
        """Rehearsal transport (tests only; include/isle_hip.h): every collective is staged through host memory and
        handed to exchange(kind, array, count) -> None, which must complete it in place.  kind: 0 all-reduce sum,
        1 all-reduce max, 2 all-gather (array has world * count elements, this rank's part filled in)."""
        dts = [np.float32, np.float64, np.int32, np.uint32, np.uint64]

        def tramp(user, kind, buf, count, dtype):
            try:
                n = count * (world if kind == 2 else 1)
                dt = np.dtype(dts[dtype])
                a = np.frombuffer((C.c_char * (n * dt.itemsize)).from_address(buf), dtype=dt)
                exchange(kind, a, count)
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1

        self._xchg_cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_int)(tramp)
        self._chk(self._lib.isle_hip_comm_init_host(self._h, world, rank, C.cast(self._xchg_cb, C.c_void_p), None))

    @staticmethod
    def gloo_exchange(dist, world, rank):
        """exchange function for comm_init_host over an initialised torch.distributed (gloo) process group."""
        import torch

        def exchange(kind, a, count):
            if kind == 2:
                # all-gather: a.view(world, count), own row filled; gloo has no unsigned types -> move the bytes
                t = torch.from_numpy(a.view(np.uint8).reshape(world, -1))
                parts = [torch.empty_like(t[0]) for _ in range(world)]
                dist.all_gather(parts, t[rank].clone())
                for r in range(world):
                    t[r].copy_(parts[r])
                return
            signed = {np.dtype(np.uint32): np.int32, np.dtype(np.uint64): np.int64}.get(a.dtype)
            t = torch.from_numpy(a.view(signed) if signed else a)
            dist.all_reduce(t, op=dist.ReduceOp.MAX if kind == 1 else dist.ReduceOp.SUM)
        return exchange

    @staticmethod
    def plan_shards(offs, parts):
        offs = np.ascontiguousarray(offs, np.int64)
        bounds = np.zeros(parts + 1, np.uint64)
        rc = load_library().isle_hip_plan_shards(offs.shape[0] - 1, _p(offs), parts, _p(bounds))
        if rc != 0:
            raise IsleHipError("plan_shards failed")
        return bounds

    # ---- input ----------------------------------------------------------------------------
    def upload_csc(self, V, vals, rows, offs, doc_offset=0, docs_global=0):
        vals = np.ascontiguousarray(vals, np.float32)
        offs = np.ascontiguousarray(offs, np.int64)
        D = offs.shape[0] - 1
        nnz = int(offs[-1])
        if rows.dtype == np.uint64:
            rows = np.ascontiguousarray(rows)
            fn = self._lib.isle_hip_upload_csc_u64
        else:
            rows = np.ascontiguousarray(rows, np.uint32)
            fn = self._lib.isle_hip_upload_csc_u32
        self._chk(fn(self._h, V, D, nnz, _p(vals), _p(rows), _p(offs), doc_offset, docs_global))
        self.V, self.D, self.nnz, self.doc_offset = int(V), D, nnz, int(doc_offset)
        self.D_global = int(docs_global) if docs_global else D

    # ---- upstream stage: thresholding on the device ---------------------------------------
    def upload_counts(self, V, counts, rows, offs, doc_offset=0, docs_global=0):
        """A = word-document counts in CSC (SparseMatrix::populate_CSC, src/sparseMatrix.cpp:58-133)."""
        counts = np.ascontiguousarray(counts, np.float32)
        rows = np.ascontiguousarray(rows, np.uint32)
        offs = np.ascontiguousarray(offs, np.int64)
        D = offs.shape[0] - 1
        self._chk(self._lib.isle_hip_upload_counts_u32(self._h, V, D, int(offs[-1]), _p(counts), _p(rows), _p(offs),
                                                       doc_offset, docs_global))
        self._a_shape = (int(V), D, int(offs[-1]))

    def ingest_tdf(self, text, vocab_size, num_docs, max_entries=0):
        """tdf text (bytes) -> the context's count matrix A, on the device (include/utils.h:158-228,
        src/trainer.cpp:236-247, src/sparseMatrix.cpp:58-87).  -> dict(entries_read, nnz)."""
        buf = np.frombuffer(text, dtype=np.uint8) if not isinstance(text, np.ndarray) else text
        nr, nz = C.c_uint64(), C.c_uint64()
        self._chk(self._lib.isle_hip_ingest_tdf(self._h, _p(buf) if buf.size else None, int(buf.size), int(vocab_size), int(num_docs),
                                                int(max_entries), C.byref(nr), C.byref(nz)))
        self._a_shape = (int(vocab_size), int(num_docs), int(nz.value))
        return dict(entries_read=int(nr.value), nnz=int(nz.value))

    def get_A(self):
        V, D, nnz = self._a_shape
        cnt, rows, offs = np.empty(nnz, np.float32), np.empty(nnz, np.uint32), np.empty(D + 1, np.int64)
        self._chk(self._lib.isle_hip_get_A(self._h, _p(cnt), _p(rows), _p(offs)))
        return cnt, rows, offs

    def threshold(self, num_topics, sample_rate=0.0, sample_seed=0):
        """normalize_docs + compute_thresholds + (sampled_)threshold_and_copy on the device
        (src/trainer.cpp:430-485); B becomes this context's matrix.  Returns a dict of scalars."""
        dk, nk, ab = C.c_uint64(), C.c_uint64(), C.c_uint64()
        avg = C.c_float()
        self._chk(self._lib.isle_hip_threshold(self._h, int(num_topics), float(sample_rate), int(sample_seed),
                                               C.byref(dk), C.byref(nk), C.byref(ab), C.byref(avg)))
        V, D, nnz, off, glob = (C.c_uint64() for _ in range(5))
        self._chk(self._lib.isle_hip_shape(self._h, C.byref(V), C.byref(D), C.byref(nnz), C.byref(off), C.byref(glob)))
        self.V, self.D, self.nnz = int(V.value), int(D.value), int(nnz.value)
        self.doc_offset, self.D_global = int(off.value), int(glob.value)
        return dict(docs_kept=int(dk.value), nnz_kept=int(nk.value), entries_above_threshold=int(ab.value),
                    avg_doc_sz=float(avg.value))

    def get_B(self, with_threshold_outputs=True):
        """Host copy of the context's B: dict(V, D, nnz, vals, rows, offs[, original_cols, zetas])."""
        out = dict(V=self.V, D=self.D, nnz=self.nnz, vals=np.empty(self.nnz, np.float32), rows=np.empty(self.nnz, np.uint32),
                   offs=np.empty(self.D + 1, np.int64))
        oc = ze = None
        if with_threshold_outputs:
            oc = out["original_cols"] = np.empty(self.D, np.uint64)
            ze = out["zetas"] = np.empty(self.V, np.float32)
        self._chk(self._lib.isle_hip_get_B(self._h, _p(out["vals"]), _p(out["rows"]), _p(out["offs"]),
                                           _p(oc) if oc is not None else None, _p(ze) if ze is not None else None))
        return out

    # ---- downstream stage: catchwords, topic model, edge topics ---------------------------
    def find_catchwords(self, num_topics, r, assign=None, rho=1.1, fetch_thresholds=True):
        """rth_highest_element per topic + find_catchwords (src/trainer.cpp:586-627).
        -> dict(thresholds (V,k) F-order or None, catch_topic int32[V], num_catchwords)."""
        V = self.V
        thr = np.empty((V, num_topics), np.float32, order="F") if fetch_thresholds else None
        ct = np.empty(V, np.int32)
        n = C.c_uint64()
        a = None if assign is None else np.ascontiguousarray(assign, np.uint32)
        self._chk(self._lib.isle_hip_catchwords(self._h, int(num_topics), _p(a), int(r), float(rho), _p(thr), _p(ct), C.byref(n)))
        return dict(thresholds=thr, catch_topic=ct, num_catchwords=int(n.value))

    def construct_topic_model(self, num_topics, rank_threshold, num_docs_A, fetch_sums=True):
        """SparseMatrix::construct_topic_model (src/sparseMatrix.cpp:597-838) on the device."""
        V = self.V
        M = np.empty((V, num_topics), np.float32, order="F")
        mt = np.empty(num_topics, np.float32)
        t1 = np.empty(num_docs_A, np.int32)
        t2 = np.empty(num_docs_A, np.int32)
        n = C.c_uint64()
        self._chk(self._lib.isle_hip_topic_model(self._h, int(num_topics), int(rank_threshold), _p(M), _p(mt), _p(t1), _p(t2), C.byref(n)))
        out = dict(model=M, model_threshold=mt, top1=t1, top2=t2, num_sums=int(n.value))
        if fetch_sums:
            off = np.empty(num_docs_A + 1, np.int64)
            tp = np.empty(out["num_sums"], np.uint32)
            va = np.empty(out["num_sums"], np.float32)
            self._chk(self._lib.isle_hip_get_doc_topic_sums(self._h, _p(off), _p(tp), _p(va)))
            out.update(dts_off=off, dts_topic=tp, dts_val=va)
        return out

    def edge_topics(self, pairs, primary_ratio=0.7):
        """FPaxpy pair of construct_edge_topics_v2 (src/trainer.cpp:1152-1159): pairs (n,2) -> (V,n) F-order."""
        pairs = np.ascontiguousarray(pairs, np.int64).reshape(-1, 2)
        n = pairs.shape[0]
        E = np.empty((self.V, n), np.float32, order="F")
        self._chk(self._lib.isle_hip_edge_topics(self._h, _p(pairs), n, float(primary_ratio), _p(E)))
        return E

    def frobenius(self):
        out = C.c_float()
        self._chk(self._lib.isle_hip_frobenius(self._h, C.byref(out)))
        return out.value

    # ---- eigensolver ----------------------------------------------------------------------
    def gram_apply(self, X):
        X = np.asfortranarray(X, dtype=np.float32)
        assert X.shape[0] == self.V
        Z = np.empty_like(X, order="F")
        self._chk(self._lib.isle_hip_gram_apply(self._h, _p(X), X.shape[1], _p(Z)))
        return Z

    def operator_form(self):
        """1 = LDS-banded Gram apply (row-constant B), 0 = gather form, -1 = operator not built yet."""
        f = C.c_int()
        self._chk(self._lib.isle_hip_operator_form(self._h, C.byref(f)))
        return f.value

    def compute_block_ks(self, num_topics, blk=BLOCK_KS_BLOCK_SIZE, ncv=None, maxit=BLOCK_KS_MAX_ITERS,
                         tol=BLOCK_KS_TOLERANCE, seed=1, allow_noconv=False):
        """FPSparseMatrix::compute_block_ks(num_topics, evalues) — src/sparseMatrix.cpp:1195-1220."""
        ncv = 2 * num_topics + BLOCK_KS_BLOCK_SIZE if ncv is None else ncv
        ev = np.empty(num_topics, np.float32)
        nconv, rst, nap = C.c_int(), C.c_int(), C.c_int()
        rc = self._lib.isle_hip_block_ks(self._h, num_topics, ncv, maxit, blk, tol, seed, _p(ev), C.byref(nconv), C.byref(rst),
                                         C.byref(nap))
        self._chk(rc, allow=(-3,) if allow_noconv else ())
        return dict(rc=rc, evals=ev, nconv=nconv.value, restarts=rst.value, napplies=nap.value)

    def block_ks_dense(self, A, nev, blk=BLOCK_KS_BLOCK_SIZE, ncv=None, maxit=BLOCK_KS_MAX_ITERS, tol=BLOCK_KS_TOLERANCE, seed=1,
                       start_block=None, allow_noconv=False):
        """BlockKs<utils::ArmaMatProdOp> (block-ks/ks_utils.h:167-182): the solver of compute_block_ks on a dense symmetric A."""
        A = np.asfortranarray(A, dtype=np.float32)
        n = A.shape[0]
        assert A.shape == (n, n)
        ncv = 2 * nev + BLOCK_KS_BLOCK_SIZE if ncv is None else ncv
        ev = np.empty(nev, np.float32)
        U = np.empty((n, nev), np.float32, order="F")
        sb = None if start_block is None else np.asfortranarray(start_block, dtype=np.float32)
        nconv, nref, rst, nap = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        rc = self._lib.isle_hip_block_ks_dense(self._h, _p(A), n, nev, ncv, maxit, blk, tol, seed, _p(sb), _p(ev), _p(U), C.byref(nconv),
                                               C.byref(nref), C.byref(rst), C.byref(nap))
        self._chk(rc, allow=(-3,) if allow_noconv else ())
        return dict(rc=rc, evals=ev, U=U, nconv=nconv.value, nconv_ref_rule=nref.value, restarts=rst.value, napplies=nap.value)

    def get_U(self, k):
        U = np.empty((self.V, k), np.float32, order="F")
        self._chk(self._lib.isle_hip_get_U(self._h, _p(U)))
        return U

    def set_U(self, U):
        U = np.asfortranarray(U, dtype=np.float32)
        self._chk(self._lib.isle_hip_set_U(self._h, _p(U), U.shape[1]))

    def eig_sym(self, S):
        S = np.asfortranarray(S, dtype=np.float32)
        n = S.shape[0]
        e = np.empty(n, np.float32)
        v = np.empty((n, n), np.float32, order="F")
        self._chk(self._lib.isle_hip_eig_sym(self._h, _p(S), n, _p(e), _p(v)))
        return e, v

    # ---- k-means --------------------------------------------------------------------------
    def kmeans_init_on_projected_space(self, k, inject_seeds=None, rng_seed=1):
        seeds = np.empty(k, np.uint64)
        Cl = np.empty((k, k), np.float32)
        res, rounds = C.c_float(), C.c_int()
        inj = None if inject_seeds is None else np.ascontiguousarray(inject_seeds, np.uint64)
        self._chk(self._lib.isle_hip_kmeanspp_projected(self._h, k, _p(inj), rng_seed, _p(seeds), _p(Cl), C.byref(res),
                                                       C.byref(rounds)))
        return dict(seeds=seeds, C_lowd=Cl, residual=res.value, rounds=rounds.value)

    def get_min_dist(self):
        md = np.empty(self.D, np.float32)
        self._chk(self._lib.isle_hip_get_min_dist(self._h, _p(md)))
        return md

    def run_lloyds_on_projected_space(self, k, C_lowd, max_reps=MAX_KMEANS_LOWD_REPS):
        Cl = np.array(C_lowd, dtype=np.float32, order="C", copy=True)
        it = C.c_int()
        assign = np.empty(self.D, np.uint32)
        self._chk(self._lib.isle_hip_lloyds_projected(self._h, k, _p(Cl), max_reps, C.byref(it), _p(assign)))
        return dict(C_lowd=Cl, iters=it.value, assign=assign)

    def left_multiply_by_U(self, C_lowd, fetch=True):
        """centers (V x n, F-order) = U * C_lowd^T, centre c = row c of C_lowd (ld_in = k)."""
        Cl = np.ascontiguousarray(C_lowd, dtype=np.float32)
        n, k = Cl.shape
        out = np.empty((self.V, n), np.float32, order="F") if fetch else None
        self._chk(self._lib.isle_hip_lift_centers(self._h, _p(Cl), k, n, _p(out)))
        return out

    def run_lloyds(self, k, centers=None, max_reps=MAX_KMEANS_REPS, fetch_centers=True):
        cin = None if centers is None else np.asfortranarray(centers, dtype=np.float32)
        cout = np.empty((self.V, k), np.float32, order="F") if fetch_centers else None
        assign = np.empty(self.D, np.uint32)
        it = C.c_int()
        self._chk(self._lib.isle_hip_lloyds_sparse(self._h, k, _p(cin), _p(cout), _p(assign), max_reps, C.byref(it)))
        return dict(centers=cout, assign=assign, iters=it.value)

    # ---- measurement ------------------------------------------------------------------------
    # ---- inference -------------------------------------------------------
    def infer(self, model_by_word, offs, rows, counts, iters=15, Lf=10.0, avg_doc_sz=None, want_weights=True):
        """ISLEInfer over a count matrix in CSC (drivers/ISLEInfer.cpp:60-112, src/infer.cpp:361-492).
        model_by_word: V x k row-major.  Returns weights (D x k), top_topic / top_weight (D x 5), llh (D x 2), nconverged."""
        M = np.ascontiguousarray(model_by_word, np.float32)
        V, k = M.shape
        offs = np.ascontiguousarray(offs, np.int64)
        rows = np.ascontiguousarray(rows, np.uint32)
        counts = np.ascontiguousarray(counts, np.float32)
        D = offs.shape[0] - 1
        if avg_doc_sz is None:  # populate_CSC, src/sparseMatrix.cpp:87-98
            nz = int((np.diff(offs) > 0).sum())
            avg_doc_sz = float(int(counts.astype(np.float64).sum()) // max(nz, 1))
        W = np.empty((D, k), np.float32) if want_weights else None
        tt = np.empty((D, 5), np.int32)
        tw = np.empty((D, 5), np.float32)
        llh = np.empty((D, 2), np.float32)
        nc = C.c_uint64()
        self._chk(self._lib.isle_hip_infer(self._h, V, k, _p(M), D, rows.shape[0], _p(counts), _p(rows), _p(offs), int(iters), float(Lf),
                                           float(avg_doc_sz), _p(W) if want_weights else None, _p(tt), _p(tw), _p(llh), C.byref(nc)))
        return dict(weights=W, top_topic=tt, top_weight=tw, llh=llh, nconverged=int(nc.value), avg_doc_sz=avg_doc_sz)

    def timing_enable(self, on=True):
        """0 / False: off; 1 / True: events around every launch; 2: around the Gram-apply launches only."""
        self._chk(self._lib.isle_hip_timing_enable(self._h, int(on)))

    def timing_reset(self):
        self._chk(self._lib.isle_hip_timing_reset(self._h))

    def timing_get(self):
        n = len(TIMING_FAMILIES)
        ms = np.zeros(n, np.float64)
        cnt = np.zeros(n, np.uint64)
        self._chk(self._lib.isle_hip_timing_get(self._h, _p(ms), _p(cnt)))
        return {f: (float(ms[i]), int(cnt[i])) for i, f in enumerate(TIMING_FAMILIES)}

    def synchronize(self):
        self._chk(self._lib.isle_hip_synchronize(self._h))
