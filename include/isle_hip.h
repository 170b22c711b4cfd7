/* include/isle_hip.h — C ABI of libisle_hip.so: the MI355X (gfx950) implementation of ISLE's
 * training hot path (truncated SVD of the thresholded word-document matrix B by restarted block
 * Krylov-Schur on B*B^T, k-means++ / Lloyd in the projected space, lift, Lloyd on sparse B).
 *
 * This is the drop-in boundary: each entry point replaces one public method of the reference's
 * ISLE::FPSparseMatrix<float> (or the ProdOp plug-in of BlockKs) — cited per function as
 * file:line relative to the reference root.  The reference-side binding a maintainer would add
 * is shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain C types only; every function returns 0 on success or a negative ISLE_E_* code and
 *    records a message retrievable with isle_hip_last_error(); no exception crosses the ABI.
 *  - all pointers are HOST pointers owned by the caller unless a name ends in _dev.
 *  - one context = one GPU = one process (multi-GPU: one process per GPU, documents
 *    column-sharded; see isle_hip_comm_init).  A context is not thread-safe.
 *  - matrices named *_colmajor are column-major with leading dimension = number of rows, as
 *    in the reference (Armadillo fmat / cblas ColMajor).
 *  - there is NO CPU fallback: without a visible gfx950 device isle_hip_create fails.
 */
#ifndef ISLE_HIP_H
#define ISLE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct isle_ctx isle_ctx;

enum {
  ISLE_OK = 0,
  ISLE_E_ARG = -1,      /* bad argument / state */
  ISLE_E_HIP = -2,      /* HIP runtime error */
  ISLE_E_NOCONV = -3,   /* eigensolver exhausted maxit restarts (Ritz values still returned) */
  ISLE_E_NUMERIC = -4,  /* breakdown: rank repair / small EVD failed */
  ISLE_E_COMM = -5      /* RCCL error */
};

/* ---- context --------------------------------------------------------------------------- */
/* Creates a context on HIP device `device_id`.  Returns NULL (and prints to stderr) on failure. */
isle_ctx* isle_hip_create(int device_id);
void isle_hip_destroy(isle_ctx* ctx);
const char* isle_hip_last_error(isle_ctx* ctx);

/* Multi-GPU (documents column-sharded, one process per GPU).  rank 0 calls
 * isle_hip_comm_unique_id (128 bytes), distributes the bytes by any host channel
 * (bench.py uses torch.distributed), then every rank calls isle_hip_comm_init.
 * doc_offset / docs_global place this rank's shard in the global column numbering. */
int isle_hip_comm_unique_id(void* out128);
int isle_hip_comm_init(isle_ctx* ctx, int world_size, int rank, const void* unique_id128);

/* Rehearsal transport for tests: RCCL refuses two ranks on one device, so to run the sharded path with several
 * ranks on ONE GPU (tests/test_gpu_multirank.py) every collective is staged through host memory and handed to
 * `fn` (the tests pass a torch.distributed/gloo exchange).  kind: ISLE_XCHG_*; dtype: 0 f32, 1 f64, 2 i32, 3 u32,
 * 4 u64.  All-reduce: `buf` holds `count` elements, reduced in place.  All-gather: `buf` holds world * count
 * elements with this rank's part already at rank * count; fill the rest.  Return 0 on success.  Collectives then
 * cost two PCIe copies and two stream synchronisations each: never use it for measurements (bench.py does not). */
enum { ISLE_XCHG_ALLREDUCE_SUM = 0, ISLE_XCHG_ALLREDUCE_MAX = 1, ISLE_XCHG_ALLGATHER = 2 };
typedef int (*isle_host_exchange_fn)(void* user, int kind, void* buf, uint64_t count, int dtype);
int isle_hip_comm_init_host(isle_ctx* ctx, int world_size, int rank, isle_host_exchange_fn fn, void* user);

/* Contiguous, nnz-balanced document ranges for `parts` shards (pure host function, no GPU):
 * bounds[p] .. bounds[p+1] are the columns of shard p; bounds has parts+1 entries. */
int isle_hip_plan_shards(uint64_t num_docs, const int64_t* offsets_CSC, int parts, uint64_t* bounds);

/* ---- input: the matrix B -------------------------------------------------------------- */
/* Uploads this rank's column shard of B (CSC: vals_CSC / rows_CSC / offsets_CSC of
 * include/sparseMatrix.h:23-56; rows ascending within a column, as threshold_and_copy builds
 * them, src/sparseMatrix.cpp:1328-1361).  `num_docs` columns local to this rank;
 * offsets[0] == 0, offsets[num_docs] == nnz.  rows are the reference's 8-byte word_id_t
 * (include/types.h:24) in the _u64 flavour, or 4-byte in the _u32 flavour.
 * doc_offset = global id of local column 0, docs_global = total columns over all ranks
 * (pass 0 and num_docs for single-GPU). */
int isle_hip_upload_csc_u64(isle_ctx* ctx, uint64_t vocab_size, uint64_t num_docs, uint64_t nnz,
                            const float* vals, const uint64_t* rows, const int64_t* offsets,
                            uint64_t doc_offset, uint64_t docs_global);
int isle_hip_upload_csc_u32(isle_ctx* ctx, uint64_t vocab_size, uint64_t num_docs, uint64_t nnz,
                            const float* vals, const uint32_t* rows, const int64_t* offsets,
                            uint64_t doc_offset, uint64_t docs_global);

/* ---- upstream stage: thresholding on the device ------------------- */
/* Uploads this rank's column shard of A, the word-document COUNT matrix in CSC as
 * SparseMatrix<T>::populate_CSC builds it (src/sparseMatrix.cpp:58-133: rows ascending within a
 * column, duplicates merged, counts > 0).  doc_offset / docs_global as for isle_hip_upload_csc. */
int isle_hip_upload_counts_u32(isle_ctx* ctx, uint64_t vocab_size, uint64_t num_docs, uint64_t nnz,
                               const float* counts, const uint32_t* rows, const int64_t* offsets,
                               uint64_t doc_offset, uint64_t docs_global);

/* tdf ingest on the device: DocWordEntriesReader::fill_doc_word_entries
 * (include/utils.h:158-228) + the sort / de-duplication of ISLETrainer::finalize_data
 * (src/trainer.cpp:236-247) + SparseMatrix::populate_CSC (src/sparseMatrix.cpp:58-87).
 * `text`: the bytes of a tdf file ("<doc> <word> <count>" per line, 1-based ids, blanks or tabs between
 * fields, optional '\r', last newline optional).  The result is this context's count matrix, exactly
 * as if it had been passed to isle_hip_upload_counts_u32 (single rank: doc_offset 0).  max_entries:
 * the reference asserts the file holds exactly that many lines; 0 = do not check.  Of several
 * lines with the same (doc, word) the first in the file survives (the reference keeps an unspecified
 * one).  entries_read / nnz (nullable): lines parsed / entries after de-duplication. */
int isle_hip_ingest_tdf(isle_ctx* ctx, const char* text, uint64_t nbytes, uint64_t vocab_size, uint64_t num_docs,
                        uint64_t max_entries, uint64_t* entries_read, uint64_t* nnz);
/* Copies the context's count matrix to the host (any pointer may be NULL); nnz via the call above
 * or offsets[num_docs]. */
int isle_hip_get_A(isle_ctx* ctx, float* counts, uint32_t* rows, int64_t* offsets);

/* normalize_docs (src/sparseMatrix.cpp:136-167) + list_word_freqs / compute_thresholds (:289-485)
 * + FPSparseMatrix(A, zetas) = threshold_and_copy (:1285-1361), or sampled_threshold_and_copy
 * (:1365-1435) when 0 < sample_rate < 1 — what ISLETrainer::train does at src/trainer.cpp:430-485.
 * B is built in device memory and becomes the context's matrix exactly as if it had been passed
 * to isle_hip_upload_csc (empty columns removed; its doc_offset / docs_global follow from the
 * shards' surviving column counts).  Thresholds use the GLOBAL corpus (token total, non-empty
 * documents and per-word histograms are all-reduced).  Sampling keys are drawn on the host from
 * sample_seed and the document's GLOBAL number (the reference uses unseeded rand()); with several
 * ranks all keys are gathered and every rank selects the same pivot, so the shards keep what a
 * single-rank run keeps.
 * Outputs (any may be NULL): docs_kept / nnz_kept describe this rank's shard of B;
 * entries_above_threshold is the global count before sampling (the reference's log line);
 * avg_doc_sz as computed at src/sparseMatrix.cpp:98. */
int isle_hip_threshold(isle_ctx* ctx, uint64_t num_topics, double sample_rate, uint64_t sample_seed,
                       uint64_t* docs_kept, uint64_t* nnz_kept, uint64_t* entries_above_threshold,
                       float* avg_doc_sz);

/* Copies the context's B (and, after isle_hip_threshold, original_cols[D] = global column of A
 * behind each column of B and zetas[V]) to the host; any pointer may be NULL.  Sizes: query with
 * isle_hip_shape. */
int isle_hip_get_B(isle_ctx* ctx, float* vals, uint32_t* rows, int64_t* offsets,
                   uint64_t* original_cols, float* zetas);
int isle_hip_shape(isle_ctx* ctx, uint64_t* vocab_size, uint64_t* num_docs, uint64_t* nnz,
                   uint64_t* doc_offset, uint64_t* docs_global);

/* FPSparseMatrix::frobenius  src/sparseMatrix.cpp:1096-1100  (sum of squares of all entries,
 * over all ranks). */
int isle_hip_frobenius(isle_ctx* ctx, float* out);

/* ---- eigensolver ----------------------------------------------------------------------- */
/* ProdOp::multiply of MKL_SpSpTrProd  include/matUtils.h:336-365:
 * Z (V x b, col-major) = B * (B^T * X), X V x b col-major, 1 <= b <= 32. */
int isle_hip_gram_apply(isle_ctx* ctx, const float* X_colmajor, int b, float* Z_colmajor);

/* Which form of the operator the last build chose for the current B (the operator is built by the first
 * isle_hip_gram_apply / isle_hip_block_ks after an upload, like the MKL_SpSpTrProd constructor
 * include/matUtils.h:52-273): *form = 1 LDS-banded form (every row of B holds one value, as threshold_and_copy
 * src/sparseMatrix.cpp:1285-1321 produces), 0 gather form (any CSC matrix), -1 not built yet.
 * Environment ISLE_GRAM_LDS=0 forces the gather form. */
int isle_hip_operator_form(isle_ctx* ctx, int* form);

/* The environment switches the library honours (no reference counterpart: the reference's choices are compile-time macros,
 * include/hyperparams.h).  Every switch is in ONE table (isle_amd/csrc/common.h IsleKnob); entry `index` of it: its name, its kind
 * ("form": selects between exact forms of one computation, "tuning", "diagnostic", "test hook") and a sentence on its effect.
 * Returns the number of switches (also for index out of range, with the outputs untouched).  Needs no context and no GPU. */
int isle_hip_switch_info(int index, const char** name, const char** kind, const char** what);

/* FPSparseMatrix::compute_block_ks  src/sparseMatrix.cpp:1195-1220  driving
 * BlockKs<ProdOp>(op, nev, ncv, maxit, blk, tol) init()+compute()
 * block-ks/restarted_block_ks.h:190-321.  The reference passes
 * (k, 2k + BLOCK_KS_BLOCK_SIZE, 100, 10, 1e-4) (include/hyperparams.h:38-40).
 * evals: nev Ritz values, descending (= sigma_i^2).  U (V x nev) stays on the device for the
 * k-means calls (fetch with isle_hip_get_U).  seed: start-block RNG seed (the reference uses
 * unseeded rand(); parity does not depend on it).  nev / ncv need not be multiples of blk: a
 * decomposition grows block by block until it has at least ncv rows (the reference overruns its
 * basis in that case).  Returns ISLE_E_NOCONV (not 0) if maxit
 * restarts were exhausted — the reference reports full convergence in that case
 * (SURVEY.md App. C #7); evals/U are still the last Ritz pairs.
 * nconv/restarts/napplies may be NULL. */
int isle_hip_block_ks(isle_ctx* ctx, int nev, int ncv, int maxit, int blk, float tol, uint64_t seed,
                      float* evals, int* nconv, int* restarts, int* napplies);

/* The same solver on a caller-supplied dense symmetric operator: BlockKs<ProdOp> is a template over any symmetric
 * operator with multiply()/rows()/cols() (block-ks/restarted_block_ks.h:18-40); this entry is its instantiation with
 * utils::ArmaMatProdOp (block-ks/ks_utils.h:167-182, multiply(X) = A * X), the operator the reference pairs with its
 * known-spectrum recipe utils::get_seed_eigs (ks_utils.h:136-165).  It runs the SAME host loop and device kernels as
 * isle_hip_block_ks (init / expand / truncate / compute, panel QR, rank repair, small EVD, Ritz rotation); only the
 * operator application is a dense product.  A: n x n col-major symmetric (host).  start_block: NULL, or n x blk
 * col-major values for the first draw of init()'s start block (:211-218 redraws at random while it is rank deficient).
 * evals: nev Ritz values, descending.  U (nullable): n x nev col-major Ritz vectors.  nconv: Ritz pairs that passed the
 * residual test of the last restart.  nconv_ref_rule (nullable): what the reference itself would report — equal to nconv
 * on convergence; after maxit restarts its rule (:303-317) looks at the expanded H without dividing and therefore says
 * nev, while this library returns ISLE_E_NOCONV with the honest count.  The context's B, U and
 * k-means state are not touched.  Single rank only. */
int isle_hip_block_ks_dense(isle_ctx* ctx, const float* A_colmajor, uint64_t n, int nev, int ncv, int maxit, int blk,
                            float tol, uint64_t seed, const float* start_block, float* evals, float* U_colmajor,
                            int* nconv, int* nconv_ref_rule, int* restarts, int* napplies);

/* U_colmajor (V x nev), what compute_block_ks memcpy's at src/sparseMatrix.cpp:1214. */
int isle_hip_get_U(isle_ctx* ctx, float* U_colmajor);
/* Test hook: install a caller-provided U (V x k col-major) instead of running the eigensolver. */
int isle_hip_set_U(isle_ctx* ctx, const float* U_colmajor, int k);

/* Dense symmetric eigendecomposition used inside truncate() (arma::eig_sym,
 * block-ks/restarted_block_ks.h:150-161): S n x n col-major symmetric -> evals descending,
 * vecs col-major.  Exposed for parity tests. */
int isle_hip_eig_sym(isle_ctx* ctx, const float* S_colmajor, int n, float* evals_desc, float* vecs_colmajor);

/* ---- k-means --------------------------------------------------------------------------- */
/* FPSparseMatrix::kmeans_init_on_projected_space(k, reps = 1, seeds, centers_coords)
 * src/sparseMatrix.cpp:2212-2238 -> kmeanspp_on_projected_space :2133-2209.
 * inject_seeds: NULL, or k global doc ids that replace the D^2 draws (test hook; the round
 * schedule and min-distance updates still run).  seeds_out: k global doc ids.
 * C_lowd: k x k, centre c at offset c*k (the reference's centers_lowd).  residual: as
 * returned by the reference (App. C #9).  rng_seed seeds the host draw RNG (rand() stand-in). */
int isle_hip_kmeanspp_projected(isle_ctx* ctx, int k, const uint64_t* inject_seeds, uint64_t rng_seed,
                                uint64_t* seeds_out, float* C_lowd, float* residual, int* rounds);

/* Test hook, no context needed: the first n values of the host generator that stands in for the reference's rand() calls
 * (src/sparseMatrix.cpp:2150, include/matUtils.h:473-477) — glibc's rand() after srand(seed); seed 1 = never seeded. */
int isle_hip_host_rand(uint64_t seed, int n, uint32_t* out);

/* Optional test hook: copies the current min-distance array (local docs) after kmeanspp. */
int isle_hip_get_min_dist(isle_ctx* ctx, float* min_dist);

/* FPSparseMatrix::run_lloyds_on_projected_space(k, C_lowd, NULL, max_reps)
 * src/sparseMatrix.cpp:2016-2072.  C_lowd in/out.  assign_out (local docs, may be NULL). */
int isle_hip_lloyds_projected(isle_ctx* ctx, int k, float* C_lowd, int max_reps, int* iters_run,
                              uint32_t* assign_out);

/* FPSparseMatrix::left_multiply_by_U_Spectra(out, in, ld_in, ncols)
 * src/sparseMatrix.cpp:1438-1450: centers (V x ncols col-major) = U (V x k) * in (ld_in x ncols).
 * centers may be NULL: the result then only stays on the device as the start point of
 * isle_hip_lloyds_sparse. */
int isle_hip_lift_centers(isle_ctx* ctx, const float* in, int ld_in, int ncols, float* centers);

/* FPSparseMatrix::run_lloyds(k, centers, closest_docs, max_reps)  src/sparseMatrix.cpp:1690-1746.
 * centers_in: V x k col-major start centres, or NULL to use the device-resident result of
 * isle_hip_lift_centers.  centers_out: V x k col-major (may be NULL).  assign: local docs ->
 * centre index (the caller buckets it into closest_docs[k], ascending doc id). */
int isle_hip_lloyds_sparse(isle_ctx* ctx, int k, const float* centers_in, float* centers_out,
                           uint32_t* assign, int max_reps, int* iters_run);

/* ---- downstream stage: catchwords, topic model, edge topics ---
 * These operate on the count matrix A left in device memory by isle_hip_upload_counts_u32 (single
 * rank only for now) and on the partition of B's columns, mapped back to A's documents through
 * original_cols exactly as src/trainer.cpp:572-575 does.
 *
 * isle_hip_catchwords = SparseMatrix::rth_highest_element for every topic (src/sparseMatrix.cpp:491-524,
 * called at src/trainer.cpp:586-590) + SparseMatrix::find_catchwords (:573-595).
 *   assign: this context's B columns -> topic (what isle_hip_lloyds_sparse returned), or NULL to use
 *           the partition still resident from the last isle_hip_lloyds_sparse call.
 *   r:      the rank of src/trainer.cpp:579-583 (>= 1).   rho: rho_c (include/hyperparams.h:11).
 *   thresholds (nullable): vocab x num_topics column-major, the reference's catchword_thresholds.
 *   catch_topic (nullable): vocab entries, the topic a word is a catchword of or -1 (the rule admits
 *           at most one topic per word); catchwords[t] of the reference = { w : catch_topic[w] == t }
 *           ascending. */
int isle_hip_catchwords(isle_ctx* ctx, int num_topics, const uint32_t* assign, uint64_t r, double rho,
                        float* thresholds, int32_t* catch_topic, uint64_t* num_catchwords);

/* SparseMatrix::construct_topic_model (src/sparseMatrix.cpp:597-838) after isle_hip_catchwords.
 *   rank_threshold: src/sparseMatrix.cpp:720.
 *   model (nullable): vocab x num_topics column-major, L1-normalised topic vectors (DenseMatrix Model).
 *   model_threshold (nullable): num_topics.  top1/top2 (nullable): per document of A, the two heaviest
 *   catchword topics (top_topic_pairs, :687-708) or -1/-1.  doc_topic_sums (nullable): number of
 *   non-zero (document, topic) catchword sums; fetch them with isle_hip_get_doc_topic_sums. */
int isle_hip_topic_model(isle_ctx* ctx, int num_topics, uint64_t rank_threshold, float* model,
                         float* model_threshold, int32_t* top1, int32_t* top2, uint64_t* doc_topic_sums);
/* doc_offsets: docs(A) + 1 entries; topic / val: doc_topic_sums entries, (document, topic) ascending. */
int isle_hip_get_doc_topic_sums(isle_ctx* ctx, int64_t* doc_offsets, uint32_t* topic, float* val);

/* The FPaxpy pair of ISLETrainer::construct_edge_topics_v2 (src/trainer.cpp:1152-1159) on the
 * device-resident Model: edge[:, e] = primary_ratio * Model[:, pairs[2e]] +
 * (1 - primary_ratio) * Model[:, pairs[2e+1]]; edge is vocab x n column-major. */
int isle_hip_edge_topics(isle_ctx* ctx, const int64_t* pairs, int n, float primary_ratio, float* edge);

/* ---- inference ---------------------------------------------------- */
/* ISLEInfer over a batch of documents: drivers/ISLEInfer.cpp:60-112 (normalize_docs(true, true),
 * infer_doc_in_file per document, heaviest topics) with ISLEInfer::mwu / grad / calculate_llh
 * src/infer.cpp:361-492.  model_by_word: vocab x num_topics ROW-major (element (word, topic) at
 * word * num_topics + topic), what load_model_from_sparse_file src/infer.cpp:32-70 builds from
 * M_hat_catch_sparse.  The documents are a count matrix in CSC (word ids ascending per document).
 * iters / Lf_guess: INFER_ITERS_DEFAULT 15 / INFER_LF_DEAFULT 10.0 (include/hyperparams.h:81-82).
 * avg_doc_sz: SparseMatrix::avg_doc_sz of the documents (src/sparseMatrix.cpp:98).
 * Outputs (host, each nullable): weights docs x num_topics row-major (1 / num_topics where inference did
 * not converge, as the dense writer prints them); top_topic / top_weight docs x 5, topics with weight >
 * 1 / num_topics in decreasing weight, -1 / 0 where there are fewer; llh docs x 2 (first = sum * avg_doc_sz,
 * second = sum * words in the document; 0, 0 when not converged); nconverged = documents with llh.first != 0.
 * Independent of the matrices held by the context. */
int isle_hip_infer(isle_ctx* ctx, uint64_t vocab_size, int num_topics, const float* model_by_word,
                   uint64_t num_docs, uint64_t nnz, const float* counts, const uint32_t* rows,
                   const int64_t* offsets, int iters, float Lf_guess, float avg_doc_sz, float* weights,
                   int32_t* top_topic, float* top_weight, float* llh, uint64_t* nconverged);

/* ---- measurement ----------------------------------------------------------------------- */
/* Per-kernel-family device time accumulated with HIP events on the context's stream since the
 * last reset (only while enabled; enabling adds event records around each launch).
 * Families: see ISLE_T_* below. */
enum {
  ISLE_T_GRAM_PASS1 = 0,   /* Y = B^T X   (CSC gather)            */
  ISLE_T_GRAM_PASS2 = 1,   /* Z = B Y     (chunked-CSR gather + chunk reduce) */
  ISLE_T_ORTHO = 2,        /* V^T F, F -= V H                      */
  ISLE_T_QR = 3,           /* panel QR (Gram, apply)               */
  ISLE_T_EVD = 4,          /* small symmetric EVD                  */
  ISLE_T_ROTATE = 5,       /* Ritz rotation (every restart + the final extraction of U) */
  ISLE_T_PROJECT = 6,      /* P = U^T B                            */
  ISLE_T_KMPP = 7,         /* k-means++ rounds                     */
  ISLE_T_LLOYD_PROJ = 8,   /* projected Lloyd assign + update      */
  ISLE_T_SPARSE_ASSIGN = 9,/* sparse Lloyd: distances + argmin     */
  ISLE_T_SPARSE_UPDATE = 10,/* sparse Lloyd: centroid update       */
  ISLE_T_BAND_BUILD = 11,  /* chunked-CSR copy of B (per solve)   */
  ISLE_T_COMM = 12,        /* collectives                          */
  ISLE_T_THRESHOLD = 13,   /* A -> B thresholding (upstream stage) */
  ISLE_T_POST = 14,        /* catchwords / topic model / edge topics (downstream stage) */
  ISLE_T_INGEST = 15,      /* tdf text -> count matrix */
  ISLE_T_INFER = 16,       /* ISLEInfer: multiplicative-weights inference */
  ISLE_T_LIFT = 17,        /* centres = U C_lowd (left_multiply_by_U_Spectra)  */
  ISLE_T_COUNT = 18
};
/* on: 0 = off, 1 = events around every launch, 2 = around the Gram applications only (what bench.py's timed region uses): the
 * LDS-banded form then books one event pair per application — both passes and the gap between them — under ISLE_T_GRAM_PASS1. */
int isle_hip_timing_enable(isle_ctx* ctx, int on);
int isle_hip_timing_reset(isle_ctx* ctx);
/* ms[ISLE_T_COUNT], launches[ISLE_T_COUNT] */
int isle_hip_timing_get(isle_ctx* ctx, double* ms, uint64_t* launches);
int isle_hip_synchronize(isle_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* ISLE_HIP_H */
