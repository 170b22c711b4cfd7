// isle_amd/csrc/common.h — internal declarations shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>

// ------------------------------------------------------------------------------------------
// Launch guard.  On this stack a kernel launch of 2^32 threads or more (grid x block) does not run and reports NO error — round 3's
// yy2_scan_k at 10 M documents left collapsed partitions that way.  Every launch of the library goes through this redefinition of
// hipLaunchKernelGGL: a launch that does not fit is not issued and is remembered (file:line), and the HIPCHK that follows every launch
// turns it into ISLE_E_ARG with that location.  Index spaces that can exceed the limit (D x k = 10^10 at config 3) are walked by
// grid-stride loops or chunked launches at their call sites; this guard is what makes a missed one loud.
// ------------------------------------------------------------------------------------------
struct IsleLaunchRefused {
  const char* file = nullptr;
  int line = 0;
  unsigned long long threads = 0;
};
inline IsleLaunchRefused& isle_launch_refused() {
  static thread_local IsleLaunchRefused r;
  return r;
}
inline bool isle_launch_fits(dim3 g, dim3 b, const char* file, int line) {
  const unsigned long long blocks = (unsigned long long)g.x * g.y * g.z, threads = blocks * ((unsigned long long)b.x * b.y * b.z);
  if (threads < (1ull << 32)) return true;  // (and with it every dimension's work-item count, which is what the dispatch packet holds in 32 bits)
  IsleLaunchRefused& r = isle_launch_refused();
  if (!r.file) {
    r.file = file;
    r.line = line;
    r.threads = threads;
  }
  fprintf(stderr, "[isle_hip] %s:%d: a launch of %llu threads (grid %u x %u x %u) was refused: 2^32 threads or more do not run on this stack\n", file, line, threads,
          g.x, g.y, g.z);
  return false;
}
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...)                                   \
  do {                                                                                            \
    if (isle_launch_fits(dim3(grid), dim3(block), __FILE__, __LINE__)) kern<<<dim3(grid), dim3(block), (lds), (stream)>>>(__VA_ARGS__); \
  } while (0)

#include "../../include/isle_hip.h"

#define ISLE_WAVE 64

struct isle_event_pair {
  hipEvent_t a, b;
  int fam;
};

// Device buffer with explicit capacity (grows, never shrinks).
template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipMalloc((void**)&p, n * sizeof(T));
    if (e == hipSuccess) cap = n;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }  // every buffer of a context dies with it
};

// One side of the LDS-banded Gram apply (gram_lds.hip): a sliced-ELL stream of band-local u16 source ids.
// ------------------------------------------------------------------------------------------
// Environment switches.  Every switch the library honours is in this table; they are read into the context when it is created and
// again at the entry of every C-ABI call (isle_enter, api.cpp) — never inside a loop or a launch path — and the code asks the context
// (isle_ctx::knob).  DESIGN.md section "Environment switches" is this table with the measurements behind the defaults.
// ------------------------------------------------------------------------------------------
enum IsleKnob {
  KN_GRAM_LDS, KN_GL_G1, KN_GL_G2, KN_GL_PLACE, KN_GL_FILL_BUCKETS, KN_GL_ROUNDS, KN_GL_COLUMNS, KN_GL_PANEL, KN_GL_WIDE_GROUPED, KN_WIDE_GATHER, KN_WIDE_LDS,
  KN_KS_ROWSHARD, KN_KS_SYNC, KN_KS_ORTHO_PASSES, KN_UPDATE_MFMA, KN_EVD_JACOBI, KN_TD_CHAIN, KN_TD_BAR, KN_TD_BACK, KN_EVD_SPLIT,
  KN_KMPP_HOST_DICE, KN_KMPP_SPARSE, KN_KMPP_TRACK, KN_NO_HAMERLY, KN_KMEANS_BOUNDS, KN_PROJ_BOUNDS, KN_PROJ_FULL, KN_FIRST_ASSIGN, KN_GEMM_BF16X3, KN_GEMM_EPILOGUE, KN_GEMM_TERMS, KN_GEMM_DMA, KN_YY_MODE, KN_YY_FUSED, KN_YY_MOVERS, KN_YY_REGROUP, KN_YY_ORDER, KN_PT_SORT, KN_PROJ_ACTIVE, KN_PROJ_SUMS, KN_CENTERS_FRESH,
  KN_INFER_CAP_ROWS, KN_CHUNK_COLS, KN_COMM_TIMEOUT, KN_COMM_SELFTEST, KN_FORCE_COMM, KN_TEST_STALL_MS,
  KN_ROCTX, KN_HOST_TRACE, KN_DEBUG_HAMERLY, KN_DEBUG_EVD, KN_GL_VERBOSE, KN_TD_FORCE_BAIL_RANK, KN_GL_TEST_CUS, KN_GL_ABLATE_SKIP,
  KN_COUNT
};
struct IsleKnobInfo {
  const char* name;
  const char* kind;  // "form" (selects between exact forms of a computation), "tuning", "diagnostic", "test hook"
  const char* what;
};
extern const IsleKnobInfo isle_knob_table[KN_COUNT];

struct GlDesc {  // one workgroup: 16 waves wave0 + i*wstride (i < nw), source bands [b0, b1), output slab
  uint32_t wave0, wstride, nw, b0, b1, slab, pos_base, pad;
};
struct GlSide {
  uint32_t n_out = 0, n_src = 0, NB = 0, nslice = 0, nwv = 0, ndesc = 0;
  int G = 4;                  // groups of a wave = output items per lane (4 ... 8)
  uint32_t wpg = 16;          // waves per workgroup of the apply kernel on this side
  DevBuf<uint32_t> slice_of;  // nwv x G: slice (64 consecutive output positions) of (wave, group), 0xffffffff = none
  DevBuf<int64_t> roff;       // nwv x NB + 1: first super-round of (wave, band)
  DevBuf<uint16_t> cnt;       // nwv x NB x 8: super-rounds (4 nonzeros per lane) of (wave, band, group), zero beyond G
  DevBuf<uint2> ids;          // super-rounds x 64 lanes: four u16 band-local source ids per lane (+ prefetch slack)
  DevBuf<GlDesc> desc;
  int64_t total_sr = 0;
};

struct isle_ctx {
  // environment switches as read at the last C-ABI entry (isle_refresh_knobs)
  std::string knob_val[KN_COUNT];
  bool knob_set[KN_COUNT] = {};
  const char* knob(int id) const { return knob_set[id] ? knob_val[id].c_str() : nullptr; }
  bool knob_on(int id) const { return knob_set[id]; }                                         // set at all (any value)
  bool knob_is(int id, const char* v) const { return knob_set[id] && knob_val[id] == v; }
  bool knob_zero(int id) const { return knob_set[id] && atoi(knob_val[id].c_str()) == 0; }
  int device = 0;
  size_t total_mem = 0;  // bytes of device memory (isle_scratch_ok)
  hipStream_t stream = nullptr;
  std::string err;
  int num_cus = 256;

  // --- page-locked host memory for the small per-iteration copies of the control loops (mailboxes, scalars, flags): a
  // hipMemcpyAsync to or from pageable memory blocks the host until the copy has run, which serialises the enqueue-ahead
  // loops (api.cpp); PIN_* are fixed regions of it
  char* pin = nullptr;
  // growable page-locked staging for the k x k centre matrices that cross the boundary at the ends of the k-means calls (4 MB at
  // k = 1000): hipMemcpyAsync from freshly allocated pageable memory was seen to block for 26 ms there (api.cpp, lloyds_projected)
  char* pin_stage = nullptr;
  size_t pin_stage_cap = 0;
  hipError_t pin_stage_reserve(size_t bytes) {
    if (bytes <= pin_stage_cap) return hipSuccess;
    if (pin_stage) (void)hipHostFree(pin_stage);
    pin_stage = nullptr;
    pin_stage_cap = 0;
    hipError_t e = hipHostMalloc((void**)&pin_stage, bytes, hipHostMallocDefault);
    if (e == hipSuccess) pin_stage_cap = bytes;
    return e;
  }
  static constexpr size_t PIN_MAIL = 0, PIN_MAIL_SLOT = 1u << 20, PIN_SMALL = 2u << 20, PIN_BYTES = (2u << 20) + (256u << 10);

  // --- communicator (null for single GPU)
  ncclComm_t comm = nullptr;
  int world = 1, rank = 0;
  // rehearsal transport (isle_hip_comm_init_host): collectives staged through host memory and a caller-provided exchange
  // function, so that several ranks can share one GPU in tests (RCCL refuses two ranks on one device)
  isle_host_exchange_fn host_xchg = nullptr;
  void* host_xchg_user = nullptr;
  bool multi() const { return comm != nullptr || host_xchg != nullptr; }
  // watchdog of the RCCL collectives (api.cpp, wd_*): behind every collective the device writes its sequence number into a page-locked
  // word; a thread started with the communicator compares it with the number issued and, when collectives are pending and none has
  // completed for ISLE_COMM_TIMEOUT_S seconds, aborts the communicator — the stream drains, every HIPCHK from then on returns ISLE_E_COMM
  // and the caller gets an error instead of a hang (a rank that died, or a rank whose replicated state diverged and left the sequence)
  std::atomic<uint32_t> wd_issued{0};
  volatile uint32_t* wd_done = nullptr;  // page-locked, device-visible
  std::atomic<int> comm_dead{0};
  std::atomic<bool> wd_stop{false};
  std::thread wd_thread;
  double wd_timeout_s = 300.0;

  // --- B (this rank's column shard), CSC
  uint64_t V = 0, D = 0, nnz = 0, doc_offset = 0, D_global = 0;
  DevBuf<float> vals;
  DevBuf<uint32_t> rows;
  DevBuf<int64_t> offs;

  // --- A (word-document counts, this rank's column shard) and the thresholding workspaces (threshold.hip)
  uint64_t a_V = 0, a_D = 0, a_nnz = 0, a_doc_offset = 0, a_D_global = 0;
  bool a_ready = false;
  DevBuf<float> a_cnt;
  DevBuf<uint32_t> a_rows;
  DevBuf<int64_t> a_offs;
  DevBuf<uint16_t> a_q;        // rounded normalised counts, clamped to maxv
  DevBuf<uint32_t> a_hist;     // V x (maxv + 1)
  DevBuf<uint32_t> a_kept, a_flag;
  DevBuf<float> a_wgt;
  DevBuf<int64_t> a_off_all, a_col_of, a_scan;
  DevBuf<float> zetas;             // V
  DevBuf<uint64_t> original_cols;  // D (B column -> global document id of A)
  bool b_from_threshold = false;
  float a_avg = 0.f;               // avg_doc_sz of the whole corpus (src/sparseMatrix.cpp:98)
  bool a_avg_valid = false;

  // --- downstream stage (post.hip): catchwords, topic model, edge topics
  DevBuf<float> a_nv;              // normalised values of A
  DevBuf<int32_t> p_cluster_of;    // a_D
  DevBuf<uint32_t> p_cnt, p_min, p_slot;   // V x k, word-major
  DevBuf<float> p_thr;             // V x k, word-major catchword thresholds
  DevBuf<uint64_t> p_seg_id, p_seg_off, p_counters;
  DevBuf<uint32_t> p_seg_len, p_seg_rank, p_seg_cur;
  DevBuf<float> p_segvals;
  DevBuf<int32_t> p_catch;         // V: topic the word is a catchword of, or -1
  DevBuf<uint32_t> p_nz;
  DevBuf<int64_t> p_dts_off;       // a_D + 1
  DevBuf<uint32_t> p_dts_topic;
  DevBuf<float> p_dts_val;
  uint64_t p_dts_n = 0;
  DevBuf<int32_t> p_top1, p_top2;  // a_D
  DevBuf<uint32_t> p_tcnt;
  DevBuf<int64_t> p_toff;
  DevBuf<float> p_mthr;            // k
  DevBuf<float> p_model;           // V x k col-major
  int p_k = 0;                     // num_topics of the last catchword pass
  bool p_catch_ready = false, p_model_ready = false, assign_valid = false;

  // --- chunked-CSR copy of B for Z = B*Y (built per eigensolve, like the reference's operator ctor)
  bool band_ready = false;
  uint32_t band_rows = 0;  // requested columns per chunk (0 = default), env ISLE_CHUNK_COLS
  uint32_t nbands = 0;     // number of column chunks (multiple of 8)
  uint32_t chunk_cols = 0;
  DevBuf<uint32_t> bcol;   // nnz   doc id
  DevBuf<float> bval;      // nnz
  DevBuf<int64_t> seg_off; // nchunks*V + 1
  DevBuf<float> Zpart;     // nchunks x V x BP partial rows

  // --- LDS-banded Gram apply (gram_lds.hip): used when every row of B holds a single value (B = diag(s) * pattern, which is
  // what threshold_and_copy produces, src/sparseMatrix.cpp:1285-1321); otherwise the gather kernels of spmm.hip run.
  int gl_mode = -1;          // -1 not decided for the current B, 0 gather path, 1 LDS path
  GlSide gl1, gl2;           // pass 1 (outputs = documents, sources = words), pass 2 (outputs = words, sources = documents)
  DevBuf<float> rowval;      // V: the value of row w
  DevBuf<int> gl_flag;
  DevBuf<uint32_t> dperm, dpos, wperm, wpos;  // position -> document, document -> position, position -> word, word -> position
  DevBuf<uint64_t> gl_key_a, gl_key_b;
  DevBuf<uint32_t> gl_val_a, gl_val_b;
  DevBuf<uint32_t> gl_bst;   // D x (NB1 + 1): first entry of each word band inside a document's column
  DevBuf<uint16_t> gl_cellcnt;   // V x NB2: entries of (word, document band)
  DevBuf<uint32_t> gl_srsum, gl_sbase;
  DevBuf<uint32_t> gl_fb_cnt, gl_fb_tmp;  // pass-2 fill by buckets of word positions: entries per (band, bucket); the packed entries (nnz words = 4 GB at config 3).
                                          // KEPT between builds, like gl_pscratch below (7.7 GB at 10 M documents): every solve builds the operator again and a
                                          // hipFree / hipMalloc pair of that size per step costs more than the memory is worth on a 288 GB device; both count
                                          // against what isle_scratch_ok sees as free — it decides from the device's TOTAL memory, so the routes do not depend on them
  DevBuf<int64_t> gl_fb_off;
  DevBuf<uint16_t> gl_scnt;  // super-rounds of (slice, band) of pass 2 (gl_sbase's indexing)
  DevBuf<uint32_t> gl_biglist;   // [count | (wave, band, group) triples whose pass-2 cells are too long for the register sort]
  DevBuf<uint32_t> ccount;       // k x V, centre-major: members of centre c that contain word w (sparse Lloyd centroid update)
  DevBuf<uint32_t> ccounted;     // D: the centre under which document d is counted in ccount
  DevBuf<int64_t> gl_scan;
  DevBuf<unsigned long long> gl_blocktot;
  DevBuf<uint32_t> gl_slab0, gl_nch;  // per word block: first partial slab, number of slabs
  uint32_t gl_block_items = 4096;     // words per word block of pass 2 (256 per wave of the block)
  DevBuf<float> gl_Xs, gl_part;       // scaled panel diag(s) X; partial rows of Z per (word block, band chunk)
  DevBuf<float> gl_pscratch;          // k-wide product, grouped form: up to 16 panels' rows, whole and in position order (D x 12 floats each), and the
                                      // rows' squared-norm partials per group of panels (k_gl_wide)
  DevBuf<uint32_t> rs_hist;           // radix sort scratch (ingest.hip: k_sort_pairs_u64)
  DevBuf<int64_t> rs_hist_off, rs_scratch;

  // --- gram-apply workspaces
  DevBuf<float> Xrm, Yrm, Zrm;   // V*BP, D*BP, V*BP
  DevBuf<float> Xcm, Zcm;        // staging for the host-pointer API

  // --- eigensolver state
  DevBuf<float> basis;     // V x (ncv + blk) col-major
  DevBuf<float> Fbuf;      // V x blk
  DevBuf<float> Tmp;       // V x nev (rotation target)
  DevBuf<double> part;     // partial reductions
  DevBuf<float> coef;      // 3 x (m x blk)
  DevBuf<double> gram;     // blk x blk
  DevBuf<double> pq_part, pq_R1;  // panel QR: partial Gram matrices, first triangular factor
  DevBuf<float> pq_T;             // panel QR: T (32 x 32) and R (32 x 32)
  DevBuf<int> pq_meta;            // panel QR: rank, status, pivots
  DevBuf<float> small;     // misc small device scratch
  DevBuf<double> jacW, jacV;  // n x n each
  DevBuf<double> jacS;        // per-pair Gram / rotation scratch
  DevBuf<float> Wf;        // n x n float eigenvectors
  DevBuf<float> evd_in;    // n x n: the fp32 matrix handed to the small EVD
  int U_k = 0;             // number of columns of U available
  DevBuf<float> Ucm;       // V x k col-major  (copy of basis[:, :k])
  DevBuf<float> Urm;       // V x ldk row-major
  int ldk = 0;

  // --- k-means state
  DevBuf<float> P;         // D x ldk
  DevBuf<float> pnorm;     // D
  bool P_ready = false;
  DevBuf<float> Pt;        // ldk x D coordinate-major copy (projected assignment), only for ldk <= 256
  DevBuf<float> dotsT;     // D x k centre-major dot products (first assignment of Lloyd on B through the projection)
  DevBuf<uint4> gemm_b3;   // the small operand of k_gemm_nn_assign split into three bf16 terms (gemm_bf16x3.h)
  DevBuf<float> yy_gmax2;     // G floats: the groups' largest movements without the movers
  DevBuf<float> yy_mdots;     // D x 12: dot products of every document with the movers' centres (YyMovers)
  DevBuf<float> cmax_buf;     // one float: the largest centre norm of a full projected pass
  DevBuf<uint32_t> ga_redo;   // rows the two-term pass of an assignment product left open (last = count), dense.hip gemm_assign_two_pass
  DevBuf<float> ga_bn;             // squared norms of the product's columns and their maximum
  DevBuf<float> ga_rows, ga_rown;  // those rows gathered coordinate-major, their squared norms
  bool ts_open = false;       // a TimeScope is open (nested scopes are not timed again)
  bool roctx_open = false;    // ... and a roctx range (ISLE_ROCTX; nested scopes push no second one)
  uint32_t ga_last_redo = 0;  // their number in the last product (diagnostic: isle_hip_measure)
  DevBuf<float> assign_part;  // per document and 64-column slot the best centre of the slot (16 bytes: k_gemm_assign_yy / _tiles, dense.hip)
  DevBuf<float> lift_C;    // the k x k coefficients the device-resident centres were lifted from (centres = U lift_C^T)
  int lift_ld = 0, lift_k = 0;
  bool lift_valid = false;
  bool Pt_ready = false;
  DevBuf<uint4> Pt2;       // the coordinate-major copy split in two bf16 terms, as the LDS image of every (row block, slab): the A operand of the
  bool Pt2_ready = false;  // LDS-DMA assignment products (gemm_bf16x3.h, gemm_bf16x2_dma_k)
  bool Pt2_pos = false;    // its rows are POSITIONS of the length order (row m = document dperm[m]): made by the grouped projection on its way
  DevBuf<float> min_dist;  // D
  DevBuf<double> cum;      // D + 1
  DevBuf<double> scan_blk;
  DevBuf<float> Cdev;      // k x ldk centres (projected)
  DevBuf<float> cnorm;     // k
  DevBuf<float> Csum;      // k x ldk + k
  DevBuf<uint32_t> assign, assign_prev;
  DevBuf<int> counts;
  DevBuf<uint32_t> members;  // documents grouped by centre
  DevBuf<int> moff;          // k+1 offsets, k cursors
  bool members_valid = false; // members = a permutation of the local docs grouped by centre
  DevBuf<int> flags;
  DevBuf<float> centers_rm;   // V x ldk  (word-space centres, row-major)
  DevBuf<float> centers_cm;   // V x k col-major staging
  bool centers_ready = false;
  int centers_k = 0;
  DevBuf<float> dnorm;     // D   |b_d|^2
  DevBuf<float> hub, hlb;  // D   Hamerly bounds (sparse Lloyd)
  DevBuf<float> yglb;      // D x G Yinyang group bounds (sparse Lloyd)
  DevBuf<float> ptlb;      // D x TL tile bounds (projected Lloyd at k > 224)
  DevBuf<uint32_t> pneed;  // D: tiles a document has to re-examine
  DevBuf<uint32_t> pcand;  // D + 1: candidates of the first filter stage (last = count)
  DevBuf<int> seg_desc;    // projected centroid sums: chunk descriptors (beg, end) and the centres' first chunks
  DevBuf<float> seg_part;  // one partial row per chunk
  DevBuf<uint32_t> proj_counted;  // D: the centre under which a document is counted in the projected sums (k_proj_accumulate_delta)
  DevBuf<uint32_t> proj_nch;      // number of documents that changed centre
  DevBuf<float> proj_dpart;       // k x 8 x ldk partial sums of the changes
  DevBuf<float> Csum_local;       // several ranks: this rank's sums (Csum holds the all-reduced ones)
  DevBuf<uint32_t> active; // D + 1 (last = count)
  DevBuf<unsigned long long> dbg_cnt;  // diagnostics (ISLE_DEBUG_HAMERLY)
  // what the k-means++ rounds keep for Lloyd's first assignment in span(U) (kmeans.hip kmpp_min_dots_track_k)
  DevBuf<uint32_t> kmpp_arg;   // nearest seed so far
  DevBuf<float> kmpp_m2a;      // smallest distance to the other seeds of its tile
  DevBuf<float> kmpp_tmin;     // smallest distance per tile of 32 seeds, tile-major [tile][document]
  DevBuf<float> kmpp_best;     // distances to the nearest of ALL k seeds (min_dist does not include the last batch)
  bool kmpp_track = false;     // every round so far went through the tracking kernel
  int kmpp_track_seeds = 0;    // seeds folded in
  int kmpp_track_k = 0;        // 0: nothing to start from; k: state complete for the k seeds in kmpp_C_host, projection generation kmpp_P_gen
  uint64_t P_gen = 0, kmpp_P_gen = 0;
  std::vector<float> kmpp_C_host;  // the seeds' coordinates as handed to the caller (k x k)
  // Yinyang iteration ordered by group (spmm.hip, k_yy2_assign)
  DevBuf<float> yy_cg;                 // centres group-major: G tables of V x 8 floats
  DevBuf<uint32_t> yy_map;             // regrouped Yinyang groups: id_of_slot (8 G) then slot_of_id (k)
  DevBuf<float> yy_cns;                // squared centre norms by slot (8 G)
  DevBuf<float> yy_liftC;              // the lift coefficients' rows by slot (first assignment through the projection)
  DevBuf<uint32_t> yy_own, yy_res;     // YyRes (3 words) per active slot / per pair
  DevBuf<unsigned long long> yy_need;  // per active slot: bit mask of the groups to scan
  DevBuf<uint32_t> yy_cnt, yy_off;     // pairs per active slot, their exclusive scan
  DevBuf<uint8_t> yy_pgrp;             // group of every pair (slot-major order)
  DevBuf<float> centers_old;  // V x ldk
  DevBuf<float> Pa, pna, Cold; // compacted active rows (ldk x n), their norms, previous projected centres

  // block Krylov-Schur, pipelined expand loop: device mailbox [rank, status, pivots | R | coefficients] of a step, fetched by
  // one copy; the events that mark its arrival
  hipEvent_t ks_ev[2] = {nullptr, nullptr};
  hipEvent_t ks_ev_ready[2] = {nullptr, nullptr};  // the step's mailbox is complete on the main stream (the copy stream waits for it)
  hipStream_t copy_stream = nullptr;               // the mailbox's way to the host: beside the main stream, not in it (api_ks.cpp expand)
  DevBuf<float> ks_mail;
  DevBuf<float> ks_top;     // truncation: the locked rows of H next to the rotated block, and their product with the Ritz rotation
  DevBuf<float> ks_gather;  // row-sharded orthogonalisation: the ranks' slices of F (world x nloc x blk)

  // largest dynamic-LDS size requested so far per kernel ON THIS CONTEXT'S DEVICE (hipFuncSetAttribute is per device; a
  // process-wide flag would leave the kernels of a second GPU at the 64 KB default)
  std::vector<std::pair<const void*, int>> lds_attr;
  bool td_persist_failed = false;  // the persistent tridiagonalisation hit its barrier time-out once (GPU shared): launch chain from now on

  // --- timing
  bool timing = false;
  uint32_t timing_mask = 0xffffffffu;  // families whose launches are bracketed by events while timing is on
  std::vector<isle_event_pair> ev_used;
  std::vector<isle_event_pair> ev_free;
  double t_ms[ISLE_T_COUNT] = {0};
  uint64_t t_n[ISLE_T_COUNT] = {0};
};

int isle_fail(isle_ctx* c, int code, const char* fmt, ...);
// A new matrix of D documents and nnz entries replaces the context's: the largest derived buffers (projection and its copies, product scratch,
// build scratch) are released where they are sized for a far larger matrix — everything derived from the old one is void anyway (api.cpp)
void isle_trim_derived(isle_ctx* c, uint64_t D, uint64_t nnz);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (context = device, kernel, size): api.cpp
int isle_max_lds(isle_ctx* c, const void* fn, int bytes);
bool isle_scratch_ok(isle_ctx* c, size_t have_elems, double bytes);
void isle_refresh_knobs(isle_ctx* c);
int isle_enter(isle_ctx* c);  // entry of a C-ABI call: the context's device becomes current, the environment switches are read
// ISLE_HOST_TRACE=1: host wall time since the previous mark, to stderr (marks that follow within 0.2 ms stay silent).  Finds GPU-idle
// stretches that are host work, which no kernel profile shows.
void isle_host_mark(const char* what);

#define HIPCHK(ctx, call)                                                                   \
  do {                                                                                      \
    hipError_t e__ = (call);                                                                \
    if ((ctx) && (ctx)->comm_dead.load(std::memory_order_relaxed))                          \
      return isle_fail((ctx), ISLE_E_COMM, "%s:%d: a collective did not complete within %.0f s (ISLE_COMM_TIMEOUT_S): the communicator was aborted", __FILE__, __LINE__, (ctx)->wd_timeout_s); \
    if (e__ != hipSuccess)                                                                  \
      return isle_fail((ctx), ISLE_E_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
    if (isle_launch_refused().file) {                                                       \
      const IsleLaunchRefused r__ = isle_launch_refused();                                  \
      isle_launch_refused() = IsleLaunchRefused();                                          \
      return isle_fail((ctx), ISLE_E_ARG, "%s:%d: kernel launch of %llu threads refused (2^32 or more do not run)", r__.file, r__.line, r__.threads); \
    }                                                                                       \
  } while (0)

#define ISLECHK(call)            \
  do {                           \
    int rc__ = (call);           \
    if (rc__ != 0) return rc__;  \
  } while (0)

// RAII timing scope: records an event pair on the stream when timing is enabled.
struct TimeScope {
  isle_ctx* c;
  isle_event_pair ep;
  bool on;
  bool marker = false;  // a roctx range is open for this scope (ISLE_ROCTX)
  TimeScope(isle_ctx* c_, int fam);
  ~TimeScope();
};

static inline uint64_t isle_scan_scratch(uint64_t n) { return (n + 4095) / 4096 + 1; }  // = isle_scan::scan_scratch_elems
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---------------- kernel launchers (each defined in one .hip file) -------------------------
// spmm.hip
int k_pack_rm(isle_ctx* c, const float* Xcm, uint64_t V, int b, int BP, float* Xrm);
int k_unpack_cm(isle_ctx* c, const float* Zrm, uint64_t V, int b, int BP, float* Zcm);
int k_gram_pass1(isle_ctx* c, int BP);   // Yrm = B^T Xrm
int k_gram_pass2(isle_ctx* c, int BP);   // Zrm = B Yrm
int k_band_build(isle_ctx* c);
int k_band_build_chunked(isle_ctx* c);   // chunk-major cells for the gather path (spmm.hip)
// gram_lds.hip
// collectives on c->stream (api.cpp): RCCL, or the host-staged rehearsal transport; no-ops without a communicator
enum { ISLE_DT_F32 = 0, ISLE_DT_F64 = 1, ISLE_DT_I32 = 2, ISLE_DT_U32 = 3, ISLE_DT_U64 = 4 };
int isle_allreduce(isle_ctx* c, void* buf, size_t count, int dtype, bool max_op = false);  // in place
int isle_allgather(isle_ctx* c, const void* send, void* recv, size_t count_per_rank, int dtype);  // recv: world * count_per_rank
int k_dots_assign(isle_ctx* c, int k, int ldk, const float* cn, const float* dn, uint32_t* assign, float* ub, float* lb, int G);
int k_dots_assign_cm(isle_ctx* c, const float* dotsT, int k, int G, const float* cn, const float* dn, const float* cn_max_dev, uint32_t* assign, float* ub,
                     float* lb);  // the same from column-major dot products, Yinyang bounds
int k_gl_detect(isle_ctx* c);            // sets c->gl_mode for the current B (no-op once decided)
int k_centers_counts(isle_ctx* c, const uint32_t* assign, int k, int ldk, float* Crm, bool fresh);  // fresh: needs c->members grouped by `assign`
int k_gl_build(isle_ctx* c);
int k_gl_wide(isle_ctx* c, const float* Mrm, int k, int ld, float* Out, float* norms = nullptr /*also the rows' squared norms (selects the grouped form)*/,
              void* A2pos = nullptr /*grouped form only: also the split copy of Out by POSITION (k_gemm_split_a_bytes(D, k) bytes)*/, bool* a2_done = nullptr);
int k_gl_thin(isle_ctx* c, const float* Wcm, int nc, int ld, float* Out, bool by_position = false);  // Out (D x ld) = B^T W, W V x nc col-major, nc <= 32  // Out (D x ld) = B^T M, LDS-banded form only
int k_gl_apply_cm(isle_ctx* c, const float* Xcm, int b, int BP, float* Zcm);  // Zcm (V x b col-major) = B (B^T Xcm), b columns in a panel of BP in {4, 8, 12}
// ingest.hip
int k_sort_pairs_u64(isle_ctx* c, uint64_t* key_a, uint32_t* val_a, uint64_t* key_b, uint32_t* val_b, uint64_t n, int key_bits, bool* in_a);
int k_frobenius(isle_ctx* c, double* out_host);
int k_spmm_wide_project(isle_ctx* c, const float* Mrm, int k, int ldk, float* P, float* norms, void* A2pos = nullptr, bool* a2_done = nullptr);
int k_ensure_pt(isle_ctx* c);  // the coordinate-major f32 copy of the projection, made from P when a route asks for it (dense.hip)
int k_spmm_wide_assign(isle_ctx* c, const float* Mrm, int k, int ldk, const float* cn, const float* dn, uint32_t* assign,
                       const uint32_t* perm /*nullable: slot -> doc*/, const uint32_t* nslots = nullptr /*device slot count*/,
                       float* ub = nullptr, float* lb = nullptr, int G = 0 /*> 0: lb holds G Yinyang group bounds per document*/);
// The centres of a Yinyang iteration that moved far (at most ten: one thin pass): they are left out of their groups' movements and bounded by
// their exact new distances instead (yy2_filter_tighten_k)
struct YyMovers {
  int n = 0, ld = 0;   // ld = 4 ceil(n / 4): row stride of the D x n dot products
  uint32_t id[10] = {};
};
// Which centres share a Yinyang group.  Null pointers: group g = centres 8 g .. 8 g + 7 (the identity).  Otherwise the groups are made
// of SLOTS: slot s = 8 g + t holds centre id_of_slot[s] (s < k; the slots behind are padding), slot_of_id is the inverse.  The kernels
// of the by-group iteration index tables and bounds by slot and report centres by id; ties between equal distances go to the smaller ID
// whatever the slots' order, as the reference's isamin does (src/sparseMatrix.cpp:1553-1572).  ONE exception, stated in INTEGRATION.md: the
// FIRST assignment (the product through the projection, columns in slot order: tile_epilogue / yy_first_combine_k) keeps the earlier COLUMN
// on a bit-exact tie, i.e. the smaller slot = the smaller squared norm; identical centres keep their id order (the slot order is a stable
// sort by norm), so only two DIFFERENT centres at bit-equal distances from a document can fall the other way, in that one iteration.
struct YyMap {
  const uint32_t* id_of_slot = nullptr;
  const uint32_t* slot_of_id = nullptr;
#if defined(__HIPCC__)
  __device__ inline uint32_t id(uint32_t s) const { return id_of_slot ? id_of_slot[s] : s; }
  __device__ inline uint32_t slot(uint32_t i) const { return slot_of_id ? slot_of_id[i] : i; }
#endif
};
int k_yy_filter_tighten(isle_ctx* c, const uint32_t* order, const uint32_t* assign, float* ub, float* glb, int G, const float* delta_dev,
                        const float* gmax_dev, uint32_t* active, uint32_t* nactive, const float* Cg, int k, int ld, const float* cn, const float* dn,
                        const float* cn_max, const YyMovers& mv, const float* Crm, const YyMap& map = YyMap(), const float* cn_by_id = nullptr);
// the maps of a regrouping (host arrays of 8 G and k entries) on the device; cn_slot[s] = cn[id_of_slot[s]]; rows of a small matrix by slot;
// labels written by slot turned into ids
int k_yy_map_upload(isle_ctx* c, const uint32_t* id_of_slot_host, const uint32_t* slot_of_id_host, int k, int G, YyMap* map);
int k_yy_gather_by_slot(isle_ctx* c, const YyMap& map, int k, int G, const float* cn, float* cn_slot);
int k_yy_rows_by_slot(isle_ctx* c, const YyMap& map, int k, const float* in, int ld, float* out);
int k_yy_labels_to_ids(isle_ctx* c, const YyMap& map, uint32_t* assign, uint64_t D);
int k_yy_filter(isle_ctx* c, const uint32_t* order, const uint32_t* assign, float* ub, float* glb, int G, const float* delta_dev, const float* gmax_dev,
                uint32_t* active, uint32_t* nactive);
int k_yy_dbg_margins(isle_ctx* c, const uint32_t* active, const uint32_t* nactive, const uint32_t* assign, const float* ub, const float* glb, int G,
                     const YyMap& map, unsigned long long* hist_host);
int k_yy_pack_groups(isle_ctx* c, const float* Crm, int ld, int G, const YyMap& map = YyMap());  // c->yy_cg = the centres group-major (V x 8 floats per group)
int k_yy_scan(isle_ctx* c, const float* Crm, const float* Cg /*nullable*/, int k, int ld, int G, const float* cn, const float* dn, const float* cn_max_dev,
              const uint32_t* active, const uint32_t* nactive, uint32_t* assign, float* ub, float* glb, unsigned long long* dbg = nullptr,
              const YyMap& map = YyMap());
int k_yy2_assign(isle_ctx* c, const float* Cg, int k, int ld, int G, const float* cn, const float* dn, const float* cn_max_dev, const uint32_t* active,
                 const uint32_t* nactive, uint32_t* assign, float* ub, float* glb, bool* done, unsigned long long* pairs_out = nullptr, bool pre_tightened = false,
                 const YyMap& map = YyMap());
struct HamTop {  // largest and second largest centre movement of an iteration (Hamerly's bound update), device resident
  uint32_t amax;
  float d1, d2, pad;
};
int k_ham_delta(isle_ctx* c, float* delta_dev /*in: squared movements, out: rounded-up movements*/, int k, HamTop* top_dev);
int k_hamerly_filter(isle_ctx* c, const uint32_t* order, const uint32_t* assign, float* ub, float* lb, const float* delta_dev, const HamTop* top_dev,
                     uint32_t* active, uint32_t* nactive, int fam = ISLE_T_SPARSE_ASSIGN);
int k_member_lists(isle_ctx* c, const uint32_t* assign, uint64_t D, int k, const int* counts_dev, int* max_out, std::vector<int>* counts_host = nullptr);
int k_member_lists_dev(isle_ctx* c, const uint32_t* assign, uint64_t D, int k, const int* counts_dev);  // no host round trip
int k_yy_delta(isle_ctx* c, float* delta_dev, int k, int G, int group, float* gmax_dev, const uint32_t* id_of_slot = nullptr);
int k_max_f32(isle_ctx* c, const float* v, int n, float* out_dev);
int k_csc_validate(isle_ctx* c, unsigned long long* err_host2);  // the uploaded CSC arrays on the device: [0] first bad column + 1 (0 = fine), [1] what (spmm.hip)
int k_doc_norms(isle_ctx* c, float* dn);
int k_centers_from_rows(isle_ctx* c, const uint32_t* assign, int k, int ldk, float* Crm, bool first_of_run = true);

// threshold.hip
int k_th_stats(isle_ctx* c, uint64_t* tokens_nz_dev);
int k_th_round_hist(isle_ctx* c, float avg, uint32_t maxv);
int k_th_zetas(isle_ctx* c, uint32_t maxv, uint64_t count_gr, uint64_t count_eq);
int k_th_count(isle_ctx* c, bool with_weights);
int k_th_drop(isle_ctx* c, const uint8_t* drop_dev);
int k_th_scans(isle_ctx* c);
int k_th_emit(isle_ctx* c, uint64_t doc_base);

// ingest.hip
int k_ingest_tdf(isle_ctx* c, const unsigned char* text_dev, uint64_t n, uint64_t V, uint64_t D, uint64_t* entries_read, uint64_t* err_out);

// infer.hip
int k_infer(isle_ctx* c, uint64_t V, int k, const float* model_by_word, uint64_t D, uint64_t nnz, const float* counts, const uint32_t* rows,
            const int64_t* offs, int iters, float Lfguess, float avg_doc_sz, float* weights, int32_t* top_topic, float* top_weight, float* llh,
            uint64_t* nconverged);

// post.hip
int k_post_normalize(isle_ctx* c, float avg);
int k_post_cluster_of(isle_ctx* c, const uint32_t* assign_dev, bool identity);
int k_post_catch_thresholds(isle_ctx* c, uint32_t k, uint32_t r, const int* sizes_dev);
int k_post_find_catchwords(isle_ctx* c, uint32_t k, double rho, uint64_t* ncatch_host);
int k_post_thr_colmajor(isle_ctx* c, uint32_t k, float* out_dev);
int k_post_doc_topic_sums(isle_ctx* c, uint32_t k, uint64_t* n_out);
int k_post_model_thresholds(isle_ctx* c, uint32_t k, uint32_t rank);
int k_post_model(isle_ctx* c, uint32_t k);
int k_post_edge(isle_ctx* c, const int64_t* pairs_dev, int n, float a, float b, float* edge_dev);

// dense.hip
int k_vtf(isle_ctx* c, const float* Vb, uint64_t n, int m, const float* F, int b, float* coef /*m x b col-major dev*/, uint64_t ld = 0);
int k_update(isle_ctx* c, float* F, uint64_t n, int b, const float* Vb, int m, const float* coef, uint64_t ld = 0);
int k_panel_qr(isle_ctx* c, float* F, uint64_t n, int w, float* Qdst, float* R_host /*w*w*/, int* rank_out);
int k_panel_qr_kernels(isle_ctx* c, float* F, uint64_t n, int w, float* Qdst, int* meta_dev /*2 + 32*/, float* Rout_dev /*w*w*/);
int k_randu(isle_ctx* c, float* F, uint64_t count, uint64_t seed);
int k_gemm_nn(isle_ctx* c, const float* A, uint64_t M, int K, const float* B, int ldb, int N, float* C, int family = ISLE_T_ROTATE);  // col-major, lda = ldc = M
// the same product for the D x k x k dot products of the assignment steps: bf16 matrix cores, operands split in three bf16 terms (gemm_bf16x3.h)
int k_gemm_nn_assign(isle_ctx* c, const float* A, uint64_t M, int K, const float* B, int ldb, int N, float* C, int family);
bool k_gemm_assign_fused_ok(isle_ctx* c, uint64_t M, int K, int N);
int k_gemm_assign_yy(isle_ctx* c, const float* A, const float* Arm, int lda_rm, const float* an, uint64_t M, int K, const float* B, int ldb, int N, int G,
                     const float* cn, const float* dn, const float* cn_max, uint32_t* assign, float* ub, float* lb, int family, const void* A2 = nullptr,
                     const uint32_t* map2 = nullptr /*row of A2 -> document when A2 lies by position; A may be null then (made on demand)*/);
int k_gemm_assign_tiles(isle_ctx* c, const float* A, const float* Arm, int lda_rm, uint64_t M, int K, const float* B, int ldb, int N, int TL, const float* cn,
                        const float* pn, const float* cmax, uint32_t* assign, float* ub, float* tlb, int family, const uint32_t* map0 = nullptr,
                        const void* A2 = nullptr, const uint32_t* map2 = nullptr);
// A2 = the two bf16 terms of a coordinate-major M x K operand in the layout gemm_bf16x2_dma_k stages by LDS-DMA (gemm_bf16x3.h); bytes it needs
int k_gemm_split_a(isle_ctx* c, const float* A, uint64_t M, int K, void* A2);
size_t k_gemm_split_a_bytes(uint64_t M, int K);
int k_compact_rows(isle_ctx* c, const float* P, const float* pn, int ldk, const uint32_t* active, uint32_t n, float* Pa, float* pna);
int k_transpose(isle_ctx* c, const float* in, uint64_t rows, uint64_t cols, uint64_t ld_in, float* out, uint64_t ld_out);  // out[c*ld_out + r]... see impl
int k_jacobi_eig(isle_ctx* c, const float* S_host, int n, float* evals_host, float* vecs_dev /*n x n col-major*/);
int k_tridiag_eig(isle_ctx* c, const float* S_host, int n, float* evals_host, float* vecs_dev, int nvec);  // evd_tridiag.hip; 1 = use another solver; evals_host[nvec..n) = 0
int k_eig_small(isle_ctx* c, const float* S_host, int n, float* evals_host, float* vecs_dev /*n x nvec col-major*/, int nvec);
int k_colnorms_rm(isle_ctx* c, const float* Mrm, uint64_t rows, int k, int ldk, float* out, const float* Sub = nullptr);
int k_scale_centers(isle_ctx* c, float* Crm, uint64_t rows, int k, int ldk, const int* counts);

// kmeans.hip
// s_old = seeds before the nc new ones; track: also keep the nearest seed and the tile minima where the route allows (c->kmpp_track)
int k_kmpp_update(isle_ctx* c, const float* P, const float* pn, uint64_t D, int k, int ldk, const float* newC, int nc, float* min_dist, int s_old = -1,
                  bool track = false);
int k_kmpp_to_tiles(isle_ctx* c, uint64_t D, int k, const float* pn, const float* cn, const float* best, uint32_t* assign, float* ub, int TL);
int k_gl_panel_width(const isle_ctx* c);  // columns per pass of the k-wide / thin products through the pass-1 stream (gram_lds.hip)
int k_scan_f2d(isle_ctx* c, const float* in, uint64_t n, double* cum /*n+1*/);
int k_search(isle_ctx* c, const double* cum, uint64_t n, const double* dice_dev, int nd, uint64_t* out_dev);
int k_search_args(isle_ctx* c, const double* cum, uint64_t n, const double* dice_host, int nd /*<= 16*/, uint64_t* out_dev);
int k_fetch_rows(isle_ctx* c, const float* P, int ldk, const uint64_t* local_ids /*~0: not on this rank*/, int n, float* dst);
int k_pack2(isle_ctx* c, const double* a, const float* b /*nullable*/, double* out2);
int k_search_frac(isle_ctx* c, const double* cum, uint64_t n, const float* last /*nullable*/, const double* frac_host, int nd /*<= 40*/,
                  uint64_t* out_dev /*42 x 8 bytes: positions, then {cum[n], last[0]} as doubles*/);
int k_proj_assign(isle_ctx* c, const float* P, const float* pn, uint64_t D, int k, int ldk, const float* C, const float* cn, uint32_t* assign,
                  float* ub = nullptr, float* lb = nullptr);
int k_proj_assign_active(isle_ctx* c, const float* P, const float* pn, int k, int ldk, const float* C, const float* cn,
                         const uint32_t* active, uint32_t n, float* Pa, float* pna, uint32_t* assign, float* ub, float* lb);
bool k_proj_full_by_gemm(isle_ctx* c, uint64_t D, int k);  // the full tile-bound pass goes through the library GEMM (kmeans.hip)
int k_proj_assign_tiles(isle_ctx* c, const float* P, const float* pn, uint64_t D, int k, int ldk, const float* C, const float* cn, uint32_t* assign,
                        float* ub, float* tlb, int TL, const uint32_t* active, uint32_t n, const uint32_t* need, float* Pa, float* pna);
int k_pt_filter(isle_ctx* c, const uint32_t* order, const uint32_t* assign, float* ub, float* tlb, int T, int TL, const float* delta_dev,
                const float* tmove_dev, uint32_t* need, uint32_t* active, uint32_t* nactive, const YyMovers& mv, const float* mdots, const float* cn, const float* pn);
int k_pt_tighten(isle_ctx* c, const float* P, const float* pn, int ldk, const float* C, const float* cn, const uint32_t* assign, const uint32_t* cand,
                 const uint32_t* ncand, float* ub, const float* tlb, int T, int TL, uint32_t* need, uint32_t* active, uint32_t* nactive);
int k_rownorms_diff(isle_ctx* c, const float* A, const float* B, int rows, int k, int ldk, float* out);
int k_rownorms(isle_ctx* c, const float* M, int rows, int k, int ldk, float* out);
int k_proj_accumulate(isle_ctx* c, const float* P, uint64_t D, int k, int ldk, const uint32_t* assign, float* Csum, int* counts);
int k_proj_accumulate_delta(isle_ctx* c, const float* P, uint64_t D, int k, int ldk, const uint32_t* assign, uint32_t* counted, float* Csum,
                            const int* counts, bool* done);
int k_proj_finalize(isle_ctx* c, const float* Csum, const int* counts, int k, int ldk, float* C);
int k_count_sizes(isle_ctx* c, const uint32_t* assign, uint64_t D, int k, int* counts);
int k_compare_u32(isle_ctx* c, const uint32_t* a, const uint32_t* b, uint64_t n, int* flag_dev);
int k_fill_f32(isle_ctx* c, float* p, uint64_t n, float v);
