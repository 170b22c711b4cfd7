// isle_amd/csrc/api.cpp — host orchestration behind the C ABI of include/isle_hip.h.
//
// The control flow mirrors the reference exactly where it defines the result:
//   BlockKs::init / expand / truncate / compute   block-ks/restarted_block_ks.h:62-321
//   kmeanspp_on_projected_space                   src/sparseMatrix.cpp:2133-2209
//   run_lloyds_on_projected_space / run_lloyds    src/sparseMatrix.cpp:2016-2072 / :1690-1746 (stop rule)
// All arithmetic on V- or D-sized data runs in the HIP kernels of spmm.hip / dense.hip / kmeans.hip;
// the host keeps only the small projected matrix H (<= (2k+b) x 2k floats) and scalars.
// There is no CPU fallback.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <thread>

#include <dlfcn.h>

#include "api_internal.h"

// ------------------------------------------------------------------------------------------
// errors, timing
// ------------------------------------------------------------------------------------------
int isle_fail(isle_ctx* c, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  return code;
}

// every environment switch of the library (common.h IsleKnob); DESIGN.md tables them with the measurements behind the defaults
const IsleKnobInfo isle_knob_table[KN_COUNT] = {
    {"ISLE_GRAM_LDS", "form", "0: Gram apply and k-wide products by the row-gather kernels (any CSC matrix) instead of the LDS-banded form (row-constant B)"},
    {"ISLE_GL_G1", "tuning", "4..8: output items per lane in pass 1 of the LDS-banded form (default: makespan model, gram_lds.hip)"},
    {"ISLE_GL_G2", "tuning", "4..8: output items per lane in pass 2 (default 4, 6 beyond 1024 document bands)"},
    {"ISLE_GL_PLACE", "tuning", "0: a lane's entries stay packed at the front of its slots in ascending order instead of the bank-aware placement (gl_place_k)"},
    {"ISLE_GL_FILL_BUCKETS", "tuning", "0: the pass-2 id stream is filled by one scatter over a band's whole region (gl_hist_fill_k) instead of by buckets of word positions (same stream)"},
    {"ISLE_GL_ROUNDS", "tuning", "0: pass 1 keeps ceil(slices / items per lane) waves in workgroups strided over the length order instead of filling whole rounds of the CUs with workgroups of adjacent waves"},
    {"ISLE_GL_COLUMNS", "tuning", "0: pass 2 chunks its document bands per word block instead of walking band columns shared through one XCD's L2"},
    {"ISLE_GL_PANEL", "tuning", "8 | 10: columns per pass of the k-wide / thin products (default 10, 8 at 8 items per lane in pass 1)"},
    {"ISLE_GL_WIDE_GROUPED", "tuning", "0: every panel pass of the projection writes its 40 bytes straight into the documents' rows of P (default on a large shard: the panels' rows go whole and in position order into a scratch, sixteen at a time, and a second kernel assembles 640-byte pieces of the document-major rows and the rows' squared norms)"},
    {"ISLE_WIDE_GATHER", "form", "k-wide products (projection, first word-space assignment) by the row-gather kernel"},
    {"ISLE_WIDE_LDS", "form", "k-wide products through the LDS-banded pass-1 stream whatever the vocabulary size"},
    {"ISLE_KS_ROWSHARD", "form", "0: every rank orthogonalises the whole Krylov block (default with several ranks: row slices, all-reduced coefficients, all-gathered block)"},
    {"ISLE_KS_SYNC", "form", "expand loop without the speculative pipeline (one host synchronisation per step)"},
    {"ISLE_KS_ORTHO_PASSES", "form", "3: the reference's three Gram-Schmidt passes per Krylov step instead of two"},
    {"ISLE_UPDATE_MFMA", "form", "0: the orthogonalisation's update F -= V H by FMA chains with the coefficients read from LDS (update_k) instead of on the matrix cores (same sums in another order)"},
    {"ISLE_EVD_JACOBI", "form", "small symmetric EVD by block Jacobi instead of tridiagonalisation"},
    {"ISLE_TD_CHAIN", "form", "tridiagonalisation as a launch chain instead of the persistent kernel"},
    {"ISLE_TD_BAR", "tuning", "flat | hier: the persistent tridiagonalisation crosses its grid barriers on one counter (gb_barrier) or through the hierarchical barrier (gbh_barrier) instead of the sharded counters polled together (gbs_barrier, gridbar.h; round 6: 36.4 / 37.0 / 40.7 ms per EVD at n = 2000 sharded / hier / flat); same bits"},
    {"ISLE_TD_BACK", "form", "seq: the eigenvectors' back-transformation applies the reflectors one by one (td_back_k) instead of by blocks of four in compact WY form"},
    {"ISLE_EVD_SPLIT", "form", "1: with several ranks rank r computes the eigenvectors [r kc, (r + 1) kc) of the small EVD and the columns are all-gathered (same bits on every rank; default since round 6: every rank computes all of them — the stage's kernels are chains of n steps whatever the number of vectors, the split saves no time)"},
    {"ISLE_KMPP_HOST_DICE", "form", "k-means++ dice scaled and searched through the host round trip (the multi-rank form) on one rank too"},
    {"ISLE_KMPP_SPARSE", "form", "0 / 1: k-means++ rounds on the projection / through thin products of B (default: by cost)"},
    {"ISLE_KMPP_TRACK", "form", "0: Lloyd in span(U) always starts with a full assignment pass (default at k > 224 on a large shard: it starts from the nearest seeds and tile minima the k-means++ rounds kept)"},
    {"ISLE_NO_HAMERLY", "form", "both Lloyd loops without distance bounds (every document re-examined every iteration)"},
    {"ISLE_KMEANS_BOUNDS", "form", "hamerly | none: bounds of Lloyd on B (default Yinyang group bounds)"},
    {"ISLE_PROJ_BOUNDS", "form", "hamerly: single lower bound in the projected Lloyd loop at k > 224 instead of tile bounds"},
    {"ISLE_PROJ_FULL", "form", "gemm | fused: full passes of the projected Lloyd loop as GEMM + epilogue or as the fused register kernel"},
    {"ISLE_FIRST_ASSIGN", "form", "sparse | projection: first assignment of Lloyd on B through the sparse product or through the projection"},
    {"ISLE_GEMM_BF16X3", "form", "0: the D x k x k dot products of the assignment steps on the f32 matrix cores (gemm_f32.h) instead of the bf16 ones with operands split in three terms (gemm_bf16x3.h)"},
    {"ISLE_GEMM_EPILOGUE", "form", "0: the D x k x k products of the assignment steps are written to memory and read by dots_assign_cm_k / proj_dots_tiles_k instead of the epilogues inside the product (same bits)"},
    {"ISLE_GEMM_TERMS", "form", "3: the assignment products run once with three bf16 terms per operand (default: two terms first, the rows whose arg-min that leaves open again with three; same assignment)"},
    {"ISLE_GEMM_DMA", "form", "0: the two-term pass of the assignment products splits A on the fly and stages it through registers (gemm_bf16x3_k) instead of reading the projection's pre-split copy by LDS-DMA through a ring of stages (gemm_bf16x2_dma_k; same products in the same order: same bits)"},
    {"ISLE_YY_MODE", "form", "doc | docg | group: Yinyang iteration by document (row-major / group-major centres) or ordered by group (default: group at k >= 256)"},
    {"ISLE_YY_FUSED", "form", "0: the by-group Yinyang iteration lowers the bounds (yy_filter_k) and tightens the active documents (yy2_tighten_k) in two launches instead of one (same bits)"},
    {"ISLE_YY_MOVERS", "form", "0: every centre's movement lowers its Yinyang group's bound (default: up to ten centres that moved far beyond the rest are bounded by their exact new distances instead)"},
    {"ISLE_YY_REGROUP", "form", "0: a Yinyang group is eight consecutive centres (default at k >= 256 behind the product's first assignment: the centres in the order of their squared norms, so that the few centres every far-off document is near share groups; same partitions, a third of the (document, group) pairs)"},
    {"ISLE_YY_ORDER", "form", "doc | member: the Yinyang filter visits the documents in their own order or in the member lists' (default: doc for the fused by-group launch, member otherwise; same bits)"},
    {"ISLE_PT_SORT", "form", "0: the active documents of a projected Lloyd iteration keep the order of the member lists (default: ordered by the set of tiles they have to re-examine, so that a workgroup's documents ask for the same tiles; same partitions)"},
    {"ISLE_PROJ_ACTIVE", "form", "tiles: the active documents of a projected Lloyd iteration re-examine only the tiles their bounds name (register kernel, f32 matrix cores); default gemm: all centres through the assignment product on their gathered rows wherever that product is taken (every tile bound refreshed, the arithmetic of the full passes)"},
    {"ISLE_PROJ_SUMS", "form", "fresh: the centroid sums of Lloyd in span(U) are formed from all member rows every iteration (default: kept up to date by the documents that changed centre; both bitwise reproducible, the two differ in rounding)"},
    {"ISLE_CENTERS_FRESH", "form", "centroid counts recounted from the member lists every iteration instead of updated by the documents that moved"},
    {"ISLE_INFER_CAP_ROWS", "form", "inference: stage at most this many model rows per document in LDS (default 0: rows read through L2)"},
    {"ISLE_CHUNK_COLS", "tuning", "gather form: rows per chunk of the chunked-CSR copy"},
    {"ISLE_COMM_TIMEOUT_S", "tuning", "seconds without a completed collective, while collectives are pending, after which the watchdog aborts the RCCL communicator and the call returns ISLE_E_COMM (default 300; 0: no watchdog)"},
    {"ISLE_COMM_SELFTEST", "diagnostic", "0 / 1: behind the communicator's creation every rank runs one all-reduce (sum, max) and one all-gather of every (datatype, size class) the step issues on patterns with known results, and logs its rank, device and PCI bus id (default: on with an RCCL communicator of more than one rank, off for the host-staged test transport and the forced 1-rank communicator)"},
    {"ISLE_FORCE_COMM", "test hook", "create a 1-rank RCCL communicator so that every collective call site runs on one GPU"},
    {"ISLE_TEST_STALL_MS", "test hook", "a kernel that spins for this many milliseconds is queued ahead of every RCCL collective (the watchdog's test)"},
    {"ISLE_ROCTX", "diagnostic", "1: every kernel family's launches are wrapped in a roctx range (isle:gram_pass1, isle:ortho, ...) for rocprofv3 --marker-trace; the roctx library is opened at run time"},
    {"ISLE_HOST_TRACE", "diagnostic", "print host wall time between marks of the control loops"},
    {"ISLE_DEBUG_HAMERLY", "diagnostic", "print active documents / group scans per Lloyd iteration"},
    {"ISLE_DEBUG_EVD", "diagnostic", "print sweeps / orthogonality defect of the small EVD"},
    {"ISLE_GL_VERBOSE", "diagnostic", "print the geometry of the LDS-banded operator build"},
    {"ISLE_TD_FORCE_BAIL_RANK", "test hook", "this rank behaves as if the grid barrier of its persistent EVD had timed out"},
    {"ISLE_GL_TEST_CUS", "test hook", "the operator build lays pass 1 out as for a device of this many CUs (several rounds of workgroups on a small matrix)"},
    {"ISLE_GL_ABLATE_SKIP", "test hook", "m >= 2: TIMING EXPERIMENT, results wrong by construction — every m-th (band, group) of every wave of the LDS-banded passes is not walked and its ids are not read (prices a form with 1/m fewer padded slots; round 6)"},
};
void isle_refresh_knobs(isle_ctx* c) {
  for (int i = 0; i < KN_COUNT; ++i) {
    const char* e = getenv(isle_knob_table[i].name);
    c->knob_set[i] = e != nullptr;
    if (e) c->knob_val[i] = e;
    else c->knob_val[i].clear();
  }
}
int isle_enter(isle_ctx* c) {
  HIPCHK(c, hipSetDevice(c->device));
  isle_refresh_knobs(c);
  return 0;
}

// May the optional D x k scratch of the GEMM routes be taken?  A function of the problem's size and the device's TOTAL memory only —
// never of what happens to be free — so that the route, and with it every rounding of the first assignment, is the same run after run
// and rank after rank: always up to 8 GB; beyond (all of config 3 on one GPU: 40 GB), up to a fifth of the device (57 GB on an MI355X).
bool isle_scratch_ok(isle_ctx* c, size_t have_elems, double bytes) {
  (void)have_elems;
  if (bytes <= 8e9) return true;
  if (c->total_mem == 0) {
    size_t fr = 0, tot = 0;
    if (hipSetDevice(c->device) != hipSuccess || hipMemGetInfo(&fr, &tot) != hipSuccess) return false;
    c->total_mem = tot;
  }
  return bytes <= 0.2 * (double)c->total_mem;
}

int isle_max_lds(isle_ctx* c, const void* fn, int bytes) {
  for (auto& e : c->lds_attr)
    if (e.first == fn) {
      if (e.second >= bytes) return 0;
      HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
      e.second = bytes;
      return 0;
    }
  HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  c->lds_attr.emplace_back(fn, bytes);
  return 0;
}

void isle_host_mark(const char* what) {
  static const bool on = getenv("ISLE_HOST_TRACE") != nullptr;
  if (!on) return;
  static auto last = std::chrono::steady_clock::now();
  const auto now = std::chrono::steady_clock::now();
  const double ms = std::chrono::duration<double, std::milli>(now - last).count();
  last = now;
  if (ms >= 0.2) fprintf(stderr, "[host] %8.3f ms before %s\n", ms, what);
}

extern "C" int isle_hip_switch_info(int index, const char** name, const char** kind, const char** what) {
  if (index >= 0 && index < KN_COUNT) {
    if (name) *name = isle_knob_table[index].name;
    if (kind) *kind = isle_knob_table[index].kind;
    if (what) *what = isle_knob_table[index].what;
  }
  return KN_COUNT;
}

// rocprofv3 markers per kernel family (SURVEY 5.1 "build hook"; the reference's phase timers are include/timer.h:72-85).  ISLE_ROCTX=1:
// every TimeScope also pushes / pops a roctx range named after its family (host-side API ranges around the launches: `rocprofv3
// --marker-trace --kernel-trace` shows which kernels belong to which family).  The roctx library is opened at run time — libisle_hip.so
// itself links RCCL and the HIP runtime only — and a missing library just leaves the markers out.
static const char* const kFamilyName[ISLE_T_COUNT] = {"isle:gram_pass1", "isle:gram_pass2", "isle:ortho", "isle:panel_qr", "isle:small_evd", "isle:rotate",
                                                       "isle:project", "isle:kmeanspp", "isle:lloyd_projected", "isle:lloyd_sparse_assign",
                                                       "isle:lloyd_sparse_update", "isle:operator_build", "isle:collectives", "isle:threshold",
                                                       "isle:post", "isle:ingest", "isle:infer", "isle:lift"};
typedef int (*roctx_push_fn)(const char*);
typedef int (*roctx_pop_fn)(void);
static roctx_push_fn g_roctx_push = nullptr;
static roctx_pop_fn g_roctx_pop = nullptr;
static void roctx_load_once() {
  static bool tried = false;
  if (tried) return;
  tried = true;
  for (const char* lib : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
    if (void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL)) {
      g_roctx_push = reinterpret_cast<roctx_push_fn>(dlsym(h, "roctxRangePushA"));
      g_roctx_pop = reinterpret_cast<roctx_pop_fn>(dlsym(h, "roctxRangePop"));
      if (g_roctx_push && g_roctx_pop) return;
      g_roctx_push = nullptr;
      g_roctx_pop = nullptr;
    }
  }
  fprintf(stderr, "[isle_hip] ISLE_ROCTX: no roctx library found (librocprofiler-sdk-roctx.so / libroctx64.so): no markers\n");
}

TimeScope::TimeScope(isle_ctx* c_, int fam) : c(c_), on(false) {
  if (fam >= 0 && fam < ISLE_T_COUNT && c->knob_on(KN_ROCTX) && !c->knob_zero(KN_ROCTX) && !c->roctx_open) {
    roctx_load_once();
    if (g_roctx_push) {
      g_roctx_push(kFamilyName[fam]);
      c->roctx_open = true;
      marker = true;
    }
  }
  if (fam < 0 || !c->timing || !((c->timing_mask >> fam) & 1u)) return;
  if (c->ts_open) return;  // inside another scope (a launcher called by a launcher): the outer one times both, nothing is counted twice
  if (!c->ev_free.empty()) {
    ep = c->ev_free.back();
    c->ev_free.pop_back();
  } else {
    if (hipEventCreate(&ep.a) != hipSuccess || hipEventCreate(&ep.b) != hipSuccess) return;
  }
  ep.fam = fam;
  on = (hipEventRecord(ep.a, c->stream) == hipSuccess);
  if (on) c->ts_open = true;
}
TimeScope::~TimeScope() {
  if (marker) {
    g_roctx_pop();
    c->roctx_open = false;
  }
  if (!on) return;
  c->ts_open = false;
  (void)hipEventRecord(ep.b, c->stream);
  c->ev_used.push_back(ep);
}

int drain_events(isle_ctx* c) {
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (auto& e : c->ev_used) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
      c->t_ms[e.fam] += ms;
      c->t_n[e.fam] += 1;
    }
    c->ev_free.push_back(e);
  }
  c->ev_used.clear();
  return 0;
}

#define NCCLCHK(ctx, call)                                                                              \
  do {                                                                                                  \
    ncclResult_t r__ = (call);                                                                          \
    if (r__ != ncclSuccess)                                                                             \
      return isle_fail((ctx), ISLE_E_COMM, "%s:%d %s -> %s", __FILE__, __LINE__, #call, ncclGetErrorString(r__)); \
  } while (0)

static const ncclDataType_t kNcclType[5] = {ncclFloat, ncclDouble, ncclInt, ncclUint32, ncclUint64};
static const size_t kDtSize[5] = {4, 8, 4, 4, 8};

// host-staged exchange: device -> host, the caller's function (gloo in tests/), host -> device
static int host_exchange(isle_ctx* c, int kind, void* dev, size_t count_per_rank, int dtype) {
  const size_t total = (kind == ISLE_XCHG_ALLGATHER ? (size_t)c->world : 1) * count_per_rank * kDtSize[dtype];
  std::vector<char> h(total);
  HIPCHK(c, hipMemcpyAsync(h.data(), dev, total, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const int rc = c->host_xchg(c->host_xchg_user, kind, h.data(), (uint64_t)count_per_rank, dtype);
  if (rc != 0) return isle_fail(c, ISLE_E_COMM, "host exchange function returned %d", rc);
  HIPCHK(c, hipMemcpyAsync(dev, h.data(), total, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

// ---- watchdog of the RCCL collectives (see isle_ctx::wd_*) ----
__global__ void wd_mark_k(volatile uint32_t* done, uint32_t seq) {
  *done = seq;
  __threadfence_system();
}
__global__ void wd_stall_k(unsigned long long ticks) {  // test hook: spin for `ticks` of the 100 MHz wall clock, then end
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
static void wd_loop(isle_ctx* c) {
  (void)hipSetDevice(c->device);
  uint32_t last_done = *c->wd_done;
  auto last_change = std::chrono::steady_clock::now();
  while (!c->wd_stop.load()) {
    std::this_thread::sleep_for(std::chrono::milliseconds(100));
    const uint32_t done = *c->wd_done, issued = c->wd_issued.load();
    const auto now = std::chrono::steady_clock::now();
    if (done != last_done || done == issued) {  // progress, or nothing pending
      last_done = done;
      last_change = now;
      continue;
    }
    if (std::chrono::duration<double>(now - last_change).count() > c->wd_timeout_s) {
      fprintf(stderr, "[isle_hip] rank %d: collective %u of %u issued has not completed for %.0f s: aborting the communicator\n", c->rank, done + 1, issued,
              c->wd_timeout_s);
      c->comm_dead.store(1);
      (void)ncclCommAbort(c->comm);
      return;
    }
  }
}
static int wd_start(isle_ctx* c) {
  if (const char* e = c->knob(KN_COMM_TIMEOUT)) c->wd_timeout_s = atof(e);
  if (!(c->wd_timeout_s > 0.0) || c->wd_thread.joinable()) return 0;
  c->wd_done = reinterpret_cast<volatile uint32_t*>(c->pin + isle_ctx::PIN_SMALL + (192u << 10) + 256);  // page-locked
  *c->wd_done = 0;
  c->wd_issued.store(0);
  c->wd_stop.store(false);
  c->wd_thread = std::thread(wd_loop, c);
  return 0;
}
static void wd_end(isle_ctx* c) {
  c->wd_stop.store(true);
  if (c->wd_thread.joinable()) c->wd_thread.join();
}
static int wd_before(isle_ctx* c) {  // ahead of an RCCL call
  if (c->comm_dead.load()) return isle_fail(c, ISLE_E_COMM, "the communicator was aborted after a collective timed out");
  if (const char* e = c->knob(KN_TEST_STALL_MS)) {
    hipLaunchKernelGGL(wd_stall_k, dim3(1), dim3(1), 0, c->stream, (unsigned long long)(atof(e) * 1e5));
    HIPCHK(c, hipGetLastError());
  }
  return 0;
}
static int wd_after(isle_ctx* c) {  // behind it: the device reports the collective's completion
  if (!c->wd_thread.joinable()) return 0;
  const uint32_t seq = c->wd_issued.load() + 1;
  hipLaunchKernelGGL(wd_mark_k, dim3(1), dim3(1), 0, c->stream, c->wd_done, seq);
  c->wd_issued.store(seq);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int isle_allreduce(isle_ctx* c, void* buf, size_t count, int dtype, bool max_op) {
  if (!c->multi() || !count) return 0;
  if (c->host_xchg) return host_exchange(c, max_op ? ISLE_XCHG_ALLREDUCE_MAX : ISLE_XCHG_ALLREDUCE_SUM, buf, count, dtype);
  ISLECHK(wd_before(c));
  NCCLCHK(c, ncclAllReduce(buf, buf, count, kNcclType[dtype], max_op ? ncclMax : ncclSum, c->comm, c->stream));
  return wd_after(c);
}

int isle_allgather(isle_ctx* c, const void* send, void* recv, size_t count_per_rank, int dtype) {
  if (!c->multi() || !count_per_rank) return 0;
  if (c->host_xchg) {
    const size_t bytes = count_per_rank * kDtSize[dtype];
    char* mine = (char*)recv + (size_t)c->rank * bytes;
    if ((const void*)mine != send) HIPCHK(c, hipMemcpyAsync(mine, send, bytes, hipMemcpyDeviceToDevice, c->stream));
    return host_exchange(c, ISLE_XCHG_ALLGATHER, recv, count_per_rank, dtype);
  }
  ISLECHK(wd_before(c));
  NCCLCHK(c, ncclAllGather(send, recv, count_per_rank, kNcclType[dtype], c->comm, c->stream));
  return wd_after(c);
}

// Control decisions of the replicated parts (rank of a Krylov block, restart index, form of the small EVD) are taken per rank from
// replicated data.  Identical GPUs running identical kernels on identical inputs give identical bits, but nothing else
// enforces it; a rank that decided differently would issue a different sequence of collectives and the job would hang.
// Every such decision therefore goes through an all-reduce(MAX) of (v, -v): all ranks see the same pair, so either all of
// them carry on or all of them return ISLE_E_COMM at the same point.  Device-side variant for the pipelined expand loop:
// ks_agree_pack_k + the same all-reduce on the mailbox, no extra host round trip.
int agree_i32(isle_ctx* c, int v, const char* what) {
  if (!c->multi()) return 0;
  HIPCHK(c, c->flags.reserve(16));
  int* h = reinterpret_cast<int*>(c->pin + isle_ctx::PIN_SMALL + (192u << 10) + 128);  // page-locked
  h[0] = v;
  h[1] = -v;
  HIPCHK(c, hipMemcpyAsync(c->flags.p + 8, h, 2 * sizeof(int), hipMemcpyHostToDevice, c->stream));
  {
    TimeScope ts(c, ISLE_T_COMM);
    ISLECHK(isle_allreduce(c, c->flags.p + 8, 2, ISLE_DT_I32, true));
  }
  HIPCHK(c, hipMemcpyAsync(h, c->flags.p + 8, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (h[0] != -h[1]) return isle_fail(c, ISLE_E_COMM, "ranks disagree on %s (min %d, max %d): replicated state diverged", what, -h[1], h[0]);
  return 0;
}
extern "C" int isle_hip_host_rand(uint64_t seed, int n, uint32_t* out) {
  if (!out || n < 0) return ISLE_E_ARG;
  HostRng g(seed);
  for (int i = 0; i < n; ++i) out[i] = g.next31();
  return 0;
}

// rows [r0, r0 + nl) of a column-major n x w matrix <-> a packed nloc x w block (rows beyond nl zero)
__global__ void slice_rows_k(float* __restrict__ M, uint64_t n, int w, uint64_t r0, uint64_t nl, uint64_t nloc, float* __restrict__ blk, int pack) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nloc * (uint64_t)w) return;
  const uint64_t j = i / nloc, r = i - j * nloc;
  if (pack) blk[i] = r < nl ? M[j * n + r0 + r] : 0.f;
  else if (r < nl) M[j * n + r0 + r] = blk[i];
}
int k_slice_rows(isle_ctx* c, float* M, uint64_t n, int w, uint64_t r0, uint64_t nl, uint64_t nloc, float* blk, bool pack) {
  const uint64_t tot = nloc * (uint64_t)w;
  if (!tot) return 0;
  hipLaunchKernelGGL(slice_rows_k, dim3(cdiv((long)tot, 256)), dim3(256), 0, c->stream, M, n, w, r0, nl, nloc, blk, pack ? 1 : 0);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------
extern "C" isle_ctx* isle_hip_create(int device_id) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    fprintf(stderr, "isle_hip_create: no HIP device visible (this library has no CPU fallback)\n");
    return nullptr;
  }
  if (device_id < 0 || device_id >= ndev) {
    fprintf(stderr, "isle_hip_create: device %d out of range (%d devices)\n", device_id, ndev);
    return nullptr;
  }
  if (hipSetDevice(device_id) != hipSuccess) return nullptr;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return nullptr;
  isle_ctx* c = new isle_ctx;
  c->device = device_id;
  c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (hipStreamCreate(&c->stream) != hipSuccess) {
    delete c;
    return nullptr;
  }
  isle_refresh_knobs(c);
  if (const char* br = c->knob(KN_CHUNK_COLS)) c->band_rows = (uint32_t)atoi(br);
  if (hipHostMalloc((void**)&c->pin, isle_ctx::PIN_BYTES, hipHostMallocDefault) != hipSuccess) {
    (void)hipStreamDestroy(c->stream);
    delete c;
    return nullptr;
  }
  return c;
}

extern "C" void isle_hip_destroy(isle_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  wd_end(c);
  if (c->comm && !c->comm_dead.load()) ncclCommDestroy(c->comm);  // (an aborted communicator is already gone)
  if (c->pin) (void)hipHostFree(c->pin);
  if (c->pin_stage) (void)hipHostFree(c->pin_stage);
  for (auto& e : c->ev_used) {
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  for (auto& e : c->ev_free) {
    (void)hipEventDestroy(e.a);
    (void)hipEventDestroy(e.b);
  }
  for (auto& e : c->ks_ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : c->ks_ev_ready)
    if (e) (void)hipEventDestroy(e);
  if (c->copy_stream) {
    (void)hipStreamSynchronize(c->copy_stream);
    (void)hipStreamDestroy(c->copy_stream);
  }
  (void)hipStreamSynchronize(c->stream);
  (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" const char* isle_hip_last_error(isle_ctx* c) { return c ? c->err.c_str() : "null context"; }

extern "C" int isle_hip_comm_unique_id(void* out128) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return ISLE_E_COMM;
  memcpy(out128, &id, sizeof id);
  return 0;
}

// ---- communicator self-test (ISLE_COMM_SELFTEST; default on behind ncclCommInitRank with more than one rank) ----
// One all-reduce (sum), one all-reduce (max) and one all-gather of every (datatype, size class) the step issues — scalars and flag pairs,
// coefficient blocks, the 4 MB Gram panel / k x k centroid sums, a 64 MB piece of the V x k centre sums — on patterns whose results are known
// in closed form, checked element by element on the host.  A rank on the wrong device, a transport that does not come up, a datatype or
// count that one build disagrees on, surface here, within seconds and with a message that names the operation, instead of as a hang or
// a wrong eigenvalue inside the first step.  It also takes RCCL's connection set-up out of the first timed collective.
template <typename T>
static int selftest_type(isle_ctx* c, int dtype, const char* tname, const std::vector<size_t>& counts, int* ncoll) {
  const int W = c->world, r = c->rank;
  for (size_t n : counts) {
    std::vector<T> h(n * (size_t)W);
    DevBuf<T> d;
    HIPCHK(c, d.reserve(n * (size_t)W));
    for (int op = 0; op < 3; ++op) {  // 0 sum, 1 max, 2 all-gather
      const size_t nsend = n;
      if (op < 2) for (size_t i = 0; i < n; ++i) h[i] = (T)((r + 1) * (int)(i % 251 + 1));
      else for (size_t i = 0; i < n; ++i) h[(size_t)r * n + i] = (T)(r * 131 + (int)(i % 127));
      HIPCHK(c, hipMemcpyAsync(op < 2 ? d.p : d.p + (size_t)r * n, op < 2 ? h.data() : h.data() + (size_t)r * n, nsend * sizeof(T), hipMemcpyHostToDevice, c->stream));
      if (op < 2) ISLECHK(isle_allreduce(c, d.p, n, dtype, op == 1));
      else ISLECHK(isle_allgather(c, d.p + (size_t)r * n, d.p, n, dtype));
      const size_t nback = op < 2 ? n : n * (size_t)W;
      HIPCHK(c, hipMemcpyAsync(h.data(), d.p, nback * sizeof(T), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      if (c->comm_dead.load()) return isle_fail(c, ISLE_E_COMM, "communicator self-test: the %s of %zu %s did not complete (communicator aborted)",
                                                op == 0 ? "all-reduce(sum)" : op == 1 ? "all-reduce(max)" : "all-gather", n, tname);
      for (size_t i = 0; i < nback; ++i) {
        T want;
        if (op == 0) want = (T)((W * (W + 1) / 2) * (int)(i % 251 + 1));
        else if (op == 1) want = (T)(W * (int)(i % 251 + 1));
        else want = (T)((int)(i / n) * 131 + (int)((i % n) % 127));
        if (h[i] != want)
          return isle_fail(c, ISLE_E_COMM, "communicator self-test FAILED on rank %d of %d: %s of %zu %s, element %zu is %.17g, expected %.17g", r, W,
                           op == 0 ? "all-reduce(sum)" : op == 1 ? "all-reduce(max)" : "all-gather", n, tname, i, (double)h[i], (double)want);
      }
      ++*ncoll;
    }
  }
  return 0;
}
static int comm_selftest(isle_ctx* c) {
  const auto t0 = std::chrono::steady_clock::now();
  int ncoll = 0;
  const std::vector<size_t> small = {1, 2, 16, 1000}, panel = {1, 2, 100, 20100, (size_t)1 << 20};
  std::vector<size_t> f32 = panel;
  f32.push_back((size_t)1 << 24);
  ISLECHK(selftest_type<float>(c, ISLE_DT_F32, "f32", f32, &ncoll));
  ISLECHK(selftest_type<double>(c, ISLE_DT_F64, "f64", panel, &ncoll));
  ISLECHK(selftest_type<int32_t>(c, ISLE_DT_I32, "i32", small, &ncoll));
  ISLECHK(selftest_type<uint32_t>(c, ISLE_DT_U32, "u32", panel, &ncoll));
  ISLECHK(selftest_type<uint64_t>(c, ISLE_DT_U64, "u64", small, &ncoll));
  char bus[64] = "?";
  (void)hipDeviceGetPCIBusId(bus, sizeof bus, c->device);
  int ver = 0;
  if (c->comm) (void)ncclGetVersion(&ver);
  fprintf(stderr, "[isle_hip] rank %d of %d: device %d (PCI %s), %s; communicator self-test: %d collectives (sum / max / all-gather x f32 f64 i32 u32 u64, 1 ... 16 M elements) correct in %.0f ms\n",
          c->rank, c->world, c->device, bus, c->host_xchg ? "host-staged test transport" : ("RCCL " + std::to_string(ver)).c_str(), ncoll,
          std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  return 0;
}
static bool comm_selftest_wanted(isle_ctx* c, bool dflt) {
  if (c->knob_on(KN_COMM_SELFTEST)) return !c->knob_zero(KN_COMM_SELFTEST);
  return dflt;
}

extern "C" int isle_hip_comm_init(isle_ctx* c, int world, int rank, const void* uid) {
  if (!c || world < 1 || rank < 0 || rank >= world) return isle_fail(c, ISLE_E_ARG, "bad world/rank");
  c->world = world;
  c->rank = rank;
  // world == 1 needs no communicator; ISLE_FORCE_COMM=1 creates a 1-rank one anyway so that every RCCL call site
  // of the sharded path can be exercised on a single GPU (tests/test_gpu_comm_selftest.py)
  isle_refresh_knobs(c);
  if (world == 1 && !c->knob_on(KN_FORCE_COMM)) return 0;
  ISLECHK(isle_enter(c));
  ncclUniqueId id;
  memcpy(&id, uid, sizeof id);
  NCCLCHK(c, ncclCommInitRank(&c->comm, world, id, rank));
  ISLECHK(wd_start(c));
  if (comm_selftest_wanted(c, world > 1)) return comm_selftest(c);  // (a forced 1-rank communicator: the collectives are the identity)
  return 0;
}

extern "C" int isle_hip_comm_init_host(isle_ctx* c, int world, int rank, isle_host_exchange_fn fn, void* user) {
  if (!c || world < 1 || rank < 0 || rank >= world || !fn) return isle_fail(c, ISLE_E_ARG, "bad world/rank/function");
  if (c->comm) return isle_fail(c, ISLE_E_ARG, "the context already has an RCCL communicator");
  c->world = world;
  c->rank = rank;
  c->host_xchg = fn;
  c->host_xchg_user = user;
  if (comm_selftest_wanted(c, false)) {
    ISLECHK(isle_enter(c));
    return comm_selftest(c);
  }
  return 0;
}

extern "C" int isle_hip_plan_shards(uint64_t num_docs, const int64_t* offs, int parts, uint64_t* bounds) {
  if (!offs || !bounds || parts < 1) return ISLE_E_ARG;
  const int64_t nnz = offs[num_docs];
  bounds[0] = 0;
  for (int p = 1; p < parts; ++p) {
    const int64_t target = (int64_t)(((__int128)nnz * p) / parts);
    const int64_t* it = std::lower_bound(offs, offs + num_docs + 1, target);
    uint64_t d = (uint64_t)(it - offs);
    if (d > num_docs) d = num_docs;
    if (d < bounds[p - 1]) d = bounds[p - 1];
    bounds[p] = d;
  }
  bounds[parts] = num_docs;
  return 0;
}

// ------------------------------------------------------------------------------------------
// upload
// ------------------------------------------------------------------------------------------
// A context keeps its buffers at the largest size it ever needed (no allocation inside a solve).  Six of them scale with D x k or nnz and
// reach 40 GB each at config 3; when a matrix arrives for which they could never be needed at that size (k <= 2048, so the projection is at
// most 2048 D floats, ...) they are released, so that a context that has held config 3 can go on to another large corpus without running
// out of the 288 GB (round 6: the topic model of config 5 behind config 3 in one process).  Their contents are void at this point anyway.
void isle_trim_derived(isle_ctx* c, uint64_t D, uint64_t nnz) {
  const size_t d = (size_t)std::max<uint64_t>(D, 1), z = (size_t)std::max<uint64_t>(nnz, 1);
  if (c->P.cap > 2048 * d) c->P.release();
  if (c->Pt.cap > 2048 * d) c->Pt.release();
  if (c->Pt2.cap > 512 * d) c->Pt2.release();        // 16-byte units: 4 k D bytes
  if (c->dotsT.cap > 2048 * d) c->dotsT.release();
  if (c->gl_pscratch.cap > 512 * d) c->gl_pscratch.release();  // sixteen slabs of 12 floats per document + norm partials
  if (c->gl_fb_tmp.cap > 2 * z) c->gl_fb_tmp.release();
}

static int upload_common(isle_ctx* c, uint64_t V, uint64_t D, uint64_t nnz, const float* vals, const uint32_t* rows32,
                         const int64_t* offs, uint64_t doc_offset, uint64_t docs_global) {
  if (!c) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  if (V == 0 || V > 0xfffffff0ull || D > 0xfffffff0ull) return isle_fail(c, ISLE_E_ARG, "vocab/doc count out of range");
  if (offs[0] != 0 || (uint64_t)offs[D] != nnz) return isle_fail(c, ISLE_E_ARG, "offsets[0] != 0 or offsets[D] != nnz");
  // (offsets are checked here, against nnz, before anything is copied: a bad offset array must not size a device read; the entries
  // themselves — 1 B of them at config 3 — are checked by a kernel behind the copy)
  for (uint64_t d = 0; d < D; ++d)
    if (offs[d + 1] < offs[d] || (uint64_t)offs[d + 1] > nnz) return isle_fail(c, ISLE_E_ARG, "offsets not monotone at column %llu", (unsigned long long)d);
  // everything derived from the previous matrix is void from here on, whether or not the new one is accepted
  c->band_ready = false;
  c->gl_mode = -1;
  c->P_ready = false;
  c->Pt_ready = false;
  c->Pt2_ready = false;
  c->lift_valid = false;
  c->members_valid = false;
  c->U_k = 0;
  c->centers_ready = false;
  c->assign_valid = false;
  c->p_catch_ready = false;
  c->p_model_ready = false;
  c->b_from_threshold = false;
  c->kmpp_track_k = 0;
  isle_trim_derived(c, D, nnz);
  c->V = V;
  c->D = D;
  c->nnz = nnz;
  c->doc_offset = doc_offset;
  c->D_global = docs_global ? docs_global : D;
  HIPCHK(c, c->vals.reserve(nnz ? nnz : 1));
  HIPCHK(c, c->rows.reserve(nnz ? nnz : 1));
  HIPCHK(c, c->offs.reserve(D + 1));
  if (nnz) {
    HIPCHK(c, hipMemcpy(c->vals.p, vals, nnz * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->rows.p, rows32, nnz * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  HIPCHK(c, hipMemcpy(c->offs.p, offs, (D + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
  {
    // include/matUtils.h:138-148 (the reference's constructor asserts): row ids in range and strictly ascending inside a column
    unsigned long long bad[2] = {0, 0};
    ISLECHK(k_csc_validate(c, bad));
    if (bad[0]) {
      c->V = 0;  // nothing usable was uploaded: later calls refuse with "no matrix uploaded"
      c->D = 0;
      c->nnz = 0;
      const unsigned long long col = bad[0] - 1;
      return bad[1] == 2   ? isle_fail(c, ISLE_E_ARG, "row index out of range in column %llu", col)
             : bad[1] == 1 ? isle_fail(c, ISLE_E_ARG, "offsets not monotone at column %llu", col)
                           : isle_fail(c, ISLE_E_ARG, "rows not strictly ascending in column %llu", col);
    }
  }
  return 0;
}

extern "C" int isle_hip_upload_csc_u32(isle_ctx* c, uint64_t V, uint64_t D, uint64_t nnz, const float* vals, const uint32_t* rows,
                                       const int64_t* offs, uint64_t doc_offset, uint64_t docs_global) {
  return upload_common(c, V, D, nnz, vals, rows, offs, doc_offset, docs_global);
}
extern "C" int isle_hip_upload_csc_u64(isle_ctx* c, uint64_t V, uint64_t D, uint64_t nnz, const float* vals, const uint64_t* rows,
                                       const int64_t* offs, uint64_t doc_offset, uint64_t docs_global) {
  std::vector<uint32_t> r32(nnz);
  for (uint64_t i = 0; i < nnz; ++i) {
    if (rows[i] > 0xfffffff0ull) return isle_fail(c, ISLE_E_ARG, "row index too large");
    r32[i] = (uint32_t)rows[i];
  }
  return upload_common(c, V, D, nnz, vals, r32.data(), offs, doc_offset, docs_global);
}

extern "C" int isle_hip_frobenius(isle_ctx* c, float* out) {
  if (!c || !out) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  double s = 0.0;
  ISLECHK(k_frobenius(c, &s));
  if (c->multi()) {
    HIPCHK(c, c->gram.reserve(1024));
    HIPCHK(c, hipMemcpyAsync(c->gram.p, &s, sizeof(double), hipMemcpyHostToDevice, c->stream));
    ISLECHK(allreduce_sum<double>(c, c->gram.p, 1));
    HIPCHK(c, hipMemcpyAsync(&s, c->gram.p, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  *out = (float)s;
  return 0;
}

// ------------------------------------------------------------------------------------------
// Gram apply on device pointers: Zcm (V x b col-major) = B (B^T Xcm)
// ------------------------------------------------------------------------------------------

// Zcm (V x b col-major) = B (B^T Xcm) on device pointers, 1 <= b <= 32, one pass over both copies of B per call.
int gram_apply_dev(isle_ctx* c, const float* Xcm, int b, float* Zcm) {
  if (b < 1 || b > 32) return isle_fail(c, ISLE_E_ARG, "gram_apply: b = %d not in [1, 32]", b);
  ISLECHK(k_band_build(c));  // first application of a solve: operator build (and the choice of the form)
  if (c->gl_mode == 1) {
    // LDS-banded form (gram_lds.hip), panels of at most 10 columns (40-byte rows of a planar band); column groups of a col-major block
    // are contiguous
    for (int j0 = 0; j0 < b; j0 += 10) {
      const int bg = std::min(10, b - j0);
      const int BPg = panel_width(bg);
      ISLECHK(k_gl_apply_cm(c, Xcm + (size_t)j0 * c->V, bg, BPg, Zcm + (size_t)j0 * c->V));
      ISLECHK(allreduce_sum<float>(c, Zcm + (size_t)j0 * c->V, (size_t)c->V * bg));
    }
    return 0;
  }
  const int BP = panel_width(b);
  HIPCHK(c, c->Xrm.reserve((size_t)c->V * BP));
  HIPCHK(c, c->Zrm.reserve((size_t)c->V * BP));
  HIPCHK(c, c->Yrm.reserve((size_t)(c->D ? c->D : 1) * BP));
  ISLECHK(k_pack_rm(c, Xcm, c->V, b, BP, c->Xrm.p));
  ISLECHK(k_gram_pass1(c, BP));
  ISLECHK(k_gram_pass2(c, BP));
  ISLECHK(allreduce_sum<float>(c, c->Zrm.p, (size_t)c->V * BP));
  ISLECHK(k_unpack_cm(c, c->Zrm.p, c->V, b, BP, Zcm));
  return 0;
}

extern "C" int isle_hip_gram_apply(isle_ctx* c, const float* X, int b, float* Z) {
  if (!c || !X || !Z) return ISLE_E_ARG;
  if (c->V == 0) return isle_fail(c, ISLE_E_ARG, "no matrix uploaded");
  ISLECHK(isle_enter(c));
  const size_t n = (size_t)c->V * b;
  HIPCHK(c, c->Xcm.reserve(n));
  HIPCHK(c, c->Zcm.reserve(n));
  HIPCHK(c, hipMemcpy(c->Xcm.p, X, n * sizeof(float), hipMemcpyHostToDevice));
  ISLECHK(gram_apply_dev(c, c->Xcm.p, b, c->Zcm.p));
  HIPCHK(c, hipMemcpyAsync(Z, c->Zcm.p, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int isle_hip_operator_form(isle_ctx* c, int* form) {
  if (!c || !form) return ISLE_E_ARG;
  *form = c->band_ready ? c->gl_mode : -1;
  return 0;
}

// ------------------------------------------------------------------------------------------
// measurement
// ------------------------------------------------------------------------------------------
extern "C" int isle_hip_timing_enable(isle_ctx* c, int on) {
  if (!c) return ISLE_E_ARG;
  c->timing = on != 0;
  c->timing_mask = on == 2 ? ((1u << ISLE_T_GRAM_PASS1) | (1u << ISLE_T_GRAM_PASS2)) : 0xffffffffu;
  return 0;
}
extern "C" int isle_hip_timing_reset(isle_ctx* c) {
  if (!c) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  ISLECHK(drain_events(c));
  for (int i = 0; i < ISLE_T_COUNT; ++i) {
    c->t_ms[i] = 0.0;
    c->t_n[i] = 0;
  }
  return 0;
}
extern "C" int isle_hip_timing_get(isle_ctx* c, double* ms, uint64_t* launches) {
  if (!c) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  ISLECHK(drain_events(c));
  for (int i = 0; i < ISLE_T_COUNT; ++i) {
    if (ms) ms[i] = c->t_ms[i];
    if (launches) launches[i] = c->t_n[i];
  }
  return 0;
}
extern "C" int isle_hip_synchronize(isle_ctx* c) {
  if (!c) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

