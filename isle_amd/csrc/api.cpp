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

#include "common.h"

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
    {"ISLE_GL_ROUNDS", "tuning", "0: pass 1 keeps ceil(slices / items per lane) waves in workgroups strided over the length order instead of filling whole rounds of the CUs with workgroups of adjacent waves"},
    {"ISLE_GL_COLUMNS", "tuning", "0: pass 2 chunks its document bands per word block instead of walking band columns shared through one XCD's L2"},
    {"ISLE_GL_PANEL", "tuning", "8 | 10: columns per pass of the k-wide / thin products (default 10, 8 at 8 items per lane in pass 1)"},
    {"ISLE_WIDE_GATHER", "form", "k-wide products (projection, first word-space assignment) by the row-gather kernel"},
    {"ISLE_WIDE_LDS", "form", "k-wide products through the LDS-banded pass-1 stream whatever the vocabulary size"},
    {"ISLE_KS_ROWSHARD", "form", "1: several ranks orthogonalise row slices of the Krylov block (all-reduced coefficients, all-gathered block); default 0 = replicated"},
    {"ISLE_KS_SYNC", "form", "expand loop without the speculative pipeline (one host synchronisation per step)"},
    {"ISLE_KS_ORTHO_PASSES", "form", "3: the reference's three Gram-Schmidt passes per Krylov step instead of two"},
    {"ISLE_QR_FUSED", "form", "1: panel QR as one persistent launch (bitwise equal to the kernel chain, no faster)"},
    {"ISLE_EVD_JACOBI", "form", "small symmetric EVD by block Jacobi instead of tridiagonalisation"},
    {"ISLE_TD_CHAIN", "form", "tridiagonalisation as a launch chain instead of the persistent kernel"},
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
    {"ISLE_YY_MODE", "form", "doc | docg | group: Yinyang iteration by document (row-major / group-major centres) or ordered by group (default: group at k >= 256)"},
    {"ISLE_YY_FUSED", "form", "0: the by-group Yinyang iteration lowers the bounds (yy_filter_k) and tightens the active documents (yy2_tighten_k) in two launches instead of one (same bits)"},
    {"ISLE_CENTERS_FRESH", "form", "centroid counts recounted from the member lists every iteration instead of updated by the documents that moved"},
    {"ISLE_INFER_CAP_ROWS", "form", "inference: stage at most this many model rows per document in LDS (default 0: rows read through L2)"},
    {"ISLE_CHUNK_COLS", "tuning", "gather form: rows per chunk of the chunked-CSR copy"},
    {"ISLE_FORCE_COMM", "test hook", "create a 1-rank RCCL communicator so that every collective call site runs on one GPU"},
    {"ISLE_HOST_TRACE", "diagnostic", "print host wall time between marks of the control loops"},
    {"ISLE_DEBUG_HAMERLY", "diagnostic", "print active documents / group scans per Lloyd iteration"},
    {"ISLE_DEBUG_EVD", "diagnostic", "print sweeps / orthogonality defect of the small EVD"},
    {"ISLE_GL_VERBOSE", "diagnostic", "print the geometry of the LDS-banded operator build"},
    {"ISLE_TD_FORCE_BAIL_RANK", "test hook", "this rank behaves as if the grid barrier of its persistent EVD had timed out"},
    {"ISLE_GL_TEST_CUS", "test hook", "the operator build lays pass 1 out as for a device of this many CUs (several rounds of workgroups on a small matrix)"},
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

TimeScope::TimeScope(isle_ctx* c_, int fam) : c(c_), on(false) {
  if (fam < 0 || !c->timing || !((c->timing_mask >> fam) & 1u)) return;
  if (!c->ev_free.empty()) {
    ep = c->ev_free.back();
    c->ev_free.pop_back();
  } else {
    if (hipEventCreate(&ep.a) != hipSuccess || hipEventCreate(&ep.b) != hipSuccess) return;
  }
  ep.fam = fam;
  on = (hipEventRecord(ep.a, c->stream) == hipSuccess);
}
TimeScope::~TimeScope() {
  if (!on) return;
  (void)hipEventRecord(ep.b, c->stream);
  c->ev_used.push_back(ep);
}

static int drain_events(isle_ctx* c) {
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

int isle_allreduce(isle_ctx* c, void* buf, size_t count, int dtype, bool max_op) {
  if (!c->multi() || !count) return 0;
  if (c->host_xchg) return host_exchange(c, max_op ? ISLE_XCHG_ALLREDUCE_MAX : ISLE_XCHG_ALLREDUCE_SUM, buf, count, dtype);
  NCCLCHK(c, ncclAllReduce(buf, buf, count, kNcclType[dtype], max_op ? ncclMax : ncclSum, c->comm, c->stream));
  return 0;
}

int isle_allgather(isle_ctx* c, const void* send, void* recv, size_t count_per_rank, int dtype) {
  if (!c->multi() || !count_per_rank) return 0;
  if (c->host_xchg) {
    const size_t bytes = count_per_rank * kDtSize[dtype];
    char* mine = (char*)recv + (size_t)c->rank * bytes;
    if ((const void*)mine != send) HIPCHK(c, hipMemcpyAsync(mine, send, bytes, hipMemcpyDeviceToDevice, c->stream));
    return host_exchange(c, ISLE_XCHG_ALLGATHER, recv, count_per_rank, dtype);
  }
  NCCLCHK(c, ncclAllGather(send, recv, count_per_rank, kNcclType[dtype], c->comm, c->stream));
  return 0;
}

template <class T>
struct DtOf;
template <>
struct DtOf<float> { static constexpr int v = ISLE_DT_F32; };
template <>
struct DtOf<double> { static constexpr int v = ISLE_DT_F64; };
template <>
struct DtOf<int> { static constexpr int v = ISLE_DT_I32; };
template <>
struct DtOf<uint32_t> { static constexpr int v = ISLE_DT_U32; };
template <>
struct DtOf<uint64_t> { static constexpr int v = ISLE_DT_U64; };

template <class T>
static int allreduce_sum(isle_ctx* c, T* buf, size_t count) {
  if (!c->multi()) return 0;
  TimeScope ts(c, ISLE_T_COMM);
  return isle_allreduce(c, buf, count, DtOf<T>::v);
}

// Control decisions of the replicated parts (rank of a Krylov block, restart index, form of the small EVD) are taken per rank from
// replicated data.  Identical GPUs running identical kernels on identical inputs give identical bits, but nothing else
// enforces it; a rank that decided differently would issue a different sequence of collectives and the job would hang.
// Every such decision therefore goes through an all-reduce(MAX) of (v, -v): all ranks see the same pair, so either all of
// them carry on or all of them return ISLE_E_COMM at the same point.  Device-side variant for the pipelined expand loop:
// ks_agree_pack_k + the same all-reduce on the mailbox, no extra host round trip.
static int agree_i32(isle_ctx* c, int v, const char* what) {
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
constexpr int KS_AGREE = 40;  // int slots [40, 43) of the expand mailbox: max rank, -min rank, max status over all ranks
__global__ void ks_agree_pack_k(int* meta) {
  if (threadIdx.x == 0) {
    meta[KS_AGREE] = meta[0];
    meta[KS_AGREE + 1] = -meta[0];
    meta[KS_AGREE + 2] = meta[1];
  }
}

// ------------------------------------------------------------------------------------------
// host RNG (rand() stand-in)
// ------------------------------------------------------------------------------------------
namespace {
// glibc's rand() (the TYPE_3 additive feedback generator of random_r.c: r[i] = r[i - 31] + r[i - 3], 31 state words seeded by the
// Lehmer generator 16807 mod 2^31 - 1, the first 310 outputs discarded, results shifted right by one).  The reference calls rand()
// without ever calling srand(), i.e. with seed 1: rng_seed = 1 therefore draws the numbers a reference binary linked against glibc
// draws (tests/test_abi_cpu.py checks the sequence against this machine's libc).  What still separates un-injected seeds from a
// reference run is the prefix sum they index (fp32 and sequential there, :2170-2172; fp64 and parallel here).
struct HostRng {
  uint32_t r[34];
  int k = 0;  // next output is the k-th
  explicit HostRng(uint64_t seed64) {
    uint32_t seed = (uint32_t)seed64;
    if (seed == 0) seed = 1;
    int32_t w[34];
    w[0] = (int32_t)seed;
    for (int i = 1; i < 31; ++i) {
      int64_t v = (16807LL * w[i - 1]) % 2147483647LL;
      if (v < 0) v += 2147483647LL;
      w[i] = (int32_t)v;
    }
    for (int i = 31; i < 34; ++i) w[i] = w[i - 31];
    for (int i = 0; i < 34; ++i) r[i] = (uint32_t)w[i];
    for (int i = 34; i < 344; ++i) step();  // discarded
  }
  uint32_t step() {  // the ring holds the last 34 words; word i lives at i % 34
    const uint32_t v = r[(k + 34 - 31) % 34] + r[(k + 34 - 3) % 34];
    r[k % 34] = v;
    k = (k + 1) % 34;
    return v;
  }
  uint32_t next31() { return step() >> 1; }  // rand(): 0 .. RAND_MAX = 2^31 - 1
  // include/matUtils.h:473-477: (double)rand() + (double)rand() * (RAND_MAX + 1), over (RAND_MAX + 1)^2.  The two rand() calls of that
  // expression are UNSEQUENCED in C++: which of them supplies the low word is the reference compiler's choice.  Assumed here: the left
  // operand is evaluated first (what g++ does for this expression at -O3 — the only arrangement under which rng_seed = 1 reproduces an
  // unseeded reference binary's dice); with the other order the low and high words swap.  Un-injected seeds are not promised equal to
  // a reference run's in any case, which is why the parity tests inject them.
  double fraction() {
    const double R1 = 2147483648.0;
    const double lo = (double)next31();
    const double hi = (double)next31();
    return (lo + hi * R1) / (R1 * R1);
  }
};
}  // namespace
extern "C" int isle_hip_host_rand(uint64_t seed, int n, uint32_t* out) {
  if (!out || n < 0) return ISLE_E_ARG;
  HostRng g(seed);
  for (int i = 0; i < n; ++i) out[i] = g.next31();
  return 0;
}
namespace {

struct HMat {  // small col-major float matrix on the host (the projected matrix H)
  size_t r = 0, c = 0, ld = 0, cap_c = 0;  // r x c in use inside an ld x cap_c allocation (zero outside what was written)
  std::vector<float> a;
  HMat() {}
  HMat(size_t r_, size_t c_) : r(r_), c(c_), ld(r_), cap_c(c_), a(r_ * c_, 0.f) {}
  // room to grow: the Krylov expansion appends blocks of rows and columns in place (at ncv = 2010 the matrix is 16 MB, and every
  // fresh copy of it cost the host 2 - 6 ms with the GPU idle)
  HMat(size_t r_, size_t c_, size_t cap_r_, size_t cap_c_) : r(r_), c(c_), ld(std::max(r_, cap_r_)), cap_c(std::max(c_, cap_c_)), a(ld * cap_c, 0.f) {}
  float& operator()(size_t i, size_t j) { return a[j * ld + i]; }
  float operator()(size_t i, size_t j) const { return a[j * ld + i]; }
};
HMat hsub(const HMat& m, size_t r0, size_t c0, size_t r1, size_t c1) {  // inclusive bounds (arma submat)
  HMat o(r1 - r0 + 1, c1 - c0 + 1);
  for (size_t j = c0; j <= c1; ++j)
    for (size_t i = r0; i <= r1; ++i) o(i - r0, j - c0) = m(i, j);
  return o;
}
}  // namespace

static int round4(int k) { return (k + 3) & ~3; }

// rows [r0, r0 + nl) of a column-major n x w matrix <-> a packed nloc x w block (rows beyond nl zero)
__global__ void slice_rows_k(float* __restrict__ M, uint64_t n, int w, uint64_t r0, uint64_t nl, uint64_t nloc, float* __restrict__ blk, int pack) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nloc * (uint64_t)w) return;
  const uint64_t j = i / nloc, r = i - j * nloc;
  if (pack) blk[i] = r < nl ? M[j * n + r0 + r] : 0.f;
  else if (r < nl) M[j * n + r0 + r] = blk[i];
}
static int k_slice_rows(isle_ctx* c, float* M, uint64_t n, int w, uint64_t r0, uint64_t nl, uint64_t nloc, float* blk, bool pack) {
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
  if (c->comm) ncclCommDestroy(c->comm);
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
  return 0;
}

extern "C" int isle_hip_comm_init_host(isle_ctx* c, int world, int rank, isle_host_exchange_fn fn, void* user) {
  if (!c || world < 1 || rank < 0 || rank >= world || !fn) return isle_fail(c, ISLE_E_ARG, "bad world/rank/function");
  if (c->comm) return isle_fail(c, ISLE_E_ARG, "the context already has an RCCL communicator");
  c->world = world;
  c->rank = rank;
  c->host_xchg = fn;
  c->host_xchg_user = user;
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
static int upload_common(isle_ctx* c, uint64_t V, uint64_t D, uint64_t nnz, const float* vals, const uint32_t* rows32,
                         const int64_t* offs, uint64_t doc_offset, uint64_t docs_global) {
  if (!c) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  if (V == 0 || V > 0xfffffff0ull || D > 0xfffffff0ull) return isle_fail(c, ISLE_E_ARG, "vocab/doc count out of range");
  if (offs[0] != 0 || (uint64_t)offs[D] != nnz) return isle_fail(c, ISLE_E_ARG, "offsets[0] != 0 or offsets[D] != nnz");
  for (uint64_t d = 0; d < D; ++d) {
    if (offs[d + 1] < offs[d]) return isle_fail(c, ISLE_E_ARG, "offsets not monotone at column %llu", (unsigned long long)d);
    for (int64_t i = offs[d]; i < offs[d + 1]; ++i) {
      if (rows32[i] >= V) return isle_fail(c, ISLE_E_ARG, "row index out of range at %lld", (long long)i);
      // include/matUtils.h:138-148: columns strictly increasing
      if (i > offs[d] && rows32[i] <= rows32[i - 1])
        return isle_fail(c, ISLE_E_ARG, "rows not strictly ascending in column %llu", (unsigned long long)d);
    }
  }
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
  c->band_ready = false;
  c->gl_mode = -1;
  c->P_ready = false;
  c->Pt_ready = false;
  c->lift_valid = false;
  c->members_valid = false;
  c->U_k = 0;
  c->centers_ready = false;
  c->assign_valid = false;
  c->p_catch_ready = false;
  c->p_model_ready = false;
  c->b_from_threshold = false;
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

// ------------------------------------------------------------------------------------------
// upstream stage: A -> B on the device
// ------------------------------------------------------------------------------------------
extern "C" int isle_hip_upload_counts_u32(isle_ctx* c, uint64_t V, uint64_t D, uint64_t nnz, const float* counts, const uint32_t* rows,
                                          const int64_t* offs, uint64_t doc_offset, uint64_t docs_global) {
  if (!c) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  if (V == 0 || V > 0xfffffff0ull || D > 0xfffffff0ull) return isle_fail(c, ISLE_E_ARG, "vocab/doc count out of range");
  if (offs[0] != 0 || (uint64_t)offs[D] != nnz) return isle_fail(c, ISLE_E_ARG, "offsets[0] != 0 or offsets[D] != nnz");
  for (uint64_t d = 0; d < D; ++d) {
    if (offs[d + 1] < offs[d]) return isle_fail(c, ISLE_E_ARG, "offsets not monotone at column %llu", (unsigned long long)d);
    for (int64_t i = offs[d]; i < offs[d + 1]; ++i) {
      if (rows[i] >= V) return isle_fail(c, ISLE_E_ARG, "row index out of range at %lld", (long long)i);
      if (i > offs[d] && rows[i] <= rows[i - 1])
        return isle_fail(c, ISLE_E_ARG, "rows not strictly ascending in column %llu", (unsigned long long)d);
      if (!(counts[i] > 0.f)) return isle_fail(c, ISLE_E_ARG, "count not positive at %lld", (long long)i);
    }
  }
  c->a_V = V;
  c->a_D = D;
  c->a_nnz = nnz;
  c->a_doc_offset = doc_offset;
  c->a_D_global = docs_global ? docs_global : D;
  HIPCHK(c, c->a_cnt.reserve(nnz ? nnz : 1));
  HIPCHK(c, c->a_rows.reserve(nnz ? nnz : 1));
  HIPCHK(c, c->a_offs.reserve(D + 1));
  if (nnz) {
    HIPCHK(c, hipMemcpy(c->a_cnt.p, counts, nnz * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->a_rows.p, rows, nnz * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  HIPCHK(c, hipMemcpy(c->a_offs.p, offs, (D + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
  c->a_ready = true;
  c->a_avg_valid = false;
  c->p_catch_ready = false;
  c->p_model_ready = false;
  return 0;
}

extern "C" int isle_hip_ingest_tdf(isle_ctx* c, const char* text, uint64_t nbytes, uint64_t V, uint64_t D, uint64_t max_entries,
                                   uint64_t* entries_read, uint64_t* nnz) {
  if (!c) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  if (c->world > 1) return isle_fail(c, ISLE_E_ARG, "ingest_tdf: single-rank only");
  if (V == 0 || V > 0xfffffff0ull || D == 0 || D > 0xfffffff0ull) return isle_fail(c, ISLE_E_ARG, "ingest_tdf: vocab/doc count out of range");
  if (nbytes && !text) return isle_fail(c, ISLE_E_ARG, "ingest_tdf: null text");
  c->a_ready = false;
  DevBuf<unsigned char> td;
  HIPCHK(c, td.reserve(nbytes + 16));
  hipError_t he = nbytes ? hipMemcpy(td.p, text, nbytes, hipMemcpyHostToDevice) : hipSuccess;
  uint64_t nread = 0, err[2] = {0, 0};
  int rc = 0;
  if (he == hipSuccess) rc = k_ingest_tdf(c, td.p, nbytes, V, D, &nread, err);
  (void)hipStreamSynchronize(c->stream);
  td.release();
  HIPCHK(c, he);
  ISLECHK(rc);
  if (err[0]) {
    static const char* what[] = {"", "bad character", "more than three fields", "fewer than three fields", "doc/word id is 0 or exceeds <num_docs>/<vocab_size>",
                                 "count is 0"};
    return isle_fail(c, ISLE_E_ARG, "ingest_tdf: %s on line %llu", what[err[0] < 6 ? err[0] : 0], (unsigned long long)(err[1] + 1));
  }
  if (max_entries && nread != max_entries)  // include/utils.h:227
    return isle_fail(c, ISLE_E_ARG, "ingest_tdf: file has %llu entries, <max_entries> says %llu", (unsigned long long)nread, (unsigned long long)max_entries);
  c->a_doc_offset = 0;
  c->a_D_global = D;
  c->a_ready = true;
  c->a_avg_valid = false;
  c->p_catch_ready = false;
  c->p_model_ready = false;
  if (entries_read) *entries_read = nread;
  if (nnz) *nnz = c->a_nnz;
  return 0;
}

extern "C" int isle_hip_get_A(isle_ctx* c, float* counts, uint32_t* rows, int64_t* offs) {
  if (!c) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  if (!c->a_ready) return isle_fail(c, ISLE_E_ARG, "get_A: no count matrix");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (counts && c->a_nnz) HIPCHK(c, hipMemcpy(counts, c->a_cnt.p, c->a_nnz * sizeof(float), hipMemcpyDeviceToHost));
  if (rows && c->a_nnz) HIPCHK(c, hipMemcpy(rows, c->a_rows.p, c->a_nnz * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (offs) HIPCHK(c, hipMemcpy(offs, c->a_offs.p, (c->a_D + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int isle_hip_threshold(isle_ctx* c, uint64_t num_topics, double sample_rate, uint64_t sample_seed, uint64_t* docs_kept,
                                  uint64_t* nnz_kept, uint64_t* entries_above, float* avg_out) {
  if (!c) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  if (!c->a_ready) return isle_fail(c, ISLE_E_ARG, "threshold: no count matrix uploaded");
  if (num_topics == 0) return isle_fail(c, ISLE_E_ARG, "threshold: num_topics == 0");
  const bool sampling = sample_rate > 0.0 && sample_rate < 1.0;
  if (sampling && c->world > 1) return isle_fail(c, ISLE_E_ARG, "threshold: document sampling is single-rank only");
  const uint64_t V = c->a_V, D = c->a_D;

  // corpus statistics (src/sparseMatrix.cpp:92-99), global
  HIPCHK(c, c->a_scan.reserve(isle_scan_scratch(D) + 4));
  uint64_t* st_dev = (uint64_t*)c->a_scan.p;
  ISLECHK(k_th_stats(c, st_dev));
  ISLECHK(allreduce_sum<uint64_t>(c, st_dev, 2));
  uint64_t st[2];
  HIPCHK(c, hipMemcpyAsync(st, st_dev, sizeof(st), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const uint64_t tokens = st[0], nz_docs = st[1];
  const float avg = (float)(tokens / std::max<uint64_t>(nz_docs, 1));  // :98, integer division
  if (avg_out) *avg_out = avg;
  c->a_avg = avg;
  c->a_avg_valid = true;
  const uint64_t maxv64 = (uint64_t)avg + 2;
  if (maxv64 > 65535) return isle_fail(c, ISLE_E_ARG, "threshold: average document size %g too large", (double)avg);
  const uint32_t maxv = (uint32_t)maxv64;

  // rounded normalised counts + per-word value histogram, global
  HIPCHK(c, c->a_q.reserve(c->a_nnz ? c->a_nnz : 1));
  HIPCHK(c, c->a_hist.reserve((size_t)V * (maxv + 1)));
  ISLECHK(k_th_round_hist(c, avg, maxv));
  ISLECHK(allreduce_sum<uint32_t>(c, c->a_hist.p, (size_t)V * (maxv + 1)));

  // thresholds  (src/sparseMatrix.cpp:367-368)
  uint64_t count_gr = (uint64_t)(1.0 * (float)nz_docs / (2.0 * (float)num_topics));
  uint64_t count_eq = (uint64_t)std::ceil(3.0 * (1.0 / 60.0) * 1.0 * (float)nz_docs / (float)num_topics);
  if (count_gr == 0) count_gr = 1;
  if (count_eq == 0) count_eq = 1;
  HIPCHK(c, c->zetas.reserve(V));
  ISLECHK(k_th_zetas(c, maxv, count_gr, count_eq));

  // survivors per document
  HIPCHK(c, c->a_kept.reserve(D ? D : 1));
  if (sampling) HIPCHK(c, c->a_wgt.reserve(D ? D : 1));
  ISLECHK(k_th_count(c, sampling));
  ISLECHK(k_th_scans(c));
  int64_t above_local = 0;
  HIPCHK(c, hipMemcpyAsync(&above_local, c->a_off_all.p + D, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (entries_above) {
    uint64_t g = (uint64_t)above_local;
    if (c->multi()) {
      HIPCHK(c, hipMemcpyAsync(st_dev, &g, sizeof(g), hipMemcpyHostToDevice, c->stream));
      ISLECHK(allreduce_sum<uint64_t>(c, st_dev, 1));
      HIPCHK(c, hipMemcpyAsync(&g, st_dev, sizeof(g), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    *entries_above = g;
  }

  if (sampling && D) {  // sampled_threshold_and_copy, src/sparseMatrix.cpp:1383-1415 (keys on the host, like the reference)
    std::vector<float> wgt(D), key(D), dice(D);
    HIPCHK(c, hipMemcpy(wgt.data(), c->a_wgt.p, D * sizeof(float), hipMemcpyDeviceToHost));
    for (uint64_t d = 0; d < D; ++d) {
      uint64_t z = (sample_seed + 1) * 0x9E3779B97F4A7C15ull ^ (d * 0xD1342543DE82EF95ull);
      z += 0x9E3779B97F4A7C15ull;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      z = z ^ (z >> 31);
      const double u = (double)(z >> 11) * (1.0 / 9007199254740992.0);
      key[d] = (wgt[d] == 0.f) ? 0.f : (float)std::pow(u, 1.0 / (double)wgt[d]);
      dice[d] = key[d];
    }
    const size_t nth = std::min<size_t>((size_t)((float)sample_rate * (float)D), D - 1);
    std::nth_element(dice.begin(), dice.begin() + nth, dice.end(), std::greater<float>());
    const float pivot = dice[nth];
    std::vector<uint8_t> drop(D);
    for (uint64_t d = 0; d < D; ++d) drop[d] = !(key[d] >= pivot);
    DevBuf<uint8_t> drop_dev;
    HIPCHK(c, drop_dev.reserve(D));
    HIPCHK(c, hipMemcpy(drop_dev.p, drop.data(), D, hipMemcpyHostToDevice));
    int rc = k_th_drop(c, drop_dev.p);
    if (rc == 0) rc = k_th_scans(c);
    (void)hipStreamSynchronize(c->stream);
    drop_dev.release();
    ISLECHK(rc);
  }

  int64_t tail[2];  // nnz(B), columns of B (local)
  HIPCHK(c, hipMemcpyAsync(&tail[0], c->a_off_all.p + D, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&tail[1], c->a_col_of.p + D, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const uint64_t bnnz = (uint64_t)tail[0], Db = (uint64_t)tail[1];

  // placement of this shard in B's global column numbering
  uint64_t b_off = 0, b_glob = Db;
  if (c->multi()) {
    DevBuf<uint64_t> all;
    HIPCHK(c, all.reserve((size_t)c->world + 1));
    HIPCHK(c, hipMemcpyAsync(all.p + c->world, &Db, sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    {
      TimeScope ts(c, ISLE_T_COMM);
      ISLECHK(isle_allgather(c, all.p + c->world, all.p, 1, ISLE_DT_U64));
    }
    std::vector<uint64_t> h(c->world);
    HIPCHK(c, hipMemcpyAsync(h.data(), all.p, c->world * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    all.release();
    b_glob = 0;
    for (int r = 0; r < c->world; ++r) {
      if (r == c->rank) b_off = b_glob;
      b_glob += h[r];
    }
  }

  c->V = V;
  c->D = Db;
  c->nnz = bnnz;
  c->doc_offset = b_off;
  c->D_global = b_glob;
  HIPCHK(c, c->vals.reserve(bnnz ? bnnz : 1));
  HIPCHK(c, c->rows.reserve(bnnz ? bnnz : 1));
  HIPCHK(c, c->offs.reserve(Db + 1));
  HIPCHK(c, c->original_cols.reserve(Db ? Db : 1));
  if (D == 0) HIPCHK(c, hipMemsetAsync(c->offs.p, 0, sizeof(int64_t), c->stream));
  ISLECHK(k_th_emit(c, c->a_doc_offset));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->band_ready = false;
  c->gl_mode = -1;
  c->P_ready = false;
  c->Pt_ready = false;
  c->lift_valid = false;
  c->members_valid = false;
  c->U_k = 0;
  c->centers_ready = false;
  c->assign_valid = false;
  c->p_catch_ready = false;
  c->p_model_ready = false;
  c->b_from_threshold = true;
  if (docs_kept) *docs_kept = Db;
  if (nnz_kept) *nnz_kept = bnnz;
  return 0;
}

extern "C" int isle_hip_shape(isle_ctx* c, uint64_t* V, uint64_t* D, uint64_t* nnz, uint64_t* doc_offset, uint64_t* docs_global) {
  if (!c) return ISLE_E_ARG;
  if (V) *V = c->V;
  if (D) *D = c->D;
  if (nnz) *nnz = c->nnz;
  if (doc_offset) *doc_offset = c->doc_offset;
  if (docs_global) *docs_global = c->D_global;
  return 0;
}

extern "C" int isle_hip_get_B(isle_ctx* c, float* vals, uint32_t* rows, int64_t* offs, uint64_t* original_cols, float* zetas) {
  if (!c) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  if (c->V == 0) return isle_fail(c, ISLE_E_ARG, "get_B: no matrix");
  if ((original_cols || zetas) && !c->b_from_threshold)
    return isle_fail(c, ISLE_E_ARG, "get_B: original_cols / zetas exist only after isle_hip_threshold");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (vals && c->nnz) HIPCHK(c, hipMemcpy(vals, c->vals.p, c->nnz * sizeof(float), hipMemcpyDeviceToHost));
  if (rows && c->nnz) HIPCHK(c, hipMemcpy(rows, c->rows.p, c->nnz * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (offs) HIPCHK(c, hipMemcpy(offs, c->offs.p, (c->D + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
  if (original_cols && c->D) HIPCHK(c, hipMemcpy(original_cols, c->original_cols.p, c->D * sizeof(uint64_t), hipMemcpyDeviceToHost));
  if (zetas) HIPCHK(c, hipMemcpy(zetas, c->zetas.p, c->V * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

// ------------------------------------------------------------------------------------------
// downstream stage: catchwords, topic model, edge topics
// ------------------------------------------------------------------------------------------
static int post_prepare(isle_ctx* c, const char* who) {
  if (!c->a_ready) return isle_fail(c, ISLE_E_ARG, "%s: no count matrix uploaded (isle_hip_upload_counts_u32)", who);
  if (c->world > 1) return isle_fail(c, ISLE_E_ARG, "%s: single-rank only", who);
  return 0;
}

extern "C" int isle_hip_catchwords(isle_ctx* c, int num_topics, const uint32_t* assign, uint64_t r, double rho, float* thresholds,
                                   int32_t* catch_topic, uint64_t* num_catchwords) {
  if (!c) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  ISLECHK(post_prepare(c, "catchwords"));
  if (num_topics < 1) return isle_fail(c, ISLE_E_ARG, "catchwords: num_topics < 1");
  if (r < 1 || r > 0xfffffff0ull) return isle_fail(c, ISLE_E_ARG, "catchwords: rank r = %llu out of range (too few documents per topic?)",
                                                    (unsigned long long)r);
  const bool identity = !c->b_from_threshold;
  if (identity && c->D != c->a_D) return isle_fail(c, ISLE_E_ARG, "catchwords: B was uploaded separately and its columns do not match A's");
  if (assign) {
    for (uint64_t j = 0; j < c->D; ++j)
      if (assign[j] >= (uint32_t)num_topics) return isle_fail(c, ISLE_E_ARG, "catchwords: assign[%llu] out of range", (unsigned long long)j);
    HIPCHK(c, c->assign.reserve(c->D ? c->D : 1));
    if (c->D) HIPCHK(c, hipMemcpy(c->assign.p, assign, c->D * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->assign_valid = true;
    c->members_valid = false;
  } else if (!c->assign_valid) {
    return isle_fail(c, ISLE_E_ARG, "catchwords: no partition resident (run isle_hip_lloyds_sparse or pass assign)");
  }
  if (!c->a_avg_valid) {  // B came from the host: the corpus statistics were never computed here
    HIPCHK(c, c->a_scan.reserve(isle_scan_scratch(c->a_D) + 4));
    ISLECHK(k_th_stats(c, (uint64_t*)c->a_scan.p));
    uint64_t st[2];
    HIPCHK(c, hipMemcpyAsync(st, c->a_scan.p, sizeof(st), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->a_avg = (float)(st[0] / std::max<uint64_t>(st[1], 1));
    c->a_avg_valid = true;
  }
  ISLECHK(k_post_normalize(c, c->a_avg));
  ISLECHK(k_post_cluster_of(c, c->assign.p, identity));
  HIPCHK(c, c->counts.reserve(num_topics));
  ISLECHK(k_count_sizes(c, c->assign.p, c->D, num_topics, c->counts.p));
  ISLECHK(k_post_catch_thresholds(c, (uint32_t)num_topics, (uint32_t)r, c->counts.p));
  uint64_t nc = 0;
  ISLECHK(k_post_find_catchwords(c, (uint32_t)num_topics, rho, &nc));
  if (num_catchwords) *num_catchwords = nc;
  c->p_k = num_topics;
  c->p_catch_ready = true;
  c->p_model_ready = false;
  if (thresholds) {
    HIPCHK(c, c->p_segvals.reserve((size_t)c->a_V * num_topics));
    ISLECHK(k_post_thr_colmajor(c, (uint32_t)num_topics, c->p_segvals.p));
    HIPCHK(c, hipMemcpyAsync(thresholds, c->p_segvals.p, (size_t)c->a_V * num_topics * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  }
  if (catch_topic) HIPCHK(c, hipMemcpyAsync(catch_topic, c->p_catch.p, c->a_V * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int isle_hip_topic_model(isle_ctx* c, int num_topics, uint64_t rank_threshold, float* model, float* model_threshold, int32_t* top1,
                                    int32_t* top2, uint64_t* doc_topic_sums) {
  if (!c) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  ISLECHK(post_prepare(c, "topic_model"));
  if (!c->p_catch_ready || c->p_k != num_topics) return isle_fail(c, ISLE_E_ARG, "topic_model: run isle_hip_catchwords(num_topics = %d) first", num_topics);
  if (rank_threshold < 1 || rank_threshold > 0xfffffff0ull) return isle_fail(c, ISLE_E_ARG, "topic_model: rank_threshold out of range");  // :721
  uint64_t n = 0;
  ISLECHK(k_post_doc_topic_sums(c, (uint32_t)num_topics, &n));
  ISLECHK(k_post_model_thresholds(c, (uint32_t)num_topics, (uint32_t)rank_threshold));
  ISLECHK(k_post_model(c, (uint32_t)num_topics));
  c->p_model_ready = true;
  if (doc_topic_sums) *doc_topic_sums = n;
  if (model) HIPCHK(c, hipMemcpyAsync(model, c->p_model.p, (size_t)c->a_V * num_topics * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  if (model_threshold) HIPCHK(c, hipMemcpyAsync(model_threshold, c->p_mthr.p, num_topics * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  if (top1 && c->a_D) HIPCHK(c, hipMemcpyAsync(top1, c->p_top1.p, c->a_D * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  if (top2 && c->a_D) HIPCHK(c, hipMemcpyAsync(top2, c->p_top2.p, c->a_D * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int isle_hip_get_doc_topic_sums(isle_ctx* c, int64_t* doc_offsets, uint32_t* topic, float* val) {
  if (!c) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  if (!c->p_model_ready) return isle_fail(c, ISLE_E_ARG, "get_doc_topic_sums: run isle_hip_topic_model first");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (doc_offsets) HIPCHK(c, hipMemcpy(doc_offsets, c->p_dts_off.p, (c->a_D + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
  if (topic && c->p_dts_n) HIPCHK(c, hipMemcpy(topic, c->p_dts_topic.p, c->p_dts_n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (val && c->p_dts_n) HIPCHK(c, hipMemcpy(val, c->p_dts_val.p, c->p_dts_n * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int isle_hip_edge_topics(isle_ctx* c, const int64_t* pairs, int n, float primary_ratio, float* edge) {
  if (!c) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  if (!c->p_model_ready) return isle_fail(c, ISLE_E_ARG, "edge_topics: run isle_hip_topic_model first");
  if (n < 0 || (n && (!pairs || !edge))) return isle_fail(c, ISLE_E_ARG, "edge_topics: bad arguments");
  if (n == 0) return 0;
  for (int e = 0; e < 2 * n; ++e)
    if (pairs[e] < 0 || pairs[e] >= c->p_k) return isle_fail(c, ISLE_E_ARG, "edge_topics: topic id %lld out of range", (long long)pairs[e]);
  DevBuf<int64_t> pd;
  DevBuf<float> ed;
  HIPCHK(c, pd.reserve(2 * (size_t)n));
  hipError_t e1 = ed.reserve((size_t)c->a_V * n);
  if (e1 != hipSuccess) {
    pd.release();
    HIPCHK(c, e1);
  }
  int rc = 0;
  hipError_t he = hipMemcpy(pd.p, pairs, 2 * (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice);
  if (he == hipSuccess) rc = k_post_edge(c, pd.p, n, primary_ratio, (float)(1.0 - (double)primary_ratio), ed.p);
  if (he == hipSuccess && rc == 0) he = hipStreamSynchronize(c->stream);
  if (he == hipSuccess && rc == 0) he = hipMemcpy(edge, ed.p, (size_t)c->a_V * n * sizeof(float), hipMemcpyDeviceToHost);
  pd.release();
  ed.release();
  ISLECHK(rc);
  HIPCHK(c, he);
  return 0;
}

extern "C" int isle_hip_infer(isle_ctx* c, uint64_t V, int k, const float* model_by_word, uint64_t D, uint64_t nnz, const float* counts,
                              const uint32_t* rows, const int64_t* offs, int iters, float Lf, float avg_doc_sz, float* weights,
                              int32_t* top_topic, float* top_weight, float* llh, uint64_t* nconverged) {
  if (!c || !model_by_word || !offs || (nnz && (!counts || !rows))) return ISLE_E_ARG;
  if (iters < 1 || !(Lf > 0.f)) return isle_fail(c, ISLE_E_ARG, "infer: iters = %d, Lf = %g", iters, (double)Lf);
  if (offs[0] != 0 || (uint64_t)offs[D] != nnz) return isle_fail(c, ISLE_E_ARG, "infer: offsets do not span the %llu entries", (unsigned long long)nnz);
  ISLECHK(isle_enter(c));
  return k_infer(c, V, k, model_by_word, D, nnz, counts, rows, offs, iters, Lf, avg_doc_sz, weights, top_topic, top_weight, llh, nconverged);
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
static int panel_width(int b) { return 4 * ((b + 3) / 4); }  // BP in {4, 8, ..., 32}: one float4 lane per 4 columns

// Zcm (V x b col-major) = B (B^T Xcm) on device pointers, 1 <= b <= 32, one pass over both copies of B per call.
static int gram_apply_dev(isle_ctx* c, const float* Xcm, int b, float* Zcm) {
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
// Panel QR: rank-revealing CholQR2 on an fp64 Gram matrix.
// utils::compute_qr (block-ks/ks_utils.h:43-127) is MGS in double with one DGKS correction and drops
// columns whose residual norm is < 1e-6.  Cholesky of G = F^T F processed column by column IS that MGS
// in exact arithmetic (pivot_i^2 = residual norm^2 of column i); a second pass restores orthogonality
// to working precision.  F: n x w (device, destroyed).  Q: n x rank written to Qdst.  R: rank x w.
// ------------------------------------------------------------------------------------------
static int dev_qr(isle_ctx* c, float* F, uint64_t n, int w, float* Qdst, std::vector<float>& R, int* rank_out) {
  std::vector<float> Rfull((size_t)w * w, 0.f);
  int rk = 0;
  ISLECHK(k_panel_qr(c, F, n, w, Qdst, Rfull.data(), &rk));
  ISLECHK(agree_i32(c, rk, "the rank of a start / repair block"));
  *rank_out = rk;
  R.assign(Rfull.begin(), Rfull.begin() + (size_t)rk * w);
  return 0;
}

// ------------------------------------------------------------------------------------------
// Restarted block Krylov-Schur
// ------------------------------------------------------------------------------------------
namespace {
struct Ks {
  isle_ctx* c;
  size_t nev, ncv, maxit, blk;
  uint64_t dim;
  float tol;
  HMat H;
  size_t vcols = 0, nconv = 0, n_restarts = 0, last_j = 0;
  long napplies = 0;
  uint64_t seed, draws = 0;
  // The ProdOp plug-in (block-ks/restarted_block_ks.h:18-40): the context's B B^T (MKL_SpSpTrProd, include/matUtils.h:336-365), or a
  // dense symmetric dim x dim matrix on the device (ArmaMatProdOp, block-ks/ks_utils.h:167-182) when dense_A is set.
  const float* dense_A = nullptr;
  const float* start_dev = nullptr;  // optional dim x blk start block (first try of init's draw loop)
  float* Vb() { return c->basis.p; }
  float* col(size_t j) { return c->basis.p + j * dim; }

  int randu(float* F, size_t cols) { return k_randu(c, F, dim * cols, seed + 0x1000 * (++draws)); }

  // orthogonalise F (dim x w) against the first m basis columns, `passes` times; coefficient blocks kept on device
  int ortho(float* F, int w, size_t m, int passes, float* coef_dev = nullptr) {
    HIPCHK(c, c->coef.reserve(3 * (c->basis.cap / dim) * 32));
    float* base = coef_dev ? coef_dev : c->coef.p;
    // Several ranks: the basis is replicated, but nobody needs to orthogonalise ALL rows.  Rank r takes rows [r nloc, (r+1) nloc):
    // its share of V^T F, an all-reduce of the m x w coefficients (80 kB at m = 2000), the update of its rows — for every pass —
    // and at the end the slices of F are all-gathered (4 MB at V = 100k), so every rank again holds the whole, bitwise equal F for
    // the replicated panel QR.  The step is HBM-bound on reading the basis (0.15 s of a 1.15 s C3-shard step): it now divides by
    // the number of ranks at the price of passes + 1 small collectives per step.  ISLE_KS_ROWSHARD=0 keeps it replicated.
    const bool shard = c->multi() && !dense_A && c->knob_on(KN_KS_ROWSHARD) && !c->knob_zero(KN_KS_ROWSHARD);
    if (!shard) {
      for (int p = 0; p < passes; ++p) {
        float* cf = base + (size_t)p * m * w;
        ISLECHK(k_vtf(c, Vb(), dim, (int)m, F, w, cf));
        ISLECHK(k_update(c, F, dim, w, Vb(), (int)m, cf));
      }
      return 0;
    }
    const uint64_t nloc = (((dim + c->world - 1) / c->world) + 3) & ~3ull;  // rows per rank, a multiple of 4 (16-byte aligned slices)
    const uint64_t r0 = std::min<uint64_t>(dim, (uint64_t)c->rank * nloc), r1 = std::min<uint64_t>(dim, r0 + nloc);
    const uint64_t nl = r1 - r0;
    for (int p = 0; p < passes; ++p) {
      float* cf = base + (size_t)p * m * w;
      ISLECHK(k_vtf(c, Vb() + r0, nl, (int)m, F + r0, w, cf, dim));
      ISLECHK(allreduce_sum<float>(c, cf, m * (size_t)w));
      ISLECHK(k_update(c, F + r0, nl, w, Vb() + r0, (int)m, cf, dim));
    }
    HIPCHK(c, c->ks_gather.reserve((size_t)c->world * nloc * w));
    float* mine = c->ks_gather.p + (size_t)c->rank * nloc * w;
    ISLECHK(k_slice_rows(c, F, dim, w, r0, nl, nloc, mine, true));  // pack my rows (zero padded to nloc)
    {
      TimeScope ts(c, ISLE_T_COMM);
      ISLECHK(isle_allgather(c, mine, c->ks_gather.p, nloc * (size_t)w, ISLE_DT_F32));
    }
    for (int r = 0; r < c->world; ++r) {
      const uint64_t q0 = std::min<uint64_t>(dim, (uint64_t)r * nloc), q1 = std::min<uint64_t>(dim, q0 + nloc);
      if (r != c->rank && q1 > q0) ISLECHK(k_slice_rows(c, F, dim, w, q0, q1 - q0, nloc, c->ks_gather.p + (size_t)r * nloc * w, false));
    }
    return 0;
  }

  // rank repair shared by init (:238-258) and expand (:106-132)
  int repair(size_t& nvecs, size_t target, size_t width) {
    size_t tries = 0;
    while (nvecs < target && tries < 100) {
      tries++;
      ISLECHK(randu(c->Fbuf.p, width));
      ISLECHK(ortho(c->Fbuf.p, (int)width, nvecs, 2));
      std::vector<float> R2;
      int rk2 = 0;
      const size_t room = target - nvecs;
      // Q lands in Tmp first: only `room` columns may be appended
      ISLECHK(dev_qr(c, c->Fbuf.p, dim, (int)width, c->Tmp.p, R2, &rk2));
      const size_t take = std::min<size_t>((size_t)rk2, room);
      if (take) HIPCHK(c, hipMemcpyAsync(col(nvecs), c->Tmp.p, take * dim * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
      nvecs += take;
    }
    if (nvecs < target) return isle_fail(c, ISLE_E_NUMERIC, "unable to find new starting basis for Arnoldi expansion");
    return 0;
  }

  int apply(const float* X, float* Z) {
    napplies++;
    if (dense_A) return k_gemm_nn(c, dense_A, dim, (int)dim, X, (int)dim, (int)blk, Z, ISLE_T_GRAM_PASS1);
    return gram_apply_dev(c, X, (int)blk, Z);
  }

  int init() {  // :203-259
    std::vector<float> R;
    int rank = 0;
    bool first = true;
    do {  // :211-218: redrawn until the start block has full rank
      if (first && start_dev) HIPCHK(c, hipMemcpyAsync(c->Fbuf.p, start_dev, dim * blk * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
      else ISLECHK(randu(c->Fbuf.p, blk));
      first = false;
      ISLECHK(dev_qr(c, c->Fbuf.p, dim, (int)blk, col(0), R, &rank));
    } while ((size_t)rank < blk);
    float* V1 = c->Fbuf.p;
    ISLECHK(apply(col(0), V1));
    ISLECHK(ortho(V1, (int)blk, blk, 2));  // H = V^T V1; V1 -= V H; C = V^T V1; H += C; V1 -= V C
    std::vector<float> hc(2 * blk * blk);
    HIPCHK(c, hipMemcpyAsync(hc.data(), c->coef.p, hc.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    ISLECHK(dev_qr(c, V1, dim, (int)blk, col(blk), R, &rank));  // synchronises the stream
    H = HMat(2 * blk, blk, std::max<size_t>(ncv, 2 * blk) + blk, blk + (std::max<size_t>(ncv, 2 * blk) + blk - 2 * blk));
    for (size_t j = 0; j < blk; ++j) {
      for (size_t i = 0; i < blk; ++i) H(i, j) = hc[j * blk + i] + hc[blk * blk + j * blk + i];
      for (int i = 0; i < rank; ++i) H(blk + i, j) = R[j * rank + i];
    }
    vcols = blk + rank;
    if ((size_t)rank < blk) ISLECHK(repair(vcols, 2 * blk, blk - rank));
    vcols = 2 * blk;
    return 0;
  }

  int expand() {  // :62-136
    // H grows by blk rows and columns per step; it is kept in a work matrix of the final size while the loop runs (copying
    // the whole of H at every step cost ~10 ms of host time per solve at ncv = 410, with the GPU idle behind the QR's sync)
    isle_host_mark("expand: entry");
    const size_t cap_r = std::max<size_t>(ncv, H.r) + blk, cap_c = H.c + (cap_r - H.r);
    if (H.ld < cap_r || H.cap_c < cap_c) {  // init() and truncate() allocate with this room, so this copy is the exception
      HMat W(H.r, H.c, cap_r, cap_c);
      for (size_t j = 0; j < H.c; ++j)
        for (size_t i = 0; i < H.r; ++i) W(i, j) = H(i, j);
      H = std::move(W);
    }
    HMat& W = H;  // grows in place: rows hr.. and columns hcn.. are zero until a step writes them
    size_t hr = H.r, hcn = H.c;
    auto shrink = [&]() {
      H.r = hr;
      H.c = hcn;
    };
    // Pipelined: after the QR of step i is enqueued, the operator application and orthogonalisation of step i + 1 are
    // enqueued too (they only need Q on the device, assuming full rank), and the host then waits for an event recorded
    // behind the QR to fold R and the coefficients into H.  Everything the host needs from a step — rank and status of the QR,
    // R, the three coefficient blocks — is written into one device mailbox and comes back as ONE copy (every small copy
    // costs ~20 us of queue time).  A rank-deficient panel (never seen on thresholded matrices) discards the speculative
    // work and repairs, as the synchronous form (ISLE_KS_SYNC=1) does.
    const bool pipelined = !c->knob_on(KN_KS_SYNC);
    // Passes of block Gram-Schmidt against the basis per step.  The reference makes three (CGS + 2 DGKS, :83-91); the second
    // already leaves coefficients at rounding level ("twice is enough"), so two are made here and the third
    // block of coefficients that the reference adds into H is zero.  ISLE_KS_ORTHO_PASSES=3 restores the reference's count.
    int npass = 2;
    if (const char* e = c->knob(KN_KS_ORTHO_PASSES)) npass = std::max(2, std::min(3, atoi(e)));
    constexpr size_t MB_R = 64, MB_COEF = 64 + 32 * 32;  // mailbox offsets (floats): [meta ints | R | coefficients]
    const size_t mb_floats = MB_COEF + 3 * cap_r * blk;
    HIPCHK(c, c->ks_mail.reserve(mb_floats));
    for (int i = 0; i < 2; ++i)
      if (!c->ks_ev[i]) HIPCHK(c, hipEventCreateWithFlags(&c->ks_ev[i], hipEventDisableTiming));
    std::vector<float> host_mail_pageable[2];
    float* host_mail[2];
    for (int i = 0; i < 2; ++i) {
      if (mb_floats * sizeof(float) <= isle_ctx::PIN_MAIL_SLOT) {  // page-locked: the copy below then really is asynchronous
        host_mail[i] = reinterpret_cast<float*>(c->pin + isle_ctx::PIN_MAIL + (size_t)i * isle_ctx::PIN_MAIL_SLOT);
      } else {
        host_mail_pageable[i].resize(mb_floats);
        host_mail[i] = host_mail_pageable[i].data();
      }
    }
    isle_host_mark("expand: work matrix ready");
    float* mail = c->ks_mail.p;
    bool spec = false;  // apply + ortho of the current step already enqueued
    int slot = 0;
    while (hr < ncv) {
      const size_t m = hr;
      float* F = c->Fbuf.p;
      if (!spec) {
        ISLECHK(apply(col(hcn), F));
        ISLECHK(ortho(F, (int)blk, m, npass, mail + MB_COEF));
      }
      spec = false;
      if (m + blk > cap_r || hcn + blk > cap_c) return isle_fail(c, ISLE_E_NUMERIC, "expand: projected matrix outgrew its work space");
      ISLECHK(k_panel_qr_kernels(c, F, dim, (int)blk, col(hcn + blk), reinterpret_cast<int*>(mail), mail + MB_R));
      if (c->multi()) {  // the ranks' verdicts on this block travel with the mailbox (see agree_i32)
        hipLaunchKernelGGL(ks_agree_pack_k, dim3(1), dim3(64), 0, c->stream, reinterpret_cast<int*>(mail));
        HIPCHK(c, hipGetLastError());
        TimeScope ts(c, ISLE_T_COMM);
        ISLECHK(isle_allreduce(c, reinterpret_cast<int*>(mail) + KS_AGREE, 3, ISLE_DT_I32, true));
      }
      float* hm = host_mail[slot];
      HIPCHK(c, hipMemcpyAsync(hm, mail, (MB_COEF + (size_t)npass * m * blk) * sizeof(float), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipEventRecord(c->ks_ev[slot], c->stream));
      const bool more = m + blk < ncv;
      if (pipelined && more) {  // speculate: full rank -> next step works on the blk new columns with m + blk basis vectors
        ISLECHK(apply(col(hcn + blk), F));
        ISLECHK(ortho(F, (int)blk, m + blk, npass, mail + MB_COEF));  // behind the copy on the same stream: no hazard
        spec = true;
      }
      HIPCHK(c, hipEventSynchronize(c->ks_ev[slot]));
      const int* meta = reinterpret_cast<const int*>(hm);
      if (c->multi() && meta[KS_AGREE] != -meta[KS_AGREE + 1])
        return isle_fail(c, ISLE_E_COMM, "ranks disagree on the rank of a Krylov block (min %d, max %d): replicated state diverged",
                         -meta[KS_AGREE + 1], meta[KS_AGREE]);
      if (meta[1] == 2) {  // the speculative next step has already overwritten the block: give up on this solve, the next one runs the five kernels
        ISLECHK(k_panel_qr_fused_lost(c));
        return isle_fail(c, ISLE_E_NUMERIC, "panel QR: the persistent kernel lost residency at a grid barrier; this context now uses the five-kernel form");
      }
      if (meta[1] || (c->multi() && meta[KS_AGREE + 2]))
        return isle_fail(c, ISLE_E_NUMERIC, "CholQR2: second Gram matrix not positive definite");
      const int rk = meta[0];
      const float* hc = hm + MB_COEF;
      const float* Rfull = hm + MB_R;
      for (size_t j = 0; j < blk; ++j)
        for (size_t i = 0; i < m; ++i) {
          float h = hc[j * m + i];
          h = h + hc[m * blk + j * m + i];
          if (npass > 2) h = h + hc[2 * m * blk + j * m + i];
          W(i, hcn + j) = h;
        }
      for (size_t j = 0; j < blk; ++j)
        for (int i = 0; i < rk; ++i) W(m + i, hcn + j) = Rfull[j * rk + i];
      hr = m + blk;
      hcn += blk;
      if ((size_t)rk < blk) {
        if (spec) {  // the speculative step used columns that are about to be replaced
          HIPCHK(c, hipStreamSynchronize(c->stream));
          spec = false;
          napplies--;
        }
        shrink();  // repair() reads H.r / H.c
        size_t nvecs = H.c + rk;
        ISLECHK(repair(nvecs, H.r, blk - rk));
      }
      slot ^= 1;
    }
    isle_host_mark("expand: loop done");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    isle_host_mark("expand: synchronised");
    shrink();
    isle_host_mark("expand: shrink");
    vcols = H.r;
    return 0;
  }

  int truncate() {  // :138-187
    isle_host_mark("truncate: entry");
    const size_t n = H.c - nconv;
    const size_t keep = nev - nconv;  // only the leading `keep` eigenvectors are used below
    // Everything that crosses the bus here lives in the context's page-locked staging area — [subH n x n | vH n x keep | locked
    // rows of H nconv x n | top nconv x keep], 36 MB at k = 1000: copies from freshly allocated pageable vectors blocked the host
    // (registration with the driver) and made their release slow, with the GPU idle in between.
    HIPCHK(c, c->pin_stage_reserve((n * n + n * keep + nconv * n + nconv * keep) * sizeof(float)));
    float* subH = reinterpret_cast<float*>(c->pin_stage);
    float* vH = subH + n * n;
    float* blkH = vH + n * keep;
    float* top = blkH + nconv * n;
    for (size_t j = 0; j < n; ++j) memcpy(subH + j * n, &H(nconv, nconv + j), n * sizeof(float));  // H(nconv:, nconv:), square
    isle_host_mark("truncate: subH extracted");
    std::vector<float> eH(n);
    HIPCHK(c, c->Wf.reserve(n * n));
    ISLECHK(k_eig_small(c, subH, (int)n, eH.data(), c->Wf.p, (int)keep));
    isle_host_mark("truncate: eig_small returned");
    HIPCHK(c, hipMemcpyAsync(vH, c->Wf.p, n * keep * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    // V = [ V(:, :nconv) | V(:, nconv : ncols-blk) * vH(:, :keep) | V(:, tail blk) ]
    ISLECHK(k_gemm_nn(c, col(nconv), dim, (int)n, c->Wf.p, (int)n, (int)keep, c->Tmp.p));
    // top = H(0:nconv, nconv:) * vH(:, :keep)  (:176-178), the coupling of the locked columns with the rotated block: nconv x n x keep
    // multiply-adds — 0.3 G at k = 1000 with 600 pairs locked, 31 ms of host time with the GPU idle when it was a host loop
    if (nconv > 0) {
      HIPCHK(c, c->ks_top.reserve(nconv * n + nconv * keep));
      for (size_t t = 0; t < n; ++t) memcpy(blkH + t * nconv, &H(0, nconv + t), nconv * sizeof(float));
      HIPCHK(c, hipMemcpyAsync(c->ks_top.p, blkH, nconv * n * sizeof(float), hipMemcpyHostToDevice, c->stream));
      ISLECHK(k_gemm_nn(c, c->ks_top.p, nconv, (int)n, c->Wf.p, (int)n, (int)keep, c->ks_top.p + nconv * n));
      HIPCHK(c, hipMemcpyAsync(top, c->ks_top.p + nconv * n, nconv * keep * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipMemcpyAsync(c->Fbuf.p, col(vcols - blk), blk * dim * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(col(nconv), c->Tmp.p, keep * dim * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(col(nev), c->Fbuf.p, blk * dim * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    vcols = nev + blk;
    isle_host_mark("truncate: device part synchronised");
    // Transform H (:169-184)
    auto vh = [&](size_t i, size_t j) { return vH[j * n + i]; };
    HMat last = hsub(H, H.r - blk, H.c - blk, H.r - 1, H.c - 1);  // blk x blk
    HMat newrows(blk, keep);
    for (size_t j = 0; j < keep; ++j)
      for (size_t t = 0; t < blk; ++t) {
        const float v = vh(n - blk + t, j);
        for (size_t i = 0; i < blk; ++i) newrows(i, j) += last(i, t) * v;
      }
    const size_t grow_r = std::max<size_t>(ncv, nev + blk) + blk;  // what the next expand() asks for
    HMat Hn(nev + blk, nev, grow_r, nev + (grow_r - (nev + blk)));
    for (size_t j = 0; j < nconv; ++j) {  // locked columns keep their entries (rows < nev from the old H; residual rows too)
      for (size_t i = 0; i < nev; ++i) Hn(i, j) = H(i, j);
      for (size_t i = 0; i < blk; ++i) Hn(nev + i, j) = H(nev + i, j);
    }
    for (size_t j = nconv; j < nev; ++j) {
      Hn(j, j) = eH[j - nconv];
      for (size_t i = 0; i < blk; ++i) Hn(nev + i, j) = newrows(i, j - nconv);
      for (size_t i = 0; i < nconv; ++i) Hn(i, j) = top[(j - nconv) * nconv + i];
    }
    H = std::move(Hn);
    isle_host_mark("truncate: H transformed");
    return 0;
  }

  size_t first_unconverged(bool divide) const {  // :278-293
    for (size_t j = 0; j < H.c; ++j) {
      float s = 0.f;
      for (size_t i = H.r - blk; i < H.r; ++i) s += H(i, j) * H(i, j);
      float nrm = std::sqrt(s);
      if (divide) nrm = nrm / H(j, j);
      if (nrm >= tol) return j;
    }
    return H.c;
  }

  int compute() {  // :261-321
    n_restarts = 0;
    nconv = 0;
    ISLECHK(expand());
    while (n_restarts < maxit) {
      ISLECHK(truncate());
      const size_t j = first_unconverged(true);
      ISLECHK(agree_i32(c, (int)j, "the number of converged Ritz pairs"));
      last_j = j;
      if (j == H.c) {
        nconv = H.c;
        break;
      }
      nconv = j;
      ++n_restarts;
      ISLECHK(expand());
    }
    return 0;
  }
};
}  // namespace

static int install_U(isle_ctx* c, const float* Ucm_dev, int k) {
  c->ldk = round4(k);
  HIPCHK(c, c->Ucm.reserve((size_t)c->V * k));
  HIPCHK(c, c->Urm.reserve((size_t)c->V * c->ldk));
  if (Ucm_dev != c->Ucm.p)
    HIPCHK(c, hipMemcpyAsync(c->Ucm.p, Ucm_dev, (size_t)c->V * k * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(c->Urm.p, 0, (size_t)c->V * c->ldk * sizeof(float), c->stream));
  ISLECHK(k_transpose(c, c->Ucm.p, c->V, k, c->V, c->Urm.p, c->ldk));  // compute_U_rowmajor :1223-1231
  c->U_k = k;
  c->P_ready = false;
  c->Pt_ready = false;
  c->lift_valid = false;
  c->centers_ready = false;
  return 0;
}

// Shared driver of both eigensolver entries: BlockKs(op, nev, ncv, maxit, blk, tol); init(); compute()  (:190-321).
// ncv and nev need not be multiples of the block size: a decomposition grows by whole blocks until it has AT LEAST ncv
// rows (the reference sizes V for exactly ncv columns and overruns it in that case), so the basis holds up to
// ncv + blk - 1 vectors.
static int ks_solve(isle_ctx* c, Ks& ks, int nev, int ncv, int maxit, int blk, float tol, uint64_t seed, float* evals, int* nconv,
                    int* restarts, int* napplies, int* nconv_ref_rule) {
  if (nev < 1 || blk < 1 || blk > 32 || maxit < 1) return isle_fail(c, ISLE_E_ARG, "bad nev/blk/maxit (nev >= 1, 1 <= blk <= 32, maxit >= 1)");
  ks.c = c;
  ks.nev = nev;
  ks.ncv = ncv;
  ks.maxit = maxit;
  ks.blk = (blk < nev) ? blk : 1;  // block-ks/restarted_block_ks.h:198
  ks.tol = tol;
  ks.seed = seed;
  if ((size_t)ncv < (size_t)nev + 2 * ks.blk || (uint64_t)ncv + ks.blk > ks.dim)
    return isle_fail(c, ISLE_E_ARG, "need nev + 2*blk <= ncv and ncv + blk <= operator dimension (nev=%d ncv=%d blk=%zu dim=%llu)", nev, ncv,
                     ks.blk, (unsigned long long)ks.dim);
  HIPCHK(c, c->basis.reserve((size_t)ks.dim * (ncv + 2 * ks.blk)));
  HIPCHK(c, c->Fbuf.reserve((size_t)ks.dim * ks.blk));
  HIPCHK(c, c->Tmp.reserve((size_t)ks.dim * std::max<size_t>(nev, ks.blk)));
  isle_host_mark("ks_solve: entry");
  ISLECHK(ks.init());
  isle_host_mark("ks_solve: init done");
  ISLECHK(ks.compute());
  isle_host_mark("ks_solve: compute done");
  int rc = 0;
  size_t nc = ks.nconv, nc_ref = ks.nconv;
  if (ks.n_restarts == (size_t)maxit) {
    // The reference recomputes residuals from the EXPANDED H without dividing by the Ritz value (:303-317); the last blk rows of
    // an expanded H are [0 ... 0 R], so that rule reports min(first column of the last block, nev) = nev whatever happened
    // (SURVEY App. C #7).  Here: the count of the last restart's residual test, status ISLE_E_NOCONV, and the same Ritz pairs;
    // the reference's figure is available through nconv_ref_rule.
    nc_ref = ks.first_unconverged(false);
    nc = std::min(ks.last_j, (size_t)nev);
    if (nc < (size_t)nev) rc = ISLE_E_NOCONV;
  }
  nc = std::min(nc, (size_t)nev);
  nc_ref = std::min(nc_ref, (size_t)nev);
  for (int i = 0; i < nev; ++i) evals[i] = ks.H(i, i);  // src/sparseMatrix.cpp:1212-1213
  if (nconv) *nconv = (int)nc;
  if (nconv_ref_rule) *nconv_ref_rule = (int)nc_ref;
  if (restarts) *restarts = (int)ks.n_restarts;
  if (napplies) *napplies = (int)ks.napplies;
  return rc;
}

extern "C" int isle_hip_block_ks(isle_ctx* c, int nev, int ncv, int maxit, int blk, float tol, uint64_t seed, float* evals, int* nconv,
                                 int* restarts, int* napplies) {
  if (!c || !evals) return ISLE_E_ARG;
  if (c->V == 0) return isle_fail(c, ISLE_E_ARG, "no matrix uploaded");
  ISLECHK(isle_enter(c));
  Ks ks;
  ks.dim = c->V;
  c->band_ready = false;  // the operator (CSR copy) is rebuilt per solve, as in src/sparseMatrix.cpp:1199
  int nc = 0;
  const int rc = ks_solve(c, ks, nev, ncv, maxit, blk, tol, seed, evals, &nc, restarts, napplies, nullptr);
  if (nconv) *nconv = nc;
  if (rc != 0 && rc != ISLE_E_NOCONV) return rc;
  ISLECHK(install_U(c, c->basis.p, nev));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  isle_host_mark("block_ks: U installed, exit");
  if (rc == ISLE_E_NOCONV) return isle_fail(c, rc, "block KS: %d restarts exhausted, %d of %d Ritz pairs converged", maxit, nc, nev);
  return 0;
}

extern "C" int isle_hip_block_ks_dense(isle_ctx* c, const float* A, uint64_t n, int nev, int ncv, int maxit, int blk, float tol,
                                       uint64_t seed, const float* start_block, float* evals, float* U, int* nconv, int* nconv_ref_rule,
                                       int* restarts, int* napplies) {
  if (!c || !A || !evals || n < 2 || n > 46340) return isle_fail(c, ISLE_E_ARG, "block_ks_dense: bad arguments (2 <= n <= 46340)");
  ISLECHK(isle_enter(c));
  if (c->multi()) return isle_fail(c, ISLE_E_ARG, "block_ks_dense: the dense operator is not sharded (single rank only)");
  const int b_eff = (blk < nev) ? blk : 1;
  DevBuf<float> Adev, Sdev;
  HIPCHK(c, Adev.reserve((size_t)n * n));
  HIPCHK(c, hipMemcpy(Adev.p, A, (size_t)n * n * sizeof(float), hipMemcpyHostToDevice));
  Ks ks;
  ks.dim = n;
  ks.dense_A = Adev.p;
  if (start_block && b_eff >= 1) {
    HIPCHK(c, Sdev.reserve((size_t)n * b_eff));
    HIPCHK(c, hipMemcpy(Sdev.p, start_block, (size_t)n * b_eff * sizeof(float), hipMemcpyHostToDevice));
    ks.start_dev = Sdev.p;
  }
  int nc = 0;
  const int rc = ks_solve(c, ks, nev, ncv, maxit, blk, tol, seed, evals, &nc, restarts, napplies, nconv_ref_rule);
  if (nconv) *nconv = nc;
  hipError_t he = hipStreamSynchronize(c->stream);
  if (he == hipSuccess && (rc == 0 || rc == ISLE_E_NOCONV) && U)
    he = hipMemcpy(U, c->basis.p, (size_t)n * nev * sizeof(float), hipMemcpyDeviceToHost);
  HIPCHK(c, he);
  if (rc == ISLE_E_NOCONV) return isle_fail(c, rc, "block KS (dense operator): %d restarts exhausted, %d of %d Ritz pairs converged", maxit, nc, nev);
  return rc;
}

extern "C" int isle_hip_get_U(isle_ctx* c, float* U) {
  if (!c || !U || c->U_k == 0) return isle_fail(c, ISLE_E_ARG, "no U available");
  ISLECHK(isle_enter(c));
  HIPCHK(c, hipMemcpyAsync(U, c->Ucm.p, (size_t)c->V * c->U_k * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
extern "C" int isle_hip_set_U(isle_ctx* c, const float* U, int k) {
  if (!c || !U || k < 1 || c->V == 0) return isle_fail(c, ISLE_E_ARG, "set_U: bad arguments");
  ISLECHK(isle_enter(c));
  HIPCHK(c, c->Ucm.reserve((size_t)c->V * k));
  HIPCHK(c, hipMemcpy(c->Ucm.p, U, (size_t)c->V * k * sizeof(float), hipMemcpyHostToDevice));
  ISLECHK(install_U(c, c->Ucm.p, k));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int isle_hip_eig_sym(isle_ctx* c, const float* S, int n, float* evals, float* vecs) {
  if (!c || !S || !evals || !vecs || n < 1) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  HIPCHK(c, c->Wf.reserve((size_t)n * n));
  ISLECHK(k_eig_small(c, S, n, evals, c->Wf.p, n));
  HIPCHK(c, hipMemcpy(vecs, c->Wf.p, (size_t)n * n * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

// ------------------------------------------------------------------------------------------
// k-means in the projected space
// ------------------------------------------------------------------------------------------
static int ensure_P(isle_ctx* c, int k) {
  if (c->U_k != k) return isle_fail(c, ISLE_E_ARG, "U has %d columns, k = %d (run isle_hip_block_ks / set_U first)", c->U_k, k);
  if (c->P_ready) return 0;
  const size_t D = c->D ? c->D : 1;
  HIPCHK(c, c->P.reserve(D * c->ldk));
  HIPCHK(c, c->pnorm.reserve(D));
  ISLECHK(k_spmm_wide_project(c, c->Urm.p, k, c->ldk, c->P.p, c->pnorm.p));
  c->P_ready = true;
  c->P_gen++;
  c->Pt_ready = false;
  if (c->D) {  // coordinate-major copy for the register-resident MFMA distance kernels
    HIPCHK(c, c->Pt.reserve((size_t)c->D * c->ldk));
    ISLECHK(k_transpose(c, c->P.p, c->ldk, c->D, c->ldk, c->Pt.p, c->D));
    c->Pt_ready = true;
  }
  return 0;
}

// dst (n x ldk, device) <- P rows of the given GLOBAL doc ids (owner contributes, others zero, then all-reduce)
static int fetch_rows(isle_ctx* c, const uint64_t* ids, int n, float* dst) {
  if (n == 0) return 0;
  const bool multi = c->multi();
  std::vector<uint64_t> local(n);
  for (int i = 0; i < n; ++i) {
    const uint64_t g = ids[i];
    if (g >= c->doc_offset && g < c->doc_offset + c->D) local[i] = g - c->doc_offset;
    else if (multi) local[i] = ~0ull;  // another rank's document: zeros here, the all-reduce brings the row
    else return isle_fail(c, ISLE_E_ARG, "seed doc id %llu out of range", (unsigned long long)g);
  }
  ISLECHK(k_fetch_rows(c, c->P.p, c->ldk, local.data(), n, dst));  // one kernel (the ids travel as arguments), not one copy per row
  if (multi) ISLECHK(allreduce_sum<float>(c, dst, (size_t)n * c->ldk));
  return 0;
}

extern "C" int isle_hip_kmeanspp_projected(isle_ctx* c, int k, const uint64_t* inject, uint64_t rng_seed, uint64_t* seeds_out,
                                           float* C_lowd, float* residual, int* rounds_out) {
  if (!c || !seeds_out || !C_lowd || k < 1) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  if ((uint64_t)k > c->D_global) return isle_fail(c, ISLE_E_ARG, "k > number of documents");
  isle_host_mark("kmeanspp: entry");
  ISLECHK(ensure_P(c, k));  // compute_projected_docs_l2sq :2144
  isle_host_mark("kmeanspp: projection enqueued");
  const uint64_t D = c->D, Dg = c->D_global;
  const int ldk = c->ldk;
  const bool multi = c->multi();
  HIPCHK(c, c->min_dist.reserve(D ? D : 1));
  HIPCHK(c, c->cum.reserve(D + 1));
  HIPCHK(c, c->Cdev.reserve((size_t)k * ldk));
  HIPCHK(c, c->gram.reserve(1024));
  HIPCHK(c, c->small.reserve(4096));
  ISLECHK(k_fill_f32(c, c->min_dist.p, D, 3.402823466e+38f));  // FP_MAX :2148
  HostRng rng(rng_seed);
  std::vector<uint64_t> centers;
  const uint64_t first = inject ? inject[0] : (uint64_t)(((size_t)rng.next31() * (size_t)84619573) % (size_t)Dg);  // :2150
  centers.push_back(first);
  ISLECHK(fetch_rows(c, &first, 1, c->Cdev.p));
  int new_added = 1, rounds = 0;
  double grand = 0.0, last_md = 0.0;
  // k > 224 (Lloyd in span(U) keeps tile bounds): the rounds also keep every document's nearest seed and tile minima, so that Lloyd's
  // first assignment — a D x k x k pass against exactly these seeds — need not be computed again (kmeans.hip kmpp_min_dots_track_k)
  const bool track = k > 224 && (k + 31) / 32 <= 32 && !c->knob_zero(KN_KMPP_TRACK);
  c->kmpp_track_k = 0;
  const int maxdraw = 2 + (int)std::ceil(std::sqrt((double)k));
  std::vector<double> dice(maxdraw);
  // page-locked staging for the per-round scalars: [my 2 | tot 2 * world | local maxdraw] doubles, then drawn maxdraw u64
  double* pin_d = reinterpret_cast<double*>(c->pin + isle_ctx::PIN_SMALL);
  if ((size_t)(2 + 2 * c->world + 2 * maxdraw + 42) * 8 > (128u << 10)) return isle_fail(c, ISLE_E_ARG, "k-means++: staging area too small");
  double* my = pin_d;
  double* tot = pin_d + 2;
  double* local = tot + 2 * c->world;
  uint64_t* drawn = reinterpret_cast<uint64_t*>(local + maxdraw);
  while ((int)centers.size() < k) {
    rounds++;
    ISLECHK(k_kmpp_update(c, c->P.p, c->pnorm.p, D, k, ldk, c->Cdev.p + (centers.size() - new_added) * (size_t)ldk, new_added,
                          c->min_dist.p, (int)(centers.size() - new_added), track));
    ISLECHK(k_scan_f2d(c, c->min_dist.p, D, c->cum.p));  // :2170-2172 (double, parallel; the reference's is fp32 sequential)
    const int s = (int)centers.size();
    int ndraw = 0;
    for (int cc = 0; cc < 1 + std::sqrt((double)(s - 5 > 0 ? s - 5 : 0)); ++cc) ndraw++;  // :2183 (upper bound on draws)
    ndraw = std::min(ndraw, maxdraw);
    if (!multi && !inject && ndraw <= 40 && !c->knob_on(KN_KMPP_HOST_DICE)) {  // the switch: for the test that holds both forms to the same seeds
      // one rank: the dice are products of the total with host-drawn fractions, so the device can throw them itself — the totals, the
      // dice and their search come back in one copy (search_frac_k), one host round trip per round
      for (int i = 0; i < ndraw; ++i) dice[i] = rng.fraction();  // :2184
      uint64_t* res = drawn + maxdraw;  // page-locked, 42 entries
      // the kernel writes its 42 words straight into the page-locked area (host memory mapped into the device's address space): no
      // copy kernel, and one gap less, between the search and the host's wake-up
      ISLECHK(k_search_frac(c, c->cum.p, D, D > 0 ? c->min_dist.p + (D - 1) : nullptr, dice.data(), ndraw, res));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      memcpy(my, res + 40, 2 * sizeof(double));
      grand = my[0];
      last_md = my[1];
      for (int i = 0; i < ndraw; ++i) drawn[i] = std::min<uint64_t>(res[i], D - 1) + c->doc_offset;
    } else {
      // totals (per rank) -> offsets
      my[0] = my[1] = 0.0;
      ISLECHK(k_pack2(c, c->cum.p + D, D > 0 ? c->min_dist.p + (D - 1) : nullptr, c->gram.p + 200));
      HIPCHK(c, hipMemcpyAsync(my, c->gram.p + 200, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));  // one copy, one round trip for both scalars
      for (int r = 0; r < 2 * c->world; ++r) tot[r] = 0.0;
      if (multi) {
        double* dv = c->gram.p;
        HIPCHK(c, hipMemcpyAsync(dv + 2 * c->world, my, 2 * sizeof(double), hipMemcpyHostToDevice, c->stream));
        {
          TimeScope ts(c, ISLE_T_COMM);
          ISLECHK(isle_allgather(c, dv + 2 * c->world, dv, 2, ISLE_DT_F64));
        }
        HIPCHK(c, hipMemcpyAsync(tot, dv, 2 * c->world * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
      } else {
        tot[0] = my[0];
        tot[1] = my[1];
      }
      grand = 0.0;
      double my_off = 0.0;
      for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) my_off = grand;
        grand += tot[2 * r];
      }
      last_md = tot[2 * (c->world - 1) + 1];
      if (!inject) {
        // all ranks draw the same dice; the owner of the interval searches its local prefix sums
        for (int i = 0; i < ndraw; ++i) {
          dice[i] = grand * rng.fraction();  // :2184
          const double x = dice[i] - my_off;
          const bool mine = (x >= 0.0 && x < my[0]) || (c->world == 1);
          local[i] = mine ? std::min(std::max(x, 0.0), my[0]) : -1.0;
        }
        double* dd = c->gram.p + 64;
        uint64_t* od = (uint64_t*)(c->gram.p + 128);
        if (ndraw <= 16) {
          ISLECHK(k_search_args(c, c->cum.p, D, local, ndraw, od));  // dice as kernel arguments
        } else {
          HIPCHK(c, hipMemcpyAsync(dd, local, ndraw * sizeof(double), hipMemcpyHostToDevice, c->stream));  // `local` outlives the sync below
          ISLECHK(k_search(c, c->cum.p, D, dd, ndraw, od));
        }
        HIPCHK(c, hipMemcpyAsync(drawn, od, ndraw * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int i = 0; i < ndraw; ++i) {
          if (local[i] < 0.0 || D == 0) drawn[i] = 0;
          else drawn[i] = std::min<uint64_t>(drawn[i], D - 1) + c->doc_offset + 1;  // +1: zero means "not mine"
        }
        if (multi) {
          HIPCHK(c, hipMemcpyAsync(od, drawn, ndraw * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
          ISLECHK(allreduce_sum<uint64_t>(c, od, ndraw));
          HIPCHK(c, hipMemcpyAsync(drawn, od, ndraw * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
          HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        for (int i = 0; i < ndraw; ++i) drawn[i] = drawn[i] ? drawn[i] - 1 : 0;
      }
    }
    new_added = 0;
    std::vector<uint64_t> fresh;
    for (int cc = 0; cc < ndraw && (int)centers.size() < k; ++cc) {
      const uint64_t nc = inject ? inject[centers.size()] : drawn[cc];
      if (std::find(centers.begin(), centers.end(), nc) == centers.end()) {  // duplicates skipped, not redrawn :2189
        centers.push_back(nc);
        fresh.push_back(nc);
        new_added++;
      }
    }
    if (new_added) ISLECHK(fetch_rows(c, fresh.data(), new_added, c->Cdev.p + (centers.size() - new_added) * (size_t)ldk));
    if (inject && new_added == 0) return isle_fail(c, ISLE_E_ARG, "injected seeds contain duplicates");
    if (rounds > 100 * k) return isle_fail(c, ISLE_E_NUMERIC, "k-means++ cannot find %d distinct seeds", k);
  }
  isle_host_mark("kmeanspp: rounds done");
  // the last batch of seeds is never folded into min_dist (the loop ends when the k-th seed is drawn, :2163-2207); for Lloyd's first
  // assignment it is folded into a COPY of the distances
  if (track && c->kmpp_track && new_added > 0 && c->kmpp_track_seeds == k - new_added && D > 0) {
    HIPCHK(c, c->kmpp_best.reserve(D));
    HIPCHK(c, hipMemcpyAsync(c->kmpp_best.p, c->min_dist.p, D * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    ISLECHK(k_kmpp_update(c, c->P.p, c->pnorm.p, D, k, ldk, c->Cdev.p + (size_t)(k - new_added) * ldk, new_added, c->kmpp_best.p, k - new_added, true));
  }
  // best_centers_coords[c] = U^T b_seed[c]  (:2232-2234)
  const size_t ch_bytes = (size_t)k * ldk * sizeof(float);
  HIPCHK(c, c->pin_stage_reserve(ch_bytes));
  const float* Ch = reinterpret_cast<const float*>(c->pin_stage);
  HIPCHK(c, hipMemcpyAsync(c->pin_stage, c->Cdev.p, ch_bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int cc = 0; cc < k; ++cc) {
    seeds_out[cc] = centers[cc];
    memcpy(C_lowd + (size_t)cc * k, Ch + (size_t)cc * ldk, (size_t)k * sizeof(float));
  }
  if (residual) *residual = (float)(grand - last_md);  // dist_cumul[num_docs - 1]  (:2208; App. C #9)
  if (rounds_out) *rounds_out = rounds;
  if (track && c->kmpp_track && c->kmpp_track_seeds == k) {  // complete: Lloyd may start from it if it is handed exactly these centres
    c->kmpp_C_host.assign(C_lowd, C_lowd + (size_t)k * k);
    c->kmpp_P_gen = c->P_gen;
    c->kmpp_track_k = k;
  }
  isle_host_mark("kmeanspp: exit");
  return 0;
}

extern "C" int isle_hip_get_min_dist(isle_ctx* c, float* out) {
  if (!c || !out) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  if (c->D) HIPCHK(c, hipMemcpy(out, c->min_dist.p, c->D * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

// The reference's stop rule (src/sparseMatrix.cpp:2044-2064 / :1718-1738): converged when the cluster
// sizes equal the previous iteration's AND the partition equals the last partition stored on an
// iteration whose sizes matched.
namespace {
struct StopRule {
  isle_ctx* c;
  int k;
  std::vector<long long> prev_sizes;
  bool have_prev = false;
  StopRule(isle_ctx* c_, int k_) : c(c_), k(k_), prev_sizes(k_, 0) {}
  // sizes: GLOBAL cluster sizes of this iteration.  assign: device, local docs.
  int converged(const std::vector<long long>& sizes, const uint32_t* assign, bool* out) {
    bool changed = false;
    for (int i = 0; i < k; ++i)
      if (prev_sizes[i] != sizes[i]) changed = true;
    prev_sizes = sizes;
    if (!changed) {
      if (!have_prev) {
        changed = c->D_global > 0;  // prev_closest_docs are k empty lists
      } else {
        HIPCHK(c, c->flags.reserve(16));
        ISLECHK(k_compare_u32(c, assign, c->assign_prev.p, c->D, c->flags.p));
        ISLECHK(allreduce_sum<int>(c, c->flags.p, 1));
        int* f = reinterpret_cast<int*>(c->pin + isle_ctx::PIN_SMALL + (192u << 10));  // page-locked
        HIPCHK(c, hipMemcpyAsync(f, c->flags.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        changed = *f != 0;
      }
      HIPCHK(c, c->assign_prev.reserve(c->D ? c->D : 1));
      if (c->D) HIPCHK(c, hipMemcpyAsync(c->assign_prev.p, assign, c->D * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
      have_prev = true;
    }
    *out = !changed;
    return 0;
  }
};
}  // namespace

static int fetch_sizes(isle_ctx* c, int k, std::vector<long long>& sizes) {
  ISLECHK(allreduce_sum<int>(c, c->counts.p, k));
  std::vector<int> hv;
  int* h = reinterpret_cast<int*>(c->pin + isle_ctx::PIN_SMALL + (128u << 10));  // page-locked, 64 KB
  if ((size_t)k * sizeof(int) > (64u << 10)) {
    hv.resize(k);
    h = hv.data();
  }
  HIPCHK(c, hipMemcpyAsync(h, c->counts.p, (size_t)k * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  sizes.assign(h, h + k);
  return 0;
}

extern "C" int isle_hip_lloyds_projected(isle_ctx* c, int k, float* C_lowd, int max_reps, int* iters_run, uint32_t* assign_out) {
  if (!c || !C_lowd || k < 1) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  isle_host_mark("lloyds_projected: entry");
  ISLECHK(ensure_P(c, k));  // compute_projected_docs_l2sq :2032
  const uint64_t D = c->D;
  const int ldk = c->ldk;
  HIPCHK(c, c->Cdev.reserve((size_t)k * ldk));
  HIPCHK(c, c->Csum.reserve((size_t)k * ldk));
  HIPCHK(c, c->cnorm.reserve(k));
  HIPCHK(c, c->counts.reserve(k));
  HIPCHK(c, c->assign.reserve(D ? D : 1));
  c->assign_valid = false;
  const size_t ch_bytes = (size_t)k * ldk * sizeof(float);
  HIPCHK(c, c->pin_stage_reserve(ch_bytes));
  float* Ch = reinterpret_cast<float*>(c->pin_stage);
  if (ldk != k) memset(Ch, 0, ch_bytes);
  for (int cc = 0; cc < k; ++cc) memcpy(Ch + (size_t)cc * ldk, C_lowd + (size_t)cc * k, (size_t)k * sizeof(float));
  HIPCHK(c, hipMemcpyAsync(c->Cdev.p, Ch, ch_bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));  // the staging buffer is written again at the end of this call
  isle_host_mark("lloyds_projected: centres uploaded");
  // Hamerly bounds (exact skip of documents whose closest centre provably did not change), as in the sparse Lloyd
  const bool hamerly = !c->knob_on(KN_NO_HAMERLY) && c->Pt_ready;
  if (hamerly) {
    HIPCHK(c, c->hub.reserve(D ? D : 1));
    HIPCHK(c, c->hlb.reserve(D ? D : 1));
    HIPCHK(c, c->active.reserve(D + 1));
    HIPCHK(c, c->Pa.reserve((size_t)(D ? D : 1) * ldk));
    HIPCHK(c, c->pna.reserve(D ? D : 1));
    HIPCHK(c, c->Cold.reserve((size_t)k * ldk + k + 8));
  }
  float* delta_dev = hamerly ? c->Cold.p + (size_t)k * ldk : nullptr;
  HamTop* top_dev = hamerly ? reinterpret_cast<HamTop*>(c->Cold.p + (size_t)k * ldk + ((k + 3) & ~3)) : nullptr;
  // k > 224 (more than 7 tiles of 32 centres): one lower bound per tile instead of Hamerly's single one, which prunes nothing at
  // k = 1000 (kmeans.hip PR_TILES, spmm.hip pt_filter_k).  ISLE_PROJ_BOUNDS=hamerly keeps the single bound.
  const int T = (k + 31) / 32, TL = (T + 3) & ~3;
  const char* pbm = c->knob(KN_PROJ_BOUNDS);
  const bool tiles = hamerly && k > 224 && T <= 32 && !(pbm && !strcmp(pbm, "hamerly"));
  float* tmove_dev = nullptr;
  if (tiles) {
    HIPCHK(c, c->ptlb.reserve((size_t)(D ? D : 1) * TL));
    HIPCHK(c, c->pneed.reserve(D ? D : 1));
    HIPCHK(c, c->pcand.reserve(D + 1));
    HIPCHK(c, c->small.reserve(4096));
    tmove_dev = c->small.p;  // T floats
  }
  const bool from_kmpp = tiles && c->kmpp_track_k == k && c->kmpp_P_gen == c->P_gen && c->P_ready && !c->knob_zero(KN_KMPP_TRACK) &&
                         c->kmpp_C_host.size() == (size_t)k * k && memcmp(c->kmpp_C_host.data(), C_lowd, (size_t)k * k * sizeof(float)) == 0;
  c->kmpp_track_k = 0;  // used (the tile minima become bounds in place) or stale
  if (c->knob_on(KN_DEBUG_HAMERLY)) fprintf(stderr, "[projected Lloyd] first assignment %s\n", from_kmpp ? "taken from the k-means++ rounds" : "computed");
  StopRule stop(c, k);
  int it = 0;
  isle_host_mark("lloyds_projected: loop starts");
  for (; it < max_reps; ++it) {
    ISLECHK(k_rownorms(c, c->Cdev.p, k, k, ldk, c->cnorm.p));                                          // :1938
    if (tiles) {
      if (it == 0 && from_kmpp) {
        // the centres are the k-means++ seeds and the rounds kept every document's nearest seed, its tile's runner-up and the minimum of
        // every other tile: exactly this assignment (up to the rounding of the two distance evaluations, inside the bounds' slack)
        ISLECHK(k_kmpp_to_tiles(c, D, k, c->pnorm.p, c->cnorm.p, c->kmpp_best.p, c->assign.p, c->hub.p, TL));
      } else if (it == 0) {
        ISLECHK(k_proj_assign_tiles(c, c->P.p, c->pnorm.p, D, k, ldk, c->Cdev.p, c->cnorm.p, c->assign.p, c->hub.p, c->ptlb.p, TL, nullptr, 0,
                                    nullptr, nullptr, nullptr));                                           // :1947
      } else {
        uint32_t* nact = c->active.p + D;
        // documents are taken grouped by their centre (member lists of the previous iteration): a workgroup of the re-examination
        // then holds neighbours, whose needed tiles coincide
        {  // candidates by the grown upper bounds, then the exact distance to the own centre for those (pt_tighten_k)
          uint32_t* ncand = c->pcand.p + D;
          ISLECHK(k_pt_filter(c, c->members_valid ? c->members.p : nullptr, c->assign.p, c->hub.p, c->ptlb.p, T, TL, delta_dev, tmove_dev,
                              c->pneed.p, c->pcand.p, ncand));
          ISLECHK(k_pt_tighten(c, c->P.p, c->pnorm.p, ldk, c->Cdev.p, c->cnorm.p, c->assign.p, c->pcand.p, ncand, c->hub.p, c->ptlb.p, T, TL,
                               c->pneed.p, c->active.p, nact));
        }
        uint32_t* na_pin = reinterpret_cast<uint32_t*>(c->pin + isle_ctx::PIN_SMALL + (192u << 10) + 64);  // page-locked
        HIPCHK(c, hipMemcpyAsync(na_pin, nact, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const uint32_t na = *na_pin;
        if (c->knob_on(KN_DEBUG_HAMERLY)) {  // debug only: how many tiles the active documents ask for
          std::vector<uint32_t> act(na), need(D);
          if (na) HIPCHK(c, hipMemcpy(act.data(), c->active.p, na * sizeof(uint32_t), hipMemcpyDeviceToHost));
          if (D) HIPCHK(c, hipMemcpy(need.data(), c->pneed.p, D * sizeof(uint32_t), hipMemcpyDeviceToHost));
          double tiles_sum = 0, union_sum = 0;
          for (uint32_t i = 0; i < na; i += 128) {
            uint32_t u = 0;
            for (uint32_t j = i; j < std::min(na, i + 128); ++j) {
              tiles_sum += __builtin_popcount(need[act[j]]);
              u |= need[act[j]];
            }
            union_sum += __builtin_popcount(u);
          }
          fprintf(stderr, "[tile bounds, projected] iter %d active %u of %llu, tiles per document %.1f, per workgroup (union) %.1f of %d\n", it, na,
                  (unsigned long long)D, na ? tiles_sum / na : 0.0, na ? union_sum / ((na + 127) / 128) : 0.0, T);
        }
        if ((uint64_t)na * 2 > D && k_proj_full_by_gemm(c, D, k))  // most documents are up for re-examination: the full GEMM pass costs less than
          ISLECHK(k_proj_assign_tiles(c, c->P.p, c->pnorm.p, D, k, ldk, c->Cdev.p, c->cnorm.p, c->assign.p, c->hub.p, c->ptlb.p, TL, nullptr, 0,
                                      nullptr, nullptr, nullptr));  // compacting them and walking their tiles, and refreshes every bound
        else
          ISLECHK(k_proj_assign_tiles(c, c->P.p, c->pnorm.p, D, k, ldk, c->Cdev.p, c->cnorm.p, c->assign.p, c->hub.p, c->ptlb.p, TL, c->active.p, na,
                                      c->pneed.p, c->Pa.p, c->pna.p));
      }
    } else if (it == 0 || !hamerly) {
      ISLECHK(k_proj_assign(c, c->P.p, c->pnorm.p, D, k, ldk, c->Cdev.p, c->cnorm.p, c->assign.p,
                            hamerly ? c->hub.p : nullptr, hamerly ? c->hlb.p : nullptr));                // :1947
    } else {
      uint32_t* nact = c->active.p + D;
      ISLECHK(k_hamerly_filter(c, nullptr, c->assign.p, c->hub.p, c->hlb.p, delta_dev, top_dev, c->active.p, nact, ISLE_T_LLOYD_PROJ));
      uint32_t* na_pin = reinterpret_cast<uint32_t*>(c->pin + isle_ctx::PIN_SMALL + (192u << 10) + 64);  // page-locked
      HIPCHK(c, hipMemcpyAsync(na_pin, nact, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      const uint32_t na = *na_pin;
      if (c->knob_on(KN_DEBUG_HAMERLY)) fprintf(stderr, "[hamerly, projected] iter %d active %u of %llu\n", it, na, (unsigned long long)D);
      ISLECHK(k_proj_assign_active(c, c->P.p, c->pnorm.p, k, ldk, c->Cdev.p, c->cnorm.p, c->active.p, na, c->Pa.p, c->pna.p, c->assign.p,
                                   c->hub.p, c->hlb.p));
    }
    ISLECHK(k_count_sizes(c, c->assign.p, D, k, c->counts.p));
    ISLECHK(k_proj_accumulate(c, c->P.p, D, k, ldk, c->assign.p, c->Csum.p, c->counts.p));              // :1957-1984
    ISLECHK(allreduce_sum<float>(c, c->Csum.p, (size_t)k * ldk));
    std::vector<long long> sizes;
    ISLECHK(fetch_sizes(c, k, sizes));
    if (hamerly) HIPCHK(c, hipMemcpyAsync(c->Cold.p, c->Cdev.p, (size_t)k * ldk * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    ISLECHK(k_proj_finalize(c, c->Csum.p, c->counts.p, k, ldk, c->Cdev.p));                             // :1988-1992
    if (hamerly && it + 1 < max_reps) {
      ISLECHK(k_rownorms_diff(c, c->Cdev.p, c->Cold.p, k, k, ldk, delta_dev));
      if (tiles) ISLECHK(k_yy_delta(c, delta_dev, k, T, 32, tmove_dev));  // rounded-up movements and their maxima per tile
      else ISLECHK(k_ham_delta(c, delta_dev, k, top_dev));  // rounded-up movements and their top two, on the device
    }
    bool conv = false;
    ISLECHK(stop.converged(sizes, c->assign.p, &conv));
    if (conv) {
      ++it;
      break;
    }
  }
  isle_host_mark("lloyds_projected: loop done");
  HIPCHK(c, hipMemcpyAsync(Ch, c->Cdev.p, ch_bytes, hipMemcpyDeviceToHost, c->stream));
  if (assign_out && D) HIPCHK(c, hipMemcpyAsync(assign_out, c->assign.p, D * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int cc = 0; cc < k; ++cc) memcpy(C_lowd + (size_t)cc * k, Ch + (size_t)cc * ldk, (size_t)k * sizeof(float));
  if (iters_run) *iters_run = it;
  isle_host_mark("lloyds_projected: exit");
  return 0;
}

// ------------------------------------------------------------------------------------------
// lift + Lloyd on the sparse matrix
// ------------------------------------------------------------------------------------------
static int install_centers(isle_ctx* c, int ncols) {  // centers_cm (V x ncols) -> centers_rm (V x ld), zero padded
  const int ld = round4(ncols);
  HIPCHK(c, c->centers_rm.reserve((size_t)c->V * ld));
  HIPCHK(c, hipMemsetAsync(c->centers_rm.p, 0, (size_t)c->V * ld * sizeof(float), c->stream));
  ISLECHK(k_transpose(c, c->centers_cm.p, c->V, ncols, c->V, c->centers_rm.p, ld));
  c->centers_ready = true;
  c->centers_k = ncols;
  return 0;
}

extern "C" int isle_hip_lift_centers(isle_ctx* c, const float* in, int ld_in, int ncols, float* centers) {
  if (!c || !in || ncols < 1) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  if (c->U_k == 0 || ld_in < c->U_k) return isle_fail(c, ISLE_E_ARG, "lift: need U and ld_in >= k");
  isle_host_mark("lift: entry");
  HIPCHK(c, c->Csum.reserve((size_t)ld_in * ncols));
  {
    const size_t in_bytes = (size_t)ld_in * ncols * sizeof(float);
    HIPCHK(c, c->pin_stage_reserve(in_bytes));
    memcpy(c->pin_stage, in, in_bytes);
    HIPCHK(c, hipMemcpyAsync(c->Csum.p, c->pin_stage, in_bytes, hipMemcpyHostToDevice, c->stream));  // the call synchronises before it returns
  }
  HIPCHK(c, c->centers_cm.reserve((size_t)c->V * ncols));
  ISLECHK(k_gemm_nn(c, c->Ucm.p, c->V, c->U_k, c->Csum.p, ld_in, ncols, c->centers_cm.p));
  ISLECHK(install_centers(c, ncols));
  // the centres lie in span(U): Lloyd on B can take its first assignment from the projection (isle_hip_lloyds_sparse)
  HIPCHK(c, c->lift_C.reserve((size_t)ld_in * ncols));
  HIPCHK(c, hipMemcpyAsync(c->lift_C.p, c->Csum.p, (size_t)ld_in * ncols * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  c->lift_ld = ld_in;
  c->lift_k = ncols;
  c->lift_valid = true;
  if (centers) HIPCHK(c, hipMemcpyAsync(centers, c->centers_cm.p, (size_t)c->V * ncols * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  isle_host_mark("lift: exit");
  return 0;
}

extern "C" int isle_hip_lloyds_sparse(isle_ctx* c, int k, const float* centers_in, float* centers_out, uint32_t* assign, int max_reps,
                                      int* iters_run) {
  if (!c || k < 1) return ISLE_E_ARG;
  if (c->V == 0) return isle_fail(c, ISLE_E_ARG, "no matrix uploaded");
  ISLECHK(isle_enter(c));
  isle_host_mark("lloyds_sparse: entry");
  const uint64_t D = c->D, V = c->V;
  const int ld = round4(k);
  if (centers_in) {
    HIPCHK(c, c->centers_cm.reserve((size_t)V * k));
    HIPCHK(c, hipMemcpy(c->centers_cm.p, centers_in, (size_t)V * k * sizeof(float), hipMemcpyHostToDevice));
    ISLECHK(install_centers(c, k));
    c->lift_valid = false;
  } else if (!c->centers_ready || c->centers_k != k) {
    return isle_fail(c, ISLE_E_ARG, "lloyds_sparse: no device-resident centres for k = %d (call isle_hip_lift_centers)", k);
  }
  HIPCHK(c, c->dnorm.reserve(D ? D : 1));
  HIPCHK(c, c->cnorm.reserve(k));
  HIPCHK(c, c->counts.reserve(k));
  HIPCHK(c, c->assign.reserve(D ? D : 1));
  c->assign_valid = false;
  ISLECHK(k_doc_norms(c, c->dnorm.p));  // :1680-1687
  // Distance bounds: exact accelerations of the assignment step (documents whose bounds prove "unchanged" are skipped).
  // Default: Yinyang group bounds (groups of 8 centres); ISLE_KMEANS_BOUNDS=hamerly|none selects the others.
  const char* bmode = c->knob(KN_KMEANS_BOUNDS);
  const bool nobounds = c->knob_on(KN_NO_HAMERLY) || (bmode && !strcmp(bmode, "none"));
  const bool hamerly = !nobounds;                                   // any bound-based mode
  const bool yinyang = hamerly && !(bmode && !strcmp(bmode, "hamerly"));
  const int G = (k + 7) / 8;
  int yy_mode_env = -1;  // form of the Yinyang iteration: 0 = by document over the row-major centres, 1 = by document over the group-major copy, 2 = by group
  if (const char* e = c->knob(KN_YY_MODE)) yy_mode_env = !strcmp(e, "doc") ? 0 : !strcmp(e, "docg") ? 1 : !strcmp(e, "group") ? 2 : -1;
  if (yinyang) HIPCHK(c, c->yglb.reserve((size_t)(D ? D : 1) * G + 64));
  float* gmax_dev = nullptr;
  HIPCHK(c, c->hub.reserve(D ? D : 1));
  HIPCHK(c, c->hlb.reserve(D ? D : 1));
  HIPCHK(c, c->active.reserve(D + 1));
  HIPCHK(c, c->centers_old.reserve((size_t)V * ld));
  HIPCHK(c, c->Csum.reserve((size_t)2 * k + 16 + G));
  float* delta_dev = c->Csum.p;  // k floats
  gmax_dev = c->Csum.p + 2 * k + 16;  // G floats
  HamTop* top_dev = reinterpret_cast<HamTop*>(c->Csum.p + 2 * k + 12);
  StopRule stop(c, k);
  // first assignment through the projection: only for centres that came from isle_hip_lift_centers with the current U and P, and
  // while the dense product is cheaper than the sparse one: always up to k = 384; beyond, by the measured rates — the D x k x k
  // product runs at ~130 TFLOP/s (rocBLAS), a panel pass of the sparse product takes ~2.8 ps per nonzero (C3 shard, k = 1000: 19 against
  // 44 ms) — and while its D x k scratch can be had (isle_scratch_ok) (ISLE_FIRST_ASSIGN=sparse|projection forces)
  const char* fa = c->knob(KN_FIRST_ASSIGN);
  const double t_dense = 2.0 * (double)D * k * k / 130e12, t_sparse = (double)((k + 7) / 8) * (double)c->nnz * 2.8e-12;
  const bool fused_first = yinyang && k_gemm_assign_fused_ok(c, D, k, k);  // the product's epilogue forms the assignment: no D x k scratch
  const bool dense_pays = k <= 384 || (t_dense < t_sparse && (fused_first || isle_scratch_ok(c, c->dotsT.cap, (double)D * k * sizeof(float))));
  bool via_projection = !centers_in && c->lift_valid && c->lift_k == k && c->U_k == k && c->P_ready && c->Pt_ready && c->ldk == ld &&
                        D > 0 && (dense_pays || (fa && !strcmp(fa, "projection"))) && !(fa && !strcmp(fa, "sparse"));
  if (via_projection && !fused_first && c->dotsT.reserve((size_t)D * k) != hipSuccess) {
    // the route is chosen from sizes alone (isle_scratch_ok), but on a device shared with other work the D x k scratch may still not be
    // had: the sparse product gives the same assignment up to dot-product rounding, so take it instead of failing the call
    (void)hipGetLastError();
    fprintf(stderr, "[isle_hip] lloyds_sparse: no memory for the %.1f GB product of the first assignment; taking the sparse route\n",
            (double)D * k * sizeof(float) / 1e9);
    via_projection = false;
  }
  c->lift_valid = false;  // the centres move below
  int it = 0;
  isle_host_mark("lloyds_sparse: loop starts");
  for (; it < max_reps; ++it) {
    {
      TimeScope ts(c, ISLE_T_SPARSE_ASSIGN);
      ISLECHK(k_colnorms_rm(c, c->centers_rm.p, V, k, ld, c->cnorm.p));  // :1604
    }
    if (it == 0 && via_projection) {
      // B^T (U C^T) = (U^T B)^T C^T: the k-wide sparse product of the first assignment (distsq_docs_to_centers, :1494-1550) is a dense
      // D x k x k product on the projection that k-means++ / Lloyd in span(U) left on the device — one MFMA GEMM, a transposition into
      // the doc-major layout and the same distance / bound epilogue (norms of centres and documents are the word-space ones)
      TimeScope ts(c, ISLE_T_SPARSE_ASSIGN);
      if (yinyang && fused_first) {  // distances, group bounds and candidates formed inside the product: no D x k matrix in memory
        float* cn_max_dev = c->Csum.p + 2 * k + 8;
        ISLECHK(k_max_f32(c, c->cnorm.p, k, cn_max_dev));
        ISLECHK(k_gemm_assign_yy(c, c->Pt.p, D, k, c->lift_C.p, c->lift_ld, k, G, c->cnorm.p, c->dnorm.p, cn_max_dev, c->assign.p, c->hub.p, c->yglb.p,
                                 ISLE_T_SPARSE_ASSIGN));
      } else if (yinyang) {  // assignment and group bounds straight from the column-major product (the projection stays valid)
        HIPCHK(c, c->dotsT.reserve((size_t)D * k));
        ISLECHK(k_gemm_nn_assign(c, c->Pt.p, D, k, c->lift_C.p, c->lift_ld, k, c->dotsT.p, ISLE_T_SPARSE_ASSIGN));
        float* cn_max_dev = c->Csum.p + 2 * k + 8;
        ISLECHK(k_max_f32(c, c->cnorm.p, k, cn_max_dev));
        ISLECHK(k_dots_assign_cm(c, c->dotsT.p, k, G, c->cnorm.p, c->dnorm.p, cn_max_dev, c->assign.p, c->hub.p, c->yglb.p));
      } else {
        HIPCHK(c, c->dotsT.reserve((size_t)D * k));
        ISLECHK(k_gemm_nn_assign(c, c->Pt.p, D, k, c->lift_C.p, c->lift_ld, k, c->dotsT.p, ISLE_T_SPARSE_ASSIGN));
        c->P_ready = false;  // P now holds the dot products (as with the LDS-banded wide product)
        c->Pt_ready = false;
        if (ld != k) HIPCHK(c, hipMemsetAsync(c->P.p, 0, (size_t)D * ld * sizeof(float), c->stream));
        ISLECHK(k_transpose(c, c->dotsT.p, D, (uint64_t)k, D, c->P.p, (uint64_t)ld));
        ISLECHK(k_dots_assign(c, k, ld, c->cnorm.p, c->dnorm.p, c->assign.p, c->hub.p, c->hlb.p, 0));
      }
    } else if (it == 0 || !hamerly) {
      // documents are visited grouped by their previous centre (cache locality of the centre rows); results are order-independent
      ISLECHK(k_spmm_wide_assign(c, c->centers_rm.p, k, ld, c->cnorm.p, c->dnorm.p, c->assign.p,
                                 c->members_valid ? c->members.p : nullptr, nullptr, c->hub.p, yinyang ? c->yglb.p : c->hlb.p,
                                 yinyang ? G : 0));  // :1606
    } else if (yinyang) {
      // all bookkeeping of the Yinyang iteration stays on the device (largest centre norm, movements, group maxima, member
      // offsets): the only host round trip of an iteration is the one the stop rule needs
      float* cn_max_dev = c->Csum.p + 2 * k + 8;
      {
        TimeScope ts(c, ISLE_T_SPARSE_ASSIGN);
        ISLECHK(k_max_f32(c, c->cnorm.p, k, cn_max_dev));
      }
      uint32_t* nact = c->active.p + D;
      // large k: the centres also group-major (one 32-byte-row table per group) and the active documents grouped by their own group
      // (the member lists), so that waves running together gather from one table in L2; ISLE_YY_MODE = doc | docg | group picks the form
      // (measured, Lloyd on B per step: C3 shard 176 ms by document -> 112 ms by group, all of config 3 on one GPU 825 -> 588 ms; at C2,
      // G = 25 and a 40 MB table, the three forms are within 10 % of each other and the plain one stays)
      // the forms that visit documents in member order (docg, group) hold a document's group bounds four per lane: at most 256 groups
      // (k <= 2048); beyond, by document over the row-major centres, whatever ISLE_YY_MODE asks for
      const int yy_mode = G > 256 ? 0 : yy_mode_env >= 0 ? yy_mode_env : (G >= 32 ? 2 : 0);
      const uint32_t* order = yy_mode && c->members_valid ? c->members.p : nullptr;
      if (yy_mode) ISLECHK(k_yy_pack_groups(c, c->centers_rm.p, ld, G));
      // by group: the bounds are lowered and the active documents tightened in one launch (the D x G bounds read once), ISLE_YY_FUSED=0: in two
      const bool fused = yy_mode == 2 && !c->knob_zero(KN_YY_FUSED);
      if (fused)
        ISLECHK(k_yy_filter_tighten(c, order, c->assign.p, c->hub.p, c->yglb.p, G, delta_dev, gmax_dev, c->active.p, nact, c->yy_cg.p, k, ld, c->cnorm.p, c->dnorm.p,
                                    cn_max_dev));
      else
        ISLECHK(k_yy_filter(c, order, c->assign.p, c->hub.p, c->yglb.p, G, delta_dev, gmax_dev, c->active.p, nact));
      const bool dbg = c->knob_on(KN_DEBUG_HAMERLY);
      unsigned long long* dbg_dev = nullptr;
      if (dbg) {  // diagnostic only: group scans and gathered nonzeros of this iteration
        HIPCHK(c, c->dbg_cnt.reserve(2));
        HIPCHK(c, hipMemsetAsync(c->dbg_cnt.p, 0, 16, c->stream));
        dbg_dev = c->dbg_cnt.p;
      }
      bool done = false;
      unsigned long long npairs = 0;
      if (yy_mode == 2)
        ISLECHK(k_yy2_assign(c, c->yy_cg.p, k, ld, G, c->cnorm.p, c->dnorm.p, cn_max_dev, c->active.p, nact, c->assign.p, c->hub.p, c->yglb.p, &done, &npairs, fused));
      if (!done)
        ISLECHK(k_yy_scan(c, c->centers_rm.p, yy_mode ? c->yy_cg.p : nullptr, k, ld, G, c->cnorm.p, c->dnorm.p, cn_max_dev, c->active.p, nact, c->assign.p,
                          c->hub.p, c->yglb.p, dbg_dev));
      if (dbg) {
        uint32_t na = 0;
        unsigned long long cnt[2] = {0, 0};
        HIPCHK(c, hipMemcpy(&na, nact, 4, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(cnt, dbg_dev, 16, hipMemcpyDeviceToHost));
        {
          std::vector<float> dl(k), gm(G), cn(k);
          float cm = 0.f;
          HIPCHK(c, hipMemcpy(dl.data(), delta_dev, k * sizeof(float), hipMemcpyDeviceToHost));
          HIPCHK(c, hipMemcpy(gm.data(), gmax_dev, G * sizeof(float), hipMemcpyDeviceToHost));
          HIPCHK(c, hipMemcpy(cn.data(), c->cnorm.p, k * sizeof(float), hipMemcpyDeviceToHost));
          HIPCHK(c, hipMemcpy(&cm, cn_max_dev, sizeof(float), hipMemcpyDeviceToHost));
          std::vector<float> sd(dl), sc(cn);
          std::sort(sd.begin(), sd.end());
          std::sort(sc.begin(), sc.end());
          std::vector<long long> szs;
          ISLECHK(fetch_sizes(c, k, szs));
          long long smin = szs[0], smax = szs[0], empty = 0;
          for (auto v : szs) { smin = std::min(smin, v); smax = std::max(smax, v); empty += v == 0; }
          fprintf(stderr, "[yinyang] iter %d: movement median %.3g max %.3g; |c|^2 median %.3g max %.3g (cn_max %.3g); cluster sizes %lld..%lld, %lld empty\n", it,
                  sd[k / 2], sd[k - 1], sc[k / 2], sc[k - 1], cm, smin, smax, empty);
        }
        if (done)
          fprintf(stderr, "[yinyang] iter %d active %u of %llu; by group: %llu pairs beside the own-group scans (%.1f per active document, of %d)\n", it, na,
                  (unsigned long long)D, npairs, na ? (double)npairs / na : 0.0, G);
        else
          fprintf(stderr, "[yinyang] iter %d active %u of %llu; group scans %llu (%.1f per active document, of %d), gathered nonzeros %llu\n", it, na,
                  (unsigned long long)D, cnt[0], na ? (double)cnt[0] / na : 0.0, G, cnt[1]);
      }
    } else {
      uint32_t* nact = c->active.p + D;
      ISLECHK(k_hamerly_filter(c, c->members_valid ? c->members.p : nullptr, c->assign.p, c->hub.p, c->hlb.p, delta_dev, top_dev, c->active.p,
                               nact));
      ISLECHK(k_spmm_wide_assign(c, c->centers_rm.p, k, ld, c->cnorm.p, c->dnorm.p, c->assign.p, c->active.p, nact, c->hub.p, c->hlb.p));
      if (c->knob_on(KN_DEBUG_HAMERLY)) {
        uint32_t na = 0;
        HIPCHK(c, hipMemcpy(&na, nact, 4, hipMemcpyDeviceToHost));
        fprintf(stderr, "[hamerly] iter %d active %u of %llu\n", it, na, (unsigned long long)D);
      }
    }
    ISLECHK(k_count_sizes(c, c->assign.p, D, k, c->counts.p));
    {  // documents grouped by centre: visiting order of the next assignment, and what the counting centroid update walks
      TimeScope ts(c, ISLE_T_SPARSE_ASSIGN);
      ISLECHK(k_member_lists_dev(c, c->assign.p, D, k, c->counts.p));
    }
    if (hamerly) HIPCHK(c, hipMemcpyAsync(c->centers_old.p, c->centers_rm.p, (size_t)V * ld * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    ISLECHK(k_centers_from_rows(c, c->assign.p, k, ld, c->centers_rm.p, it == 0));                 // :1613-1638
    ISLECHK(allreduce_sum<float>(c, c->centers_rm.p, (size_t)V * ld));
    std::vector<long long> sizes;
    ISLECHK(fetch_sizes(c, k, sizes));
    ISLECHK(k_scale_centers(c, c->centers_rm.p, V, k, ld, c->counts.p));  // :1641-1646
    if (hamerly && it + 1 < max_reps) {  // centre movements for the next filter
      TimeScope ts(c, ISLE_T_SPARSE_ASSIGN);
      ISLECHK(k_colnorms_rm(c, c->centers_rm.p, V, k, ld, delta_dev, c->centers_old.p));
      if (yinyang) ISLECHK(k_yy_delta(c, delta_dev, k, G, 8, gmax_dev));  // movements and group maxima stay on the device
      else ISLECHK(k_ham_delta(c, delta_dev, k, top_dev));
    }
    bool conv = false;
    ISLECHK(stop.converged(sizes, c->assign.p, &conv));
    if (conv) {
      ++it;
      break;
    }
  }
  isle_host_mark("lloyds_sparse: loop done");
  c->assign_valid = true;  // the partition stays resident for isle_hip_catchwords
  if (assign && D) HIPCHK(c, hipMemcpyAsync(assign, c->assign.p, D * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  if (centers_out) {
    HIPCHK(c, c->centers_cm.reserve((size_t)V * k));
    // row-major (V x ld) -> col-major (V x k): view as a k x V col-major matrix with ld_in = ld
    ISLECHK(k_transpose(c, c->centers_rm.p, k, V, ld, c->centers_cm.p, V));
    HIPCHK(c, hipMemcpyAsync(centers_out, c->centers_cm.p, (size_t)V * k * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  isle_host_mark("lloyds_sparse: exit");
  if (iters_run) *iters_run = it;
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
