// isle_amd/csrc/kmeans.hip — k-means kernels on the materialised projection P = U^T B (D x ldk, doc-major).
//
// The reference never materialises P (USE_EXPLICIT_PROJECTED_MATRIX=false, include/hyperparams.h:44):
// every use re-multiplies the sparse block by U or by -2*U*C^T (src/sparseMatrix.cpp:1794-1849).  With
// 288 GB of HBM per GPU P is kept resident instead, and the distance matrix is never written:
//   k_kmpp_update      update_min_distsq_to_projected_centers   src/sparseMatrix.cpp:2075-2130
//   k_scan_f2d/search  D^2 prefix sums + upper_bound draws      src/sparseMatrix.cpp:2170-2188
//   k_proj_assign      projected_closest_centers (f32 MFMA distance tiles + fused isamin)  :1852-1871
//   k_proj_accumulate  centroid sums (saxpy loop)               src/sparseMatrix.cpp:1975-1992
#include <algorithm>
#include <cstdlib>
#include <vector>

#include <cstring>

#include "common.h"
#include "hamerly.h"
#include "scan.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int CS_ITEMS = 16;  // documents per thread in the histogram-style kernels
enum { PR_ARGMIN = 0, PR_MINDIST = 1, PR_TILES = 2 };
template <int MODE>
static int launch_proj_reg(isle_ctx* c, uint64_t D, int k, int ldk, const float* C, const float* cn, const float* pn, uint32_t* assign,
                           float* min_dist, bool* done, const float* Pt = nullptr, const uint32_t* map = nullptr, float* ub = nullptr,
                           float* lb = nullptr, const uint32_t* need = nullptr, int TL = 0);

// ------------------------------------------------------------------------------------------
// min_dist[d] = min(min_dist[d], max(|p_d|^2 + |c|^2 - 2 p_d.c, 0)) over the nc newest centres.
// One wave per document; lane owns coordinates {lane + 64*it}.
// ------------------------------------------------------------------------------------------
template <int NIT>
__global__ __launch_bounds__(256) void kmpp_update_k(const float* __restrict__ P, const float* __restrict__ pn, uint32_t D, int ldk,
                                                      const float* __restrict__ newC, const float* __restrict__ cn, int nc,
                                                      float* __restrict__ min_dist) {
  const int lane = threadIdx.x & 63;
  const uint32_t d = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (d >= D) return;
  float p[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int j = lane + 64 * it;
    p[it] = (j < ldk) ? P[(size_t)d * ldk + j] : 0.f;
  }
  const float nd = pn[d];
  float best = min_dist[d];
  for (int cc = 0; cc < nc; ++cc) {
    const float* cr = newC + (size_t)cc * ldk;
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int j = lane + 64 * it;
      if (j < ldk) s = fmaf(p[it], cr[j], s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float t = fmaxf((-2.0f * s + cn[cc]) + nd, 0.0f);  // :1838-1846 order, clamp :2117
    best = fminf(best, t);
  }
  if (lane == 0) min_dist[d] = best;
}

// The same update on the coordinate-major copy Pt (Pt[j * D + d]): one lane per document, the loop over the coordinates reads
// 256 contiguous bytes per wave and step, the (at most 16) new centres sit in LDS as [coordinate][centre] and are read as
// broadcast float4 — no cross-lane reduction at all.  A k-means++ round is then one streaming pass over Pt
// (4 k D bytes, the figure of SURVEY 8d) instead of a pass through the MFMA distance kernel built for hundreds of centres.
template <int NQ>  // float4 of centres per coordinate: nc <= 4 NQ
__global__ __launch_bounds__(256) void kmpp_min_pt_k(const float* __restrict__ Pt, const float* __restrict__ pn, uint32_t D, int k, int ldk,
                                                      const float* __restrict__ newC, const float* __restrict__ cn, int nc,
                                                      float* __restrict__ min_dist) {
  extern __shared__ float4 Cq[];  // k x NQ float4: Cq[j * NQ + q] = centres 4q..4q+3 at coordinate j (0 beyond nc)
  for (int idx = threadIdx.x; idx < k * NQ * 4; idx += 256) {
    const int j = idx / (NQ * 4), cc = idx - j * (NQ * 4);
    reinterpret_cast<float*>(Cq)[idx] = cc < nc ? newC[(size_t)cc * ldk + j] : 0.f;
  }
  __syncthreads();
  const uint32_t d = blockIdx.x * 256 + threadIdx.x;
  const uint32_t dd = d < D ? d : D - 1;  // clamped: loads stay unconditional
  float4 acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* col = Pt + dd;
#pragma unroll 8
  for (int j = 0; j < k; ++j) {
    const float x = col[(size_t)j * D];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const float4 cv = Cq[j * NQ + q];
      acc[q].x = fmaf(x, cv.x, acc[q].x);
      acc[q].y = fmaf(x, cv.y, acc[q].y);
      acc[q].z = fmaf(x, cv.z, acc[q].z);
      acc[q].w = fmaf(x, cv.w, acc[q].w);
    }
  }
  if (d >= D) return;
  const float nd = pn[d];
  float best = min_dist[d];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const float a[4] = {acc[q].x, acc[q].y, acc[q].z, acc[q].w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int cc = 4 * q + u;
      if (cc < nc) best = fminf(best, fmaxf((-2.0f * a[u] + cn[cc]) + nd, 0.0f));  // :1838-1846 order, clamp :2117
    }
  }
  min_dist[d] = best;
}

// out[r] = sum_j M[r*ldk + j]^2 over j < k   (compute_projected_centers_l2sq :1874-1884)
__global__ __launch_bounds__(256) void rownorms_k(const float* __restrict__ M, int rows, int k, int ldk, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float s = 0.f;
  for (int j = lane; j < k; j += 64) s = fmaf(M[(size_t)r * ldk + j], M[(size_t)r * ldk + j], s);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) out[r] = s;
}
// out[r] = sum_j (A[r][j] - B[r][j])^2 : squared movement of every projected centre
__global__ __launch_bounds__(256) void rownorms_diff_k(const float* __restrict__ A, const float* __restrict__ B, int rows, int k, int ldk,
                                                        float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float s = 0.f;
  for (int j = lane; j < k; j += 64) {
    const float x = A[(size_t)r * ldk + j] - B[(size_t)r * ldk + j];
    s = fmaf(x, x, s);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) out[r] = s;
}
int k_rownorms_diff(isle_ctx* c, const float* A, const float* B, int rows, int k, int ldk, float* out) {
  if (rows == 0) return 0;
  hipLaunchKernelGGL(rownorms_diff_k, dim3(cdiv(rows, 4)), dim3(256), 0, c->stream, A, B, rows, k, ldk, out);
  HIPCHK(c, hipGetLastError());
  return 0;
}
int k_rownorms(isle_ctx* c, const float* M, int rows, int k, int ldk, float* out) {
  if (rows == 0) return 0;
  hipLaunchKernelGGL(rownorms_k, dim3(cdiv(rows, 4)), dim3(256), 0, c->stream, M, rows, k, ldk, out);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// min_dist[d] = min(min_dist[d], max(pn[d] + cn[j] - 2 dots[d][j], 0)) over the nc new seeds
// (dpos: the thin product left document d's row at dpos[d], k_gl_thin by_position; null: at d)
// (NQ = ld / 4 float4 per row, a template parameter since round 5: the row's loads are all in flight before the first is used — with a
// run-time trip count a thread walked its row one dependent load at a time)
template <int NQ>
__global__ __launch_bounds__(256) void kmpp_min_dots_k(const float* __restrict__ dots, const float* __restrict__ pn, const float* __restrict__ cn,
                                                        int nc, uint32_t D, float* __restrict__ min_dist, const uint32_t* __restrict__ dpos) {
  constexpr int ld = 4 * NQ;
  const uint32_t d = blockIdx.x * 256 + threadIdx.x;
  if (d >= D) return;
  const float4* row = reinterpret_cast<const float4*>(dots + (size_t)(dpos ? dpos[d] : d) * ld);
  float4 vv[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) vv[q] = row[q];
  const float nd = pn[d];
  float m = min_dist[d];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const float4 v = vv[q];
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * q + e < nc) m = fminf(m, fmaxf((-2.0f * x[e] + cn[4 * q + e]) + nd, 0.0f));
  }
  min_dist[d] = m;
}

// The same update that also keeps what Lloyd's first assignment in span(U) needs (k > 224, tile bounds): the nearest seed so far
// (arg; first index among equal distances), per tile of 32 seeds the smallest distance (tmin, TILE-major [tile][document]: a round touches
// one or two floats per document, coalesced across documents — document-major rows cost a 128-byte line each) and, for the tile that
// holds the nearest seed, the smallest distance to its OTHER seeds (m2a).  The seeds arrive in index order, so a round touches the two
// or three tiles its seeds fall into.  After the last seed this is, document by document, what a full assignment against the k seeds
// produces — nearest centre, runner-up of its tile, minimum of every other tile — and run_lloyds_on_projected_space can start from it
// instead of a D x k x k pass (kmpp_to_tiles_k).  `best` = the running minimum (min_dist during the rounds; a copy for the last batch,
// which the reference never folds into min_dist).
template <int NQ>
__global__ __launch_bounds__(256) void kmpp_min_dots_track_k(const float* __restrict__ dots, const float* __restrict__ pn,
                                                              const float* __restrict__ cn, int nc, uint32_t D, float* __restrict__ best,
                                                              uint32_t* __restrict__ arg, float* __restrict__ m2a, float* __restrict__ tmin,
                                                              uint32_t s_old, const uint32_t* __restrict__ dpos) {
  constexpr int ld = 4 * NQ;
  const uint32_t d = blockIdx.x * 256 + threadIdx.x;
  if (d >= D) return;
  const float4* row = reinterpret_cast<const float4*>(dots + (size_t)(dpos ? dpos[d] : d) * ld);
  float4 vv[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) vv[q] = row[q];
  const float nd = pn[d];
  float m = best[d], m2 = m2a[d];
  uint32_t a = arg[d];  // meaningless while m is FP_MAX (no seed seen): the first seed replaces it
  float* tm_col = tmin + d;  // tile T of this document at tm_col[T * D]
  uint32_t curT = s_old >> 5;
  float tm = tm_col[(size_t)curT * D];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const float4 v = vv[q];
    const float xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = 4 * q + e;
      if (j < nc) {
        const uint32_t sj = s_old + (uint32_t)j, T = sj >> 5;
        if (T != curT) {
          tm_col[(size_t)curT * D] = tm;
          curT = T;
          tm = tm_col[(size_t)T * D];
        }
        const float x = fmaxf((-2.0f * xs[e] + cn[j]) + nd, 0.0f);
        if (x < m) {  // strictly: the earlier seed keeps a tie
          m2 = (T == (a >> 5) && m < 3.0e38f) ? m : tm;  // same tile: the old nearest becomes its runner-up; another tile: that tile's minimum so far
          m = x;
          a = sj;
        } else if (T == (a >> 5)) {
          m2 = fminf(m2, x);
        }
        tm = fminf(tm, x);
      }
    }
  }
  tm_col[(size_t)curT * D] = tm;
  best[d] = m;
  arg[d] = a;
  m2a[d] = m2;
}
// tracked state -> the outputs of the full tile-bound assignment (proj_dots_tiles_k's tail): assign, ub, one lower bound per tile
__global__ __launch_bounds__(256) void kmpp_to_tiles_k(uint32_t D, int k, const float* __restrict__ pn, const float* __restrict__ cn, const float* __restrict__ best,
                                                        const uint32_t* __restrict__ arg, const float* __restrict__ m2a, const float* __restrict__ tmin,
                                                        uint32_t* __restrict__ assign, float* __restrict__ ub, float* __restrict__ tlb, int TL) {
  __shared__ float tcn[33];  // largest |centre|^2 per tile, and over all
  const int T = (k + 31) / 32;
  if (threadIdx.x < 33) {
    float m = 0.f;
    if ((int)threadIdx.x < T)
      for (int j = 32 * threadIdx.x; j < min(k, 32 * (int)threadIdx.x + 32); ++j) m = fmaxf(m, cn[j]);
    tcn[threadIdx.x] = m;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = 0.f;
    for (int t = 0; t < T; ++t) m = fmaxf(m, tcn[t]);
    tcn[32] = m;
  }
  __syncthreads();
  // The tile minima lie tile-major (a tile's D values in a run), the bounds document-major (rows of TL floats): a thread computes its
  // document's T bounds into an LDS tile [document][tile] and the workgroup writes its 256 rows as one run (a thread writing its own row
  // touches a different line per lane and store: 5.5 ms at config 3, round 5)
  __shared__ float rows_s[256][33];
  const uint32_t d0 = blockIdx.x * 256, d = d0 + threadIdx.x;
  if (d < D) {
    const float nd = pn[d], b = best[d];
    const uint32_t a = arg[d], Ta = a >> 5;
    float uu, ll;
    for (int t = 0; t < T; ++t) {
      const float m1 = tmin[(size_t)t * D + d];
      hamerly_store_bounds(m1, m1, nd + tcn[t], &uu, &ll);
      rows_s[threadIdx.x][t] = ll;
    }
    hamerly_store_bounds(b, m2a[d], nd + tcn[Ta], &uu, &ll);
    rows_s[threadIdx.x][Ta] = ll;  // the assigned centre's tile: closest OTHER centre in it
    hamerly_store_bounds(b, b, nd + tcn[32], &uu, &ll);
    ub[d] = uu;
    assign[d] = a;
  }
  __syncthreads();
  const uint32_t nd_blk = min(256u, D - d0);
  for (uint32_t i = threadIdx.x; i < nd_blk * (uint32_t)TL; i += 256) {
    const uint32_t j = i / (uint32_t)TL, t = i - j * (uint32_t)TL;
    if ((int)t < T) tlb[(size_t)d0 * TL + i] = rows_s[j][t];  // (the columns T .. TL - 1 stay as they are: nobody reads them)
  }
}

// the rows of the thin product (c->dotsT, ld = 4 .. 32 floats, by position) folded into the running minima (and, tracked, the nearest seed and tile minima)
static void launch_min_dots(isle_ctx* c, bool track, int ld, const float* pn, const float* cn, int ncj, uint64_t D, float* min_dist, uint32_t s0) {
  const dim3 g(cdiv(D, 256)), b(256);
#define MD(N)                                                                                                                                     \
  if (track)                                                                                                                                      \
    hipLaunchKernelGGL(kmpp_min_dots_track_k<N>, g, b, 0, c->stream, c->dotsT.p, pn, cn, ncj, (uint32_t)D, min_dist, c->kmpp_arg.p, c->kmpp_m2a.p, \
                       c->kmpp_tmin.p, s0, c->dpos.p);                                                                                            \
  else                                                                                                                                            \
    hipLaunchKernelGGL(kmpp_min_dots_k<N>, g, b, 0, c->stream, c->dotsT.p, pn, cn, ncj, (uint32_t)D, min_dist, c->dpos.p)
  switch (ld / 4) {
    case 1: MD(1); break;
    case 2: MD(2); break;
    case 3: MD(3); break;
    case 4: MD(4); break;
    case 5: MD(5); break;
    case 6: MD(6); break;
    case 7: MD(7); break;
    default: MD(8); break;
  }
#undef MD
}
int k_kmpp_update(isle_ctx* c, const float* P, const float* pn, uint64_t D, int k, int ldk, const float* newC, int nc,
                  float* min_dist, int s_old, bool track) {
  TimeScope ts(c, ISLE_T_KMPP);
  if (D == 0 || nc == 0) return 0;
  if (s_old == 0) c->kmpp_track = false;
  HIPCHK(c, c->cnorm.reserve((size_t)std::max(k, nc)));
  ISLECHK(k_rownorms(c, newC, nc, k, ldk, c->cnorm.p));
  // The reference's own formulation: P_d . P_c = b_d^T (U P_c), a thin product of B with the V x nc matrix U C_new^T —
  // 8 nnz bytes per 12 new seeds instead of the 4 k D bytes of the materialised projection.  Pays on a large shard at k = 1000
  // (1 GB per 8 seeds against 5 GB), not at C2 (0.8 GB against 0.8 GB); ISLE_KMPP_SPARSE=0/1 forces the choice.  The rule compares
  // measured rates: a pass of the pass-1 stream costs ~2.8 ps per nonzero (= 15.4 bytes of streaming at 5.5 TB/s), the materialised
  // projection streams at 5.5 TB/s for <= 16 new seeds (kmpp_min_pt_k) and 2.3x slower through the matrix-core tile beyond.
  {
    const char* e = c->knob(KN_KMPP_SPARSE);
    const int PW = k_gl_panel_width(c);
    const int passes = (nc + PW - 1) / PW;  // columns per pass of the pass-1 stream (gram_lds.hip gl_panel_width)
    const bool can = c->gl_mode == 1 && c->band_ready && c->U_k == k;
    bool sparse = can && 15.4 * (double)c->nnz * passes < 4.0 * (double)ldk * (double)D * (nc <= 16 ? 1.0 : 2.3);
    if (e) sparse = atoi(e) != 0 && can;
    if (sparse) {
      // (Round 3 also built the pruned round of accelerated k-means++ seeding — only the documents a new seed can reach by the
      // triangle inequality, from the materialised projection.  On this corpus the documents of topics without a seed stay within
      // reach: 98 % at 20 seeds, 50 % at 570, 28 % at 906 of 1000, and a visited document costs a 4 kB row plus 2 k flops per new seed
      // against 0.25 ns per document and pass here: k-means++ 56 -> 64 ms per C3-shard step.  Removed.)
      for (int j0 = 0; j0 < nc; j0 += 32) {  // at most 32 new seeds per thin product
        const int ncj = std::min(32, nc - j0);
        const int ld = (ncj + 3) & ~3;
        HIPCHK(c, c->Tmp.reserve((size_t)c->V * 32));
        HIPCHK(c, c->dotsT.reserve((size_t)D * 32));
        ISLECHK(k_gemm_nn(c, c->Ucm.p, c->V, k, newC + (size_t)j0 * ldk, ldk, ncj, c->Tmp.p, ISLE_T_KMPP));  // W = U C_new^T  (V x ncj col-major)
        ISLECHK(k_gl_thin(c, c->Tmp.p, ncj, ld, c->dotsT.p, true));  // rows by position: the kernels below read them through dpos
        if (track) {  // also the nearest seed and the tile minima (kmpp_min_dots_track_k)
          const int T = (k + 31) / 32;
          if (s_old == 0 && j0 == 0) {
            HIPCHK(c, c->kmpp_arg.reserve(D));
            HIPCHK(c, c->kmpp_m2a.reserve(D));
            HIPCHK(c, c->kmpp_tmin.reserve((size_t)D * T));
            ISLECHK(k_fill_f32(c, c->kmpp_tmin.p, (size_t)D * T, 3.402823466e+38f));
            ISLECHK(k_fill_f32(c, c->kmpp_m2a.p, D, 3.402823466e+38f));
            HIPCHK(c, hipMemsetAsync(c->kmpp_arg.p, 0, D * sizeof(uint32_t), c->stream));
            c->kmpp_track = true;
          }
          if (c->kmpp_track) launch_min_dots(c, true, ld, pn, c->cnorm.p + j0, ncj, D, min_dist, (uint32_t)(s_old + j0));
          else launch_min_dots(c, false, ld, pn, c->cnorm.p + j0, ncj, D, min_dist, 0u);
        } else {
          launch_min_dots(c, false, ld, pn, c->cnorm.p + j0, ncj, D, min_dist, 0u);
        }
        HIPCHK(c, hipGetLastError());
      }
      if (track && c->kmpp_track) c->kmpp_track_seeds = s_old + nc;
      return 0;
    }
  }
  c->kmpp_track = false;  // the routes below do not keep the nearest seed and the tile minima
  if (nc <= 16 && (c->Pt_ready || c->Pt2_ready) && (size_t)k * ((nc + 3) / 4) * sizeof(float4) <= 64 * 1024) {
    ISLECHK(k_ensure_pt(c));
    // streaming pass over the coordinate-major copy (the centres fit the default 64 KB of dynamic LDS)
    const int nq = (nc + 3) / 4;
    const dim3 g(cdiv(D, 256)), b(256);
    const size_t lds = (size_t)k * nq * sizeof(float4);
#define LP(N) hipLaunchKernelGGL(kmpp_min_pt_k<N>, g, b, lds, c->stream, c->Pt.p, pn, (uint32_t)D, k, ldk, newC, c->cnorm.p, nc, min_dist)
    if (nq == 1) LP(1);
    else if (nq == 2) LP(2);
    else if (nq == 3) LP(3);
    else LP(4);
#undef LP
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  if (nc <= 32) {  // distance tile on the matrix cores: the nc newest centres are one 32-row MFMA tile
    bool done = false;
    ISLECHK(launch_proj_reg<PR_MINDIST>(c, D, nc, ldk, newC, c->cnorm.p, pn, nullptr, min_dist, &done));
    if (done) return 0;
  }
  const int nit = cdiv(ldk, 64);
  dim3 g(cdiv(D, 4)), b(256);
#define LK(N) hipLaunchKernelGGL(kmpp_update_k<N>, g, b, 0, c->stream, P, pn, (uint32_t)D, ldk, newC, c->cnorm.p, nc, min_dist)
  if (nit <= 1) LK(1);
  else if (nit <= 2) LK(2);
  else if (nit <= 4) LK(4);
  else if (nit <= 8) LK(8);
  else if (nit <= 16) LK(16);
  else if (nit <= 32) LK(32);
  else return isle_fail(c, ISLE_E_ARG, "k too large");
#undef LK
  HIPCHK(c, hipGetLastError());
  return 0;
}

// Small-count variants for the k-means++ rounds: every hipMemcpy of a few bytes costs ~20 us of queue time, so the dice travel
// as kernel arguments and the seeds' rows are gathered by one kernel instead of one copy each.
struct KmDice {
  double x[16];
};
__global__ void search_args_k(const double* __restrict__ cum, uint64_t n, KmDice dice, int nd, uint64_t* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nd) return;
  const double x = dice.x[t];
  uint64_t lo = 0, hi = n + 1;  // first index with cum[idx] > x
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (cum[mid] > x) hi = mid; else lo = mid + 1;
  }
  out[t] = lo - 1;
}
int k_search_args(isle_ctx* c, const double* cum, uint64_t n, const double* dice_host, int nd, uint64_t* out_dev) {
  if (nd == 0) return 0;
  if (nd > 16) return isle_fail(c, ISLE_E_ARG, "k_search_args: %d > 16 dice", nd);
  KmDice dd;
  for (int i = 0; i < 16; ++i) dd.x[i] = i < nd ? dice_host[i] : 0.0;
  hipLaunchKernelGGL(search_args_k, dim3(1), dim3(64), 0, c->stream, cum, n, dd, nd, out_dev);
  HIPCHK(c, hipGetLastError());
  return 0;
}
// One launch for everything a single-rank k-means++ round needs from the device: dice[t] = total * frac[t] (the host's
// `grand * rng.fraction()` of src/sparseMatrix.cpp:2184 — the same IEEE double product, taken where the total lives), its
// position in the prefix sums, and the two scalars {total, last min-distance}: one copy and one round trip per round instead of two.
struct KmFrac {
  double f[40];
};
// One wave per die: the 64 lanes probe 64 evenly spaced positions of the bracket at once (a binary search by one lane is 21 dependent
// loads at D = 1M, ~20 us per round; this is 4).  The answer is the same: the first index whose prefix sum exceeds the die.
__global__ __launch_bounds__(64) void search_frac_k(const double* __restrict__ cum, uint64_t n, const float* __restrict__ last, KmFrac frac, int nd,
                                                     uint64_t* __restrict__ out /*40 positions, then the 2 scalars as doubles*/) {
  const int lane = threadIdx.x, t = blockIdx.x;
  const double total = cum[n];
  if (t == 0 && lane == 0) {
    double* o2 = reinterpret_cast<double*>(out + 40);
    o2[0] = total;
    o2[1] = last ? (double)last[0] : 0.0;
  }
  if (t >= nd) return;
  double fr = 0.0;
#pragma unroll
  for (int i = 0; i < 40; ++i)  // frac lives in kernel-argument registers: a run-time index would spill it
    if (i == t) fr = frac.f[i];
  const double x = fmin(fmax(total * fr, 0.0), total);
  uint64_t lo = 0, hi = n + 1;  // the first index with cum[idx] > x lies in [lo, hi]; cum[idx] > x for all idx >= hi
  while (lo < hi) {
    const uint64_t len = hi - lo;  // candidates lo .. hi - 1, and hi itself if none of them qualifies
    const uint64_t step = (len + 63) / 64;
    const uint64_t idx = lo + (uint64_t)lane * step;  // lanes probe lo, lo + step, ...
    const bool in = idx < hi;
    const bool gt = in && cum[idx] > x;
    const unsigned long long m = __ballot(gt);
    if (m) {  // the first probing lane that exceeds x: the answer is in (previous probe, that probe]
      const int f = __ffsll((long long)m) - 1;
      hi = lo + (uint64_t)f * step;
      lo = f ? lo + (uint64_t)(f - 1) * step + 1 : lo;
      if (f == 0) hi = lo;  // cum[lo] > x already
    } else {  // no probe exceeds x: the answer lies behind the last probe
      const unsigned long long inm = __ballot(in);
      const int lastl = 63 - __builtin_clzll(inm);
      lo = lo + (uint64_t)lastl * step + 1;
    }
  }
  if (lane == 0) out[t] = lo - 1;
}
int k_search_frac(isle_ctx* c, const double* cum, uint64_t n, const float* last, const double* frac_host, int nd, uint64_t* out_dev) {
  if (nd > 40) return isle_fail(c, ISLE_E_ARG, "k_search_frac: %d > 40 dice", nd);
  KmFrac ff;
  for (int i = 0; i < 40; ++i) ff.f[i] = i < nd ? frac_host[i] : 0.0;
  hipLaunchKernelGGL(search_frac_k, dim3(nd > 0 ? nd : 1), dim3(64), 0, c->stream, cum, n, last, ff, nd, out_dev);
  HIPCHK(c, hipGetLastError());
  return 0;
}
struct KmIds {
  uint64_t id[16];  // local row, or ~0 for "not on this rank" (destination row zeroed)
};
__global__ __launch_bounds__(256) void fetch_rows_k(const float* __restrict__ P, int ldk, KmIds ids, int n, float* __restrict__ dst) {
  const int r = blockIdx.x;
  if (r >= n) return;
  const uint64_t id = ids.id[r];
  for (int j = threadIdx.x; j < ldk; j += 256) dst[(size_t)r * ldk + j] = id == ~0ull ? 0.f : P[(size_t)id * ldk + j];
}
int k_fetch_rows(isle_ctx* c, const float* P, int ldk, const uint64_t* local_ids /*~0: absent*/, int n, float* dst) {
  for (int i0 = 0; i0 < n; i0 += 16) {
    KmIds ids;
    const int m = std::min(16, n - i0);
    for (int i = 0; i < 16; ++i) ids.id[i] = i < m ? local_ids[i0 + i] : ~0ull;
    hipLaunchKernelGGL(fetch_rows_k, dim3(m), dim3(256), 0, c->stream, P, ldk, ids, m, dst + (size_t)i0 * ldk);
  }
  HIPCHK(c, hipGetLastError());
  return 0;
}
// {cum[n], last[0]} -> out2 (doubles): the two scalars a k-means++ round needs on the host, one copy instead of two
__global__ void pack2_k(const double* __restrict__ a, const float* __restrict__ b, double* __restrict__ out2) {
  out2[0] = a[0];
  out2[1] = b ? (double)b[0] : 0.0;
}
int k_pack2(isle_ctx* c, const double* a, const float* b, double* out2) {
  hipLaunchKernelGGL(pack2_k, dim3(1), dim3(1), 0, c->stream, a, b, out2);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int k_scan_f2d(isle_ctx* c, const float* in, uint64_t n, double* cum) {
  TimeScope ts(c, ISLE_T_KMPP);
  HIPCHK(c, c->scan_blk.reserve(isle_scan::scan_scratch_elems(n)));
  HIPCHK(c, (isle_scan::exclusive_scan<float, double>(c->stream, in, n, cum, c->scan_blk.p)));
  return 0;
}

// upper_bound(cum[0..n], dice) - 1   (src/sparseMatrix.cpp:2186-2188); cum has n+1 entries
__global__ void search_k(const double* __restrict__ cum, uint64_t n, const double* __restrict__ dice, int nd, uint64_t* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nd) return;
  const double x = dice[t];
  uint64_t lo = 0, hi = n + 1;  // first index with cum[idx] > x
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (cum[mid] > x) hi = mid; else lo = mid + 1;
  }
  out[t] = lo - 1;
}
int k_search(isle_ctx* c, const double* cum, uint64_t n, const double* dice_dev, int nd, uint64_t* out_dev) {
  if (nd == 0) return 0;
  hipLaunchKernelGGL(search_k, dim3(cdiv(nd, 64)), dim3(64), 0, c->stream, cum, n, dice_dev, nd, out_dev);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// Projected assignment: 128 documents per workgroup (32 per wave), centres in tiles of 32, K staged
// through LDS in slabs of 32 coordinates.  v_mfma_f32_32x32x2_f32 with centres on the MFMA row index and
// documents on the MFMA column (= lane) index, so each lane keeps a running (|dist|, index) for ONE
// document across all centre tiles; the D x k distance matrix never exists.
// ------------------------------------------------------------------------------------------
constexpr int PA_DOCS = 128, PA_CT = 32, PA_BK = 32;
__global__ __launch_bounds__(256) void proj_assign_k(const float* __restrict__ P, const float* __restrict__ pn, uint32_t D, int k, int ldk,
                                                      const float* __restrict__ C, const float* __restrict__ cn,
                                                      uint32_t* __restrict__ assign) {
  __shared__ float Ps[PA_DOCS][PA_BK + 1];
  __shared__ float Cs[PA_CT][PA_BK + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const uint32_t d0 = blockIdx.x * PA_DOCS;
  const uint32_t myd = d0 + 32 * wave + l31;
  const float nd = (myd < D) ? pn[myd] : 0.f;
  float best = 3.4e38f;
  uint32_t bidx = 0xffffffffu;
  for (int c0 = 0; c0 < k; c0 += PA_CT) {
    floatx16 acc = {0};
    for (int k0 = 0; k0 < ldk; k0 += PA_BK) {
      __syncthreads();
#pragma unroll
      for (int u = 0; u < (PA_DOCS * PA_BK) / 256; ++u) {
        const int idx = threadIdx.x + 256 * u;
        const int row = idx / PA_BK, col = idx - row * PA_BK;
        const uint32_t dd = d0 + row;
        const float pv = P[(size_t)min(dd, D - 1) * ldk + min(k0 + col, ldk - 1)];
        Ps[row][col] = pv * ((dd < D && k0 + col < ldk) ? 1.f : 0.f);
      }
#pragma unroll
      for (int u = 0; u < (PA_CT * PA_BK) / 256; ++u) {
        const int idx = threadIdx.x + 256 * u;
        const int row = idx / PA_BK, col = idx - row * PA_BK;
        const float cv = C[(size_t)min(c0 + row, k - 1) * ldk + min(k0 + col, ldk - 1)];
        Cs[row][col] = cv * ((c0 + row < k && k0 + col < ldk) ? 1.f : 0.f);
      }
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < PA_BK; kk += 2) {
        const float a = Cs[l31][kk + h];
        const float b = Ps[32 * wave + l31][kk + h];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cc = c0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (cc < k) {
        const float dist = fabsf((-2.0f * acc[r] + cn[cc]) + nd);
        if (dist < best || (dist == best && (uint32_t)cc < bidx)) {
          best = dist;
          bidx = (uint32_t)cc;
        }
      }
    }
  }
  const float ob = __shfl_xor(best, 32);
  const uint32_t oi = __shfl_xor(bidx, 32);
  if (ob < best || (ob == best && oi < bidx)) {
    best = ob;
    bidx = oi;
  }
  if (h == 0 && myd < D) assign[myd] = bidx;
}
// Register-resident variant for ldk <= 256: every lane keeps its half-row of P (KH = ldk/2 coordinates: MFMA k-slot h
// of step i carries coordinate h*KH + i) in registers for the whole kernel, so P is read from HBM exactly once per
// iteration; centres stream through LDS in slabs of 2 x 16 coordinates and all centre tiles accumulate side by side.
constexpr int PR_SL = 16;
// MODE PR_ARGMIN : assign[d] = first index of min |dist|        (Lloyd, cblas_isamin semantics)
// MODE PR_MINDIST: min_dist[d] = min(min_dist[d], max(dist, 0)) (k-means++ round against the newest <= 32*CTMAX centres)
template <int NSLAB, int CTMAX, int MODE>
__global__ __launch_bounds__(256, (NSLAB * 16 + CTMAX * 16 > 224) ? 1 : 2) void proj_assign_reg_k(const float* __restrict__ Pt /*ldk x D*/, const float* __restrict__ pn,
                                                             uint32_t D, int k, int ldk, const float* __restrict__ C,
                                                             const float* __restrict__ cn, uint32_t* __restrict__ assign,
                                                             float* __restrict__ min_dist, const uint32_t* __restrict__ map,
                                                             float* __restrict__ ub, float* __restrict__ lb,
                                                             const uint32_t* __restrict__ need /*PR_TILES: tiles to examine per document, null = all*/,
                                                             int TL /*PR_TILES: row stride of lb = tile bounds*/) {
  const int dbg = TL >> 16;  // timing experiments only (ISLE_PT_DBG): 1 = no tile-bound stores, 2 = no tile epilogue at all
  TL &= 0xffff;
  constexpr int KHC = NSLAB * PR_SL;  // coordinates per lane half held in registers at a time
  constexpr int CG = CTMAX * 32;      // centres per group (their accumulators live side by side)
  extern __shared__ float Cs[];       // [2][CG][PR_SL + 1]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int KH = ldk >> 1;
  const uint32_t myd = blockIdx.x * 128 + 32 * wave + l31;
  const float live = (myd < D) ? 1.f : 0.f;
  const float nd = (myd < D) ? pn[myd] : 0.f;
  const float* col = Pt + (size_t)h * KH * D + min(myd, D - 1);
  float best = 3.4e38f, second = 3.4e38f, cmax = 0.f;
  uint32_t bidx = 0xffffffffu;
  // PR_TILES (bounds per tile of 32 centres, see pt_filter_k): the workgroup examines the union of the tiles its documents need
  const uint32_t dst = (myd < D) ? (map ? map[myd] : myd) : 0u;
  uint32_t wneed = 0xffffffffu, btile = 0;
  float btc = 0.f;
  if (MODE == PR_TILES && need) {
    __shared__ uint32_t sh_need;
    if (threadIdx.x == 0) sh_need = 0u;
    __syncthreads();
    const uint32_t mine = (myd < D && h == 0) ? need[dst] : 0u;
    if (mine) atomicOr(&sh_need, mine);
    __syncthreads();
    wneed = sh_need;
  }
  // k <= CG and KH <= KHC (e.g. k = 200) is a single pass: P is then read from HBM exactly once per call.
  for (int cg0 = 0; cg0 < k; cg0 += CG) {
    const uint32_t gmask = (wneed >> (cg0 >> 5)) & ((CTMAX >= 32) ? 0xffffffffu : ((1u << CTMAX) - 1u));
    if (MODE == PR_TILES && gmask == 0u) continue;  // uniform over the workgroup
    floatx16 acc[CTMAX];
#pragma unroll
    for (int t = 0; t < CTMAX; ++t) acc[t] = (floatx16){0};
    for (int kc0 = 0; kc0 < KH; kc0 += KHC) {
      float p[KHC];
      // clamped addresses + mask multiply: every load is unconditional and independent (a guarded load makes hipcc
      // branch around each one and wait for it separately)
#pragma unroll
      for (int i = 0; i < KHC; ++i) p[i] = col[(size_t)min(kc0 + i, KH - 1) * D] * ((kc0 + i < KH) ? live : 0.f);
#pragma unroll
      for (int s = 0; s < NSLAB; ++s) {
        const int cb0 = kc0 + s * PR_SL;  // first coordinate (within a half) of this slab
        if (cb0 < KH) {
          __syncthreads();
          if (MODE != PR_TILES || gmask == ((CTMAX >= 32) ? 0xffffffffu : ((1u << CTMAX) - 1u))) {
            if (NSLAB * 16 + CTMAX * 16 > 224) {
              // 2 halves x CG centres x 16 coordinates: thread (ii = tid % 16, c16 = tid / 16) takes centres c16, c16 + 16, ... of both
              // halves.  All loads of a batch are issued before the first LDS store (a rolled loop waited for every load in turn:
              // 8 us of staging per 3.4 us of matrix-core work at k = 1000)
              const int ii = threadIdx.x % PR_SL, c16 = threadIdx.x / PR_SL;
              const int coord = cb0 + ii;
              const float cmask = coord < KH ? 1.f : 0.f;
              constexpr int NLD = 2 * CG / 16;  // loads per thread
              constexpr int NBW = 16;
              constexpr int NB_ = NLD < NBW ? NLD : NBW;
  #pragma unroll
              for (int u0 = 0; u0 < NLD; u0 += NB_) {
                float v[NB_];
  #pragma unroll
                for (int u = 0; u < NB_; ++u) {
                  if (u0 + u >= NLD) break;
                  const int hh = (u0 + u) / (CG / 16), cc = c16 + 16 * ((u0 + u) % (CG / 16));
                  v[u] = C[(size_t)min(cg0 + cc, k - 1) * ldk + hh * KH + min(coord, KH - 1)] * ((cg0 + cc < k) ? cmask : 0.f);  // unconditional, masked
                }
  #pragma unroll
                for (int u = 0; u < NB_; ++u) {
                  if (u0 + u >= NLD) break;
                  const int hh = (u0 + u) / (CG / 16), cc = c16 + 16 * ((u0 + u) % (CG / 16));
                  Cs[(hh * CG + cc) * (PR_SL + 1) + ii] = v[u];
                }
              }
            } else {  // two workgroups per CU: no registers to spare for a batch (it spills), and the other workgroup hides the latency
              for (int idx = threadIdx.x; idx < 2 * CG * PR_SL; idx += 256) {
                const int ii = idx % PR_SL;
                const int cc = (idx / PR_SL) % CG;
                const int hh = idx / (PR_SL * CG);
                const int coord = cb0 + ii;
                const float cv = C[(size_t)min(cg0 + cc, k - 1) * ldk + hh * KH + min(coord, KH - 1)];  // unconditional, masked below
                Cs[(hh * CG + cc) * (PR_SL + 1) + ii] = cv * ((cg0 + cc < k && coord < KH) ? 1.f : 0.f);
              }
            }
          } else {
            // only the tiles that are examined: a tile is 2 halves x 32 centres x 16 coordinates = 1024 floats, four per thread,
            // loaded together (the tile test is uniform over the workgroup: a scalar branch around unconditional loads)
            const int ii = threadIdx.x % PR_SL, c16 = threadIdx.x / PR_SL;  // 16 centres per 256 threads
            const int coord = cb0 + ii;
            const float cmask = coord < KH ? 1.f : 0.f;
#pragma unroll
            for (int t = 0; t < CTMAX; ++t) {
              if ((gmask >> t) & 1u) {
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const int hh = q >> 1, cc = 32 * t + 16 * (q & 1) + c16;
                  v[q] = C[(size_t)min(cg0 + cc, k - 1) * ldk + hh * KH + min(coord, KH - 1)] * ((cg0 + cc < k) ? cmask : 0.f);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const int hh = q >> 1, cc = 32 * t + 16 * (q & 1) + c16;
                  Cs[(hh * CG + cc) * (PR_SL + 1) + ii] = v[q];
                }
              }
            }
          }
          __syncthreads();
#pragma unroll
          for (int t = 0; t < CTMAX; ++t) {
            if (cg0 + 32 * t < k && (MODE != PR_TILES || ((gmask >> t) & 1u))) {
              const float* cb = &Cs[(h * CG + 32 * t + l31) * (PR_SL + 1)];
#pragma unroll
              for (int ii = 0; ii < PR_SL; ++ii) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(cb[ii], p[s * PR_SL + ii], acc[t], 0, 0, 0);
            }
          }
        }
      }
    }
    if (MODE == PR_TILES) {
      // per tile: the two smallest |distances| and the first index of the smallest over its 32 centres (both lane halves); the
      // tile's bound is stored at once, the running best keeps the runner-up of ITS tile (a tile bound excludes the assigned centre)
#pragma unroll
      for (int t = 0; t < CTMAX; ++t) {
        if (cg0 + 32 * t < k && ((gmask >> t) & 1u) && !(dbg & 2)) {
          float m1 = 3.4e38f, m2 = 3.4e38f, tc = 0.f;
          uint32_t i1 = 0xffffffffu;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int cc = cg0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (cc < k) {
              tc = fmaxf(tc, cn[cc]);
              const float dist = fabsf((-2.0f * acc[t][r] + cn[cc]) + nd);
              if (dist < m1 || (dist == m1 && (uint32_t)cc < i1)) {
                m2 = m1;
                m1 = dist;
                i1 = (uint32_t)cc;
              } else {
                m2 = fminf(m2, dist);
              }
            }
          }
          const float om1 = __shfl_xor(m1, 32), om2 = __shfl_xor(m2, 32);
          const uint32_t oi1 = (uint32_t)__shfl_xor((int)i1, 32);
          tc = fmaxf(tc, __shfl_xor(tc, 32));
          if (om1 < m1 || (om1 == m1 && oi1 < i1)) {
            m2 = fminf(m1, om2);
            m1 = om1;
            i1 = oi1;
          } else {
            m2 = fminf(m2, om1);
          }
          const uint32_t T = (uint32_t)(cg0 >> 5) + (uint32_t)t;
          if (m1 < best || (m1 == best && i1 < bidx)) {
            best = m1;
            bidx = i1;
            second = m2;
            btile = T;
            btc = tc;
          }
          cmax = fmaxf(cmax, tc);
          if (h == 0 && myd < D && !(dbg & 1)) {
            float uu, ll;
            hamerly_store_bounds(m1, m1, nd + tc, &uu, &ll);
            lb[(size_t)dst * TL + T] = ll;
          }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < CTMAX; ++t) {
      if ((MODE != PR_TILES || (dbg & 2)) && cg0 + 32 * t < k) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cc = cg0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (cc < k) {
            cmax = fmaxf(cmax, cn[cc]);
            const float raw = (-2.0f * acc[t][r] + cn[cc]) + nd;
            if (MODE == PR_ARGMIN || MODE == PR_TILES) {
              const float dist = fabsf(raw);
              if (dist < best || (dist == best && (uint32_t)cc < bidx)) {
                second = best;
                best = dist;
                bidx = (uint32_t)cc;
              } else {
                second = fminf(second, dist);
              }
            } else {
              best = fminf(best, fmaxf(raw, 0.0f));
            }
          }
        }
      }
    }
  }
  if (MODE == PR_TILES && !(dbg & 2)) {  // both lane halves hold the same (best, index, runner-up of the best's tile)
    if (h == 0 && myd < D) {
      assign[dst] = bidx;
      float uu, ll, l2;
      hamerly_store_bounds(best, second, nd + btc, &uu, &l2);
      hamerly_store_bounds(best, best, nd + cmax, &uu, &ll);
      ub[dst] = uu;
      lb[(size_t)dst * TL + btile] = l2;  // the assigned centre's tile: closest OTHER centre in it
    }
    return;
  }
  const float ob = __shfl_xor(best, 32);
  const float os = __shfl_xor(second, 32);
  const uint32_t oi = __shfl_xor(bidx, 32);
  cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
  if (MODE == PR_ARGMIN || MODE == PR_TILES) {
    if (ob < best || (ob == best && oi < bidx)) {
      second = fminf(best, os);
      best = ob;
      bidx = oi;
    } else {
      second = fminf(second, ob);
    }
    if (h == 0 && myd < D) {
      const uint32_t dst = map ? map[myd] : myd;  // compacted (active-list) launches write through the slot -> doc map
      assign[dst] = bidx;
      if (ub) hamerly_store_bounds(best, second, nd + cmax, &ub[dst], &lb[dst]);
    }
  } else {
    best = fminf(best, ob);
    if (h == 0 && myd < D) min_dist[myd] = fminf(min_dist[myd], best);
  }
}

// dispatch over (coordinate slabs, centre tiles); returns false if the shape is not covered
template <int MODE>
static int launch_proj_reg(isle_ctx* c, uint64_t D, int k, int ldk, const float* C, const float* cn, const float* pn, uint32_t* assign,
                           float* min_dist, bool* done, const float* Pt, const uint32_t* map, float* ub, float* lb, const uint32_t* need, int TL) {
  *done = false;
  if (!D) return 0;
  if (!Pt) {  // the context's projection, coordinate-major: made from P if the default routes have not needed it yet
    if (!(c->Pt_ready || c->Pt2_ready)) return 0;
    ISLECHK(k_ensure_pt(c));
    Pt = c->Pt.p;
  }
  const int kpad = (k + 31) & ~31;
  const int ct = kpad / 32;
  const int nslab = cdiv(ldk / 2, PR_SL);
  dim3 g(cdiv(D, 128)), b(256);
#define LR(NS, CM)                                                                                                            \
  do {                                                                                                                        \
    const size_t lds = (size_t)2 * (CM * 32) * (PR_SL + 1) * sizeof(float);                                                   \
    ISLECHK(isle_max_lds(c, (const void*)proj_assign_reg_k<NS, CM, MODE>, (int)lds));                                        \
    hipLaunchKernelGGL((proj_assign_reg_k<NS, CM, MODE>), g, b, lds, c->stream, Pt, pn, (uint32_t)D, k, ldk, C, cn, assign, min_dist, map, ub, lb, \
                       need, TL);                                                                                             \
    *done = true;                                                                                                             \
  } while (0)
  if (MODE == PR_TILES) {  // only the shape that needs it: more than 7 tiles or slabs (k > 224), at most 32 tiles
    if (ct > 32) return 0;
    LR(8, 8);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  if (ct <= 1) {
    if (nslab <= 2) LR(2, 1);
    else if (nslab <= 4) LR(4, 1);
    else if (nslab <= 7) LR(7, 1);
    else LR(8, 1);  // loops over coordinate chunks of 128 per half when ldk > 256
  } else if (ct <= 2 && nslab <= 2) LR(2, 2);
  else if (ct <= 4 && nslab <= 4) LR(4, 4);
  else if (ct <= 7 && nslab <= 7) LR(7, 7);
  else LR(8, 8);  // loops over centre groups of 256 and coordinate chunks of 128 per half
#undef LR
  HIPCHK(c, hipGetLastError());
  return 0;
}

// Compacts the projected rows of the active documents into a coordinate-major panel (ldk x n) so that the
// register-resident kernel can run on them unchanged: Pa[j * n + i] = P[active[i] * ldk + j], pna[i] = pn[active[i]].
__global__ __launch_bounds__(256) void compact_rows_k(const float* __restrict__ P, const float* __restrict__ pn, int ldk,
                                                       const uint32_t* __restrict__ active, uint32_t n, float* __restrict__ Pa,
                                                       float* __restrict__ pna) {
  __shared__ float t[32][33];
  const uint32_t i0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j0 = 0; j0 < ldk; j0 += 32) {
    __syncthreads();
    for (int yy = ty; yy < 32; yy += 8) {  // row yy of the tile = active doc i0+yy, coordinates j0..j0+31 (coalesced 128 B)
      const uint32_t i = i0 + yy;
      const uint32_t d = active[min(i, n - 1)];
      t[yy][tx] = P[(size_t)d * ldk + min(j0 + tx, ldk - 1)];
    }
    __syncthreads();
    for (int yy = ty; yy < 32; yy += 8) {
      const int j = j0 + yy;
      const uint32_t i = i0 + tx;
      if (j < ldk && i < n) Pa[(size_t)j * n + i] = t[tx][yy];
    }
  }
  if (threadIdx.x < 32 && i0 + threadIdx.x < n) pna[i0 + threadIdx.x] = pn[active[i0 + threadIdx.x]];
}

// The same for long lists: 64 documents per workgroup and 256 coordinates at a time — a row is read in 1 kB pieces (one float4 per lane,
// sixteen rows in flight per wave) and a coordinate's 64 documents leave as one 256-byte store, where the 32 x 32 tiles above move 128 bytes
// either way (config 3, 2 M active documents of 4 kB: 7.8 -> ms).
__global__ __launch_bounds__(256) void compact_rows64_k(const float* __restrict__ P, const float* __restrict__ pn, int ldk,
                                                         const uint32_t* __restrict__ active, uint32_t n, float* __restrict__ Pa,
                                                         float* __restrict__ pna) {
  extern __shared__ float t64[];  // 64 x 257
  const uint32_t i0 = blockIdx.x * 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t myd = active[min(i0 + (uint32_t)lane, n - 1)];
  for (int j0 = 0; j0 < ldk; j0 += 256) {
    __syncthreads();
    const int j = j0 + 4 * lane;  // ldk is a multiple of 4: a float4 is inside the row or outside it
    float4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const uint32_t d = (uint32_t)__shfl((int)myd, w * 16 + u);
      v[u] = *reinterpret_cast<const float4*>(P + (size_t)d * ldk + min(j, ldk - 4));
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      float* r = t64 + (size_t)(w * 16 + u) * 257 + 4 * lane;
      r[0] = v[u].x;
      r[1] = v[u].y;
      r[2] = v[u].z;
      r[3] = v[u].w;
    }
    __syncthreads();
    if (i0 + lane < n) {
#pragma unroll 8
      for (int cc = 0; cc < 64; ++cc) {
        const int cj = j0 + w * 64 + cc;
        if (cj < ldk) Pa[(size_t)cj * n + i0 + lane] = t64[(size_t)lane * 257 + w * 64 + cc];
      }
    }
  }
  if (threadIdx.x < 64 && i0 + threadIdx.x < n) pna[i0 + threadIdx.x] = pn[myd];
}

int k_compact_rows(isle_ctx* c, const float* P, const float* pn, int ldk, const uint32_t* active, uint32_t n, float* Pa, float* pna) {
  if (n >= 4096 && ldk >= 256 && ldk % 4 == 0) {
    const size_t lds = (size_t)64 * 257 * sizeof(float);
    ISLECHK(isle_max_lds(c, (const void*)compact_rows64_k, (int)lds));
    hipLaunchKernelGGL(compact_rows64_k, dim3(cdiv(n, 64)), dim3(256), lds, c->stream, P, pn, ldk, active, n, Pa, pna);
  } else {
    hipLaunchKernelGGL(compact_rows_k, dim3(cdiv(n, 32)), dim3(256), 0, c->stream, P, pn, ldk, active, n, Pa, pna);
  }
  HIPCHK(c, hipGetLastError());
  return 0;
}

// Keys of the active documents of a tile-bound iteration: the set of tiles a document has to re-examine (pt_filter_k / pt_tighten_k).
// A workgroup of proj_assign_reg_k<.., PR_TILES> examines the UNION of its 128 documents' sets; in member order that union was 9 of the
// 32 tiles at config 3 where a document asks for 2 - 3, so the active list is sorted by the set first (as a number: largest tile, then the
// next ...).  The order of the list changes no document's result: every distance a workgroup forms is exact and only tightens bounds.
__global__ __launch_bounds__(256) void pt_need_keys_k(const uint32_t* __restrict__ active, uint32_t n, const uint32_t* __restrict__ need,
                                                       uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const uint32_t d = active[i];
    key[i] = need[d];
    val[i] = d;
  }
}

// assignment of the n active documents (Hamerly); results are written through `active` into assign / ub / lb
int k_proj_assign_active(isle_ctx* c, const float* P, const float* pn, int k, int ldk, const float* C, const float* cn,
                         const uint32_t* active, uint32_t n, float* Pa, float* pna, uint32_t* assign, float* ub, float* lb) {
  TimeScope ts(c, ISLE_T_LLOYD_PROJ);
  if (n == 0) return 0;
  ISLECHK(k_compact_rows(c, P, pn, ldk, active, n, Pa, pna));
  bool done = false;
  ISLECHK(launch_proj_reg<PR_ARGMIN>(c, n, k, ldk, C, cn, pna, assign, nullptr, &done, Pa, active, ub, lb));
  if (!done) return isle_fail(c, ISLE_E_ARG, "projected assignment: register kernel unavailable");
  return 0;
}

// Tile bounds (pt_filter_k, spmm.hip): full assignment that also leaves, per document, an upper bound on the distance to its
// centre and one lower bound per tile of 32 centres (row stride TL); need == null examines every tile, otherwise the n documents
// of `active` (compacted into Pa) examine the tiles their masks name.
// Epilogue of the full tile-bound assignment when the D x k dot products come from one plain GEMM (k_proj_assign_tiles): one thread per
// document walks its row of the coordinate-major dot matrix (column j = centre j, so a wave reads consecutive documents of one centre)
// and forms exactly what the PR_TILES epilogue of proj_assign_reg_k forms — per tile of 32 centres the smallest distance, its first index
// and the runner-up; the bound of every tile; the assignment; the upper bound; the runner-up bound for the assigned centre's tile.
__global__ __launch_bounds__(256) void proj_dots_tiles_k(const float* __restrict__ dotsT, uint32_t D, int k, const float* __restrict__ pn,
                                                          const float* __restrict__ cn, uint32_t* __restrict__ assign, float* __restrict__ ub,
                                                          float* __restrict__ lb, int TL) {
  const uint32_t d = blockIdx.x * 256 + threadIdx.x;
  if (d >= D) return;
  const float nd = pn[d];
  float best = 3.4e38f, second = 3.4e38f, btc = 0.f, cmax = 0.f;
  uint32_t bidx = 0xffffffffu, btile = 0;
  for (int c0 = 0; c0 < k; c0 += 32) {
    float m1 = 3.4e38f, m2 = 3.4e38f, tc = 0.f;
    uint32_t i1 = 0xffffffffu;
    float dot[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) dot[j] = dotsT[(size_t)min(c0 + j, k - 1) * D + d];  // thirty-two loads in flight
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const int cc = c0 + j;
      if (cc < k) {
        const float cnj = cn[cc];
        tc = fmaxf(tc, cnj);
        const float dist = fabsf((-2.0f * dot[j] + cnj) + nd);
        if (dist < m1) {  // ascending index: a tie keeps the earlier centre
          m2 = m1;
          m1 = dist;
          i1 = (uint32_t)cc;
        } else {
          m2 = fminf(m2, dist);
        }
      }
    }
    const uint32_t T = (uint32_t)(c0 >> 5);
    if (m1 < best) {
      best = m1;
      bidx = i1;
      second = m2;
      btile = T;
      btc = tc;
    }
    cmax = fmaxf(cmax, tc);
    float uu, ll;
    hamerly_store_bounds(m1, m1, nd + tc, &uu, &ll);
    lb[(size_t)d * TL + T] = ll;
  }
  assign[d] = bidx;
  float uu, ll, l2;
  hamerly_store_bounds(best, second, nd + btc, &uu, &l2);
  hamerly_store_bounds(best, best, nd + cmax, &uu, &ll);
  ub[d] = uu;
  lb[(size_t)d * TL + btile] = l2;  // the assigned centre's tile: closest OTHER centre in it
}

// Lloyd's first assignment in span(U) from what the k-means++ rounds kept (kmpp_min_dots_track_k): best = distances to the nearest of all k
// seeds; the tile minima (tile-major) become the tile bounds (document-major rows of c->ptlb).
int k_kmpp_to_tiles(isle_ctx* c, uint64_t D, int k, const float* pn, const float* cn, const float* best, uint32_t* assign, float* ub, int TL) {
  TimeScope ts(c, ISLE_T_LLOYD_PROJ);
  if (D == 0) return 0;
  hipLaunchKernelGGL(kmpp_to_tiles_k, dim3(cdiv(D, 256)), dim3(256), 0, c->stream, (uint32_t)D, k, pn, cn, best, c->kmpp_arg.p, c->kmpp_m2a.p, c->kmpp_tmin.p, assign,
                     ub, c->ptlb.p, TL);
  HIPCHK(c, hipGetLastError());
  return 0;
}

bool k_proj_full_by_gemm(isle_ctx* c, uint64_t D, int k) {
  const char* pf = c->knob(KN_PROJ_FULL);
  return (c->Pt_ready || c->Pt2_ready) && D > 0 && k >= 64 && (k_gemm_assign_fused_ok(c, D, k, k) || isle_scratch_ok(c, c->dotsT.cap, (double)D * k * sizeof(float))) &&
         ((2.0 * (double)D * k * k >= 2e10 && !(pf && !strcmp(pf, "fused"))) || (pf && !strcmp(pf, "gemm")));
}
int k_proj_assign_tiles(isle_ctx* c, const float* P, const float* pn, uint64_t D, int k, int ldk, const float* C, const float* cn, uint32_t* assign,
                        float* ub, float* tlb, int TL, const uint32_t* active, uint32_t n, const uint32_t* need, float* Pa, float* pna) {
  TimeScope ts(c, ISLE_T_LLOYD_PROJ);
  bool done = false;
  if (!active) {
    // The full pass is 2 D k^2 flop of plain matrix product: above ~20 GFLOP it goes through the library GEMM (135 TFLOP/s against the
    // 69 of the fused kernel's full pass, k_gemm_nn) on the coordinate-major copy of P, followed by the epilogue above over the D x k
    // dot products (5 GB at a C3 shard: 18 + 2 ms instead of 36).  ISLE_PROJ_FULL=gemm|fused forces a route.
    if (k_proj_full_by_gemm(c, D, k)) {
      if (k_gemm_assign_fused_ok(c, D, k, k)) {  // distances, tile bounds and candidates formed inside the product: no D x k matrix in memory
        HIPCHK(c, c->cmax_buf.reserve(4));
        ISLECHK(k_max_f32(c, cn, k, c->cmax_buf.p));
        const bool a2 = c->Pt2_ready && P == c->P.p;
        return k_gemm_assign_tiles(c, c->Pt_ready ? c->Pt.p : nullptr, P, ldk, D, k, C, ldk, k, TL, cn, pn, c->cmax_buf.p, assign, ub, tlb, ISLE_T_LLOYD_PROJ, nullptr,
                                   a2 ? c->Pt2.p : nullptr, a2 && c->Pt2_pos ? c->dperm.p : nullptr);
      }
      HIPCHK(c, c->dotsT.reserve((size_t)D * k));
      ISLECHK(k_ensure_pt(c));
      ISLECHK(k_gemm_nn_assign(c, c->Pt.p, D, k, C, ldk, k, c->dotsT.p, ISLE_T_LLOYD_PROJ));
      hipLaunchKernelGGL(proj_dots_tiles_k, dim3(cdiv(D, 256)), dim3(256), 0, c->stream, c->dotsT.p, (uint32_t)D, k, pn, cn, assign, ub, tlb, TL);
      HIPCHK(c, hipGetLastError());
      return 0;
    }
    ISLECHK(launch_proj_reg<PR_TILES>(c, D, k, ldk, C, cn, pn, assign, nullptr, &done, nullptr, nullptr, ub, tlb, nullptr, TL));
  } else {
    if (n == 0) return 0;
    {
      // All centres for the active documents through the assignment product on their gathered rows (two bf16 terms first): 5.9 us per
      // 1000 documents against 6 - 11 for the tiles they name through the register kernel — and EVERY tile bound of these documents is
      // refreshed, so fewer of them come back in the next iteration.
      const char* pa = c->knob(KN_PROJ_ACTIVE);
      const bool by_gemm = pa ? !strcmp(pa, "gemm") : true;
      if (by_gemm && k_gemm_assign_fused_ok(c, n, k, k)) {
        ISLECHK(k_compact_rows(c, P, pn, ldk, active, n, Pa, pna));
        HIPCHK(c, c->cmax_buf.reserve(4));
        ISLECHK(k_max_f32(c, cn, k, c->cmax_buf.p));
        return k_gemm_assign_tiles(c, Pa, P, ldk, n, k, C, ldk, k, TL, cn, pn, c->cmax_buf.p, assign, ub, tlb, ISLE_T_LLOYD_PROJ, active);
      }
    }
    if (need && n > 256 && !c->knob_zero(KN_PT_SORT)) {  // documents that ask for the same tiles next to each other (pt_need_keys_k)
      HIPCHK(c, c->gl_key_a.reserve(n));
      HIPCHK(c, c->gl_key_b.reserve(n));
      HIPCHK(c, c->gl_val_a.reserve(n));
      HIPCHK(c, c->gl_val_b.reserve(n));
      hipLaunchKernelGGL(pt_need_keys_k, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, active, n, need, c->gl_key_a.p, c->gl_val_a.p);
      HIPCHK(c, hipGetLastError());
      bool in_a = true;
      ISLECHK(k_sort_pairs_u64(c, c->gl_key_a.p, c->gl_val_a.p, c->gl_key_b.p, c->gl_val_b.p, n, ((k + 31) / 32 + 7) & ~7, &in_a));
      active = in_a ? c->gl_val_a.p : c->gl_val_b.p;
    }
    ISLECHK(k_compact_rows(c, P, pn, ldk, active, n, Pa, pna));
    ISLECHK(launch_proj_reg<PR_TILES>(c, n, k, ldk, C, cn, pna, assign, nullptr, &done, Pa, active, ub, tlb, need, TL));
  }
  if (!done) return isle_fail(c, ISLE_E_ARG, "projected assignment with tile bounds: shape not covered (k = %d)", k);
  return 0;
}

int k_proj_assign(isle_ctx* c, const float* P, const float* pn, uint64_t D, int k, int ldk, const float* C, const float* cn,
                  uint32_t* assign, float* ub, float* lb) {
  {
    TimeScope ts(c, ISLE_T_LLOYD_PROJ);
    bool done = false;
    ISLECHK(launch_proj_reg<PR_ARGMIN>(c, D, k, ldk, C, cn, pn, assign, nullptr, &done, nullptr, nullptr, ub, lb));
    if (done) return 0;
  }
  if (ub) return isle_fail(c, ISLE_E_ARG, "generic projected assignment has no bounds output");
  TimeScope ts(c, ISLE_T_LLOYD_PROJ);
  if (D == 0) return 0;
  hipLaunchKernelGGL(proj_assign_k, dim3(cdiv(D, PA_DOCS)), dim3(256), 0, c->stream, P, pn, (uint32_t)D, k, ldk, C, cn, assign);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// Member lists: members[off[c] .. off[c+1]) = documents assigned to centre c (order = placement atomics).
__global__ __launch_bounds__(256) void member_fill_k(const uint32_t* __restrict__ assign, uint32_t D, int k, const int* __restrict__ off,
                                                      int* __restrict__ cursor, uint32_t* __restrict__ members) {
  extern __shared__ int sh[];  // hist[k], base[k]
  int* hist = sh;
  int* base = sh + k;
  for (int j = threadIdx.x; j < k; j += 256) hist[j] = 0;
  __syncthreads();
  const uint32_t b0 = blockIdx.x * (256 * CS_ITEMS);
  uint32_t mine[CS_ITEMS];
  int rank[CS_ITEMS];
#pragma unroll
  for (int u = 0; u < CS_ITEMS; ++u) {
    const uint32_t d = b0 + threadIdx.x + 256 * u;
    mine[u] = (d < D) ? assign[d] : 0xffffffffu;
    rank[u] = (d < D) ? atomicAdd(&hist[mine[u]], 1) : 0;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < k; j += 256) base[j] = hist[j] ? off[j] + atomicAdd(&cursor[j], hist[j]) : 0;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < CS_ITEMS; ++u) {
    const uint32_t d = b0 + threadIdx.x + 256 * u;
    if (d < D) members[base[mine[u]] + rank[u]] = d;
  }
}

// Member sums of the projected Lloyd update in a FIXED order (the reference's centres are bitwise reproducible; float atomics over
// chunks that finish in any order are not).  The member lists are in ascending document order (k_member_lists); every centre's
// list is cut into chunks of 256 members, one workgroup per chunk (clusters differ 10x in size: a fixed split per centre left the
// largest one on a single workgroup), every wave sums whole rows (coalesced 4k-byte reads) in member order, the four waves are
// combined in LDS, and proj_segsum_reduce_k adds a centre's chunk rows in chunk order.  No atomics.
constexpr int SEG_MC = 256;
struct SegChunk {
  int beg, end;  // members [beg, end) of one centre
};
template <int NIT>  // float4 chunks per lane: ldk / 4 <= 64 NIT
__global__ __launch_bounds__(256) void proj_segsum_k(const float* __restrict__ P, int ldk, const SegChunk* __restrict__ chunks,
                                                      const uint32_t* __restrict__ members, float* __restrict__ part /*[chunk][ldk]*/) {
  extern __shared__ float red[];  // 4 x ldk
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nq = ldk / 4;
  const int beg = chunks[blockIdx.x].beg, end = chunks[blockIdx.x].end;
  float4 acc[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) acc[it] = make_float4(0.f, 0.f, 0.f, 0.f);
  // this wave's members: beg + wave, + 4, ... (at most 64): ids by one load and shuffles, four 16-byte row loads in flight
  const int myi = beg + wave + 4 * lane;
  const uint32_t mym = myi < end ? members[myi] : 0u;
  const int cnt = end > beg + wave ? min(64, (end - beg - wave + 3) / 4) : 0;
  for (int j = 0; j < cnt; j += 4) {
    float4 v[4][NIT];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t mj = (uint32_t)__shfl((int)mym, min(j + u, cnt - 1));
      const float4* row = reinterpret_cast<const float4*>(P + (size_t)mj * ldk);
      const float live = j + u < cnt ? 1.f : 0.f;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int q = lane + 64 * it;
        const float4 x = q < nq ? row[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        v[u][it] = make_float4(x.x * live, x.y * live, x.z * live, x.w * live);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        acc[it].x += v[u][it].x;
        acc[it].y += v[u][it].y;
        acc[it].z += v[u][it].z;
        acc[it].w += v[u][it].w;
      }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int q = lane + 64 * it;
    if (q < nq) reinterpret_cast<float4*>(red + (size_t)wave * ldk)[q] = acc[it];
  }
  __syncthreads();
  float* out = part + (size_t)blockIdx.x * ldk;
  for (int j = threadIdx.x; j < ldk; j += 256) out[j] = (red[j] + red[ldk + j]) + (red[2 * ldk + j] + red[3 * ldk + j]);
}
// Csum[c][:] = sum of the chunk rows [c0[c], c0[c + 1]) in order (zero for an empty centre)
__global__ __launch_bounds__(256) void proj_segsum_reduce_k(const float* __restrict__ part, const int* __restrict__ c0, int ldk, float* __restrict__ Csum) {
  const int cc = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= ldk) return;
  float s = 0.f;
  int q = c0[cc];
  const int qe = c0[cc + 1];
  for (; q + 8 <= qe; q += 8) {  // eight chunk rows in flight, added in chunk order (a rolled loop kept one: 177 us per call at C2)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(q + u) * ldk + j];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; q < qe; ++q) s += part[(size_t)q * ldk + j];
  Csum[(size_t)cc * ldk + j] = s;
}
__global__ __launch_bounds__(256) void member_keys_k(const uint32_t* __restrict__ assign, uint32_t D, uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
  const uint32_t d = blockIdx.x * 256 + threadIdx.x;
  if (d < D) {
    key[d] = assign[d];
    val[d] = d;
  }
}

// Device-only variant: offsets by a one-workgroup scan of the counts, no host round trip (the sparse Lloyd loop calls it every
// iteration and needs nothing of it on the host).
__global__ __launch_bounds__(256) void member_offsets_k(const int* __restrict__ counts, int k, int* __restrict__ off /*k+1*/, int* __restrict__ cursor /*k*/) {
  __shared__ int sh[256];
  int carry = 0;
  for (int base = 0; base < k; base += 256) {
    const int i = base + threadIdx.x;
    const int v = i < k ? counts[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      const int add = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
      __syncthreads();
      sh[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < k) {
      off[i] = carry + sh[threadIdx.x] - v;
      cursor[i] = 0;
    }
    carry += sh[255];
    __syncthreads();
  }
  if (threadIdx.x == 0) off[k] = carry;
}
int k_member_lists_dev(isle_ctx* c, const uint32_t* assign, uint64_t D, int k, const int* counts_dev) {
  HIPCHK(c, c->members.reserve(D ? D : 1));
  HIPCHK(c, c->moff.reserve(2 * (size_t)k + 2));
  int* offd = c->moff.p;
  int* cur = c->moff.p + k + 1;
  hipLaunchKernelGGL(member_offsets_k, dim3(1), dim3(256), 0, c->stream, counts_dev, k, offd, cur);
  if (D)
    hipLaunchKernelGGL(member_fill_k, dim3(cdiv(D, 256 * CS_ITEMS)), dim3(256), 2 * (size_t)k * sizeof(int), c->stream, assign, (uint32_t)D, k,
                       offd, cur, c->members.p);
  HIPCHK(c, hipGetLastError());
  c->members_valid = true;
  return 0;
}

// Yinyang bookkeeping on the device: delta[i] <- rounded-up movement of centre i, gmax[g] <- largest movement in group g
// (id_of_slot: the group's members are the centres its slots hold, see YyMap)
__global__ void yy_delta_k(float* __restrict__ delta, int k, int G, int group, float* __restrict__ gmax, const uint32_t* __restrict__ id_of_slot) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= G) return;
  float m = 0.f;
  for (int s = group * g; s < min(k, group * g + group); ++s) {
    const int i = id_of_slot ? (int)id_of_slot[s] : s;
    const float dv = sqrtf(fmaxf(delta[i], 0.f)) * (1.0f + 1e-5f) + 1e-7f;  // rounded up
    delta[i] = dv;
    m = fmaxf(m, dv);
  }
  gmax[g] = m;
}
int k_yy_delta(isle_ctx* c, float* delta_dev, int k, int G, int group, float* gmax_dev, const uint32_t* id_of_slot) {
  hipLaunchKernelGGL(yy_delta_k, dim3(cdiv(G, 64)), dim3(64), 0, c->stream, delta_dev, k, G, group, gmax_dev, id_of_slot);
  HIPCHK(c, hipGetLastError());
  return 0;
}
__global__ void max_f32_k(const float* __restrict__ v, int n, float* __restrict__ out) {
  float m = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) m = fmaxf(m, v[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if (threadIdx.x == 0) *out = m;
}
int k_max_f32(isle_ctx* c, const float* v, int n, float* out_dev) {
  hipLaunchKernelGGL(max_f32_k, dim3(1), dim3(64), 0, c->stream, v, n, out_dev);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// members = local documents grouped by centre (counts_dev = LOCAL cluster sizes from k_count_sizes).
// members[off[c] .. off[c+1]) = documents assigned to centre c in ASCENDING document order: a stable radix sort of the documents by
// centre (k_sort_pairs_u64, ingest.hip), so that everything summed over a list has one order, run after run.
int k_member_lists(isle_ctx* c, const uint32_t* assign, uint64_t D, int k, const int* counts_dev, int* max_out, std::vector<int>* counts_host) {
  std::vector<int> h(k);
  HIPCHK(c, hipMemcpyAsync(h.data(), counts_dev, (size_t)k * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int mx = 0;
  for (int i = 0; i < k; ++i) mx = std::max(mx, h[i]);
  if (max_out) *max_out = mx;
  if (counts_host) *counts_host = h;
  HIPCHK(c, c->members.reserve(D ? D : 1));
  HIPCHK(c, c->moff.reserve(2 * (size_t)k + 2));
  int* offd = c->moff.p;
  int* cur = c->moff.p + k + 1;
  hipLaunchKernelGGL(member_offsets_k, dim3(1), dim3(256), 0, c->stream, counts_dev, k, offd, cur);
  HIPCHK(c, hipGetLastError());
  if (D) {
    HIPCHK(c, c->gl_key_a.reserve(D));
    HIPCHK(c, c->gl_key_b.reserve(D));
    HIPCHK(c, c->gl_val_a.reserve(D));
    HIPCHK(c, c->gl_val_b.reserve(D));
    int bits = 1;
    while ((1 << bits) < k) ++bits;
    // the sort ping-pongs between two payload buffers, one pass per 8 key bits: `members` is placed so that the last pass lands in it
    const bool odd = D >= 2 && (((bits + 7) / 8) & 1) != 0;  // (a single document is not sorted at all: it stays in the first buffer)
    uint32_t* va = odd ? c->gl_val_a.p : c->members.p;
    uint32_t* vb = odd ? c->members.p : c->gl_val_a.p;
    hipLaunchKernelGGL(member_keys_k, dim3(cdiv(D, 256)), dim3(256), 0, c->stream, assign, (uint32_t)D, c->gl_key_a.p, va);
    HIPCHK(c, hipGetLastError());
    bool in_a = true;
    ISLECHK(k_sort_pairs_u64(c, c->gl_key_a.p, va, c->gl_key_b.p, vb, D, bits, &in_a));
    if ((in_a ? va : vb) != c->members.p) return isle_fail(c, ISLE_E_NUMERIC, "member lists: unexpected number of sort passes");
  }
  c->members_valid = true;
  return 0;
}

int k_proj_accumulate(isle_ctx* c, const float* P, uint64_t D, int k, int ldk, const uint32_t* assign, float* Csum, int* counts) {
  TimeScope ts(c, ISLE_T_LLOYD_PROJ);
  const size_t n = (size_t)k * ldk;
  if (D == 0) {
    HIPCHK(c, hipMemsetAsync(Csum, 0, n * sizeof(float), c->stream));
    return 0;
  }
  int mx = 0;
  std::vector<int> h;
  ISLECHK(k_member_lists(c, assign, D, k, counts, &mx, &h));
  // chunk descriptors: centre c owns chunks [c0[c], c0[c + 1]) of <= 256 members each
  std::vector<SegChunk> ch;
  std::vector<int> c0(k + 1, 0);
  ch.reserve(D / SEG_MC + k);
  int at = 0;
  for (int cc = 0; cc < k; ++cc) {
    for (int b0 = 0; b0 < h[cc]; b0 += SEG_MC) ch.push_back(SegChunk{at + b0, at + std::min(h[cc], b0 + SEG_MC)});
    at += h[cc];
    c0[cc + 1] = (int)ch.size();
  }
  const size_t nch = ch.size();
  static_assert(sizeof(SegChunk) == 2 * sizeof(int), "");
  HIPCHK(c, c->seg_desc.reserve(2 * nch + (size_t)k + 1));
  HIPCHK(c, c->seg_part.reserve((nch ? nch : 1) * (size_t)ldk));
  int* c0_dev = c->seg_desc.p + 2 * nch;
  // descriptors travel through the page-locked mailbox area (free outside the eigensolver): the copies are then really asynchronous
  const size_t desc_bytes = nch * sizeof(SegChunk) + ((size_t)k + 1) * sizeof(int);
  const bool pinned = desc_bytes <= 2 * isle_ctx::PIN_MAIL_SLOT;
  char* stage = c->pin + isle_ctx::PIN_MAIL;
  if (pinned) {
    if (nch) memcpy(stage, ch.data(), nch * sizeof(SegChunk));
    memcpy(stage + nch * sizeof(SegChunk), c0.data(), ((size_t)k + 1) * sizeof(int));
    HIPCHK(c, hipMemcpyAsync(c->seg_desc.p, stage, desc_bytes, hipMemcpyHostToDevice, c->stream));
  } else {
    if (nch) HIPCHK(c, hipMemcpyAsync(c->seg_desc.p, ch.data(), nch * sizeof(SegChunk), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c0_dev, c0.data(), ((size_t)k + 1) * sizeof(int), hipMemcpyHostToDevice, c->stream));
  }
  const size_t lds = 4 * (size_t)ldk * sizeof(float);
  const int nit = cdiv(ldk / 4, 64);  // float4 chunks per lane
  if (nch) {
    dim3 g((unsigned)nch), b(256);
#define LS(N) hipLaunchKernelGGL(proj_segsum_k<N>, g, b, lds, c->stream, P, ldk, reinterpret_cast<const SegChunk*>(c->seg_desc.p), c->members.p, c->seg_part.p)
    if (nit <= 1) LS(1);
    else if (nit <= 2) LS(2);
    else if (nit <= 4) LS(4);
    else if (nit <= 8) LS(8);
    else return isle_fail(c, ISLE_E_ARG, "k too large");
#undef LS
    HIPCHK(c, hipGetLastError());
  }
  hipLaunchKernelGGL(proj_segsum_reduce_k, dim3(cdiv(ldk, 256), k), dim3(256), 0, c->stream, c->seg_part.p, c0_dev, ldk, Csum);
  HIPCHK(c, hipGetLastError());
  if (!pinned) HIPCHK(c, hipStreamSynchronize(c->stream));  // ch / c0 are pageable host memory
  return 0;
}

// ------------------------------------------------------------------------------------------
// The same sums kept up to date by the documents that changed centre (iterations after the first: 1 - 5 % of the documents at config 3,
// where the fresh sums read all 40 GB of P: 6.7 ms an iteration).  `counted[d]` is the centre under which document d sits in Csum.
// A changed document contributes two entries, +row under its new centre and -row under its old one; the entries are sorted by
// (centre, document, sign) — so their order does not depend on the order in which workgroups appended them — every centre's run is cut
// in PD_PARTS equal parts, one workgroup per part sums its rows in that order (proj_delta_sum_k, the wave pattern of proj_segsum_k), and
// proj_delta_apply_k adds the parts to the centre's row in a fixed order.  No atomics on the sums: the centres stay bitwise reproducible.
// They are no longer the bits of the fresh sums (the additions associate differently); a centre that loses all members is reset to zero.
// ------------------------------------------------------------------------------------------
constexpr int PD_PARTS = 8;
__global__ __launch_bounds__(256) void proj_changed_k(const uint32_t* __restrict__ assign, uint32_t* __restrict__ counted, uint32_t D, int dbits,
                                                       uint64_t* __restrict__ key, uint32_t* __restrict__ val, uint32_t cap /*entries*/,
                                                       uint32_t* __restrict__ counter) {
  const uint32_t d = blockIdx.x * 256 + threadIdx.x;
  const uint32_t a = d < D ? assign[d] : 0u, o = d < D ? counted[d] : 0u;
  const bool ch = d < D && a != o;
  const uint32_t slot = block_append_slot(ch, counter);
  if (ch) {
    counted[d] = a;
    if (2 * (uint64_t)slot + 1 < cap) {  // beyond the capacity the caller recomputes the sums from scratch (the counter tells)
      key[2 * (size_t)slot] = ((uint64_t)a << (dbits + 1)) | ((uint64_t)d << 1);
      key[2 * (size_t)slot + 1] = ((uint64_t)o << (dbits + 1)) | ((uint64_t)d << 1) | 1ull;
      val[2 * (size_t)slot] = d;
      val[2 * (size_t)slot + 1] = d;
    }
  }
}
template <int NIT>
__global__ __launch_bounds__(256) void proj_delta_sum_k(const float* __restrict__ P, int ldk, const uint64_t* __restrict__ key, uint32_t n, int dbits,
                                                         float* __restrict__ part /*[centre][PD_PARTS][ldk]*/) {
  extern __shared__ float red[];  // 4 x ldk
  __shared__ uint32_t range[2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nq = ldk / 4;
  const uint32_t cc = blockIdx.x, pt = blockIdx.y;
  if (threadIdx.x < 2) {  // first entry of centre cc (thread 0) and of centre cc + 1 (thread 1)
    const uint64_t target = (uint64_t)(cc + threadIdx.x) << (dbits + 1);
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
      const uint32_t mid = lo + ((hi - lo) >> 1);
      if (key[mid] < target) lo = mid + 1;
      else hi = mid;
    }
    range[threadIdx.x] = lo;
  }
  __syncthreads();
  const uint32_t r0 = range[0], len = range[1] - range[0];
  const uint32_t beg = r0 + (uint32_t)(((uint64_t)len * pt) / PD_PARTS), end = r0 + (uint32_t)(((uint64_t)len * (pt + 1)) / PD_PARTS);
  float4 acc[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) acc[it] = make_float4(0.f, 0.f, 0.f, 0.f);
  const uint64_t dmask = (1ull << dbits) - 1ull;
  for (uint32_t b0 = beg; b0 < end; b0 += 256) {  // 256 entries at a time: this wave takes b0 + wave, + 4, ...
    const uint32_t e0 = min(end, b0 + 256);
    const uint32_t myi = b0 + wave + 4 * lane;
    const uint64_t myk = myi < e0 ? key[myi] : 0ull;
    const int cnt = e0 > b0 + wave ? (int)min(64u, (e0 - b0 - wave + 3) / 4) : 0;
    for (int j = 0; j < cnt; j += 4) {
      float4 v[4][NIT];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int src = min(j + u, cnt - 1);
        const uint32_t klo = (uint32_t)__shfl((int)(uint32_t)myk, src), khi = (uint32_t)__shfl((int)(uint32_t)(myk >> 32), src);
        const uint64_t kk = ((uint64_t)khi << 32) | klo;
        const float4* row = reinterpret_cast<const float4*>(P + (size_t)((kk >> 1) & dmask) * ldk);
        const float sgn = j + u < cnt ? ((kk & 1ull) ? -1.f : 1.f) : 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int q = lane + 64 * it;
          const float4 x = q < nq ? row[q] : make_float4(0.f, 0.f, 0.f, 0.f);
          v[u][it] = make_float4(x.x * sgn, x.y * sgn, x.z * sgn, x.w * sgn);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          acc[it].x += v[u][it].x;
          acc[it].y += v[u][it].y;
          acc[it].z += v[u][it].z;
          acc[it].w += v[u][it].w;
        }
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int q = lane + 64 * it;
    if (q < nq) reinterpret_cast<float4*>(red + (size_t)wave * ldk)[q] = acc[it];
  }
  __syncthreads();
  float* out = part + ((size_t)cc * PD_PARTS + pt) * ldk;
  for (int j = threadIdx.x; j < ldk; j += 256) out[j] = (red[j] + red[ldk + j]) + (red[2 * ldk + j] + red[3 * ldk + j]);
}
__global__ __launch_bounds__(256) void proj_delta_apply_k(const float* __restrict__ part, const int* __restrict__ counts, int ldk, float* __restrict__ Csum) {
  const int cc = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= ldk) return;
  const float* p = part + (size_t)cc * PD_PARTS * ldk + j;
  float v[PD_PARTS];
#pragma unroll
  for (int u = 0; u < PD_PARTS; ++u) v[u] = p[(size_t)u * ldk];
  const float dsum = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  static_assert(PD_PARTS == 8, "the sum above is written out for eight parts");
  Csum[(size_t)cc * ldk + j] = counts[cc] > 0 ? Csum[(size_t)cc * ldk + j] + dsum : 0.f;
}

// Csum (this rank's sums, `counted` = the assignment they hold) brought up to `assign`.  *done = false: too many documents changed (or the
// lists do not fit): the caller computes the sums afresh.
int k_proj_accumulate_delta(isle_ctx* c, const float* P, uint64_t D, int k, int ldk, const uint32_t* assign, uint32_t* counted, float* Csum,
                            const int* counts, bool* done) {
  TimeScope ts(c, ISLE_T_LLOYD_PROJ);
  *done = false;
  if (D == 0 || D >= (1ull << 31)) return 0;
  int dbits = 1, cbits = 1;
  while ((1ull << dbits) < D) ++dbits;
  while ((1 << cbits) < k) ++cbits;
  const uint32_t cap = (uint32_t)std::min<uint64_t>(2 * (D / 8 + 1), 0x7fffffffull);  // entries: at most an eighth of the documents changing
  HIPCHK(c, c->gl_key_a.reserve(cap));
  HIPCHK(c, c->gl_key_b.reserve(cap));
  HIPCHK(c, c->gl_val_a.reserve(cap));
  HIPCHK(c, c->gl_val_b.reserve(cap));
  HIPCHK(c, c->proj_dpart.reserve((size_t)k * PD_PARTS * ldk));
  HIPCHK(c, c->proj_nch.reserve(4));
  HIPCHK(c, hipMemsetAsync(c->proj_nch.p, 0, sizeof(uint32_t), c->stream));
  hipLaunchKernelGGL(proj_changed_k, dim3(cdiv(D, 256)), dim3(256), 0, c->stream, assign, counted, (uint32_t)D, dbits, c->gl_key_a.p, c->gl_val_a.p, cap,
                     c->proj_nch.p);
  HIPCHK(c, hipGetLastError());
  uint32_t* n_pin = reinterpret_cast<uint32_t*>(c->pin + isle_ctx::PIN_SMALL + (192u << 10) + 128);  // page-locked
  HIPCHK(c, hipMemcpyAsync(n_pin, c->proj_nch.p, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const uint64_t n = 2 * (uint64_t)*n_pin;
  if (n > cap) return 0;  // `counted` already holds the new assignment: the fresh sums the caller computes agree with it
  *done = true;
  if (n == 0) return 0;
  bool in_a = true;
  ISLECHK(k_sort_pairs_u64(c, c->gl_key_a.p, c->gl_val_a.p, c->gl_key_b.p, c->gl_val_b.p, n, dbits + 1 + cbits, &in_a));
  const uint64_t* keys = in_a ? c->gl_key_a.p : c->gl_key_b.p;
  const size_t lds = 4 * (size_t)ldk * sizeof(float);
  const int nit = cdiv(ldk / 4, 64);
  dim3 g((unsigned)k, PD_PARTS), b(256);
#define LD(N) hipLaunchKernelGGL(proj_delta_sum_k<N>, g, b, lds, c->stream, P, ldk, keys, (uint32_t)n, dbits, c->proj_dpart.p)
  if (nit <= 1) LD(1);
  else if (nit <= 2) LD(2);
  else if (nit <= 4) LD(4);
  else if (nit <= 8) LD(8);
  else return isle_fail(c, ISLE_E_ARG, "k too large");
#undef LD
  HIPCHK(c, hipGetLastError());
  hipLaunchKernelGGL(proj_delta_apply_k, dim3(cdiv(ldk, 256), k), dim3(256), 0, c->stream, c->proj_dpart.p, counts, ldk, Csum);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// centre = sum * (1/count) if count > 0 else 0   (src/sparseMatrix.cpp:1988-1992, FPscal with 1/div)
__global__ void proj_finalize_k(const float* __restrict__ Csum, const int* __restrict__ counts, int k, int ldk, float* __restrict__ C) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= k * ldk) return;
  const int cc = idx / ldk;
  const int n = counts[cc];
  C[idx] = (n > 0) ? Csum[idx] * (1.0f / (float)n) : 0.f;
}
int k_proj_finalize(isle_ctx* c, const float* Csum, const int* counts, int k, int ldk, float* C) {
  TimeScope ts(c, ISLE_T_LLOYD_PROJ);
  hipLaunchKernelGGL(proj_finalize_k, dim3(cdiv((long)k * ldk, 256)), dim3(256), 0, c->stream, Csum, counts, k, ldk, C);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// cluster sizes: per-workgroup histogram in LDS, one global atomic per (workgroup, non-empty bin)
__global__ __launch_bounds__(256) void count_sizes_k(const uint32_t* __restrict__ assign, uint64_t D, int k, int* __restrict__ counts) {
  extern __shared__ int hist[];
  for (int j = threadIdx.x; j < k; j += 256) hist[j] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * (256 * CS_ITEMS);
#pragma unroll
  for (int u = 0; u < CS_ITEMS; ++u) {
    const uint64_t i = base + threadIdx.x + 256 * u;
    if (i < D) atomicAdd(&hist[assign[i]], 1);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < k; j += 256)
    if (hist[j]) atomicAdd(&counts[j], hist[j]);
}
int k_count_sizes(isle_ctx* c, const uint32_t* assign, uint64_t D, int k, int* counts) {
  HIPCHK(c, hipMemsetAsync(counts, 0, (size_t)k * sizeof(int), c->stream));
  if (D == 0) return 0;
  hipLaunchKernelGGL(count_sizes_k, dim3(cdiv(D, 256 * CS_ITEMS)), dim3(256), (size_t)k * sizeof(int), c->stream, assign, D, k, counts);
  HIPCHK(c, hipGetLastError());
  return 0;
}

__global__ void compare_u32_k(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint64_t n, int* __restrict__ flag) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && a[i] != b[i]) *flag = 1;
}
int k_compare_u32(isle_ctx* c, const uint32_t* a, const uint32_t* b, uint64_t n, int* flag_dev) {
  HIPCHK(c, hipMemsetAsync(flag_dev, 0, sizeof(int), c->stream));
  if (n == 0) return 0;
  hipLaunchKernelGGL(compare_u32_k, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, a, b, n, flag_dev);
  HIPCHK(c, hipGetLastError());
  return 0;
}

__global__ void fill_f32_k(float* __restrict__ p, uint64_t n, float v) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
int k_fill_f32(isle_ctx* c, float* p, uint64_t n, float v) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(fill_f32_k, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, p, n, v);
  HIPCHK(c, hipGetLastError());
  return 0;
}
