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

#include "common.h"
#include "scan.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

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
int k_rownorms(isle_ctx* c, const float* M, int rows, int k, int ldk, float* out) {
  if (rows == 0) return 0;
  hipLaunchKernelGGL(rownorms_k, dim3(cdiv(rows, 4)), dim3(256), 0, c->stream, M, rows, k, ldk, out);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int k_kmpp_update(isle_ctx* c, const float* P, const float* pn, uint64_t D, int k, int ldk, const float* newC, int nc,
                  float* min_dist) {
  TimeScope ts(c, ISLE_T_KMPP);
  if (D == 0 || nc == 0) return 0;
  HIPCHK(c, c->cnorm.reserve((size_t)std::max(k, nc)));
  ISLECHK(k_rownorms(c, newC, nc, k, ldk, c->cnorm.p));
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
        Ps[row][col] = (dd < D && k0 + col < ldk) ? P[(size_t)dd * ldk + k0 + col] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < (PA_CT * PA_BK) / 256; ++u) {
        const int idx = threadIdx.x + 256 * u;
        const int row = idx / PA_BK, col = idx - row * PA_BK;
        Cs[row][col] = (c0 + row < k && k0 + col < ldk) ? C[(size_t)(c0 + row) * ldk + k0 + col] : 0.f;
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
int k_proj_assign(isle_ctx* c, const float* P, const float* pn, uint64_t D, int k, int ldk, const float* C, const float* cn,
                  uint32_t* assign) {
  TimeScope ts(c, ISLE_T_LLOYD_PROJ);
  if (D == 0) return 0;
  hipLaunchKernelGGL(proj_assign_k, dim3(cdiv(D, PA_DOCS)), dim3(256), 0, c->stream, P, pn, (uint32_t)D, k, ldk, C, cn, assign);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// Csum[assign[d]][:] += P[d][:]   (float atomics, contiguous dwords per wave-instruction); counts[c]++
__global__ __launch_bounds__(256) void proj_accumulate_k(const float* __restrict__ P, uint32_t D, int k, int ldk,
                                                          const uint32_t* __restrict__ assign, float* __restrict__ Csum,
                                                          int* __restrict__ counts) {
  const int lane = threadIdx.x & 63;
  const uint32_t d = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (d >= D) return;
  const uint32_t cc = assign[d];
  for (int j = lane; j < k; j += 64) atomicAdd(&Csum[(size_t)cc * ldk + j], P[(size_t)d * ldk + j]);
  if (lane == 0) atomicAdd(&counts[cc], 1);
}
int k_proj_accumulate(isle_ctx* c, const float* P, uint64_t D, int k, int ldk, const uint32_t* assign, float* Csum, int* counts) {
  TimeScope ts(c, ISLE_T_LLOYD_PROJ);
  HIPCHK(c, hipMemsetAsync(Csum, 0, (size_t)k * ldk * sizeof(float), c->stream));
  HIPCHK(c, hipMemsetAsync(counts, 0, (size_t)k * sizeof(int), c->stream));
  if (D == 0) return 0;
  hipLaunchKernelGGL(proj_accumulate_k, dim3(cdiv(D, 4)), dim3(256), 0, c->stream, P, (uint32_t)D, k, ldk, assign, Csum, counts);
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

__global__ __launch_bounds__(256) void count_sizes_k(const uint32_t* __restrict__ assign, uint64_t D, int* __restrict__ counts) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < D) atomicAdd(&counts[assign[i]], 1);
}
int k_count_sizes(isle_ctx* c, const uint32_t* assign, uint64_t D, int k, int* counts) {
  HIPCHK(c, hipMemsetAsync(counts, 0, (size_t)k * sizeof(int), c->stream));
  if (D == 0) return 0;
  hipLaunchKernelGGL(count_sizes_k, dim3(cdiv(D, 256)), dim3(256), 0, c->stream, assign, D, counts);
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
