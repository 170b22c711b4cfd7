// isle_amd/csrc/scan.h — deterministic device exclusive scan (out[0]=0 ... out[n]=total),
// Tin -> Tacc accumulation (float->double for the k-means++ D^2 prefix sums,
// src/sparseMatrix.cpp:2170-2172; uint32->int64 for the band-major placement offsets).
#pragma once
#include "common.h"

namespace isle_scan {

constexpr int SCAN_T = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_T * SCAN_ITEMS;

template <class Tacc>
__device__ inline Tacc block_exclusive(Tacc v, Tacc* sh /*SCAN_T*/, Tacc* total) {
  // Hillis-Steele over SCAN_T values in LDS; returns exclusive prefix of v, *total = block sum.
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int off = 1; off < SCAN_T; off <<= 1) {
    Tacc add = (t >= off) ? sh[t - off] : (Tacc)0;
    __syncthreads();
    sh[t] += add;
    __syncthreads();
  }
  const Tacc incl = sh[t];
  *total = sh[SCAN_T - 1];
  __syncthreads();
  return incl - v;
}

template <class Tin, class Tacc>
__global__ __launch_bounds__(SCAN_T) void scan_reduce_k(const Tin* __restrict__ in, uint64_t n, Tacc* __restrict__ blk) {
  __shared__ Tacc sh[SCAN_T];
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
  Tacc s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i)
    if (base + i < n) s += (Tacc)in[base + i];
  Tacc tot;
  (void)block_exclusive<Tacc>(s, sh, &tot);
  if (threadIdx.x == 0) blk[blockIdx.x] = tot;
}

// single block: exclusive scan of blk[0..nb) in place; blk[nb] = grand total
template <class Tacc>
__global__ __launch_bounds__(SCAN_T) void scan_blk_k(Tacc* __restrict__ blk, uint64_t nb) {
  __shared__ Tacc sh[SCAN_T];
  Tacc carry = 0;
  for (uint64_t base = 0; base < nb; base += SCAN_T) {
    const uint64_t i = base + threadIdx.x;
    const Tacc v = (i < nb) ? blk[i] : (Tacc)0;
    Tacc tot;
    const Tacc ex = block_exclusive<Tacc>(v, sh, &tot);
    if (i < nb) blk[i] = carry + ex;
    carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) blk[nb] = carry;
}

template <class Tin, class Tacc>
__global__ __launch_bounds__(SCAN_T) void scan_final_k(const Tin* __restrict__ in, uint64_t n, const Tacc* __restrict__ blk,
                                                        Tacc* __restrict__ out, uint64_t nb) {
  __shared__ Tacc sh[SCAN_T];
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
  Tacc loc[SCAN_ITEMS];
  Tacc s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    loc[i] = (base + i < n) ? (Tacc)in[base + i] : (Tacc)0;
    s += loc[i];
  }
  Tacc tot;
  Tacc run = blk[blockIdx.x] + block_exclusive<Tacc>(s, sh, &tot);
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    if (base + i < n) out[base + i] = run;
    run += loc[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = blk[nb];
}

// out must hold n+1 elements; blk scratch must hold nb+1 elements, nb = ceil(n / SCAN_TILE)
template <class Tin, class Tacc>
inline hipError_t exclusive_scan(hipStream_t st, const Tin* in, uint64_t n, Tacc* out, Tacc* blk) {
  const uint64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (nb == 0) return hipMemsetAsync(out, 0, sizeof(Tacc), st);
  hipLaunchKernelGGL((scan_reduce_k<Tin, Tacc>), dim3((unsigned)nb), dim3(SCAN_T), 0, st, in, n, blk);
  hipLaunchKernelGGL((scan_blk_k<Tacc>), dim3(1), dim3(SCAN_T), 0, st, blk, nb);
  hipLaunchKernelGGL((scan_final_k<Tin, Tacc>), dim3((unsigned)nb), dim3(SCAN_T), 0, st, in, n, blk, out, nb);
  return hipGetLastError();
}
inline uint64_t scan_scratch_elems(uint64_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE + 1; }

}  // namespace isle_scan
