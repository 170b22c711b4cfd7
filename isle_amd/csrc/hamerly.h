// isle_amd/csrc/hamerly.h — conservative Hamerly bounds from computed squared distances.
//
// A squared distance is evaluated as (|b|^2 + |c|^2) - 2 b.c in fp32; with n <= ~1600 products the absolute error is at
// most E = 1e-4 * (|b|^2 + |c|^2) (linear worst case n * eps * (|b|^2 + |c|^2); the typical error is 1000x smaller).
// For the distance itself |sqrt(x + e) - sqrt(x)| <= min(sqrt(E), E / sqrt(x)): tiny for documents far from a centre,
// sqrt(E) only when the distance itself is at the cancellation level.  The upper bound is widened and the lower bound
// narrowed by that amount when they are stored, so the filter needs no further slack.
#pragma once
#include <hip/hip_runtime.h>

#ifndef ISLE_SLACK_REL
#define ISLE_SLACK_REL 1e-4f  // E = ISLE_SLACK_REL * (|b|^2 + |c|^2): the absolute error allowed for in a computed squared distance
#endif

__device__ inline void hamerly_store_bounds(float best_sq, float second_sq, float norm_sum /* |b|^2 + max |c|^2 */, float* ub, float* lb) {
  const float E = ISLE_SLACK_REL * norm_sum;
  const float sE = sqrtf(E);
  const float u = sqrtf(best_sq), l = sqrtf(second_sq);
  *ub = u + fminf(sE, E / fmaxf(u, 1e-30f));
  *lb = fmaxf(l - fminf(sE, E / fmaxf(l, 1e-30f)), 0.f);
}

__device__ inline float yy_slack_down_sq(float m, float E, float sE) {  // lower bound from a squared distance
  const float l = sqrtf(m);
  return fmaxf(l - fminf(sE, E / fmaxf(l, 1e-30f)), 0.f);
}

// slot of this thread's element in a list all workgroups append to (valid only where act): ONE atomic per workgroup.  Every thread of
// the workgroup must call it (it synchronises); order inside the list is arbitrary.
__device__ inline uint32_t block_append_slot(bool act, uint32_t* __restrict__ counter) {
  __shared__ uint32_t wcnt[16], wbase[17];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
  const unsigned long long m = __ballot(act);
  __syncthreads();  // the previous call's readers are done
  if (lane == 0) wcnt[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int w = 0; w < nw; ++w) {
      wbase[w] = t;
      t += wcnt[w];
    }
    wbase[16] = t ? atomicAdd(counter, t) : 0u;
  }
  __syncthreads();
  return wbase[16] + wbase[wave] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}
// the same, also telling the workgroup's first slot and how many slots it took (for a second phase over the workgroup's own elements)
__device__ inline uint32_t block_append_slot_range(bool act, uint32_t* __restrict__ counter, uint32_t* base, uint32_t* total) {
  __shared__ uint32_t wcnt2[16], wbase2[18];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
  const unsigned long long m = __ballot(act);
  __syncthreads();
  if (lane == 0) wcnt2[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (int w = 0; w < nw; ++w) {
      wbase2[w] = t;
      t += wcnt2[w];
    }
    wbase2[16] = t ? atomicAdd(counter, t) : 0u;
    wbase2[17] = t;
  }
  __syncthreads();
  *base = wbase2[16];
  *total = wbase2[17];
  return wbase2[16] + wbase2[wave] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}
