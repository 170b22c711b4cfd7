// isle_amd/csrc/infer.hip — ISLEInfer on the device: topic weights of documents under a given model by
// multiplicative-weights updates.
//
//   ISLEInfer::infer_doc_in_file   src/infer.cpp:361-391   words whose model row sums to <= 1e-10 are left out
//   ISLEInfer::mwu                 src/infer.cpp:394-441   w <- w * exp(eta * grad), renormalised, eta = sqrt(2 ln k / (it + 1)) / Lf;
//                                                          Lf doubles (up to ten guesses) while the weights are not finite
//   ISLEInfer::grad                src/infer.cpp:443-465   z = M w, z <- a / z, grad = M^T z      (two FPgemv)
//   ISLEInfer::calculate_llh       src/infer.cpp:467-492   sum_d a_d log (M w)_d, scaled by avg_doc_sz / by the word count
//   drivers/ISLEInfer.cpp:92-112                           top topics: weight > 1 / k, heaviest five
//
// One workgroup (4 waves) per document.  The document's slice of the model (its words' rows, k floats each) is re-read from
// L2 / Infinity Cache in every iteration (15 by default); optionally (ISLE_INFER_CAP_ROWS) slices of up to that many rows are
// staged in LDS once — slower in practice, see k_infer.  Lanes own topics (one float4 per 64 topics), waves split the rows: per row one
// wave-wide dot product (z_d), then its contribution a_d / z_d * row to the gradient; the four partial gradients are added in
// fixed order.  Every wave keeps its own, identical copy of w.  Arithmetic as the reference: fp32 products and sums, eta and
// the exponential in double.  (Sums inside a row and over topics are tree reductions, the reference's are sequential or
// MKL's: results agree to fp32 rounding, not bit for bit.)
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "common.h"

namespace {

constexpr int INF_T = 256;
constexpr uint32_t INF_LDS = 163840 - 256;  // dynamic part; the kernel also holds a few statically allocated words

__global__ __launch_bounds__(256) void inf_rowok_k(const float* __restrict__ M, uint64_t V, int k, int ld, unsigned char* __restrict__ ok) {
  const uint64_t w = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (w >= V) return;
  double s = 0.0;
  for (int t = 0; t < k; ++t) s += (double)M[w * ld + t];  // std::accumulate(..., 0.0) :376
  ok[w] = s > 1.0e-10 ? 1 : 0;
}

// wave per document: a = count / doc_sum (normalize_docs(true, true)), the words kept by :376 compacted in place of the
// document's CSC range: fw / fa[offs[d] .. offs[d] + nkeep[d])
__global__ __launch_bounds__(256) void inf_prep_k(const float* __restrict__ counts, const uint32_t* __restrict__ rows, const int64_t* __restrict__ offs,
                                                   uint64_t D, const unsigned char* __restrict__ ok, uint32_t* __restrict__ fw,
                                                   float* __restrict__ fa, uint32_t* __restrict__ nkeep) {
  const int lane = threadIdx.x & 63;
  const uint64_t d = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (d >= D) return;
  const int64_t beg = offs[d], end = offs[d + 1];
  float sum = 0.f;  // counts are small integers: exact in any order
  for (int64_t i = beg + lane; i < end; i += 64) sum += counts[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
  uint32_t n = 0;
  for (int64_t i0 = beg; i0 < end; i0 += 64) {
    const int64_t i = i0 + lane;
    const bool in = i < end;
    const uint32_t w = in ? rows[i] : 0u;
    const bool keep = in && ok[w];
    const unsigned long long m = __ballot(keep);
    if (keep) {
      const uint32_t at = n + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      fw[beg + at] = w;
      fa[beg + at] = counts[i] / sum;
    }
    n += (uint32_t)__popcll(m);
  }
  if (lane == 0) nkeep[d] = n;
}

// Sum over the 16 lanes of a DPP row, result in every lane of the row: four data-parallel-primitive moves (quad swaps, half-row
// mirror, row mirror) instead of four ds_bpermute round trips (__shfl_xor goes through the LDS crossbar: ~100 clk each, and
// the row dot products are one dependent chain of them).
__device__ inline float dpp_sum16(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
  return v;
}
__device__ inline float wave_sum_f(float v) {  // all 64 lanes, result uniform
  v = dpp_sum16(v);
  const int b = __builtin_bit_cast(int, v);
  return ((__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
          __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32))) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
}
__device__ inline double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ inline float dot4(const float4 a, const float4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

// One pass over the rows of this wave (r = wave, wave + 4, ...): z_r = row_r . w (wave-wide), then either the gradient
// contribution a_r / z_r * row_r (LLH = false) or the log-likelihood term a_r log z_r (LLH = true).  IN_LDS picks the address
// space at compile time (a run-time choice of pointer makes the compiler emit flat loads).
template <int NIT, bool IN_LDS, bool LLH>
__device__ inline float inf_rows_pass(const float4* __restrict__ M, int nq, const float4* rowbuf, const float* as, const uint32_t* __restrict__ fwd,
                                      const float* __restrict__ fad, uint32_t n, int lane, int wave, const float4 (&w)[NIT], float4 (&g)[NIT]) {
  float llh = 0.f;
  auto load = [&](uint32_t r, float4 (&rv)[NIT]) {
    const float4* src = IN_LDS ? rowbuf + (size_t)r * nq : M + (size_t)fwd[r] * nq;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = lane + 64 * it;
      rv[it] = q < nq ? src[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto aof = [&](uint32_t r) { return IN_LDS ? as[r] : fad[r]; };
  uint32_t r = wave;
  for (; r + 4 < n; r += 8) {  // two rows per pass: independent chains overlap
    float4 ra[NIT], rb[NIT];
    load(r, ra);
    load(r + 4, rb);
    float pa = 0.f, pb = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      pa += dot4(ra[it], w[it]);
      pb += dot4(rb[it], w[it]);
    }
    pa = wave_sum_f(pa);
    pb = wave_sum_f(pb);
    if (LLH) {
      llh += aof(r) * logf(pa);
      llh += aof(r + 4) * logf(pb);
    } else {
      const float qa = aof(r) / pa, qb = aof(r + 4) / pb;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        g[it].x = fmaf(ra[it].x, qa, g[it].x);
        g[it].y = fmaf(ra[it].y, qa, g[it].y);
        g[it].z = fmaf(ra[it].z, qa, g[it].z);
        g[it].w = fmaf(ra[it].w, qa, g[it].w);
        g[it].x = fmaf(rb[it].x, qb, g[it].x);
        g[it].y = fmaf(rb[it].y, qb, g[it].y);
        g[it].z = fmaf(rb[it].z, qb, g[it].z);
        g[it].w = fmaf(rb[it].w, qb, g[it].w);
      }
    }
  }
  for (; r < n; r += 4) {
    float4 rv[NIT];
    load(r, rv);
    float p = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) p += dot4(rv[it], w[it]);
    const float z = wave_sum_f(p);
    if (LLH) {
      llh += aof(r) * logf(z);
    } else {
      const float rr = aof(r) / z;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        g[it].x = fmaf(rv[it].x, rr, g[it].x);
        g[it].y = fmaf(rv[it].y, rr, g[it].y);
        g[it].z = fmaf(rv[it].z, rr, g[it].z);
        g[it].w = fmaf(rv[it].w, rr, g[it].w);
      }
    }
  }
  return llh;
}

// llh, the weights (uniform where inference failed) and the heaviest five topics of one document; called by wave 0 with the
// weights in the lane layout q = lane + 64 it.
template <int NIT>
__device__ inline void inf_emit(const float4 (&w)[NIT], int lane, int wave, uint64_t d, int k, int nq, float unif, float first, float second,
                                float* __restrict__ weights, int32_t* __restrict__ top_topic, float* __restrict__ top_weight,
                                float* __restrict__ llh, unsigned int* __restrict__ nconverged) {
  const bool good = first != 0.0f;  // drivers/ISLEInfer.cpp:93
  if (wave == 0) {
    if (lane == 0) {
      llh[2 * d] = first;
      llh[2 * d + 1] = second;
      if (good) atomicAdd(nconverged, 1u);
    }
    if (weights) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int q = lane + 64 * it;
        const float v[4] = {w[it].x, w[it].y, w[it].z, w[it].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int t = 4 * q + j;
          if (q < nq && t < k) weights[d * (uint64_t)k + t] = good ? v[j] : unif;
        }
      }
    }
    // heaviest five topics with weight > 1 / k (:100-112)
    unsigned int taken[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) taken[it] = 0;
    for (int pick = 0; pick < 5; ++pick) {
      float bv = -1.f;
      int bt = 0x7fffffff;
      if (good) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int q = lane + 64 * it;
          const float v[4] = {w[it].x, w[it].y, w[it].z, w[it].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int t = 4 * q + j;
            if (q < nq && t < k && !(taken[it] & (1u << j)) && v[j] > unif && (v[j] > bv || (v[j] == bv && t < bt))) {
              bv = v[j];
              bt = t;
            }
          }
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bv, off);
        const int ot = __shfl_xor(bt, off);
        if (ov > bv || (ov == bv && ot < bt)) {
          bv = ov;
          bt = ot;
        }
      }
      const bool have = bv > 0.f;
      if (lane == 0) {
        top_topic[d * 5 + pick] = have ? bt : -1;
        top_weight[d * 5 + pick] = have ? bv : 0.f;
      }
      if (have) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int q = lane + 64 * it;
          if (bt >= 4 * q && bt < 4 * q + 4) taken[it] |= 1u << (bt - 4 * q);
        }
      }
    }
  }
}

template <int NIT>
__global__ __launch_bounds__(INF_T) void inf_docs_k(const float4* __restrict__ M, int k, int nq, const int64_t* __restrict__ offs,
                                                     const uint32_t* __restrict__ fw, const float* __restrict__ fa,
                                                     const uint32_t* __restrict__ nkeep, uint64_t D, int iters, float Lfguess, float avg_doc_sz,
                                                     uint32_t cap_rows, float* __restrict__ weights /*nullable D x k*/,
                                                     int32_t* __restrict__ top_topic, float* __restrict__ top_weight, float* __restrict__ llh,
                                                     unsigned int* __restrict__ nconverged) {
  extern __shared__ float4 sm[];          // [cap_rows x nq] model rows | [4 x nq] partial gradients | [nq] weights | [cap_rows] a
  __shared__ float sred[4];
  const uint64_t d = blockIdx.x;
  if (d >= D) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t n = nkeep[d];
  const int64_t beg = offs[d];
  const uint32_t words_in_doc = (uint32_t)(offs[d + 1] - beg);
  float4* rowbuf = sm;
  float4* gbuf = sm + (size_t)cap_rows * nq;
  float4* wbuf4 = gbuf + 4 * (size_t)nq;
  float* wflat = reinterpret_cast<float*>(wbuf4);
  float* as = reinterpret_cast<float*>(wbuf4 + nq);
  const bool in_lds = n <= cap_rows;
  const float unif = 1.0f / (float)k;
  // validity mask of this lane's topics
  float4 mask[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int q = lane + 64 * it, t0 = 4 * q;
    mask[it] = make_float4(q < nq && t0 < k ? 1.f : 0.f, q < nq && t0 + 1 < k ? 1.f : 0.f, q < nq && t0 + 2 < k ? 1.f : 0.f,
                           q < nq && t0 + 3 < k ? 1.f : 0.f);
  }
  if (in_lds) {
    // stage the slice: wave `wave` takes rows wave, wave + 4, ...  The word ids of 64 of its rows are fetched by one coalesced-ish
    // load and handed out by shuffles, and four row loads are in flight at a time (a plain loop is a chain of two dependent
    // global loads per row: ~75 us per 100-word document)
    for (uint32_t base = 0; base < n; base += 256) {
      const uint32_t myr = base + wave + 4 * lane;
      const uint32_t myw = myr < n ? fw[beg + myr] : 0u;
      const uint32_t left = n > base + wave ? (n - base - wave + 3) / 4 : 0u;
      const uint32_t cnt = min(64u, left);
      for (uint32_t j = 0; j < cnt; j += 4) {
        float4 v[4][NIT];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t wj = (uint32_t)__shfl((int)myw, (int)min(j + u, cnt - 1));
          const float4* src = M + (size_t)wj * nq;
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            const int q = lane + 64 * it;
            v[u][it] = q < nq ? src[q] : make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (j + u < cnt) {
            const uint32_t r = base + wave + 4 * (j + u);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
              const int q = lane + 64 * it;
              if (q < nq) rowbuf[(size_t)r * nq + q] = v[u][it];
            }
          }
        }
      }
    }
    for (uint32_t r = threadIdx.x; r < n; r += INF_T) as[r] = fa[beg + r];
  }
  __syncthreads();
  float4 w[NIT];
  bool converged = false;
  float Lf = Lfguess;
  float* gflat = reinterpret_cast<float*>(gbuf);  // 4 x ld floats
  const int ld = 4 * nq;
  if (n > 0) {
    for (int guess = 0; guess < 10; ++guess) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) w[it] = make_float4(unif * mask[it].x, unif * mask[it].y, unif * mask[it].z, unif * mask[it].w);
      for (int iter = 0; iter < iters; ++iter) {
        float4 g[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) g[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        // grad :443-465
        if (in_lds) (void)inf_rows_pass<NIT, true, false>(M, nq, rowbuf, as, fw + beg, fa + beg, n, lane, wave, w, g);
        else (void)inf_rows_pass<NIT, false, false>(M, nq, rowbuf, as, fw + beg, fa + beg, n, lane, wave, w, g);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int q = lane + 64 * it;
          if (q < nq) gbuf[(size_t)wave * nq + q] = g[it];
        }
        if (wave == 0) {  // the weights of this iteration, for the per-topic update below
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            const int q = lane + 64 * it;
            if (q < nq) wbuf4[q] = w[it];
          }
        }
        __syncthreads();
        // one thread per topic: w <- w * exp(eta * grad) (:418, double), then the normaliser (:420) over the workgroup
        const double eta = sqrt(2.0 * (double)logf((float)k) / (double)(float)(iter + 1)) / (double)Lf;  // :415
        float part = 0.f;
        for (int t = threadIdx.x; t < k; t += INF_T) {
          const float gt = ((gflat[t] + gflat[ld + t]) + gflat[2 * ld + t]) + gflat[3 * ld + t];
          const float wn = (float)((double)wflat[t] * exp(eta * (double)gt));
          wflat[t] = wn;
          part += wn;
        }
        part = wave_sum_f(part);
        if (lane == 0) sred[wave] = part;
        __syncthreads();
        const float normalizer = ((sred[0] + sred[1]) + sred[2]) + sred[3];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int q = lane + 64 * it;
          if (q < nq) {
            const float4 u = wbuf4[q];
            w[it] = make_float4(u.x / normalizer * mask[it].x, u.y / normalizer * mask[it].y, u.z / normalizer * mask[it].z,
                                u.w / normalizer * mask[it].w);
          }
        }
        __syncthreads();  // gbuf / wbuf / sred are rewritten in the next iteration
      }
      double ps = 0.0;
#pragma unroll
      for (int it = 0; it < NIT; ++it) ps += ((double)w[it].x + (double)w[it].y) + ((double)w[it].z + (double)w[it].w);
      const double sumw = wave_sum_d(ps);  // :425
      const bool normal = isfinite(sumw) && fabs(sumw) >= 2.2250738585072014e-308;  // std::isnormal
      if (normal) {
        converged = !(fabs(1.0 - sumw) > 0.01);
        break;  // :427-432: a finite sum far from 1 is retried with the same Lf in the reference — same outcome every time
      }
      Lf *= 2.0f;
    }
  }
  // calculate_llh :467-492
  float first = 0.f, second = 0.f;
  if (converged) {
    float4 gdummy[NIT];
    float s = in_lds ? inf_rows_pass<NIT, true, true>(M, nq, rowbuf, as, fw + beg, fa + beg, n, lane, wave, w, gdummy)
                     : inf_rows_pass<NIT, false, true>(M, nq, rowbuf, as, fw + beg, fa + beg, n, lane, wave, w, gdummy);
    if (lane == 0) sred[wave] = s;
    __syncthreads();
    s = ((sred[0] + sred[1]) + sred[2]) + sred[3];
    second = s * (float)words_in_doc;
    first = s * avg_doc_sz;
  }
  if (wave == 0) inf_emit<NIT>(w, lane, wave, d, k, nq, unif, first, second, weights, top_topic, top_weight, llh, nconverged);
}

// k <= 256: sixteen lanes per row.  A wave works on four rows at a time (sub-group sub = lane / 16 takes rows
// 4 wave + sub, + 16, ...), lane s of a sub-group holds the float4 chunks q = s + 16 f of w, of the row and of its partial
// gradient.  The row dot products reduce over 16 lanes (four xor steps) instead of 64, sixteen rows are in flight per
// workgroup instead of eight, and the sixteen partial gradients are added in fixed order by the per-topic update.
template <int NF>
__global__ __launch_bounds__(INF_T) void inf_docs16_k(const float4* __restrict__ M, int k, int nq, const int64_t* __restrict__ offs,
                                                       const uint32_t* __restrict__ fw, const float* __restrict__ fa,
                                                       const uint32_t* __restrict__ nkeep, uint64_t D, int iters, float Lfguess, float avg_doc_sz,
                                                       uint32_t cap_rows, float* __restrict__ weights, int32_t* __restrict__ top_topic,
                                                       float* __restrict__ top_weight, float* __restrict__ llh,
                                                       unsigned int* __restrict__ nconverged) {
  extern __shared__ float4 sm[];  // [cap_rows x nq] model rows | [16 x nq] partial gradients | [nq] weights | [cap_rows] a
  __shared__ float sred[16];
  const uint64_t d = blockIdx.x;
  if (d >= D) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane >> 4, sl = lane & 15, grp = wave * 4 + sub;
  const uint32_t n = nkeep[d];
  const int64_t beg = offs[d];
  const uint32_t words_in_doc = (uint32_t)(offs[d + 1] - beg);
  float4* rowbuf = sm;
  float4* gbuf = sm + (size_t)cap_rows * nq;
  float4* wbuf4 = gbuf + 16 * (size_t)nq;
  float* wflat = reinterpret_cast<float*>(wbuf4);
  float* gflat = reinterpret_cast<float*>(gbuf);
  float* as = reinterpret_cast<float*>(wbuf4 + nq);
  const int ld = 4 * nq;
  const bool in_lds = n <= cap_rows;
  const float unif = 1.0f / (float)k;
  const uint32_t* fwd = fw + beg;
  const float* fad = fa + beg;
  if (in_lds) {  // stage the slice (same scheme as inf_docs_k: ids by one load + shuffles, four row loads in flight)
    for (uint32_t base = 0; base < n; base += 256) {
      const uint32_t myr = base + wave + 4 * lane;
      const uint32_t myw = myr < n ? fwd[myr] : 0u;
      const uint32_t left = n > base + wave ? (n - base - wave + 3) / 4 : 0u;
      const uint32_t cnt = min(64u, left);
      for (uint32_t j = 0; j < cnt; j += 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t wj = (uint32_t)__shfl((int)myw, (int)min(j + u, cnt - 1));
          v[u] = lane < nq ? M[(size_t)wj * nq + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (j + u < cnt && lane < nq) rowbuf[(size_t)(base + wave + 4 * (j + u)) * nq + lane] = v[u];
      }
    }
    for (uint32_t r = threadIdx.x; r < n; r += INF_T) as[r] = fad[r];
  }
  auto sub_sum = [](float v) { return dpp_sum16(v); };
  // one pass over this sub-group's rows; LLH = false: gradient contributions into g, true: returns sum a_r log z_r
  auto rows_pass = [&](const float4 (&wl)[NF], float4 (&g)[NF], bool want_llh) {
    float acc = 0.f;
    for (uint32_t r = grp; r < n; r += 16) {
      float4 rv[NF];
      const float4* src = in_lds ? nullptr : M + (size_t)fwd[r] * nq;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int q = sl + 16 * f;
        rv[f] = q < nq ? (in_lds ? rowbuf[(size_t)r * nq + q] : src[q]) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      float p = 0.f;
#pragma unroll
      for (int f = 0; f < NF; ++f) p += dot4(rv[f], wl[f]);
      const float z = sub_sum(p);
      const float ar = in_lds ? as[r] : fad[r];
      if (want_llh) {
        acc += ar * logf(z);
      } else {
        const float rr = ar / z;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          g[f].x = fmaf(rv[f].x, rr, g[f].x);
          g[f].y = fmaf(rv[f].y, rr, g[f].y);
          g[f].z = fmaf(rv[f].z, rr, g[f].z);
          g[f].w = fmaf(rv[f].w, rr, g[f].w);
        }
      }
    }
    return acc;
  };

  float4 wl[NF];
  bool converged = false;
  float Lf = Lfguess;
  float normalizer = 1.f;
  if (n > 0) {
    for (int guess = 0; guess < 10; ++guess) {
      __syncthreads();
      for (int t = threadIdx.x; t < ld; t += INF_T) wflat[t] = t < k ? unif : 0.f;
      normalizer = 1.f;
      __syncthreads();
      for (int iter = 0; iter < iters; ++iter) {
        float4 g[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const int q = sl + 16 * f;
          g[f] = make_float4(0.f, 0.f, 0.f, 0.f);
          const float4 u = q < nq ? wbuf4[q] : make_float4(0.f, 0.f, 0.f, 0.f);
          wl[f] = make_float4(u.x / normalizer, u.y / normalizer, u.z / normalizer, u.w / normalizer);  // :421
        }
        __syncthreads();  // every wave has read the weights before they are updated
        (void)rows_pass(wl, g, false);  // grad :443-465
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const int q = sl + 16 * f;
          if (q < nq) gbuf[(size_t)grp * nq + q] = g[f];
        }
        __syncthreads();
        const double eta = sqrt(2.0 * (double)logf((float)k) / (double)(float)(iter + 1)) / (double)Lf;  // :415
        float part = 0.f;
        for (int t = threadIdx.x; t < k; t += INF_T) {
          float gt = gflat[t];
#pragma unroll
          for (int pp = 1; pp < 16; ++pp) gt += gflat[pp * ld + t];  // fixed order
          const float wn = (float)((double)(wflat[t] / normalizer) * exp(eta * (double)gt));  // :418
          wflat[t] = wn;
          part += wn;
        }
        part = wave_sum_f(part);
        if (lane == 0) sred[wave] = part;
        __syncthreads();
        normalizer = ((sred[0] + sred[1]) + sred[2]) + sred[3];  // :420
      }
      // sum of the final weights (:425), by every sub-group on its own copy
      double ps = 0.0;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int q = sl + 16 * f;
        const float4 u = q < nq ? wbuf4[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        wl[f] = make_float4(u.x / normalizer, u.y / normalizer, u.z / normalizer, u.w / normalizer);
        ps += ((double)wl[f].x + (double)wl[f].y) + ((double)wl[f].z + (double)wl[f].w);
      }
      ps += __shfl_xor(ps, 1);
      ps += __shfl_xor(ps, 2);
      ps += __shfl_xor(ps, 4);
      ps += __shfl_xor(ps, 8);
      const double sumw = ps;
      const bool normal = isfinite(sumw) && fabs(sumw) >= 2.2250738585072014e-308;  // std::isnormal
      if (normal) {
        converged = !(fabs(1.0 - sumw) > 0.01);
        break;
      }
      Lf *= 2.0f;
    }
  }
  float first = 0.f, second = 0.f;
  if (converged) {  // calculate_llh :467-492
    float4 gd[NF];
    float sacc = rows_pass(wl, gd, true);
    __syncthreads();
    if (sl == 0) sred[grp] = sacc;
    __syncthreads();
    float tot = sred[0];
#pragma unroll
    for (int pp = 1; pp < 16; ++pp) tot += sred[pp];
    second = tot * (float)words_in_doc;
    first = tot * avg_doc_sz;
  }
  if (wave == 0) {
    float4 w1[1];
    const float4 u = lane < nq ? wbuf4[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    w1[0] = make_float4(u.x / normalizer, u.y / normalizer, u.z / normalizer, u.w / normalizer);
    inf_emit<1>(w1, lane, wave, d, k, nq, unif, first, second, weights, top_topic, top_weight, llh, nconverged);
  }
}

}  // namespace

// Host pointers in and out; the model and the documents are uploaded for the call (inference is independent of the
// context's training matrices).  Returns the number of converged documents in *nconverged.
int k_infer(isle_ctx* c, uint64_t V, int k, const float* model_by_word, uint64_t D, uint64_t nnz, const float* counts, const uint32_t* rows,
            const int64_t* offs, int iters, float Lfguess, float avg_doc_sz, float* weights, int32_t* top_topic, float* top_weight, float* llh,
            uint64_t* nconverged) {
  if (k < 1 || k > 1024) return isle_fail(c, ISLE_E_ARG, "infer: num_topics = %d not in [1, 1024]", k);
  const int ld = (k + 3) & ~3, nq = ld / 4;
  const int nit = (nq + 63) / 64;
  DevBuf<float> dM, dcounts, dfa, dW, dtw, dllh;
  DevBuf<uint32_t> drows, dfw, dnk;
  DevBuf<int64_t> doffs;
  DevBuf<unsigned char> dok;
  DevBuf<int32_t> dtt;
  DevBuf<unsigned int> dnc;
  HIPCHK(c, dM.reserve((size_t)V * ld));
  HIPCHK(c, dok.reserve(V ? V : 1));
  HIPCHK(c, dcounts.reserve(nnz ? nnz : 1));
  HIPCHK(c, drows.reserve(nnz ? nnz : 1));
  HIPCHK(c, dfa.reserve(nnz ? nnz : 1));
  HIPCHK(c, dfw.reserve(nnz ? nnz : 1));
  HIPCHK(c, doffs.reserve(D + 1));
  HIPCHK(c, dnk.reserve(D ? D : 1));
  HIPCHK(c, dtt.reserve(D ? 5 * D : 1));
  HIPCHK(c, dtw.reserve(D ? 5 * D : 1));
  HIPCHK(c, dllh.reserve(D ? 2 * D : 1));
  HIPCHK(c, dnc.reserve(1));
  if (weights) HIPCHK(c, dW.reserve(D ? D * (size_t)k : 1));
  HIPCHK(c, hipMemsetAsync(dM.p, 0, (size_t)V * ld * sizeof(float), c->stream));
  HIPCHK(c, hipMemcpy2DAsync(dM.p, (size_t)ld * sizeof(float), model_by_word, (size_t)k * sizeof(float), (size_t)k * sizeof(float), V,
                             hipMemcpyHostToDevice, c->stream));
  if (nnz) {
    HIPCHK(c, hipMemcpyAsync(dcounts.p, counts, nnz * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(drows.p, rows, nnz * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  }
  HIPCHK(c, hipMemcpyAsync(doffs.p, offs, (D + 1) * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(dnc.p, 0, sizeof(unsigned int), c->stream));
  {
    TimeScope ts(c, ISLE_T_INFER);
    hipLaunchKernelGGL(inf_rowok_k, dim3(cdiv((long)V, 256)), dim3(256), 0, c->stream, dM.p, V, k, ld, dok.p);
    if (D) hipLaunchKernelGGL(inf_prep_k, dim3(cdiv((long)D, 4)), dim3(256), 0, c->stream, dcounts.p, drows.p, doffs.p, D, dok.p, dfw.p, dfa.p, dnk.p);
    HIPCHK(c, hipGetLastError());
    // LDS: cap_rows model rows + 4 partial gradients + the weights + cap_rows values of a
    const bool lanes16 = nq <= 64;  // k <= 256: sixteen lanes per row
    const size_t fixed = (lanes16 ? 17 : 5) * (size_t)ld * sizeof(float);
    uint32_t cap_rows = (uint32_t)((INF_LDS - fixed) / ((size_t)ld * sizeof(float) + sizeof(float)));
    // Measured at C2 size (50k x 200 model, ~108 kept words per document): staging a document's slice in LDS (one workgroup per
    // CU at 160 KB) runs 3.1 M docs/s, re-reading the rows from L2 / Infinity Cache in every iteration with ~10 workgroups per
    // CU 5.3 M docs/s (6.9 TB/s of row gathers) — occupancy beats locality, so no rows are staged unless
    // ISLE_INFER_CAP_ROWS asks for it.
    {
      const char* e = c->knob(KN_INFER_CAP_ROWS);
      cap_rows = std::min<uint32_t>(cap_rows, e ? (uint32_t)atoi(e) : 0u);
    }
    const size_t lds = (size_t)cap_rows * ld * sizeof(float) + fixed + (size_t)cap_rows * sizeof(float);
    if (D) {
#define INF(KERNEL)                                                                                                                     \
  do {                                                                                                                                  \
    ISLECHK(isle_max_lds(c, (const void*)KERNEL, (int)INF_LDS));                                                                         \
    hipLaunchKernelGGL(KERNEL, dim3((unsigned)D), dim3(INF_T), lds, c->stream, (const float4*)dM.p, k, nq, doffs.p, dfw.p, dfa.p, dnk.p, \
                       D, iters, Lfguess, avg_doc_sz, cap_rows, weights ? dW.p : nullptr, dtt.p, dtw.p, dllh.p, dnc.p);                 \
  } while (0)
      if (lanes16) {
        const int nf = (nq + 15) / 16;
        if (nf <= 1) INF((inf_docs16_k<1>));
        else if (nf <= 2) INF((inf_docs16_k<2>));
        else INF((inf_docs16_k<4>));
      } else if (nit <= 1) INF((inf_docs_k<1>));
      else if (nit <= 2) INF((inf_docs_k<2>));
      else INF((inf_docs_k<4>));
#undef INF
      HIPCHK(c, hipGetLastError());
    }
  }
  unsigned int nc = 0;
  HIPCHK(c, hipMemcpyAsync(&nc, dnc.p, sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
  if (D) {
    if (top_topic) HIPCHK(c, hipMemcpyAsync(top_topic, dtt.p, 5 * D * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (top_weight) HIPCHK(c, hipMemcpyAsync(top_weight, dtw.p, 5 * D * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (llh) HIPCHK(c, hipMemcpyAsync(llh, dllh.p, 2 * D * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (weights) HIPCHK(c, hipMemcpyAsync(weights, dW.p, D * (size_t)k * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (nconverged) *nconverged = nc;
  return 0;
}
