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
// One workgroup (4 waves) per document.  The document's slice of the model (its words' rows, k floats each) is staged in LDS
// once and re-read from there in every iteration (15 by default); documents whose slice does not fit (> ~195 rows at k = 200)
// read the rows from global memory instead.  Lanes own topics (one float4 per 64 topics), waves split the rows: per row one
// wave-wide dot product (z_d), then its contribution a_d / z_d * row to the gradient; the four partial gradients are added in
// fixed order.  Every wave keeps its own, identical copy of w.  Arithmetic as the reference: fp32 products and sums, eta and
// the exponential in double.  (Sums inside a row and over topics are tree reductions, the reference's are sequential or
// MKL's: results agree to fp32 rounding, not bit for bit.)
#include <algorithm>
#include <cmath>
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

__device__ inline float wave_sum_f(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ inline double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ inline float dot4(const float4 a, const float4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

template <int NIT>
__global__ __launch_bounds__(INF_T) void inf_docs_k(const float4* __restrict__ M, int k, int nq, const int64_t* __restrict__ offs,
                                                     const uint32_t* __restrict__ fw, const float* __restrict__ fa,
                                                     const uint32_t* __restrict__ nkeep, uint64_t D, int iters, float Lfguess, float avg_doc_sz,
                                                     uint32_t cap_rows, float* __restrict__ weights /*nullable D x k*/,
                                                     int32_t* __restrict__ top_topic, float* __restrict__ top_weight, float* __restrict__ llh,
                                                     unsigned int* __restrict__ nconverged) {
  extern __shared__ float4 sm[];          // [cap_rows x nq] model rows | [4 x nq] partial gradients | [cap_rows] a
  __shared__ float sred[4];
  const uint64_t d = blockIdx.x;
  if (d >= D) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t n = nkeep[d];
  const int64_t beg = offs[d];
  const uint32_t words_in_doc = (uint32_t)(offs[d + 1] - beg);
  float4* rowbuf = sm;
  float4* gbuf = sm + (size_t)cap_rows * nq;
  float* as = reinterpret_cast<float*>(gbuf + 4 * (size_t)nq);
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
    for (uint32_t r = wave; r < n; r += 4) {
      const float4* src = M + (size_t)fw[beg + r] * nq;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int q = lane + 64 * it;
        if (q < nq) rowbuf[(size_t)r * nq + q] = src[q];
      }
    }
    for (uint32_t r = threadIdx.x; r < n; r += INF_T) as[r] = fa[beg + r];
  }
  __syncthreads();
  auto load_row = [&](uint32_t r, float4 (&rv)[NIT]) {
    const float4* src = in_lds ? rowbuf + (size_t)r * nq : M + (size_t)fw[beg + r] * nq;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = lane + 64 * it;
      rv[it] = q < nq ? src[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto a_of = [&](uint32_t r) { return in_lds ? as[r] : fa[beg + r]; };

  float4 w[NIT];
  bool converged = false;
  float Lf = Lfguess;
  if (n > 0) {
    for (int guess = 0; guess < 10; ++guess) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) w[it] = make_float4(unif * mask[it].x, unif * mask[it].y, unif * mask[it].z, unif * mask[it].w);
      for (int iter = 0; iter < iters; ++iter) {
        float4 g[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) g[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (uint32_t r = wave; r < n; r += 4) {  // grad :443-465
          float4 rv[NIT];
          load_row(r, rv);
          float p = 0.f;
#pragma unroll
          for (int it = 0; it < NIT; ++it) p += dot4(rv[it], w[it]);
          const float z = wave_sum_f(p);
          const float rr = a_of(r) / z;
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            g[it].x = fmaf(rv[it].x, rr, g[it].x);
            g[it].y = fmaf(rv[it].y, rr, g[it].y);
            g[it].z = fmaf(rv[it].z, rr, g[it].z);
            g[it].w = fmaf(rv[it].w, rr, g[it].w);
          }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int q = lane + 64 * it;
          if (q < nq) gbuf[(size_t)wave * nq + q] = g[it];
        }
        __syncthreads();
        const double eta = sqrt(2.0 * (double)logf((float)k) / (double)(float)(iter + 1)) / (double)Lf;  // :415
        float part = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int q = lane + 64 * it;
          if (q < nq) {
            const float4 g0 = gbuf[q], g1 = gbuf[(size_t)nq + q], g2 = gbuf[2 * (size_t)nq + q], g3 = gbuf[3 * (size_t)nq + q];
            const float gx = ((g0.x + g1.x) + g2.x) + g3.x, gy = ((g0.y + g1.y) + g2.y) + g3.y;
            const float gz = ((g0.z + g1.z) + g2.z) + g3.z, gw = ((g0.w + g1.w) + g2.w) + g3.w;
            w[it].x = (float)((double)w[it].x * exp(eta * (double)gx));  // :418
            w[it].y = (float)((double)w[it].y * exp(eta * (double)gy));
            w[it].z = (float)((double)w[it].z * exp(eta * (double)gz));
            w[it].w = (float)((double)w[it].w * exp(eta * (double)gw));
            part += (w[it].x + w[it].y) + (w[it].z + w[it].w);
          }
        }
        const float normalizer = wave_sum_f(part);  // :420
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          w[it].x /= normalizer;
          w[it].y /= normalizer;
          w[it].z /= normalizer;
          w[it].w /= normalizer;
        }
        __syncthreads();  // gbuf is rewritten in the next iteration
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
    float s = 0.f;
    for (uint32_t r = wave; r < n; r += 4) {
      float4 rv[NIT];
      load_row(r, rv);
      float p = 0.f;
#pragma unroll
      for (int it = 0; it < NIT; ++it) p += dot4(rv[it], w[it]);
      const float z = wave_sum_f(p);
      s += a_of(r) * logf(z);
    }
    if (lane == 0) sred[wave] = s;
    __syncthreads();
    s = ((sred[0] + sred[1]) + sred[2]) + sred[3];
    second = s * (float)words_in_doc;
    first = s * avg_doc_sz;
  }
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
    // LDS: cap_rows model rows + 4 partial gradients + cap_rows values of a
    const size_t fixed = 4 * (size_t)ld * sizeof(float);
    const uint32_t cap_rows = (uint32_t)((INF_LDS - fixed) / ((size_t)ld * sizeof(float) + sizeof(float)));
    const size_t lds = (size_t)cap_rows * ld * sizeof(float) + fixed + (size_t)cap_rows * sizeof(float);
    if (D) {
#define INF(N)                                                                                                                          \
  do {                                                                                                                                  \
    HIPCHK(c, hipFuncSetAttribute((const void*)inf_docs_k<N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)INF_LDS));              \
    hipLaunchKernelGGL((inf_docs_k<N>), dim3((unsigned)D), dim3(INF_T), lds, c->stream, (const float4*)dM.p, k, nq, doffs.p, dfw.p, dfa.p, \
                       dnk.p, D, iters, Lfguess, avg_doc_sz, cap_rows, weights ? dW.p : nullptr, dtt.p, dtw.p, dllh.p, dnc.p);          \
  } while (0)
      if (nit <= 1) INF(1);
      else if (nit <= 2) INF(2);
      else INF(4);
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
