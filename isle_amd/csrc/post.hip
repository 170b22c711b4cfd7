// isle_amd/csrc/post.hip — the stage immediately downstream of the hot path and the edge topics
// of §8a a19, on the count matrix A that isle_hip_upload_counts left in HBM.
//
//   post_normalize_k     nv = avg_doc_sz * (count / doc_sum)                     src/sparseMatrix.cpp:136-167
//   post_pair_count_k    per (word, topic): documents of the topic's cluster     src/sparseMatrix.cpp:506-510
//                        that contain the word, and the smallest value
//   post_qualify_k       pairs with more than r values get a segment             :513
//   post_scatter_k       their values, gathered per pair
//   post_select_k        r-th largest of a segment: 4-pass radix select on the   :514-516 (the reference sorts)
//                        float bits (all values are positive)
//   post_thr_final_k     the "else" arm: minimum or zero                         :517-523
//   post_catch_k         catchword rule, one wave per word: only the arg-max     src/sparseMatrix.cpp:573-595
//                        topic can pass  thr_t > rho * thr_o  for all o != t
//   post_dts_k           document-topic catchword sums, entry order preserved    src/sparseMatrix.cpp:656-683
//                        (bit-identical fp32 sums), + the two heaviest topics    :687-708
//   post_topic_*         per-topic rank selection of the sums                    :713-748
//   post_model_acc_k     Model[:, t] += document column (fp32 atomics: the sum   :783-810
//                        order differs from the reference's document order)
//   post_model_norm_k    L1 normalisation of every topic vector                  :816-820
//   post_edge_k          Edge[:, e] = a * Model[:, p] + (1 - a) * Model[:, q]    src/trainer.cpp:1152-1159
//
// Everything but the Model accumulation is integer / selection / ordered-sum work and matches the CPU restatement
// bit for bit.
#include "common.h"
#include "scan.h"

namespace {

constexpr int PT = 256;
constexpr int PW = PT / ISLE_WAVE;
constexpr uint32_t NONE32 = 0xffffffffu;

__device__ inline float wsum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

inline unsigned doc_grid(isle_ctx* c, uint64_t D, int waves_per_block = PW) {
  const uint64_t want = (D + waves_per_block - 1) / waves_per_block;
  const uint64_t cap = (uint64_t)c->num_cus * 32;
  return (unsigned)std::max<uint64_t>(1, std::min(want, cap));
}

__global__ __launch_bounds__(PT) void post_normalize_k(const float* __restrict__ cnt, const int64_t* __restrict__ offs, uint64_t D, float avg,
                                                        float* __restrict__ nv) {
  const int lane = threadIdx.x & 63;
  const uint64_t nw = (uint64_t)gridDim.x * PW;
  for (uint64_t d = (uint64_t)blockIdx.x * PW + (threadIdx.x >> 6); d < D; d += nw) {
    const int64_t s = offs[d], e = offs[d + 1];
    float sum = 0.f;
    for (int64_t i = s + lane; i < e; i += 64) sum += cnt[i];
    sum = wsum(sum);  // integer counts: exact in any order
    for (int64_t i = s + lane; i < e; i += 64) nv[i] = avg * (cnt[i] / sum);
  }
}

__global__ __launch_bounds__(PT) void post_fill_i32_k(int32_t* __restrict__ p, uint64_t n, int32_t v) {
  const uint64_t i = (uint64_t)blockIdx.x * PT + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ __launch_bounds__(PT) void post_fill_u32_k(uint32_t* __restrict__ p, uint64_t n, uint32_t v) {
  const uint64_t i = (uint64_t)blockIdx.x * PT + threadIdx.x;
  if (i < n) p[i] = v;
}

// cluster_of[original_cols[j] - doc_base] = assign[j]   (src/trainer.cpp:572-575); identity map when orig == nullptr
__global__ __launch_bounds__(PT) void post_cluster_of_k(const uint32_t* __restrict__ assign, const uint64_t* __restrict__ orig, uint64_t Db,
                                                         uint64_t doc_base, int32_t* __restrict__ cluster_of) {
  const uint64_t j = (uint64_t)blockIdx.x * PT + threadIdx.x;
  if (j < Db) cluster_of[orig ? orig[j] - doc_base : j] = (int32_t)assign[j];
}

__global__ __launch_bounds__(PT) void post_pair_count_k(const uint32_t* __restrict__ rows, const int64_t* __restrict__ offs, const float* __restrict__ nv,
                                                         const int32_t* __restrict__ cluster_of, uint64_t D, uint32_t k, uint32_t* __restrict__ cnt,
                                                         uint32_t* __restrict__ minbits) {
  const int lane = threadIdx.x & 63;
  const uint64_t nw = (uint64_t)gridDim.x * PW;
  for (uint64_t d = (uint64_t)blockIdx.x * PW + (threadIdx.x >> 6); d < D; d += nw) {
    const int32_t t = cluster_of[d];
    if (t < 0) continue;
    const int64_t s = offs[d], e = offs[d + 1];
    for (int64_t i = s + lane; i < e; i += 64) {
      const size_t idx = (size_t)rows[i] * k + (uint32_t)t;
      atomicAdd(&cnt[idx], 1u);
      atomicMin(&minbits[idx], __float_as_uint(nv[i]));  // values > 0: unsigned order == float order
    }
  }
}

// pairs with cnt > r: reserve a slot and a segment (layout order is arbitrary, results do not depend on it)
__global__ __launch_bounds__(PT) void post_qualify_k(const uint32_t* __restrict__ cnt, uint64_t npairs, uint32_t r, uint32_t* __restrict__ slot_of,
                                                      unsigned long long* __restrict__ counters /*nslots, total*/, uint64_t* __restrict__ seg_id,
                                                      uint64_t* __restrict__ seg_off, uint32_t* __restrict__ seg_len, uint32_t* __restrict__ seg_rank,
                                                      uint64_t slot_cap) {
  const uint64_t idx = (uint64_t)blockIdx.x * PT + threadIdx.x;
  if (idx >= npairs) return;
  const uint32_t n = cnt[idx];
  uint32_t slot = NONE32;
  if (n > r) {
    const unsigned long long s = atomicAdd(&counters[0], 1ull);
    if (s < slot_cap) {
      slot = (uint32_t)s;
      seg_id[s] = idx;
      seg_off[s] = atomicAdd(&counters[1], (unsigned long long)n);
      seg_len[s] = n;
      seg_rank[s] = r;
    }
  }
  slot_of[idx] = slot;
}

__global__ __launch_bounds__(PT) void post_scatter_k(const uint32_t* __restrict__ rows, const int64_t* __restrict__ offs, const float* __restrict__ nv,
                                                      const int32_t* __restrict__ cluster_of, uint64_t D, uint32_t k, const uint32_t* __restrict__ slot_of,
                                                      const uint64_t* __restrict__ seg_off, uint32_t* __restrict__ seg_cur, float* __restrict__ segvals) {
  const int lane = threadIdx.x & 63;
  const uint64_t nw = (uint64_t)gridDim.x * PW;
  for (uint64_t d = (uint64_t)blockIdx.x * PW + (threadIdx.x >> 6); d < D; d += nw) {
    const int32_t t = cluster_of[d];
    if (t < 0) continue;
    const int64_t s = offs[d], e = offs[d + 1];
    for (int64_t i = s + lane; i < e; i += 64) {
      const uint32_t slot = slot_of[(size_t)rows[i] * k + (uint32_t)t];
      if (slot != NONE32) segvals[seg_off[slot] + atomicAdd(&seg_cur[slot], 1u)] = nv[i];
    }
  }
}

// rank-th largest (1-based) of each segment of positive floats; out[seg_id] = value.  One workgroup per segment.
__global__ __launch_bounds__(PT) void post_select_k(const float* __restrict__ vals, const uint64_t* __restrict__ seg_id, const uint64_t* __restrict__ seg_off,
                                                     const uint32_t* __restrict__ seg_len, const uint32_t* __restrict__ seg_rank, float* __restrict__ out) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t sh_digit, sh_remaining;
  const uint64_t sg = blockIdx.x;
  const uint32_t n = seg_len[sg];
  const uint32_t* v = (const uint32_t*)(vals + seg_off[sg]);
  uint32_t prefix = 0, mask = 0, remaining = seg_rank[sg];
  for (int shift = 24; shift >= 0; shift -= 8) {
    hist[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += PT) {
      const uint32_t key = v[i];
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t cum = 0;
      int d = 255;
      for (; d > 0; --d) {
        if (cum + hist[d] >= remaining) break;
        cum += hist[d];
      }
      sh_digit = (uint32_t)d;
      sh_remaining = remaining - cum;
    }
    __syncthreads();
    prefix |= sh_digit << shift;
    mask |= 255u << shift;
    remaining = sh_remaining;
    __syncthreads();
  }
  if (threadIdx.x == 0) out[seg_id[sg]] = __uint_as_float(prefix);
}

// everything that did not get a segment (src/sparseMatrix.cpp:517-523, and :500-504 for empty clusters)
__global__ __launch_bounds__(PT) void post_thr_final_k(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ minbits, const int* __restrict__ sizes,
                                                        uint64_t npairs, uint32_t k, uint32_t r, float* __restrict__ thr) {
  const uint64_t idx = (uint64_t)blockIdx.x * PT + threadIdx.x;
  if (idx >= npairs) return;
  const uint32_t n = cnt[idx];
  const uint32_t S = (uint32_t)sizes[idx % k];
  if (S != 0 && n > r) return;  // written by post_select_k
  float v = 0.f;
  if (S != 0 && r >= S && n == S) v = __uint_as_float(minbits[idx]);
  thr[idx] = v;
}

__global__ __launch_bounds__(PT) void post_catch_k(const float* __restrict__ thr /*V x k word-major*/, uint64_t V, uint32_t k, double rho,
                                                    int32_t* __restrict__ catch_topic, unsigned long long* __restrict__ ncatch) {
  const int lane = threadIdx.x & 63;
  const uint64_t w = (uint64_t)blockIdx.x * PW + (threadIdx.x >> 6);
  if (w >= V) return;
  const float* row = thr + (size_t)w * k;
  float m1 = -1.f, m2 = -1.f;  // thresholds are >= 0
  uint32_t i1 = NONE32;
  for (uint32_t t = lane; t < k; t += 64) {
    const float v = row[t];
    if (v > m1) {
      m2 = m1;
      m1 = v;
      i1 = t;
    } else if (v > m2) {
      m2 = v;
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const float om1 = __shfl_xor(m1, o), om2 = __shfl_xor(m2, o);
    const uint32_t oi1 = __shfl_xor(i1, o);
    // merge two (max, second max) summaries; on a tied maximum the second maximum equals it, which is all the rule needs
    if (om1 > m1 || (om1 == m1 && oi1 < i1)) {
      m2 = fmaxf(m1, om2);
      m1 = om1;
      i1 = oi1;
    } else {
      m2 = fmaxf(m2, om1);
    }
  }
  if (lane == 0) {
    int32_t ct = -1;
    if (k >= 2 && i1 != NONE32 && (double)m1 > rho * (double)m2) ct = (int32_t)i1;
    catch_topic[w] = ct;
    if (ct >= 0) atomicAdd(ncatch, 1ull);
  }
}

// Document-topic catchword sums.  One wave per document, a k-float table per wave in LDS.  Catchword entries are folded
// into the table one at a time in entry order by lane 0, so every sum has the reference's fp32 rounding sequence.
// EMIT = false: nz[d] = number of topics with a non-zero sum.  EMIT = true: the (topic, sum) pairs in topic order at
// dts_off[d], plus the two heaviest topics of the document with the reference's strict-compare scan.
template <bool EMIT>
__global__ __launch_bounds__(PT) void post_dts_k(const uint32_t* __restrict__ rows, const int64_t* __restrict__ offs, const float* __restrict__ nv,
                                                  const int32_t* __restrict__ catch_topic, uint64_t D, uint32_t k, int waves, uint32_t* __restrict__ nz,
                                                  const int64_t* __restrict__ dts_off, uint32_t* __restrict__ dts_topic, float* __restrict__ dts_val,
                                                  int32_t* __restrict__ top1, int32_t* __restrict__ top2) {
  extern __shared__ float tab_all[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (wv >= waves) return;
  float* tab = tab_all + (size_t)wv * k;
  for (uint32_t t = lane; t < k; t += 64) tab[t] = 0.f;
  const uint64_t nw = (uint64_t)gridDim.x * waves;
  for (uint64_t d = (uint64_t)blockIdx.x * waves + wv; d < D; d += nw) {
    const int64_t s = offs[d], e = offs[d + 1];
    for (int64_t base = s; base < e; base += 64) {
      const int64_t i = base + lane;
      int32_t ct = -1;
      float v = 0.f;
      if (i < e) {
        ct = catch_topic[rows[i]];
        v = nv[i];
      }
      unsigned long long m = __ballot(ct >= 0);
      while (m) {
        const int j = __ffsll((long long)m) - 1;
        const int32_t t = __shfl(ct, j);
        const float x = __shfl(v, j);
        if (lane == 0) tab[t] += x;
        m &= m - 1;
      }
    }
    // table -> ordered output
    uint32_t count = 0;
    int64_t p = EMIT ? dts_off[d] : 0;
    float mx = 0.f, mx2 = 0.f;
    int32_t t1 = -1, t2 = -1;
    for (uint32_t t0 = 0; t0 < k; t0 += 64) {
      const uint32_t t = t0 + lane;
      const float x = (t < k) ? tab[t] : 0.f;
      const bool nzero = x != 0.f;
      const unsigned long long m = __ballot(nzero);
      if (EMIT) {
        if (nzero) {
          const int64_t at = p + __popcll(m & ((1ull << lane) - 1ull));
          dts_topic[at] = t;
          dts_val[at] = x;
        }
        unsigned long long mm = m;
        while (mm) {  // src/sparseMatrix.cpp:691-702, evaluated redundantly by all lanes
          const int j = __ffsll((long long)mm) - 1;
          const float y = __shfl(x, j);
          if (y > mx) {
            mx2 = mx;
            t2 = t1;
            mx = y;
            t1 = (int32_t)(t0 + j);
          } else if (y > mx2) {
            mx2 = y;
            t2 = (int32_t)(t0 + j);
          }
          mm &= mm - 1;
        }
        p += __popcll(m);
      }
      count += __popcll(m);
      if (nzero) tab[t] = 0.f;
    }
    if (lane == 0) {
      if (EMIT) {
        const bool both = t1 >= 0 && t2 >= 0;
        top1[d] = both ? t1 : -1;
        top2[d] = both ? t2 : -1;
      } else {
        nz[d] = count;
      }
    }
  }
}

// per-topic entry counts (LDS-privatised), then block-aggregated placement of the values into per-topic segments
__global__ __launch_bounds__(PT) void post_topic_count_k(const uint32_t* __restrict__ topic, uint64_t n, uint32_t k, uint32_t* __restrict__ tcnt) {
  extern __shared__ uint32_t shc[];
  for (uint32_t t = threadIdx.x; t < k; t += PT) shc[t] = 0;
  __syncthreads();
  const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
  const uint64_t b = (uint64_t)blockIdx.x * per, e = b + per < n ? b + per : n;
  for (uint64_t i = b + threadIdx.x; i < e; i += PT) atomicAdd(&shc[topic[i]], 1u);
  __syncthreads();
  for (uint32_t t = threadIdx.x; t < k; t += PT)
    if (shc[t]) atomicAdd(&tcnt[t], shc[t]);
}
__global__ __launch_bounds__(PT) void post_topic_scatter_k(const uint32_t* __restrict__ topic, const float* __restrict__ val, uint64_t n, uint32_t k,
                                                            const int64_t* __restrict__ toff, uint32_t* __restrict__ tcur, float* __restrict__ out) {
  extern __shared__ uint32_t shc[];  // [0,k): count then local cursor; [k,2k): base reserved in the topic's segment
  for (uint32_t t = threadIdx.x; t < 2 * k; t += PT) shc[t] = 0;
  __syncthreads();
  const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
  const uint64_t b = (uint64_t)blockIdx.x * per, e = b + per < n ? b + per : n;
  for (uint64_t i = b + threadIdx.x; i < e; i += PT) atomicAdd(&shc[topic[i]], 1u);
  __syncthreads();
  for (uint32_t t = threadIdx.x; t < k; t += PT) {
    const uint32_t m = shc[t];
    shc[k + t] = m ? atomicAdd(&tcur[t], m) : 0u;
    shc[t] = 0;
  }
  __syncthreads();
  for (uint64_t i = b + threadIdx.x; i < e; i += PT) {
    const uint32_t t = topic[i];
    out[toff[t] + shc[k + t] + atomicAdd(&shc[t], 1u)] = val[i];
  }
}
// segments = topics: rank-th largest if the topic has at least `rank` sums, else threshold 0 (:728-738)
__global__ __launch_bounds__(PT) void post_topic_segs_k(const uint32_t* __restrict__ tcnt, const int64_t* __restrict__ toff, uint32_t k, uint32_t rank,
                                                         unsigned long long* __restrict__ nseg, uint64_t* __restrict__ seg_id, uint64_t* __restrict__ seg_off,
                                                         uint32_t* __restrict__ seg_len, uint32_t* __restrict__ seg_rank, float* __restrict__ mthr) {
  const uint32_t t = blockIdx.x * PT + threadIdx.x;
  if (t >= k) return;
  mthr[t] = 0.f;
  if (tcnt[t] >= rank && rank > 0) {
    const unsigned long long s = atomicAdd(nseg, 1ull);
    seg_id[s] = t;
    seg_off[s] = (uint64_t)toff[t];
    seg_len[s] = tcnt[t];
    seg_rank[s] = rank;
  }
}

__global__ __launch_bounds__(PT) void post_model_acc_k(const uint32_t* __restrict__ rows, const int64_t* __restrict__ offs, const float* __restrict__ nv,
                                                        const int32_t* __restrict__ cluster_of, uint64_t D, uint64_t V, const int64_t* __restrict__ dts_off,
                                                        const uint32_t* __restrict__ dts_topic, const float* __restrict__ dts_val,
                                                        const float* __restrict__ mthr, float* __restrict__ model) {
  const int lane = threadIdx.x & 63;
  const uint64_t nw = (uint64_t)gridDim.x * PW;
  for (uint64_t d = (uint64_t)blockIdx.x * PW + (threadIdx.x >> 6); d < D; d += nw) {
    const int64_t s = offs[d], e = offs[d + 1];
    const int64_t js = dts_off[d], je = dts_off[d + 1];
    for (int64_t j = js; j <= je; ++j) {  // j == je: the document's own cluster (:804-806)
      int64_t t;
      if (j < je) {
        t = dts_topic[j];
        if (!(dts_val[j] > mthr[t])) continue;
      } else {
        t = cluster_of[d];
        if (t < 0) continue;
      }
      float* col = model + (size_t)t * V;
      for (int64_t i = s + lane; i < e; i += 64) atomicAdd(&col[rows[i]], nv[i]);
    }
  }
}

__global__ __launch_bounds__(PT) void post_model_norm_k(float* __restrict__ model, uint64_t V) {
  __shared__ float sh[PT];
  float* col = model + (size_t)blockIdx.x * V;
  float s = 0.f;
  for (uint64_t w = threadIdx.x; w < V; w += PT) s += fabsf(col[w]);
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = PT / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  const float a = (float)(1.0 / (double)sh[0]);  // FPscal(n, 1.0 / sum, ...): double reciprocal, float scale
  for (uint64_t w = threadIdx.x; w < V; w += PT) col[w] *= a;
}

__global__ __launch_bounds__(PT) void post_transpose_thr_k(const float* __restrict__ thr_wm, uint64_t V, uint32_t k, float* __restrict__ thr_cm) {
  __shared__ float tile[32][33];
  const uint64_t w0 = (uint64_t)blockIdx.x * 32;
  const uint32_t t0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const uint64_t w = w0 + r;
    const uint32_t t = t0 + tx;
    tile[r][tx] = (w < V && t < k) ? thr_wm[(size_t)w * k + t] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const uint32_t t = t0 + r;
    const uint64_t w = w0 + tx;
    if (w < V && t < k) thr_cm[(size_t)t * V + w] = tile[tx][r];
  }
}

__global__ __launch_bounds__(PT) void post_edge_k(const float* __restrict__ model, uint64_t V, const int64_t* __restrict__ pairs, float a, float b,
                                                   float* __restrict__ edge) {
  const int64_t p = pairs[2 * blockIdx.y], q = pairs[2 * blockIdx.y + 1];
  const uint64_t w = (uint64_t)blockIdx.x * PT + threadIdx.x;
  if (w >= V) return;
  float y = a * model[(size_t)p * V + w];  // FPaxpy into a zeroed column (src/trainer.cpp:1154-1156)
  y = fmaf(b, model[(size_t)q * V + w], y);  // second FPaxpy (:1157-1159)
  edge[(size_t)blockIdx.y * V + w] = y;
}

}  // namespace

#define LAUNCH_CHECK(c) HIPCHK(c, hipGetLastError())

int k_post_normalize(isle_ctx* c, float avg) {
  TimeScope ts(c, ISLE_T_POST);
  HIPCHK(c, c->a_nv.reserve(c->a_nnz ? c->a_nnz : 1));
  if (c->a_D == 0) return 0;
  hipLaunchKernelGGL(post_normalize_k, dim3(doc_grid(c, c->a_D)), dim3(PT), 0, c->stream, c->a_cnt.p, c->a_offs.p, c->a_D, avg, c->a_nv.p);
  LAUNCH_CHECK(c);
  return 0;
}

// cluster_of for the documents of A from the partition of B's columns (device array `assign`, c->D entries)
int k_post_cluster_of(isle_ctx* c, const uint32_t* assign_dev, bool identity) {
  TimeScope ts(c, ISLE_T_POST);
  HIPCHK(c, c->p_cluster_of.reserve(c->a_D ? c->a_D : 1));
  if (c->a_D) hipLaunchKernelGGL(post_fill_i32_k, dim3(cdiv((long)c->a_D, PT)), dim3(PT), 0, c->stream, c->p_cluster_of.p, c->a_D, -1);
  if (c->D)
    hipLaunchKernelGGL(post_cluster_of_k, dim3(cdiv((long)c->D, PT)), dim3(PT), 0, c->stream, assign_dev, identity ? (const uint64_t*)nullptr : c->original_cols.p,
                       c->D, c->a_doc_offset, c->p_cluster_of.p);
  LAUNCH_CHECK(c);
  return 0;
}

// thresholds (word-major V x k in c->p_thr) for rank r; sizes_dev = documents per topic (k ints)
int k_post_catch_thresholds(isle_ctx* c, uint32_t k, uint32_t r, const int* sizes_dev) {
  TimeScope ts(c, ISLE_T_POST);
  const uint64_t V = c->a_V, D = c->a_D, np = V * k;
  HIPCHK(c, c->p_cnt.reserve(np));
  HIPCHK(c, c->p_min.reserve(np));
  HIPCHK(c, c->p_thr.reserve(np));
  HIPCHK(c, c->p_slot.reserve(np));
  HIPCHK(c, hipMemsetAsync(c->p_cnt.p, 0, np * sizeof(uint32_t), c->stream));
  HIPCHK(c, hipMemsetAsync(c->p_min.p, 0xff, np * sizeof(uint32_t), c->stream));
  if (D) hipLaunchKernelGGL(post_pair_count_k, dim3(doc_grid(c, D)), dim3(PT), 0, c->stream, c->a_rows.p, c->a_offs.p, c->a_nv.p, c->p_cluster_of.p, D, k,
                            c->p_cnt.p, c->p_min.p);
  LAUNCH_CHECK(c);
  // at most nnz / (r + 1) pairs can hold more than r values
  const uint64_t slot_cap = c->a_nnz / ((uint64_t)r + 1) + 1;
  HIPCHK(c, c->p_seg_id.reserve(slot_cap));
  HIPCHK(c, c->p_seg_off.reserve(slot_cap));
  HIPCHK(c, c->p_seg_len.reserve(slot_cap));
  HIPCHK(c, c->p_seg_rank.reserve(slot_cap));
  HIPCHK(c, c->p_seg_cur.reserve(slot_cap));
  HIPCHK(c, c->p_counters.reserve(4));
  HIPCHK(c, hipMemsetAsync(c->p_counters.p, 0, 4 * sizeof(uint64_t), c->stream));
  hipLaunchKernelGGL(post_qualify_k, dim3(cdiv((long)np, PT)), dim3(PT), 0, c->stream, c->p_cnt.p, np, r, c->p_slot.p,
                     (unsigned long long*)c->p_counters.p, c->p_seg_id.p, c->p_seg_off.p, c->p_seg_len.p, c->p_seg_rank.p, slot_cap);
  LAUNCH_CHECK(c);
  uint64_t h[2];
  HIPCHK(c, hipMemcpyAsync(h, c->p_counters.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const uint64_t nslots = h[0], total = h[1];
  if (nslots > slot_cap) return isle_fail(c, ISLE_E_NUMERIC, "catch thresholds: %llu qualifying pairs exceed the bound %llu",
                                          (unsigned long long)nslots, (unsigned long long)slot_cap);
  if (nslots) {
    HIPCHK(c, c->p_segvals.reserve(total));
    HIPCHK(c, hipMemsetAsync(c->p_seg_cur.p, 0, nslots * sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(post_scatter_k, dim3(doc_grid(c, D)), dim3(PT), 0, c->stream, c->a_rows.p, c->a_offs.p, c->a_nv.p, c->p_cluster_of.p, D, k,
                       c->p_slot.p, c->p_seg_off.p, c->p_seg_cur.p, c->p_segvals.p);
    hipLaunchKernelGGL(post_select_k, dim3((unsigned)nslots), dim3(PT), 0, c->stream, c->p_segvals.p, c->p_seg_id.p, c->p_seg_off.p, c->p_seg_len.p,
                       c->p_seg_rank.p, c->p_thr.p);
    LAUNCH_CHECK(c);
  }
  hipLaunchKernelGGL(post_thr_final_k, dim3(cdiv((long)np, PT)), dim3(PT), 0, c->stream, c->p_cnt.p, c->p_min.p, sizes_dev, np, k, r, c->p_thr.p);
  LAUNCH_CHECK(c);
  return 0;
}

int k_post_find_catchwords(isle_ctx* c, uint32_t k, double rho, uint64_t* ncatch_host) {
  TimeScope ts(c, ISLE_T_POST);
  HIPCHK(c, c->p_catch.reserve(c->a_V));
  HIPCHK(c, c->p_counters.reserve(4));
  HIPCHK(c, hipMemsetAsync(c->p_counters.p, 0, sizeof(uint64_t), c->stream));
  hipLaunchKernelGGL(post_catch_k, dim3(cdiv((long)c->a_V, PW)), dim3(PT), 0, c->stream, c->p_thr.p, c->a_V, k, rho, c->p_catch.p,
                     (unsigned long long*)c->p_counters.p);
  LAUNCH_CHECK(c);
  HIPCHK(c, hipMemcpyAsync(ncatch_host, c->p_counters.p, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

int k_post_thr_colmajor(isle_ctx* c, uint32_t k, float* out_dev) {
  TimeScope ts(c, ISLE_T_POST);
  hipLaunchKernelGGL(post_transpose_thr_k, dim3(cdiv((long)c->a_V, 32), cdiv(k, 32)), dim3(PT), 0, c->stream, c->p_thr.p, c->a_V, k, out_dev);
  LAUNCH_CHECK(c);
  return 0;
}

// document-topic sums -> c->p_dts_*; returns their number
int k_post_doc_topic_sums(isle_ctx* c, uint32_t k, uint64_t* n_out) {
  TimeScope ts(c, ISLE_T_POST);
  const uint64_t D = c->a_D;
  int waves = (int)std::min<uint64_t>(PW, (64 * 1024) / ((uint64_t)k * sizeof(float)));
  if (waves < 1) return isle_fail(c, ISLE_E_ARG, "topic model: num_topics = %u too large for the per-wave LDS table", k);
  const size_t shmem = (size_t)waves * k * sizeof(float);
  HIPCHK(c, c->p_nz.reserve(D ? D : 1));
  HIPCHK(c, c->p_dts_off.reserve(D + 1));
  HIPCHK(c, c->p_top1.reserve(D ? D : 1));
  HIPCHK(c, c->p_top2.reserve(D ? D : 1));
  HIPCHK(c, c->a_scan.reserve(isle_scan_scratch(D) + 4));
  if (D)
    hipLaunchKernelGGL(post_dts_k<false>, dim3(doc_grid(c, D, waves)), dim3(PT), shmem, c->stream, c->a_rows.p, c->a_offs.p, c->a_nv.p, c->p_catch.p, D, k,
                       waves, c->p_nz.p, (const int64_t*)nullptr, (uint32_t*)nullptr, (float*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr);
  LAUNCH_CHECK(c);
  HIPCHK(c, (isle_scan::exclusive_scan<uint32_t, int64_t>(c->stream, c->p_nz.p, D, c->p_dts_off.p, c->a_scan.p)));
  int64_t n = 0;
  HIPCHK(c, hipMemcpyAsync(&n, c->p_dts_off.p + D, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, c->p_dts_topic.reserve(n ? n : 1));
  HIPCHK(c, c->p_dts_val.reserve(n ? n : 1));
  if (D)
    hipLaunchKernelGGL(post_dts_k<true>, dim3(doc_grid(c, D, waves)), dim3(PT), shmem, c->stream, c->a_rows.p, c->a_offs.p, c->a_nv.p, c->p_catch.p, D, k,
                       waves, (uint32_t*)nullptr, c->p_dts_off.p, c->p_dts_topic.p, c->p_dts_val.p, c->p_top1.p, c->p_top2.p);
  LAUNCH_CHECK(c);
  c->p_dts_n = (uint64_t)n;
  *n_out = (uint64_t)n;
  return 0;
}

// per-topic thresholds c->p_mthr (k floats) = rank-th largest document sum of the topic, 0 if it has fewer
int k_post_model_thresholds(isle_ctx* c, uint32_t k, uint32_t rank) {
  TimeScope ts(c, ISLE_T_POST);
  const uint64_t n = c->p_dts_n;
  HIPCHK(c, c->p_tcnt.reserve(2 * (size_t)k));
  HIPCHK(c, c->p_toff.reserve(k + 1));
  HIPCHK(c, c->p_mthr.reserve(k));
  HIPCHK(c, c->p_seg_id.reserve(k));
  HIPCHK(c, c->p_seg_off.reserve(k));
  HIPCHK(c, c->p_seg_len.reserve(k));
  HIPCHK(c, c->p_seg_rank.reserve(k));
  HIPCHK(c, c->p_segvals.reserve(n ? n : 1));
  HIPCHK(c, c->p_counters.reserve(4));
  HIPCHK(c, c->a_scan.reserve(isle_scan_scratch(k) + 4));
  HIPCHK(c, hipMemsetAsync(c->p_tcnt.p, 0, 2 * (size_t)k * sizeof(uint32_t), c->stream));
  HIPCHK(c, hipMemsetAsync(c->p_counters.p, 0, sizeof(uint64_t), c->stream));
  const unsigned g = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n + 16383) / 16384, (uint64_t)c->num_cus * 4));
  if (n) hipLaunchKernelGGL(post_topic_count_k, dim3(g), dim3(PT), (size_t)k * sizeof(uint32_t), c->stream, c->p_dts_topic.p, n, k, c->p_tcnt.p);
  LAUNCH_CHECK(c);
  HIPCHK(c, (isle_scan::exclusive_scan<uint32_t, int64_t>(c->stream, c->p_tcnt.p, k, c->p_toff.p, c->a_scan.p)));
  if (n)
    hipLaunchKernelGGL(post_topic_scatter_k, dim3(g), dim3(PT), 2 * (size_t)k * sizeof(uint32_t), c->stream, c->p_dts_topic.p, c->p_dts_val.p, n, k,
                       c->p_toff.p, c->p_tcnt.p + k, c->p_segvals.p);
  hipLaunchKernelGGL(post_topic_segs_k, dim3(cdiv(k, PT)), dim3(PT), 0, c->stream, c->p_tcnt.p, c->p_toff.p, k, rank, (unsigned long long*)c->p_counters.p,
                     c->p_seg_id.p, c->p_seg_off.p, c->p_seg_len.p, c->p_seg_rank.p, c->p_mthr.p);
  LAUNCH_CHECK(c);
  uint64_t nseg = 0;
  HIPCHK(c, hipMemcpyAsync(&nseg, c->p_counters.p, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (nseg) {
    hipLaunchKernelGGL(post_select_k, dim3((unsigned)nseg), dim3(PT), 0, c->stream, c->p_segvals.p, c->p_seg_id.p, c->p_seg_off.p, c->p_seg_len.p,
                       c->p_seg_rank.p, c->p_mthr.p);
    LAUNCH_CHECK(c);
  }
  return 0;
}

int k_post_model(isle_ctx* c, uint32_t k) {
  TimeScope ts(c, ISLE_T_POST);
  const uint64_t V = c->a_V, D = c->a_D;
  HIPCHK(c, c->p_model.reserve(V * k));
  HIPCHK(c, hipMemsetAsync(c->p_model.p, 0, V * k * sizeof(float), c->stream));
  if (D)
    hipLaunchKernelGGL(post_model_acc_k, dim3(doc_grid(c, D)), dim3(PT), 0, c->stream, c->a_rows.p, c->a_offs.p, c->a_nv.p, c->p_cluster_of.p, D, V,
                       c->p_dts_off.p, c->p_dts_topic.p, c->p_dts_val.p, c->p_mthr.p, c->p_model.p);
  hipLaunchKernelGGL(post_model_norm_k, dim3(k), dim3(PT), 0, c->stream, c->p_model.p, V);
  LAUNCH_CHECK(c);
  return 0;
}

int k_post_edge(isle_ctx* c, const int64_t* pairs_dev, int n, float a, float b, float* edge_dev) {
  TimeScope ts(c, ISLE_T_POST);
  if (n == 0) return 0;
  hipLaunchKernelGGL(post_edge_k, dim3(cdiv((long)c->a_V, PT), n), dim3(PT), 0, c->stream, c->p_model.p, c->a_V, pairs_dev, a, b, edge_dev);
  LAUNCH_CHECK(c);
  return 0;
}
