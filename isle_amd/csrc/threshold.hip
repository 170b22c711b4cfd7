// isle_amd/csrc/threshold.hip — the stage immediately upstream of the hot path:
// A (word-document counts, CSC) -> B (thresholded, CSC) built directly in HBM.
//
//   th_stats_k   token total and non-empty documents                src/sparseMatrix.cpp:92-99
//   th_round_k   per-document normalisation + rounding + per-word    src/sparseMatrix.cpp:136-167 (normalize_docs),
//                value histogram                                      :289-354 (list_word_freqs; the descending frequency
//                                                                      list of a word is held as a histogram over the
//                                                                      rounded values, which are small integers)
//   th_zeta_k    threshold selection rule per word                    src/sparseMatrix.cpp:357-485 (compute_thresholds)
//   th_count_k   surviving entries per document (+ sampling weight)   src/sparseMatrix.cpp:1285-1321, :1385-1396
//   th_place_k   original_cols / offsets of B from the two scans      src/sparseMatrix.cpp:1302-1320
//   th_emit_k    rows / sqrt(zeta) of the surviving entries           src/sparseMatrix.cpp:1328-1361
//
// All of this is HBM-bound integer work: wave per document, coalesced entry loads, one global atomic per
// entry into the V x (maxv+1) histogram.
#include "common.h"
#include "scan.h"

namespace {

constexpr int TH_T = 256;
constexpr int TH_WAVES = TH_T / ISLE_WAVE;

__device__ inline float wave_sum_f32(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ inline uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ __launch_bounds__(TH_T) void th_stats_k(const float* __restrict__ cnt, const int64_t* __restrict__ offs, uint64_t D, uint64_t nnz,
                                                    unsigned long long* __restrict__ out /*tokens, nz_docs*/) {
  __shared__ unsigned long long sh[2][TH_T];
  unsigned long long t = 0, nz = 0;
  const uint64_t stride = (uint64_t)gridDim.x * TH_T;
  for (uint64_t i = (uint64_t)blockIdx.x * TH_T + threadIdx.x; i < nnz; i += stride) t += (unsigned long long)cnt[i];
  for (uint64_t d = (uint64_t)blockIdx.x * TH_T + threadIdx.x; d < D; d += stride) nz += (offs[d + 1] > offs[d]);
  sh[0][threadIdx.x] = t;
  sh[1][threadIdx.x] = nz;
  __syncthreads();
  for (int s = TH_T / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + s];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    atomicAdd(&out[0], sh[0][0]);
    atomicAdd(&out[1], sh[1][0]);
  }
}

// q[i] = min(round(avg * (count_i / sum_d)), maxv); hist[row][q]++ for q > 0.
// The document sum is a wave reduction: counts are integers, so every summation order is exact below 2^24 tokens per
// document (the reference accumulates sequentially in fp32, src/sparseMatrix.cpp:150-153).
__global__ __launch_bounds__(TH_T) void th_round_k(const float* __restrict__ cnt, const uint32_t* __restrict__ rows,
                                                    const int64_t* __restrict__ offs, uint64_t D, float avg, uint32_t maxv,
                                                    uint16_t* __restrict__ q, uint32_t* __restrict__ hist) {
  const int lane = threadIdx.x & 63;
  const uint64_t nw = (uint64_t)gridDim.x * TH_WAVES;
  const size_t ldh = (size_t)maxv + 1;
  for (uint64_t d = (uint64_t)blockIdx.x * TH_WAVES + (threadIdx.x >> 6); d < D; d += nw) {
    const int64_t s = offs[d], e = offs[d + 1];
    float sum = 0.f;
    for (int64_t i = s + lane; i < e; i += 64) sum += cnt[i];
    sum = wave_sum_f32(sum);
    for (int64_t i = s + lane; i < e; i += 64) {
      const float r = sum > 0.f ? roundf(avg * (cnt[i] / sum)) : 0.f;  // counts are > 0 at both entry points; never let 0 / 0 through
      const uint32_t v = (uint32_t)fminf(r, (float)maxv);
      q[i] = (uint16_t)v;
      if (v > 0) atomicAdd(&hist[(size_t)rows[i] * ldh + v], 1u);
    }
  }
}

// One thread per word walks its histogram row exactly like the reference walks the descending list.
__global__ __launch_bounds__(TH_T) void th_zeta_k(const uint32_t* __restrict__ hist, uint64_t V, uint32_t maxv, unsigned long long count_gr,
                                                   unsigned long long count_eq, float* __restrict__ zetas) {
  const uint64_t w = (uint64_t)blockIdx.x * TH_T + threadIdx.x;
  if (w >= V) return;
  const uint32_t* hw = hist + (size_t)w * ((size_t)maxv + 1);
  unsigned long long size = 0;
  for (uint32_t v = 1; v <= maxv; ++v) size += hw[v];
  float z = 1.0f;  // :399-411, :477-480
  if (size != 0 && count_gr <= size) {
    uint32_t zeta = maxv;
    unsigned long long cum = 0;
    for (uint32_t v = maxv; v >= 1; --v) {  // zeta = freqs[count_gr - 1] of the descending list
      cum += hw[v];
      if (cum >= count_gr) {
        zeta = v;
        break;
      }
    }
    while (true) {  // :445-470
      if (hw[zeta] < count_eq) {
        z = (float)zeta;
        break;
      }
      uint32_t nxt = 0;
      for (uint32_t v = zeta; v-- > 1;)
        if (hw[v] > 0) {
          nxt = v;
          break;
        }
      if (nxt == 0 || zeta == 1) break;  // z stays 1
      zeta = nxt;
    }
  }
  zetas[w] = z;
}

__global__ __launch_bounds__(TH_T) void th_count_k(const uint16_t* __restrict__ q, const uint32_t* __restrict__ rows,
                                                    const int64_t* __restrict__ offs, uint64_t D, const float* __restrict__ zetas,
                                                    uint32_t* __restrict__ kept, float* __restrict__ wgt /*nullable*/) {
  const int lane = threadIdx.x & 63;
  const uint64_t nw = (uint64_t)gridDim.x * TH_WAVES;
  for (uint64_t d = (uint64_t)blockIdx.x * TH_WAVES + (threadIdx.x >> 6); d < D; d += nw) {
    const int64_t s = offs[d], e = offs[d + 1];
    uint32_t n = 0;
    float w = 0.f;
    for (int64_t i = s + lane; i < e; i += 64) {
      const float z = zetas[rows[i]];
      if ((float)q[i] >= z) {
        ++n;
        w += z;  // integer-valued: exact in any order
      }
    }
    n = wave_sum_u32(n);
    if (wgt) w = wave_sum_f32(w);
    if (lane == 0) {
      kept[d] = n;
      if (wgt) wgt[d] = w;
    }
  }
}

__global__ __launch_bounds__(TH_T) void th_flag_k(const uint32_t* __restrict__ kept, uint64_t D, uint32_t* __restrict__ flag) {
  const uint64_t d = (uint64_t)blockIdx.x * TH_T + threadIdx.x;
  if (d < D) flag[d] = kept[d] > 0;
}

__global__ __launch_bounds__(TH_T) void th_place_k(const uint32_t* __restrict__ kept, const int64_t* __restrict__ col_of, const int64_t* __restrict__ off_all,
                                                    uint64_t D, uint64_t doc_base, uint64_t* __restrict__ original_cols, int64_t* __restrict__ boffs) {
  const uint64_t d = (uint64_t)blockIdx.x * TH_T + threadIdx.x;
  if (d < D && kept[d] > 0) {
    const int64_t j = col_of[d];
    original_cols[j] = doc_base + d;
    boffs[j] = off_all[d];
  }
  if (d == 0) boffs[col_of[D]] = off_all[D];
}

__global__ __launch_bounds__(TH_T) void th_emit_k(const uint16_t* __restrict__ q, const uint32_t* __restrict__ rows, const int64_t* __restrict__ offs,
                                                   uint64_t D, const float* __restrict__ zetas, const uint32_t* __restrict__ kept,
                                                   const int64_t* __restrict__ off_all, float* __restrict__ bvals, uint32_t* __restrict__ brows) {
  const int lane = threadIdx.x & 63;
  const uint64_t nw = (uint64_t)gridDim.x * TH_WAVES;
  for (uint64_t d = (uint64_t)blockIdx.x * TH_WAVES + (threadIdx.x >> 6); d < D; d += nw) {
    if (kept[d] == 0) continue;
    const int64_t s = offs[d], e = offs[d + 1];
    int64_t p = off_all[d];
    for (int64_t base = s; base < e; base += 64) {
      const int64_t i = base + lane;
      uint32_t r = 0;
      float z = 0.f;
      bool keep = false;
      if (i < e) {
        r = rows[i];
        z = zetas[r];
        keep = (float)q[i] >= z;
      }
      const unsigned long long m = __ballot(keep);
      if (keep) {
        const int64_t at = p + __popcll(m & ((1ull << lane) - 1ull));
        brows[at] = r;
        bvals[at] = sqrtf(z);  // :1347
      }
      p += __popcll(m);
    }
  }
}

__global__ __launch_bounds__(TH_T) void th_zero_k(uint32_t* __restrict__ kept, const uint8_t* __restrict__ drop, uint64_t D) {
  const uint64_t d = (uint64_t)blockIdx.x * TH_T + threadIdx.x;
  if (d < D && drop[d]) kept[d] = 0;
}

inline unsigned doc_grid(isle_ctx* c, uint64_t D) {
  const uint64_t want = (D + TH_WAVES - 1) / TH_WAVES;
  const uint64_t cap = (uint64_t)c->num_cus * 32;
  return (unsigned)std::max<uint64_t>(1, std::min(want, cap));
}

}  // namespace

int k_th_stats(isle_ctx* c, uint64_t* tokens_nz_dev) {
  TimeScope ts(c, ISLE_T_THRESHOLD);
  HIPCHK(c, hipMemsetAsync(tokens_nz_dev, 0, 2 * sizeof(uint64_t), c->stream));
  const unsigned g = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((c->a_nnz + TH_T - 1) / TH_T, (uint64_t)c->num_cus * 8));
  hipLaunchKernelGGL(th_stats_k, dim3(g), dim3(TH_T), 0, c->stream, c->a_cnt.p, c->a_offs.p, c->a_D, c->a_nnz,
                     (unsigned long long*)tokens_nz_dev);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int k_th_round_hist(isle_ctx* c, float avg, uint32_t maxv) {
  TimeScope ts(c, ISLE_T_THRESHOLD);
  HIPCHK(c, hipMemsetAsync(c->a_hist.p, 0, (size_t)c->a_V * ((size_t)maxv + 1) * sizeof(uint32_t), c->stream));
  if (c->a_D == 0) return 0;
  hipLaunchKernelGGL(th_round_k, dim3(doc_grid(c, c->a_D)), dim3(TH_T), 0, c->stream, c->a_cnt.p, c->a_rows.p, c->a_offs.p, c->a_D, avg, maxv,
                     c->a_q.p, c->a_hist.p);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int k_th_zetas(isle_ctx* c, uint32_t maxv, uint64_t count_gr, uint64_t count_eq) {
  TimeScope ts(c, ISLE_T_THRESHOLD);
  hipLaunchKernelGGL(th_zeta_k, dim3(cdiv((long)c->a_V, TH_T)), dim3(TH_T), 0, c->stream, c->a_hist.p, c->a_V, maxv,
                     (unsigned long long)count_gr, (unsigned long long)count_eq, c->zetas.p);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int k_th_count(isle_ctx* c, bool with_weights) {
  TimeScope ts(c, ISLE_T_THRESHOLD);
  if (c->a_D == 0) return 0;
  hipLaunchKernelGGL(th_count_k, dim3(doc_grid(c, c->a_D)), dim3(TH_T), 0, c->stream, c->a_q.p, c->a_rows.p, c->a_offs.p, c->a_D, c->zetas.p,
                     c->a_kept.p, with_weights ? c->a_wgt.p : (float*)nullptr);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int k_th_drop(isle_ctx* c, const uint8_t* drop_dev) {
  TimeScope ts(c, ISLE_T_THRESHOLD);
  if (c->a_D == 0) return 0;
  hipLaunchKernelGGL(th_zero_k, dim3(cdiv((long)c->a_D, TH_T)), dim3(TH_T), 0, c->stream, c->a_kept.p, drop_dev, c->a_D);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// Scans: off_all (entries kept before each document) and col_of (B column of each document).
int k_th_scans(isle_ctx* c) {
  TimeScope ts(c, ISLE_T_THRESHOLD);
  const uint64_t D = c->a_D;
  HIPCHK(c, c->a_flag.reserve(D ? D : 1));
  HIPCHK(c, c->a_off_all.reserve(D + 1));
  HIPCHK(c, c->a_col_of.reserve(D + 1));
  HIPCHK(c, c->a_scan.reserve(isle_scan::scan_scratch_elems(D) + 1));
  if (D) {
    hipLaunchKernelGGL(th_flag_k, dim3(cdiv((long)D, TH_T)), dim3(TH_T), 0, c->stream, c->a_kept.p, D, c->a_flag.p);
    HIPCHK(c, hipGetLastError());
  }
  HIPCHK(c, (isle_scan::exclusive_scan<uint32_t, int64_t>(c->stream, c->a_kept.p, D, c->a_off_all.p, c->a_scan.p)));
  HIPCHK(c, (isle_scan::exclusive_scan<uint32_t, int64_t>(c->stream, c->a_flag.p, D, c->a_col_of.p, c->a_scan.p)));
  return 0;
}

// Writes B into the context's CSC buffers (already reserved for Db columns / bnnz entries).
int k_th_emit(isle_ctx* c, uint64_t doc_base) {
  TimeScope ts(c, ISLE_T_THRESHOLD);
  const uint64_t D = c->a_D;
  if (D == 0) return 0;
  hipLaunchKernelGGL(th_place_k, dim3(cdiv((long)D, TH_T)), dim3(TH_T), 0, c->stream, c->a_kept.p, c->a_col_of.p, c->a_off_all.p, D, doc_base,
                     c->original_cols.p, c->offs.p);
  hipLaunchKernelGGL(th_emit_k, dim3(doc_grid(c, D)), dim3(TH_T), 0, c->stream, c->a_q.p, c->a_rows.p, c->a_offs.p, D, c->zetas.p, c->a_kept.p,
                     c->a_off_all.p, c->vals.p, c->rows.p);
  HIPCHK(c, hipGetLastError());
  return 0;
}
