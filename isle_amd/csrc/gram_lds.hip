// isle_amd/csrc/gram_lds.hip — LDS-banded Gram apply  Z = B (B^T X)  for matrices whose rows hold one value each.
//
// Replaces MKL_SpSpTrProd::multiply (include/matUtils.h:336-365) on the matrices the trainer actually builds:
// threshold_and_copy (src/sparseMatrix.cpp:1285-1321) stores sqrt(zeta_w) in every entry of word row w, so
// B = diag(s) * pattern and
//     Y[d, :] = sum_{w in d} (s_w X[w, :])          pass 1, gather over the words of a document
//     Z[w, :] = s_w * sum_{d contains w} Y[d, :]    pass 2, gather over the documents of a word
// need no values in the sparse stream.  Why another form than the wave-per-segment gather of spmm.hip: that one is
// bound by the vector-L1 tag path (~4 clk per gathered 48-B panel row).  Here the gathered operand is
// staged through LDS in bands of GL_RB rows (160 KB), every lane owns whole output items (register accumulators, no
// cross-lane reduction), and the pattern is a sliced-ELL stream of band-local u16 ids, 4 per lane per "super-round",
// read fully coalesced with a 4-deep register prefetch ring.  tools/microbench/lds_band_gather.hip is the prototype
// (0.24 ms for a 99 M-nonzero pass against 0.75 ms for the gather kernel).
//
// Geometry.  Output items (documents in pass 1, words in pass 2) are ordered by decreasing nonzero count and cut into
// slices of 64 consecutive positions; a wave owns GL_G = 4 slices (one output item per lane and group), a workgroup is
// 16 waves.  Per (wave, source band, group) the stream holds cnt super-rounds = ceil(max lane count / 4); padding ids
// point at a zero row kept behind the band in LDS.  Pass 1: a workgroup walks all word bands for its 4096 documents;
// slices are dealt to waves in serpentine order over four quantile ranges so that all waves carry the same load.
// Pass 2: a word block (64 consecutive slices) is split over document-band chunks in proportion to its work; chunk
// partials go to slabs that gl_reduce_k sums in fixed order (and scales by s_w).
//
// The summation order inside a (word, document band) cell follows the placement atomics of the build (as in the
// chunked-CSR copy of spmm.hip), so Z may differ between runs by fp32 rounding only.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "common.h"
#include "scan.h"

namespace {

constexpr int GL_WAVES = 16;
constexpr int GL_THREADS = GL_WAVES * 64;
constexpr int GL_G = 4;
constexpr uint32_t GL_RB = 3412;  // source rows per band: (3412 + 1 zero row) * 48 B = 163 824 B <= 160 KiB
constexpr uint32_t GL_LDS = (GL_RB + 1) * 48;
constexpr int GL_PF = 4;  // super-rounds in flight per wave (4 x 512 B)
constexpr uint32_t GL_NONE = 0xffffffffu;
constexpr uint32_t GL_BLOCK_SLICES = GL_WAVES * GL_G;       // 64 slices
constexpr uint32_t GL_BLOCK_ITEMS = GL_BLOCK_SLICES * 64;   // 4096 output items per workgroup

// ---------------------------------------------------------------------------------------------------------------
// does every row hold a single value?
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gl_rowval_set_k(const float* __restrict__ vals, const uint32_t* __restrict__ rows, uint64_t nnz,
                                                        float* __restrict__ rowval) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < nnz) rowval[rows[i]] = vals[i];  // racing writers of one row: any of them is a valid representative
}
__global__ __launch_bounds__(256) void gl_rowval_chk_k(const float* __restrict__ vals, const uint32_t* __restrict__ rows, uint64_t nnz,
                                                        const float* __restrict__ rowval, int* __restrict__ flag) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < nnz && !(vals[i] == rowval[rows[i]])) *flag = 1;
}

// ---------------------------------------------------------------------------------------------------------------
// orderings by decreasing nonzero count (stable: ties keep index order)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gl_len_key_k(const int64_t* __restrict__ off, uint64_t n, uint64_t stride, uint64_t maxkey,
                                                     uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint64_t len = (uint64_t)(off[(i + 1) * stride] - off[i * stride]);
  key[i] = maxkey - (len < maxkey ? len : maxkey);
  val[i] = (uint32_t)i;
}
__global__ __launch_bounds__(256) void gl_invert_k(const uint32_t* __restrict__ perm, uint64_t n, uint32_t* __restrict__ pos) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) pos[perm[i]] = (uint32_t)i;
}

// ---------------------------------------------------------------------------------------------------------------
// pass 1 build: the word-band boundaries inside every document column (rows ascend within a column)
// bst[p * (NB + 1) + band] = number of entries of document dperm[p] with row < band * GL_RB
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gl_bst_k(const uint32_t* __restrict__ rows, const int64_t* __restrict__ offs,
                                                 const uint32_t* __restrict__ dperm, uint64_t D, uint32_t NB, uint32_t* __restrict__ bst) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= D * (NB + 1)) return;
  const uint64_t p = i / (NB + 1);
  const uint32_t band = (uint32_t)(i - p * (NB + 1));
  const uint32_t d = dperm[p];
  const int64_t beg = offs[d];
  const uint32_t len = (uint32_t)(offs[d + 1] - beg);
  const uint64_t target = (uint64_t)band * GL_RB;
  uint32_t lo = 0, hi = len;  // first entry with row >= target
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if ((uint64_t)rows[beg + mid] < target) lo = mid + 1;
    else hi = mid;
  }
  bst[i] = lo;
}

__device__ inline uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, m));
  return v;
}

// cnt[(wv * NB + band) * 4 + g] = super-rounds of (wave, band, group); srsum[wv * NB + band] = their sum.
// One workgroup of 4 waves per wave wv (wave g of the workgroup = group g).  PASS 1: lane count = bst differences;
// PASS 2: lane count = size of cell (word wperm[q], band) of the row-major cells.
template <int PASS>
__global__ __launch_bounds__(256) void gl_cnt_k(const uint32_t* __restrict__ slice_of, uint32_t n_out, uint32_t NB,
                                                 const uint32_t* __restrict__ bst, const int64_t* __restrict__ seg_off,
                                                 const uint32_t* __restrict__ wperm, uint16_t* __restrict__ cnt, uint32_t* __restrict__ srsum,
                                                 int* __restrict__ overflow) {
  __shared__ uint32_t sh[4];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const size_t wv = blockIdx.x;
  const uint32_t sl = slice_of[wv * 4 + g];
  const uint64_t pos = (uint64_t)sl * 64 + lane;
  const bool live = sl != GL_NONE && pos < n_out;
  size_t base = 0;
  if (live) base = PASS == 1 ? (size_t)pos * (NB + 1) : (size_t)wperm[pos] * NB;
  for (uint32_t band = 0; band < NB; ++band) {
    uint32_t n = 0;
    if (live) n = PASS == 1 ? bst[base + band + 1] - bst[base + band] : (uint32_t)(seg_off[base + band + 1] - seg_off[base + band]);
    const uint32_t sr = (wave_max_u32(n) + 3) >> 2;
    if (lane == 0) {
      if (sr > 0xffffu) *overflow = 1;
      cnt[(wv * NB + band) * 4 + g] = (uint16_t)sr;
      sh[g] = sr;
    }
    __syncthreads();
    if (threadIdx.x == 0) srsum[wv * NB + band] = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
  }
}

// ids of pass 1: one workgroup of 4 waves per (wave wv, band); wave g writes group g's super-rounds.
__global__ __launch_bounds__(256) void gl_fill1_k(const uint32_t* __restrict__ slice_of, uint32_t D, uint32_t NB,
                                                   const uint32_t* __restrict__ bst, const uint32_t* __restrict__ dperm,
                                                   const uint32_t* __restrict__ rows, const int64_t* __restrict__ offs, uint64_t nnz,
                                                   const uint16_t* __restrict__ cnt, const int64_t* __restrict__ roff, uint2* __restrict__ ids) {
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const size_t wb = blockIdx.x;  // wv * NB + band
  const size_t wv = wb / NB;
  const uint32_t band = (uint32_t)(wb - wv * NB);
  const uint16_t* cc = cnt + wb * 4;
  const uint32_t n = cc[g];
  if (n == 0) return;
  int64_t sr0 = roff[wb];
  for (int j = 0; j < g; ++j) sr0 += cc[j];
  const uint32_t sl = slice_of[wv * 4 + g];
  const uint64_t pos = (uint64_t)sl * 64 + lane;
  int64_t base = 0;
  uint32_t len = 0;
  if (sl != GL_NONE && pos < D) {
    const uint32_t b0 = bst[pos * (NB + 1) + band], b1 = bst[pos * (NB + 1) + band + 1];
    base = offs[dperm[pos]] + b0;
    len = b1 - b0;
  }
  const uint32_t r0 = band * GL_RB;
  for (uint32_t r = 0; r < n; ++r) {
    uint32_t id[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const uint32_t j = 4 * r + t;
      int64_t at = base + j;
      if (at > (int64_t)nnz - 1) at = (int64_t)nnz - 1;
      const uint32_t row = rows[at];
      id[t] = j < len ? row - r0 : GL_RB;
    }
    ids[(size_t)(sr0 + r) * 64 + lane] = make_uint2(id[0] | (id[1] << 16), id[2] | (id[3] << 16));
  }
}

// ---------------------------------------------------------------------------------------------------------------
// pass 2 build: row-major (word, document band) cells of B, band = position of the document / GL_RB.
// Stands in for the CSR copy of the reference's operator constructor (mkl_scsrcsc, include/matUtils.h:103-106); the
// cells of one word are contiguous, so centers_from_rows_k (spmm.hip) walks them as one CSR row.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gl_cell_count_k(const uint32_t* __restrict__ rows, const int64_t* __restrict__ offs,
                                                        const uint32_t* __restrict__ dpos, uint32_t D, uint32_t NB,
                                                        uint32_t* __restrict__ cellcnt) {
  const int lane = threadIdx.x & 63;
  const uint32_t d = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (d >= D) return;
  const uint32_t band = dpos[d] / GL_RB;
  for (int64_t i = offs[d] + lane; i < offs[d + 1]; i += 64) atomicAdd(&cellcnt[(size_t)rows[i] * NB + band], 1u);
}
__global__ __launch_bounds__(256) void gl_cell_fill_k(const float* __restrict__ vals, const uint32_t* __restrict__ rows,
                                                       const int64_t* __restrict__ offs, const uint32_t* __restrict__ dpos, uint32_t D,
                                                       uint32_t NB, const int64_t* __restrict__ seg_off, uint32_t* __restrict__ cursor,
                                                       uint32_t* __restrict__ ccol, float* __restrict__ cval) {
  const int lane = threadIdx.x & 63;
  const uint32_t d = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (d >= D) return;
  const uint32_t band = dpos[d] / GL_RB;
  for (int64_t i = offs[d] + lane; i < offs[d + 1]; i += 64) {
    const size_t cell = (size_t)rows[i] * NB + band;
    const int64_t at = seg_off[cell] + atomicAdd(&cursor[cell], 1u);
    ccol[at] = d;
    cval[at] = vals[i];
  }
}

__global__ __launch_bounds__(256) void gl_fill2_k(const uint32_t* __restrict__ slice_of, uint32_t V, uint32_t NB,
                                                   const int64_t* __restrict__ seg_off, const uint32_t* __restrict__ wperm,
                                                   const uint32_t* __restrict__ ccol, const uint32_t* __restrict__ dpos, uint64_t nnz,
                                                   const uint16_t* __restrict__ cnt, const int64_t* __restrict__ roff, uint2* __restrict__ ids) {
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const size_t wb = blockIdx.x;
  const size_t wv = wb / NB;
  const uint32_t band = (uint32_t)(wb - wv * NB);
  const uint16_t* cc = cnt + wb * 4;
  const uint32_t n = cc[g];
  if (n == 0) return;
  int64_t sr0 = roff[wb];
  for (int j = 0; j < g; ++j) sr0 += cc[j];
  const uint32_t sl = slice_of[wv * 4 + g];
  const uint64_t pos = (uint64_t)sl * 64 + lane;
  int64_t base = 0;
  uint32_t len = 0;
  if (sl != GL_NONE && pos < V) {
    const size_t cell = (size_t)wperm[pos] * NB + band;
    base = seg_off[cell];
    len = (uint32_t)(seg_off[cell + 1] - base);
  }
  const uint32_t r0 = band * GL_RB;
  for (uint32_t r = 0; r < n; ++r) {
    uint32_t id[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const uint32_t j = 4 * r + t;
      int64_t at = base + j;
      if (at > (int64_t)nnz - 1) at = (int64_t)nnz - 1;
      const uint32_t p = dpos[ccol[at]];
      id[t] = j < len ? p - r0 : GL_RB;
    }
    ids[(size_t)(sr0 + r) * 64 + lane] = make_uint2(id[0] | (id[1] << 16), id[2] | (id[3] << 16));
  }
}

// total super-rounds of every word block (16 waves x all bands): the weight used to size its band chunks
__global__ __launch_bounds__(256) void gl_blocktot_k(const uint32_t* __restrict__ srsum, uint32_t nwv, uint32_t NB,
                                                      unsigned long long* __restrict__ tot) {
  __shared__ unsigned long long sh[256];
  const size_t w0 = (size_t)blockIdx.x * GL_WAVES;
  const size_t w1 = (w0 + GL_WAVES < (size_t)nwv) ? w0 + GL_WAVES : (size_t)nwv;
  unsigned long long s = 0;
  for (size_t i = w0 * NB + threadIdx.x; i < w1 * NB; i += 256) s += srsum[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int m = 128; m >= 1; m >>= 1) {
    if ((int)threadIdx.x < m) sh[threadIdx.x] += sh[threadIdx.x + m];
    __syncthreads();
  }
  if (threadIdx.x == 0) tot[blockIdx.x] = sh[0];
}

// ---------------------------------------------------------------------------------------------------------------
// the apply kernel (both passes)
// ---------------------------------------------------------------------------------------------------------------
__device__ inline void add4(float4& a, const float4 b) {
  a.x += b.x;
  a.y += b.y;
  a.z += b.z;
  a.w += b.w;
}

template <int LPE>
__global__ __launch_bounds__(GL_THREADS) void gl_apply_k(const float4* __restrict__ In, uint32_t n_src, const uint2* __restrict__ ids,
                                                          const int64_t* __restrict__ roff, const uint16_t* __restrict__ cnt,
                                                          const uint32_t* __restrict__ slice_of, const GlDesc* __restrict__ desc, uint32_t NB,
                                                          float4* __restrict__ Out, size_t slab_stride, uint32_t n_out) {
  extern __shared__ float4 xs[];  // (GL_RB + 1) rows of LPE float4
  const GlDesc ds = desc[blockIdx.x];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const bool wvalid = (uint32_t)w < ds.nw;  // wave-uniform
  const size_t wv = wvalid ? (size_t)ds.wave0 + (size_t)w * ds.wstride : (size_t)ds.wave0;
  float4 acc[GL_G][LPE];
#pragma unroll
  for (int g = 0; g < GL_G; ++g)
#pragma unroll
    for (int l = 0; l < LPE; ++l) acc[g][l] = make_float4(0.f, 0.f, 0.f, 0.f);
  // the wave's id stream is contiguous over bands and groups; reads run GL_PF super-rounds ahead (slack behind the array)
  const uint2* p = ids + (size_t)roff[wv * NB + ds.b0] * 64 + lane;
  uint2 q0 = p[0], q1 = p[64], q2 = p[128], q3 = p[192];
  p += 256;
  for (uint32_t band = ds.b0; band < ds.b1; ++band) {
    __syncthreads();  // every wave is done with the previous band
    {
      const uint32_t r0 = band * GL_RB;
      const uint32_t nrow = min(GL_RB, n_src - r0);
      const float4* src = In + (size_t)r0 * LPE;
      const uint32_t n4 = nrow * LPE;  // <= 10 * 1024
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float4 tmp[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) tmp[j] = src[min(threadIdx.x + (h * 5 + j) * (uint32_t)GL_THREADS, n4 - 1)];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          const uint32_t i = threadIdx.x + (h * 5 + j) * (uint32_t)GL_THREADS;
          if (i < n4) xs[i] = tmp[j];
        }
      }
      if (threadIdx.x < LPE) xs[GL_RB * LPE + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const uint2 cc = *reinterpret_cast<const uint2*>(cnt + (wv * NB + band) * 4);
    uint32_t c01 = __builtin_amdgcn_readfirstlane(cc.x), c23 = __builtin_amdgcn_readfirstlane(cc.y);
    if (!wvalid) c01 = c23 = 0;
#pragma unroll
    for (int g = 0; g < GL_G; ++g) {
      const uint32_t n = ((g < 2 ? c01 : c23) >> (16 * (g & 1))) & 0xffffu;
      for (uint32_t r = 0; r < n; ++r) {
        const uint2 u = q0;
        q0 = q1;
        q1 = q2;
        q2 = q3;
        q3 = *p;
        p += 64;
        const uint32_t a0 = (u.x & 0xffffu) * LPE, a1 = (u.x >> 16) * LPE, a2 = (u.y & 0xffffu) * LPE, a3 = (u.y >> 16) * LPE;
#pragma unroll
        for (int l = 0; l < LPE; ++l) add4(acc[g][l], xs[a0 + l]);
#pragma unroll
        for (int l = 0; l < LPE; ++l) add4(acc[g][l], xs[a1 + l]);
#pragma unroll
        for (int l = 0; l < LPE; ++l) add4(acc[g][l], xs[a2 + l]);
#pragma unroll
        for (int l = 0; l < LPE; ++l) add4(acc[g][l], xs[a3 + l]);
      }
    }
  }
  if (!wvalid) return;
  float4* out = Out + (size_t)ds.slab * slab_stride;
#pragma unroll
  for (int g = 0; g < GL_G; ++g) {
    const uint32_t sl = slice_of[wv * 4 + g];
    const uint64_t pos = (uint64_t)sl * 64 + lane;
    if (sl != GL_NONE && pos < n_out) {
#pragma unroll
      for (int l = 0; l < LPE; ++l) out[(pos - ds.pos_base) * LPE + l] = acc[g][l];
    }
  }
}

// Xs[w, :] = s_w * X[w, :]
__global__ __launch_bounds__(256) void gl_scale_k(const float4* __restrict__ X, const float* __restrict__ rowval, size_t n4, int LPE,
                                                   float4* __restrict__ Xs) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float s = rowval[i / LPE];
  float4 v = X[i];
  v.x *= s;
  v.y *= s;
  v.z *= s;
  v.w *= s;
  Xs[i] = v;
}

// Z[wperm[q], :] = s_w * sum over the slabs of q's word block (fixed order)
__global__ __launch_bounds__(256) void gl_reduce_k(const float4* __restrict__ part, const uint32_t* __restrict__ slab0,
                                                    const uint32_t* __restrict__ nch, const uint32_t* __restrict__ wperm,
                                                    const float* __restrict__ rowval, uint32_t V, int LPE, float4* __restrict__ Z) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)V * LPE) return;
  const uint32_t q = (uint32_t)(i / LPE);
  const int l = (int)(i - (size_t)q * LPE);
  const uint32_t ob = q / GL_BLOCK_ITEMS;
  const size_t stride = (size_t)GL_BLOCK_ITEMS * LPE;
  const float4* src = part + (size_t)slab0[ob] * stride + (size_t)(q - ob * GL_BLOCK_ITEMS) * LPE + l;
  float4 s = src[0];
  const uint32_t n = nch[ob];
  for (uint32_t ch = 1; ch < n; ++ch) add4(s, src[(size_t)ch * stride]);
  const uint32_t w = wperm[q];
  const float v = rowval[w];
  s.x *= v;
  s.y *= v;
  s.z *= v;
  s.w *= v;
  Z[(size_t)w * LPE + l] = s;
}

int bits_for(uint64_t n) {
  int b = 1;
  while ((1ull << b) <= n) ++b;
  return b;
}

// order n items by decreasing length (len_i = off[(i+1)*stride] - off[i*stride]); perm[p] = item at position p
int order_by_length(isle_ctx* c, const int64_t* off, uint64_t n, uint64_t stride, uint64_t maxlen, uint32_t* perm) {
  HIPCHK(c, c->gl_key_a.reserve(n));
  HIPCHK(c, c->gl_key_b.reserve(n));
  HIPCHK(c, c->gl_val_a.reserve(n));
  HIPCHK(c, c->gl_val_b.reserve(n));
  hipLaunchKernelGGL(gl_len_key_k, dim3(cdiv((long)n, 256)), dim3(256), 0, c->stream, off, n, stride, maxlen, c->gl_key_a.p, c->gl_val_a.p);
  HIPCHK(c, hipGetLastError());
  bool in_a = true;
  ISLECHK(k_sort_pairs_u64(c, c->gl_key_a.p, c->gl_val_a.p, c->gl_key_b.p, c->gl_val_b.p, n, bits_for(maxlen), &in_a));
  HIPCHK(c, hipMemcpyAsync(perm, in_a ? c->gl_val_a.p : c->gl_val_b.p, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

// counts -> offsets -> ids for one side; the per-lane entry source differs per pass (fill kernel chosen by PASS)
template <int PASS>
int build_side(isle_ctx* c, GlSide& s, const std::vector<uint32_t>& slice_of_host) {
  HIPCHK(c, s.slice_of.reserve(slice_of_host.size()));
  HIPCHK(c, hipMemcpyAsync(s.slice_of.p, slice_of_host.data(), slice_of_host.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  const size_t nwb = (size_t)s.nwv * s.NB;
  HIPCHK(c, s.cnt.reserve(nwb * 4));
  HIPCHK(c, s.roff.reserve(nwb + 1));
  HIPCHK(c, c->gl_srsum.reserve(nwb));
  HIPCHK(c, c->gl_scan.reserve(isle_scan::scan_scratch_elems(nwb) + 8));
  HIPCHK(c, c->gl_flag.reserve(4));
  HIPCHK(c, hipMemsetAsync(c->gl_flag.p, 0, sizeof(int), c->stream));
  hipLaunchKernelGGL((gl_cnt_k<PASS>), dim3(s.nwv), dim3(256), 0, c->stream, s.slice_of.p, s.n_out, s.NB, c->gl_bst.p, c->seg_off.p,
                     c->wperm.p, s.cnt.p, c->gl_srsum.p, c->gl_flag.p);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, (isle_scan::exclusive_scan<uint32_t, int64_t>(c->stream, c->gl_srsum.p, nwb, s.roff.p, c->gl_scan.p)));
  int overflow = 0;
  HIPCHK(c, hipMemcpyAsync(&overflow, c->gl_flag.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&s.total_sr, s.roff.p + nwb, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (overflow) return isle_fail(c, ISLE_E_NUMERIC, "operator build: more than 65535 super-rounds in one (wave, band) cell");
  HIPCHK(c, s.ids.reserve(((size_t)s.total_sr + 2 * GL_PF) * 64));
  // the prefetch ring reads up to GL_PF super-rounds past the end: keep that slack defined
  HIPCHK(c, hipMemsetAsync(s.ids.p + (size_t)s.total_sr * 64, 0, (size_t)2 * GL_PF * 64 * sizeof(uint2), c->stream));
  if (nwb) {
    if (PASS == 1)
      hipLaunchKernelGGL(gl_fill1_k, dim3((unsigned)nwb), dim3(256), 0, c->stream, s.slice_of.p, s.n_out, s.NB, c->gl_bst.p, c->dperm.p,
                         c->rows.p, c->offs.p, c->nnz, s.cnt.p, s.roff.p, s.ids.p);
    else
      hipLaunchKernelGGL(gl_fill2_k, dim3((unsigned)nwb), dim3(256), 0, c->stream, s.slice_of.p, s.n_out, s.NB, c->seg_off.p, c->wperm.p,
                         c->bcol.p, c->dpos.p, c->nnz, s.cnt.p, s.roff.p, s.ids.p);
    HIPCHK(c, hipGetLastError());
  }
  return 0;
}

template <int LPE>
int launch_apply(isle_ctx* c, const GlSide& s, const float4* In, float4* Out, size_t slab_stride) {
  static bool attr_set = false;
  if (!attr_set) {
    HIPCHK(c, hipFuncSetAttribute((const void*)gl_apply_k<LPE>, hipFuncAttributeMaxDynamicSharedMemorySize, GL_LDS));
    attr_set = true;
  }
  hipLaunchKernelGGL((gl_apply_k<LPE>), dim3(s.ndesc), dim3(GL_THREADS), GL_LDS, c->stream, In, s.n_src, s.ids.p, s.roff.p, s.cnt.p,
                     s.slice_of.p, s.desc.p, s.NB, Out, slab_stride, s.n_out);
  HIPCHK(c, hipGetLastError());
  return 0;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
int k_gl_detect(isle_ctx* c) {
  c->gl_mode = 0;
  const char* e = getenv("ISLE_GRAM_LDS");
  if (e && atoi(e) == 0) return 0;
  if (c->nnz == 0 || c->D == 0 || c->V == 0) return 0;
  if (c->D >= 0xfffffff0ull || c->V >= 0xfffffff0ull) return 0;
  HIPCHK(c, c->rowval.reserve(c->V));
  HIPCHK(c, c->gl_flag.reserve(4));
  HIPCHK(c, hipMemsetAsync(c->rowval.p, 0, c->V * sizeof(float), c->stream));
  HIPCHK(c, hipMemsetAsync(c->gl_flag.p, 0, sizeof(int), c->stream));
  const dim3 g(cdiv((long)c->nnz, 256)), b(256);
  hipLaunchKernelGGL(gl_rowval_set_k, g, b, 0, c->stream, c->vals.p, c->rows.p, c->nnz, c->rowval.p);
  HIPCHK(c, hipGetLastError());
  hipLaunchKernelGGL(gl_rowval_chk_k, g, b, 0, c->stream, c->vals.p, c->rows.p, c->nnz, c->rowval.p, c->gl_flag.p);
  HIPCHK(c, hipGetLastError());
  int flag = 0;
  HIPCHK(c, hipMemcpyAsync(&flag, c->gl_flag.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->gl_mode = flag ? 0 : 1;
  return 0;
}

int k_gl_build(isle_ctx* c) {
  const uint32_t D = (uint32_t)c->D, V = (uint32_t)c->V;
  GlSide& s1 = c->gl1;
  GlSide& s2 = c->gl2;
  // ---- documents by decreasing length
  HIPCHK(c, c->dperm.reserve(D));
  HIPCHK(c, c->dpos.reserve(D));
  ISLECHK(order_by_length(c, c->offs.p, D, 1, V, c->dperm.p));
  hipLaunchKernelGGL(gl_invert_k, dim3(cdiv(D, 256)), dim3(256), 0, c->stream, c->dperm.p, (uint64_t)D, c->dpos.p);
  HIPCHK(c, hipGetLastError());

  // ---- pass 1: outputs = documents (position order), sources = words
  s1.n_out = D;
  s1.n_src = V;
  s1.NB = (V + GL_RB - 1) / GL_RB;
  s1.nslice = (D + 63) / 64;
  s1.nwv = (s1.nslice + GL_G - 1) / GL_G;
  {
    // serpentine over four quantile ranges of the length-ordered slices: every wave gets one long, two middle, one short slice
    std::vector<uint32_t> so((size_t)s1.nwv * 4);
    const uint64_t n = s1.nwv;
    for (uint64_t wv = 0; wv < n; ++wv) {
      const uint64_t cand[4] = {wv, 2 * n - 1 - wv, 2 * n + wv, 4 * n - 1 - wv};
      for (int g = 0; g < 4; ++g) so[wv * 4 + g] = cand[g] < s1.nslice ? (uint32_t)cand[g] : GL_NONE;
    }
    HIPCHK(c, c->gl_bst.reserve((size_t)D * (s1.NB + 1)));
    const uint64_t nb = (uint64_t)D * (s1.NB + 1);
    hipLaunchKernelGGL(gl_bst_k, dim3(cdiv((long)nb, 256)), dim3(256), 0, c->stream, c->rows.p, c->offs.p, c->dperm.p, (uint64_t)D, s1.NB,
                       c->gl_bst.p);
    HIPCHK(c, hipGetLastError());
    ISLECHK(build_side<1>(c, s1, so));
    // workgroup j = waves j, j + nwg, j + 2 nwg, ... : equal totals, one slab (Y itself)
    const uint32_t nwg = (s1.nwv + GL_WAVES - 1) / GL_WAVES;
    std::vector<GlDesc> ds(nwg);
    for (uint32_t j = 0; j < nwg; ++j) {
      uint32_t nw = 0;
      while (nw < GL_WAVES && (uint64_t)j + (uint64_t)nw * nwg < s1.nwv) ++nw;
      ds[j] = GlDesc{j, nwg, nw, 0u, s1.NB, 0u, 0u, 0u};
    }
    s1.ndesc = nwg;
    HIPCHK(c, s1.desc.reserve(nwg));
    HIPCHK(c, hipMemcpyAsync(s1.desc.p, ds.data(), ds.size() * sizeof(GlDesc), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // ds / so are stack-owned
  }

  // ---- row-major (word, document band) cells
  s2.n_out = V;
  s2.n_src = D;
  s2.NB = (D + GL_RB - 1) / GL_RB;
  const size_t ncell = (size_t)V * s2.NB;
  HIPCHK(c, c->gl_cellcnt.reserve(ncell));
  HIPCHK(c, c->seg_off.reserve(ncell + 1));
  HIPCHK(c, c->gl_scan.reserve(isle_scan::scan_scratch_elems(ncell) + 8));
  HIPCHK(c, c->bcol.reserve(c->nnz));
  HIPCHK(c, c->bval.reserve(c->nnz));
  HIPCHK(c, hipMemsetAsync(c->gl_cellcnt.p, 0, ncell * sizeof(uint32_t), c->stream));
  hipLaunchKernelGGL(gl_cell_count_k, dim3(cdiv(D, 4)), dim3(256), 0, c->stream, c->rows.p, c->offs.p, c->dpos.p, D, s2.NB, c->gl_cellcnt.p);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, (isle_scan::exclusive_scan<uint32_t, int64_t>(c->stream, c->gl_cellcnt.p, ncell, c->seg_off.p, c->gl_scan.p)));
  HIPCHK(c, hipMemsetAsync(c->gl_cellcnt.p, 0, ncell * sizeof(uint32_t), c->stream));
  hipLaunchKernelGGL(gl_cell_fill_k, dim3(cdiv(D, 4)), dim3(256), 0, c->stream, c->vals.p, c->rows.p, c->offs.p, c->dpos.p, D, s2.NB,
                     c->seg_off.p, c->gl_cellcnt.p, c->bcol.p, c->bval.p);
  HIPCHK(c, hipGetLastError());
  c->nbands = s2.NB;
  c->chunk_cols = GL_RB;
  c->cells_rowmajor = true;

  // ---- words by decreasing row length
  HIPCHK(c, c->wperm.reserve(V));
  ISLECHK(order_by_length(c, c->seg_off.p, V, s2.NB, D, c->wperm.p));

  // ---- pass 2: outputs = words (position order), sources = documents (position order)
  s2.nslice = (V + 63) / 64;
  const uint32_t nblk = (s2.nslice + GL_BLOCK_SLICES - 1) / GL_BLOCK_SLICES;
  s2.nwv = nblk * GL_WAVES;
  {
    // a word block = 64 consecutive slices; serpentine inside the block keeps its 16 waves level
    std::vector<uint32_t> so((size_t)s2.nwv * 4);
    for (uint32_t ob = 0; ob < nblk; ++ob)
      for (uint32_t w = 0; w < GL_WAVES; ++w) {
        const uint32_t cand[4] = {w, 31 - w, 32 + w, 63 - w};
        for (int g = 0; g < 4; ++g) {
          const uint64_t sl = (uint64_t)ob * GL_BLOCK_SLICES + cand[g];
          so[((size_t)ob * GL_WAVES + w) * 4 + g] = sl < s2.nslice ? (uint32_t)sl : GL_NONE;
        }
      }
    ISLECHK(build_side<2>(c, s2, so));
    HIPCHK(c, c->gl_blocktot.reserve(nblk));
    hipLaunchKernelGGL(gl_blocktot_k, dim3(nblk), dim3(256), 0, c->stream, c->gl_srsum.p, s2.nwv, s2.NB, c->gl_blocktot.p);
    HIPCHK(c, hipGetLastError());
    std::vector<unsigned long long> tot(nblk);
    HIPCHK(c, hipMemcpyAsync(tot.data(), c->gl_blocktot.p, nblk * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // band chunks per word block in proportion to its super-rounds; about two workgroups per CU in total
    unsigned long long all = 0;
    for (auto t : tot) all += t;
    const double target = std::max(1.0, (double)all / (2.0 * c->num_cus));
    std::vector<uint32_t> slab0(nblk), nch(nblk);
    std::vector<GlDesc> ds;
    uint32_t nslab = 0;
    for (uint32_t ob = 0; ob < nblk; ++ob) {
      uint32_t n = (uint32_t)std::min<double>((double)s2.NB, std::max(1.0, std::ceil((double)tot[ob] / target)));
      slab0[ob] = nslab;
      nch[ob] = n;
      for (uint32_t ch = 0; ch < n; ++ch) {
        const uint32_t b0 = (uint32_t)((uint64_t)ch * s2.NB / n), b1 = (uint32_t)((uint64_t)(ch + 1) * s2.NB / n);
        ds.push_back(GlDesc{ob * GL_WAVES, 1u, (uint32_t)GL_WAVES, b0, b1, nslab + ch, ob * GL_BLOCK_ITEMS, 0u});
      }
      nslab += n;
    }
    s2.ndesc = (uint32_t)ds.size();
    HIPCHK(c, s2.desc.reserve(ds.size()));
    HIPCHK(c, c->gl_slab0.reserve(nblk));
    HIPCHK(c, c->gl_nch.reserve(nblk));
    HIPCHK(c, hipMemcpyAsync(s2.desc.p, ds.data(), ds.size() * sizeof(GlDesc), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->gl_slab0.p, slab0.data(), nblk * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->gl_nch.p, nch.data(), nblk * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, c->gl_part.reserve((size_t)nslab * GL_BLOCK_ITEMS * 12));  // sized for the widest panel (BP = 12)
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return 0;
}

// Zrm (V x BP) = B (B^T Xrm); Xrm / Yrm / Zrm of the context, BP in {4, 8, 12}
int k_gl_apply(isle_ctx* c, int BP) {
  const int LPE = BP / 4;
  if (LPE < 1 || LPE > 3) return isle_fail(c, ISLE_E_ARG, "LDS Gram apply: panel width %d not in {4, 8, 12}", BP);
  const uint32_t V = (uint32_t)c->V;
  const size_t nx4 = (size_t)V * LPE;
  HIPCHK(c, c->gl_Xs.reserve((size_t)V * BP));
  {
    TimeScope ts(c, ISLE_T_GRAM_PASS1);
    hipLaunchKernelGGL(gl_scale_k, dim3(cdiv((long)nx4, 256)), dim3(256), 0, c->stream, (const float4*)c->Xrm.p, c->rowval.p, nx4, LPE,
                       (float4*)c->gl_Xs.p);
    HIPCHK(c, hipGetLastError());
    const float4* X = (const float4*)c->gl_Xs.p;
    float4* Y = (float4*)c->Yrm.p;
    if (LPE == 1) ISLECHK(launch_apply<1>(c, c->gl1, X, Y, 0));
    if (LPE == 2) ISLECHK(launch_apply<2>(c, c->gl1, X, Y, 0));
    if (LPE == 3) ISLECHK(launch_apply<3>(c, c->gl1, X, Y, 0));
  }
  {
    TimeScope ts(c, ISLE_T_GRAM_PASS2);
    const float4* Y = (const float4*)c->Yrm.p;
    float4* P = (float4*)c->gl_part.p;
    const size_t stride = (size_t)GL_BLOCK_ITEMS * LPE;
    if (LPE == 1) ISLECHK(launch_apply<1>(c, c->gl2, Y, P, stride));
    if (LPE == 2) ISLECHK(launch_apply<2>(c, c->gl2, Y, P, stride));
    if (LPE == 3) ISLECHK(launch_apply<3>(c, c->gl2, Y, P, stride));
    hipLaunchKernelGGL(gl_reduce_k, dim3(cdiv((long)nx4, 256)), dim3(256), 0, c->stream, (const float4*)c->gl_part.p, c->gl_slab0.p,
                       c->gl_nch.p, c->wperm.p, c->rowval.p, V, LPE, (float4*)c->Zrm.p);
    HIPCHK(c, hipGetLastError());
  }
  return 0;
}
