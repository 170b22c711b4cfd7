// isle_amd/csrc/gram_lds.hip — LDS-banded Gram apply  Z = B (B^T X)  for matrices whose rows hold one value each.
//
// Replaces MKL_SpSpTrProd::multiply (include/matUtils.h:336-365) on the matrices the trainer actually builds:
// threshold_and_copy (src/sparseMatrix.cpp:1285-1321) stores sqrt(zeta_w) in every entry of word row w, so
// B = diag(s) * pattern and
//     Y[d, :] = sum_{w in d} (s_w X[w, :])          pass 1, gather over the words of a document
//     Z[w, :] = s_w * sum_{d contains w} Y[d, :]    pass 2, gather over the documents of a word
// need no values in the sparse stream.  Why another form than the wave-per-segment gather of spmm.hip: that one is
// bound by the vector-L1 tag path (~4 clk per gathered panel row).  Here the gathered operand is staged through LDS in
// bands of GL_RB = 4078 rows (planar: a float4 plane per four columns, a float2 half plane for a 10-column panel — 40 bytes a row, 160 KB),
// every lane owns whole output items (register accumulators, no cross-lane reduction), and the pattern is a sliced-ELL stream of
// band-local u16 ids (stored as 8 * id), 4 per lane per "super-round", read fully coalesced through a register ring that moves by pairs.
// tools/microbench/lds_band_gather.hip is the prototype (0.24 ms for a 99 M-nonzero pass against 0.75 ms for the gather kernel).
//
// Geometry.  Output items (documents in pass 1, words in pass 2) are ordered by decreasing nonzero count and cut into
// slices of 64 consecutive positions; a wave owns G slices (one output item per lane and group; G = 4 ... 8 per side, chosen by the
// build so that the workgroups fill whole rounds of the CUs: every workgroup stages every band it walks, so fewer, fatter workgroups
// stage less), a workgroup is 16 waves.  Per (wave, source band, group) the stream holds cnt super-rounds = ceil(max lane count / 4); inside
// such a slice gl_place_k gives a lane's entries the slots in which the rows a ds_read_b128 lane group reads lie on different LDS banks, and
// the padding slots the zero row (16 of them behind every plane) of a bank class nobody uses.  Pass 1: a workgroup walks all word bands
// for its 1024 G documents; slices are dealt to waves in serpentine order over quantile ranges so that all waves carry the same load;
// Y leaves it as the packed (planar, banded) operand of pass 2.
// Pass 2: a word block (16 G consecutive slices) is split over document-band chunks in proportion to its work; chunk
// partials go to slabs that gl_reduce_cm_k sums in fixed order (and scales by s_w).
//
// The build never materialises a transposed copy of B: per document band one workgroup histograms the band's entries by
// word in LDS (two u16 counters per dword) — first to count the (word, band) cells, then again as placement cursors
// that drop every entry's band-local document id straight into its slot of the pass-2 stream; gl_sort2_k then puts the
// ids of every cell in ascending order, so the summation order — and every bit of Z — depends on B alone, not on the
// arrival order of those LDS atomics (the reference's operator is bitwise reproducible).
//
// The same row-constant structure turns the centroid update of Lloyd on B (lloyds_iter, src/sparseMatrix.cpp:1631-1646)
// into integer counting: centre_c[w] = s_w * #{members of c that contain w} / |c|  (cc_hist_k below).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "common.h"
#include "scan.h"

namespace {

constexpr int GL_WAVES = 16;
constexpr int GL_THREADS = GL_WAVES * 64;
constexpr int GL_GMAX = 8;  // groups (output items per lane) of a wave: 4 ... 8, GlSide::G; count records are GL_GMAX wide
#ifndef GL_RB_V
#define GL_RB_V 4078
#endif
#ifndef GL_APPLY_WAVES_V
#define GL_APPLY_WAVES_V 16
#endif
constexpr uint32_t GL_RB = GL_RB_V;  // source rows per band (even: the half plane is whole float4)
constexpr uint32_t GL_APPLY_WAVES = GL_APPLY_WAVES_V;  // most waves of a workgroup of gl_apply_k (experiment builds: 8 with bands of half the height, two workgroups per CU)
constexpr uint32_t GL_NZ = 16;    // zero rows behind every plane of the band, one per residue class mod 16 (= per 16-byte bank group of the plane):
                                  // a padding slot reads the zero row of a class no real lane of its ds_read_b128 lane group uses in that slot (gl_place_k)
constexpr uint32_t GL_PS = GL_RB + GL_NZ;   // rows of a plane in LDS
constexpr uint32_t GL_PSB = GL_PS * 16;     // 65 504 bytes: the plane stride still fits the 16-bit offset field of ds_read_b128
// A band of a packed operand is PLANAR: plane l holds columns 4l .. 4l+3 of its rows as float4 (16-byte bank groups: row r lies on group
// r mod 16), a panel of 4 LPE - 2 columns ends in a half plane of float2 (8-byte slots: r mod 32).  Ten columns are 40 bytes a row —
// 4078 rows in 160 KiB where 48-byte rows held 3397 (25 word bands instead of 30 at V = 100 000: larger cells, less padding, fewer
// staging rounds) — every read is a naturally aligned ds_read_b128 / b64 (rows of 40 bytes read as 8-byte pieces were measured in round 2:
// slower than 48-byte rows), and the half plane is conflict-free where row-major 48-byte rows put all b64 reads on the even 8-byte slots.
// In global memory a band image is the same planes without the zero rows: GL_RB * (16 NF + 8 HALF) bytes, bands back to back.
__host__ __device__ constexpr uint32_t gl_img_bytes(int LPE, bool half) { return GL_RB * (16u * (uint32_t)(half ? LPE - 1 : LPE) + (half ? 8u : 0u)); }
// byte offset of (row, plane l) in a packed operand; l == NF addresses the half plane (float2)
__host__ __device__ inline size_t gl_planar_off(uint32_t row, int l, int LPE, bool half) {
  const uint32_t band = row / GL_RB, r = row - band * GL_RB;
  const int NF = half ? LPE - 1 : LPE;
  return (size_t)band * gl_img_bytes(LPE, half) + (size_t)(l < NF ? l : NF) * GL_RB * 16u + (size_t)r * (l < NF ? 16u : 8u);
}
constexpr int GL_PLACE_MAXN = 8;  // slices of up to 8 super-rounds (32 slots per lane) are placed; longer ones keep their ascending order
constexpr int GL_PF = 4;  // super-rounds in flight per wave (4 x 512 B)
constexpr uint32_t GL_NONE = 0xffffffffu;
constexpr uint32_t GL_OUT_PLANAR = 0xffffffffu;  // gl_apply_k out_n2: the output is a packed (planar, banded) operand
constexpr uint32_t GL_VP = 81920;  // words per vocabulary part of the LDS histograms (two u16 counters per dword: 160 KiB)
constexpr uint32_t GL_HLDS = GL_VP / 2 * 4;
constexpr int GL_SUB = 8;  // lanes per document in the histogram kernels
constexpr double GL_BAND_COST = 256.0;  // cost of staging one 160 KB band from HBM in pass 2, in super-rounds (measured by sweep at C2)

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

// cnt[(wv * NB + band) * GL_GMAX + g] = super-rounds of (wave, band, group); srsum[wv * NB + band] = their sum.
// One workgroup of G waves per wave wv (wave g of the workgroup = group g).  PASS 1: lane count = bst differences;
// PASS 2: lane count = size of cell (word wperm[q], band).
template <int PASS>
__global__ __launch_bounds__(64 * GL_GMAX) void gl_cnt_k(const uint32_t* __restrict__ slice_of, int G, uint32_t n_out, uint32_t NB,
                                                          const uint32_t* __restrict__ bst, const uint16_t* __restrict__ cellcnt,
                                                          const uint32_t* __restrict__ wperm, uint16_t* __restrict__ cnt, uint32_t* __restrict__ srsum,
                                                          int* __restrict__ overflow) {
  // sixteen bands per step: their counts are asked for together and the workgroup meets twice per step (a band at a time, the 2453
  // document bands of pass 2 at config 3 were a chain of 2453 loads and 4906 barriers: 3.1 ms)
  constexpr int CB = 16;
  __shared__ uint32_t sh[GL_GMAX][CB];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const size_t wv = blockIdx.x;
  const uint32_t sl = slice_of[wv * G + g];
  const uint64_t pos = (uint64_t)sl * 64 + lane;
  const bool live = sl != GL_NONE && pos < n_out;
  size_t base = 0;
  if (live) base = PASS == 1 ? (size_t)pos * (NB + 1) : (size_t)wperm[pos];
  for (uint32_t b0 = 0; b0 < NB; b0 += CB) {
    uint32_t n[CB];
#pragma unroll
    for (int t = 0; t < CB; ++t) {
      const uint32_t band = min(b0 + (uint32_t)t, NB - 1);  // clamped: the loads stay unconditional
      n[t] = 0;
      if (live) n[t] = PASS == 1 ? bst[base + band + 1] - bst[base + band] : (uint32_t)cellcnt[(size_t)band * n_out + base];  // (pass 2: n_out = V)
    }
#pragma unroll
    for (int t = 0; t < CB; ++t) {
      const uint32_t sr = (wave_max_u32(n[t]) + 3) >> 2;
      if (lane == 0 && b0 + t < NB) {
        if (sr > 0x7fffu) *overflow = 1;  // bit 15 marks a half last super-round (gl_place_k)
        cnt[(wv * NB + b0 + t) * GL_GMAX + g] = (uint16_t)sr;
        sh[g][t] = sr;
      }
    }
    __syncthreads();
    if (threadIdx.x < CB && b0 + threadIdx.x < NB) {
      const uint32_t band = b0 + threadIdx.x;
      uint32_t t = 0;
      for (int j = 0; j < G; ++j) t += sh[j][threadIdx.x];
      srsum[wv * NB + band] = t;
      for (int j = G; j < GL_GMAX; ++j) cnt[(wv * NB + band) * GL_GMAX + j] = 0;
    }
    __syncthreads();
  }
}

// ids of pass 1: one workgroup of G waves per (wave wv, band); wave g writes group g's super-rounds.
__global__ __launch_bounds__(64 * GL_GMAX) void gl_fill1_k(const uint32_t* __restrict__ slice_of, int G, uint32_t D, uint32_t NB,
                                                            const uint32_t* __restrict__ bst, const uint32_t* __restrict__ dperm,
                                                            const uint32_t* __restrict__ rows, const int64_t* __restrict__ offs, uint64_t nnz,
                                                            const uint16_t* __restrict__ cnt, const int64_t* __restrict__ roff, uint2* __restrict__ ids,
                                                            size_t nwb) {
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  // wb = wv * NB + band.  Consecutive workgroups go to the eight XCDs in turn; the 25 bands of one slice of 64 documents read the SAME
  // lines of `rows` and `bst` (a document's entries of consecutive bands follow each other: ~3 entries a band, 32 a line), so XCD x takes the
  // x-th eighth of the (wave, band) pairs and a line is fetched once into one L2 instead of once per band into all eight (round 5: the
  // counters showed 29.8 GB fetched for 2.4 GB of ids; the grid has 8 ceil(nwb / 8) workgroups)
  const size_t per = gridDim.x >> 3;
  const size_t wb = (size_t)(blockIdx.x & 7u) * per + (blockIdx.x >> 3);
  if (wb >= nwb) return;
  const size_t wv = wb / NB;
  const uint32_t band = (uint32_t)(wb - wv * NB);
  const uint16_t* cc = cnt + wb * GL_GMAX;
  const uint32_t n = cc[g];
  if (n == 0) return;
  int64_t sr0 = roff[wb];
  for (int j = 0; j < g; ++j) sr0 += cc[j];
  const uint32_t sl = slice_of[wv * G + g];
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
// pass 2 build: (word, document band) cell sizes and the id stream, by LDS histograms over the band's entries.
// band = position of the document / GL_RB; grid = (bands, vocabulary parts of GL_VP words).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GL_THREADS) void gl_hist_count_k(const uint32_t* __restrict__ rows, const int64_t* __restrict__ offs,
                                                               const uint32_t* __restrict__ dperm, uint32_t D, uint32_t V, uint32_t NB,
                                                               uint16_t* __restrict__ cellcnt /* NB x V: a band's counts are a run (round 5; word-major, the 2-byte stores below each touched a line of their own: 8.8 GB written for 0.5 GB) */) {
  extern __shared__ uint32_t hist[];  // GL_VP / 2 dwords, two u16 counters each (a cell holds <= GL_RB entries)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane / GL_SUB, sl = lane % GL_SUB;
  const uint32_t band = blockIdx.x;
  const uint32_t w0 = blockIdx.y * GL_VP, w1 = min(V, w0 + GL_VP);
  for (uint32_t j = threadIdx.x; j < GL_VP / 2; j += GL_THREADS) hist[j] = 0;
  __syncthreads();
  const uint32_t p0 = band * GL_RB, p1 = min(D, p0 + GL_RB);
  // GL_SUB lanes per document: 64 / GL_SUB independent load -> atomic chains per wave (the loop is latency-bound)
  for (uint32_t p = p0 + wave * (64 / GL_SUB) + sub; p < p1; p += GL_WAVES * (64 / GL_SUB)) {
    const uint32_t d = dperm[p];
    const int64_t iend = offs[d + 1];
    for (int64_t i = offs[d] + sl; i < iend; i += 4 * GL_SUB) {  // four row ids in flight per lane
      uint32_t w[4];
      bool in[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t iu = i + (int64_t)u * GL_SUB;
        const bool live = iu < iend;
        w[u] = rows[live ? iu : iend - 1];
        in[u] = live && w[u] >= w0 && w[u] < w1;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (in[u]) atomicAdd(&hist[(w[u] - w0) >> 1], ((w[u] - w0) & 1u) ? 0x10000u : 1u);
    }
  }
  __syncthreads();
  for (uint32_t j = threadIdx.x; j < (w1 - w0 + 1) / 2; j += GL_THREADS) {
    const uint32_t v = hist[j];
    const uint32_t w = w0 + 2 * j;
    cellcnt[(size_t)band * V + w] = (uint16_t)(v & 0xffffu);
    if (w + 1 < w1) cellcnt[(size_t)band * V + w + 1] = (uint16_t)(v >> 16);
  }
}

// sort key of a word: its row length = sum of its cells
__global__ __launch_bounds__(256) void gl_rowlen_key_k(const uint16_t* __restrict__ cellcnt, uint32_t V, uint32_t NB, uint64_t maxkey,
                                                        uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
  const uint32_t w = blockIdx.x * 256 + threadIdx.x;
  if (w >= V) return;
  uint64_t len = 0;
  for (uint32_t b = 0; b < NB; ++b) len += cellcnt[(size_t)b * V + w];
  key[w] = maxkey - (len < maxkey ? len : maxkey);
  val[w] = w;
}

// sbase[slice * NB + band] = first super-round of (slice, band) in the stream
__global__ __launch_bounds__(256) void gl_sbase_k(const uint32_t* __restrict__ slice_of, int G, const uint16_t* __restrict__ cnt,
                                                   const int64_t* __restrict__ roff, size_t nwb, uint32_t NB, uint32_t* __restrict__ sbase,
                                                   uint16_t* __restrict__ scnt /*nullable: super-rounds of (slice, band), same indexing*/) {
  const size_t wb = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (wb >= nwb) return;
  const size_t wv = wb / NB;
  const uint32_t band = (uint32_t)(wb - wv * NB);
  uint32_t base = (uint32_t)roff[wb];
  for (int g = 0; g < G; ++g) {
    const uint32_t sl = slice_of[wv * G + g];
    if (sl != GL_NONE) {
      sbase[(size_t)sl * NB + band] = base;
      if (scnt) scnt[(size_t)sl * NB + band] = cnt[wb * GL_GMAX + g];
    }
    base += cnt[wb * GL_GMAX + g];
  }
}

// every entry (w, d) of the band: slot = cursor of cell (w, band) (LDS, returning atomic) -> ids of pass 2
__global__ __launch_bounds__(GL_THREADS) void gl_hist_fill_k(const uint32_t* __restrict__ rows, const int64_t* __restrict__ offs,
                                                              const uint32_t* __restrict__ dperm, uint32_t D, uint32_t V, uint32_t NB,
                                                              const uint32_t* __restrict__ wpos, const uint32_t* __restrict__ sbase,
                                                              uint16_t* __restrict__ ids16) {
  extern __shared__ uint32_t hist[];
  constexpr int NU = 16;  // entries in flight per lane
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane / GL_SUB, sl = lane % GL_SUB;
  const uint32_t band = blockIdx.x;
  const uint32_t w0 = blockIdx.y * GL_VP, w1 = min(V, w0 + GL_VP);
  for (uint32_t j = threadIdx.x; j < GL_VP / 2; j += GL_THREADS) hist[j] = 0;
  __syncthreads();
  const uint32_t p0 = band * GL_RB, p1 = min(D, p0 + GL_RB);
  for (uint32_t p = p0 + wave * (64 / GL_SUB) + sub; p < p1; p += GL_WAVES * (64 / GL_SUB)) {
    const uint32_t d = dperm[p];
    const int64_t iend = offs[d + 1];
    // sixteen entries per lane and pass — a whole document per batch of its GL_SUB lanes (four in round 1, then eight: operator build 6.3 ->
    // 5.5 -> 5.1 ms at C2): the chain row id -> cursor -> word position -> slice base -> store is five dependent round trips, so
    // independent entries are kept in flight together
    for (int64_t i = offs[d] + sl; i < iend; i += NU * GL_SUB) {
      uint32_t w[NU], q[NU];
      bool in[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int64_t iu = i + (int64_t)u * GL_SUB;
        const bool live = iu < iend;
        w[u] = rows[live ? iu : iend - 1];
        in[u] = live && w[u] >= w0 && w[u] < w1;
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) q[u] = in[u] ? wpos[w[u]] : 0u;
      uint32_t sb[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) sb[u] = in[u] ? sbase[(size_t)(q[u] >> 6) * NB + band] : 0u;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if (in[u]) {
          const uint32_t odd = (w[u] - w0) & 1u;
          const uint32_t cur = atomicAdd(&hist[(w[u] - w0) >> 1], odd ? 0x10000u : 1u);
          const uint32_t j = odd ? (cur >> 16) : (cur & 0xffffu);
          ids16[((size_t)(sb[u] + (j >> 2)) * 64 + (q[u] & 63u)) * 4 + (j & 3u)] = (uint16_t)(p - p0);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// pass 2 fill in two steps (round 5).  gl_hist_fill_k scatters 2-byte ids over a band's whole stream region (1.3 MB at config 3); with one
// workgroup per CU the regions in flight on an XCD are ten times its L2, every line leaves half written and comes back: 29.6 GB written and
// 19.4 GB fetched for 2.6 GB of ids.  Here the band's entries are first dealt into BUCKETS of word positions (gl_fb_scatter_k: a packed word
// per entry, position << 12 | document, appended to one of NBK runs per band — few open lines per workgroup, the L2 combines them), then one
// small workgroup per (band, bucket) hands out the slots (gl_fb_fill_k: cursors of <= 8192 words and the bucket's slice bases in LDS) and
// writes the ids into a region of 1 / NBK of the band's: the regions in flight fit the L2.  The bucket sizes come from the cell counts
// (gl_fb_count_k), their offsets from a scan.  The slots of a cell are handed out in arrival order as before; gl_sort2_k orders them.
// ---------------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(GL_THREADS) void gl_fb_count_k(const uint16_t* __restrict__ cellcnt /*NB x V, by word id*/, const uint32_t* __restrict__ wperm,
                                                             uint32_t V, uint32_t W, uint32_t NBK, uint32_t* __restrict__ bcnt /*NB x NBK*/) {
  // one workgroup per band (its 2 V bytes of counts are gathered by word position: fetched once and found in the caches by all sixteen waves;
  // a workgroup per (band, bucket) fetched 14 GB for the 0.5 GB table), a wave per bucket in turn
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t band = blockIdx.x;
  for (uint32_t bk = wave; bk < NBK; bk += GL_WAVES) {
    const uint32_t q0 = bk * W, q1 = min(V, q0 + W);
    uint32_t t = 0;
    for (uint32_t q = q0 + lane; q < q1; q += 64) t += cellcnt[(size_t)band * V + wperm[q]];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += (uint32_t)__shfl_xor((int)t, o);
    if (lane == 0) bcnt[(size_t)band * NBK + bk] = t;
  }
}
__global__ __launch_bounds__(GL_THREADS) void gl_fb_scatter_k(const uint32_t* __restrict__ rows, const int64_t* __restrict__ offs,
                                                               const uint32_t* __restrict__ dperm, uint32_t D, const uint32_t* __restrict__ wpos,
                                                               uint32_t W, uint32_t NBK, const int64_t* __restrict__ boff /*NB x NBK + 1*/,
                                                               uint32_t* __restrict__ tmp) {
  __shared__ uint32_t cur[128];  // NBK <= 128: cursors relative to the band's first entry
  constexpr int NU = 8;          // entries in flight per lane
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane / GL_SUB, sl = lane % GL_SUB;
  const uint32_t band = blockIdx.x;
  const int64_t base = boff[(size_t)band * NBK];
  if (threadIdx.x < NBK) cur[threadIdx.x] = (uint32_t)(boff[(size_t)band * NBK + threadIdx.x] - base);
  __syncthreads();
  const uint32_t p0 = band * GL_RB, p1 = min(D, p0 + GL_RB);
  for (uint32_t p = p0 + wave * (64 / GL_SUB) + sub; p < p1; p += GL_WAVES * (64 / GL_SUB)) {
    const uint32_t d = dperm[p];
    const int64_t iend = offs[d + 1];
    for (int64_t i = offs[d] + sl; i < iend; i += NU * GL_SUB) {
      uint32_t w[NU], q[NU];
      bool in[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int64_t iu = i + (int64_t)u * GL_SUB;
        in[u] = iu < iend;
        w[u] = rows[in[u] ? iu : iend - 1];
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) q[u] = wpos[w[u]];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if (in[u]) {
          const uint32_t at = atomicAdd(&cur[q[u] / W], 1u);
          tmp[base + at] = (q[u] << 12) | (p - p0);
        }
      }
    }
  }
}
__global__ __launch_bounds__(256) void gl_fb_fill_k(const uint32_t* __restrict__ tmp, const int64_t* __restrict__ boff, uint32_t V, uint32_t NB, uint32_t W,
                                                     uint32_t NBK, const uint32_t* __restrict__ sbase, const uint16_t* __restrict__ scnt,
                                                     uint16_t* __restrict__ ids16, uint32_t lds_sr /*super-rounds the LDS image holds*/) {
  // LDS: W cursors | the bucket's slice bases | their super-round offsets inside the image | the image: the bucket's super-rounds of this band
  // (512 bytes each), assembled here with their padding ids and written out as whole lines — a 2-byte store per id straight into the stream
  // leaves every 32-byte sector partly written (the slots no entry lands in), and the memory pays a read-modify-write for it: 15.6 GB written
  // for a 3.2 GB stream.  A bucket whose super-rounds do not fit (a very frequent word) scatters into the stream as gl_hist_fill_k does.
  extern __shared__ uint32_t fb_lds[];
  const uint32_t NSL = W / 64 + 2;
  uint32_t* cur = fb_lds;
  uint32_t* sb = cur + W;
  uint32_t* loff = sb + NSL;
  uint16_t* img = reinterpret_cast<uint16_t*>(fb_lds + ((W + 2 * NSL + 1 + 3) & ~3u));  // 16-byte aligned: read back as uint4
  const uint32_t band = blockIdx.x, bk = blockIdx.y;
  const uint32_t q0 = bk * W, q1 = min(V, q0 + W);
  if (q0 >= q1) return;
  const uint32_t s0 = q0 >> 6, ns = ((q1 - 1) >> 6) - s0 + 1;
  for (uint32_t j = threadIdx.x; j < q1 - q0; j += 256) cur[j] = 0;
  for (uint32_t j = threadIdx.x; j < ns; j += 256) {
    sb[j] = sbase[(size_t)(s0 + j) * NB + band];
    loff[j + 1] = scnt[(size_t)(s0 + j) * NB + band] & 0x7fffu;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    loff[0] = 0;
    for (uint32_t j = 0; j < ns; ++j) loff[j + 1] += loff[j];
  }
  __syncthreads();
  const uint32_t total = loff[ns];
  const bool in_lds = total <= lds_sr;  // uniform
  if (in_lds) {
    uint32_t* img32 = reinterpret_cast<uint32_t*>(img);
    for (uint32_t i = threadIdx.x; i < total * 128; i += 256) img32[i] = GL_RB | (GL_RB << 16);
  }
  __syncthreads();
  const int64_t e0 = boff[(size_t)band * NBK + bk], e1 = boff[(size_t)band * NBK + bk + 1];
  for (int64_t e = e0 + threadIdx.x; e < e1; e += 4 * 256) {  // four entries in flight per thread
    uint32_t x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = __builtin_nontemporal_load(&tmp[min(e + (int64_t)u * 256, e1 - 1)]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (e + (int64_t)u * 256 < e1) {
        const uint32_t q = x[u] >> 12, sl = (q >> 6) - s0;
        const uint32_t j = atomicAdd(&cur[q - q0], 1u);
        if (in_lds) img[((size_t)(loff[sl] + (j >> 2)) * 64 + (q & 63u)) * 4 + (j & 3u)] = (uint16_t)(x[u] & 0xfffu);
        else ids16[((size_t)(sb[sl] + (j >> 2)) * 64 + (q & 63u)) * 4 + (j & 3u)] = (uint16_t)(x[u] & 0xfffu);
      }
    }
  }
  if (!in_lds) return;
  __syncthreads();
  const uint4* src = reinterpret_cast<const uint4*>(img);
  for (uint32_t j = 0; j < ns; ++j) {  // a slice's super-rounds are one run of the stream
    const uint32_t n4 = (loff[j + 1] - loff[j]) * 32;  // uint4 per super-round: 32
    uint4* dst = reinterpret_cast<uint4*>(ids16) + (size_t)sb[j] * 32;
    for (uint32_t t = threadIdx.x; t < n4; t += 256) dst[t] = src[(size_t)loff[j] * 32 + t];
  }
}

// The cursors above hand out a cell's slots in the order the LDS atomics arrive, which differs from run to run; this pass puts
// the ids of every (word, document band) cell in ascending order, so that the stream — and with it the summation order of pass
// 2 and every bit of Z — is a function of B alone (the reference's operator is bitwise reproducible).  A lane's slots
// of one (wave, band, group) hold the cell's ids (distinct band-local document positions < GL_RB) followed by padding (GL_RB).
// gl_sort2_k: up to 16 super-rounds (64 slots) are sorted in registers by a bitonic network (no LDS, full occupancy); longer
// cells (frequent words) go on a list for gl_sort2_big_k, where a lane marks its ids in a bitmap of its own (one LDS column per
// lane, no atomics, no cross-lane traffic) and writes them back in bit order.
template <int N>
__device__ inline void gl_sort_regs(uint2* __restrict__ s, uint32_t n) {  // n <= N / 4 super-rounds of this lane, stride 64
  uint32_t v[N];
#pragma unroll
  for (int r = 0; r < N / 4; ++r) {
    const uint2 u = s[(size_t)min((uint32_t)r, n - 1) * 64];  // clamped address, masked value: the loads stay unconditional
    const bool live = (uint32_t)r < n;
    v[4 * r] = live ? (u.x & 0xffffu) : 0xffffu;
    v[4 * r + 1] = live ? (u.x >> 16) : 0xffffu;
    v[4 * r + 2] = live ? (u.y & 0xffffu) : 0xffffu;
    v[4 * r + 3] = live ? (u.y >> 16) : 0xffffu;
  }
#pragma unroll
  for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int l = i ^ j;
        if (l > i) {
          const uint32_t a = v[i], b = v[l];
          const uint32_t lo = min(a, b), hi = max(a, b);
          const bool up = (i & k) == 0;
          v[i] = up ? lo : hi;
          v[l] = up ? hi : lo;
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < N / 4; ++r)
    if ((uint32_t)r < n) s[(size_t)r * 64] = make_uint2(v[4 * r] | (v[4 * r + 1] << 16), v[4 * r + 2] | (v[4 * r + 3] << 16));
}

// one workgroup of G waves per (wave wv, band); wave g sorts group g's slots.  big[0] = number of entries that follow.
__global__ __launch_bounds__(64 * GL_GMAX) void gl_sort2_k(uint32_t NB, const uint16_t* __restrict__ cnt, const int64_t* __restrict__ roff,
                                                            uint2* __restrict__ ids, uint32_t* __restrict__ big) {
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const size_t wb = blockIdx.x;  // wv * NB + band
  const uint16_t* cc = cnt + wb * GL_GMAX;
  const uint32_t n = cc[g];
  if (n == 0) return;
  int64_t sr0 = roff[wb];
  for (int j = 0; j < g; ++j) sr0 += cc[j];
  uint2* s = ids + (size_t)sr0 * 64 + lane;
  if (n <= 2) gl_sort_regs<8>(s, n);
  else if (n <= 4) gl_sort_regs<16>(s, n);
  else if (n <= 8) gl_sort_regs<32>(s, n);
  else if (n <= 16) gl_sort_regs<64>(s, n);
  else if (lane == 0) big[1 + atomicAdd(&big[0], 1u)] = (uint32_t)(wb * GL_GMAX + g);  // the list's order does not matter
}

constexpr int GL_BMW = ((GL_RB + 31) / 32 + 3) & ~3;  // bitmap words per lane, a multiple of 4
__global__ __launch_bounds__(128) void gl_sort2_big_k(uint32_t NB, const uint16_t* __restrict__ cnt, const int64_t* __restrict__ roff,
                                                       uint2* __restrict__ ids, const uint32_t* __restrict__ big) {
  __shared__ uint32_t bm[2][GL_BMW][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t nbig = big[0];
  for (uint32_t it = blockIdx.x * 2 + wave; it < nbig; it += gridDim.x * 2) {
    const uint32_t e = big[1 + it];
    const size_t wb = e / GL_GMAX;
    const int g = (int)(e % GL_GMAX);
    const uint16_t* cc = cnt + wb * GL_GMAX;
    const uint32_t n = cc[g];
    int64_t sr0 = roff[wb];
    for (int j = 0; j < g; ++j) sr0 += cc[j];
    for (int wd = 0; wd < GL_BMW; ++wd) bm[wave][wd][lane] = 0u;
    uint2* s = ids + (size_t)sr0 * 64 + lane;
    for (uint32_t r = 0; r < n; ++r) {
      const uint2 u = s[(size_t)r * 64];
      const uint32_t id[4] = {u.x & 0xffffu, u.x >> 16, u.y & 0xffffu, u.y >> 16};
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (id[t] < GL_RB) bm[wave][id[t] >> 5][lane] |= 1u << (id[t] & 31u);
    }
    uint16_t* s16 = reinterpret_cast<uint16_t*>(s);
    uint32_t j = 0;
    for (int wd0 = 0; wd0 < GL_BMW; wd0 += 4) {  // four bitmap words in flight per step (a chain of LDS latencies otherwise)
      uint32_t bw[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) bw[u] = bm[wave][wd0 + u][lane];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        uint32_t bits = bw[u];
        while (bits) {
          const uint32_t b = (uint32_t)__ffs((int)bits) - 1u;
          bits &= bits - 1u;
          s16[(size_t)(j >> 2) * 256 + (j & 3u)] = (uint16_t)((wd0 + u) * 32 + b);  // slot j of this lane: super-round j / 4, entry j % 4
          ++j;
        }
      }
    }
  }
}

// Bank-aware placement of a slice's entries.  A gathered panel row is read as one ds_read_b128 per plane (+ a ds_read_b64 from the half
// plane of the 10-column panel).  The LDS serves a wave's b128 in four fixed groups of 16 lanes, one cycle per group when the 16 rows lie
// in 16 different 16-byte bank groups — row r of a plane lies on group r mod 16 — and one more cycle for every further distinct row on a
// busy one; a b64 is served per 32-lane half, row r of the half plane on 8-byte slot r mod 32.  With a lane's entries packed at the front
// of its slots in ascending order the LDS array spent 18.7 cycles per wave-row at a C3 shard (SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS, 46 % of
// them SQ_LDS_BANK_CONFLICT) against 10 conflict-free.  The order of a lane's entries inside a slice is free (a sum), and so is the slot a
// lane leaves empty: every (lane group, slot) is given rows of DISTINCT classes mod 16, every (half, slot) rows of distinct classes mod 32,
// where the slice has room — an edge colouring of the bipartite multigraph lanes x classes with the slots as colours — and the padding
// slots read the zero row of a class that is free in both lane groups of their half.
// Deterministic: a class mod 16 of a half is claimed per step by its lowest pending lane (ds_min), which takes the first slot at or
// behind its cursor that neither it nor the class (mod 16 in its lane group, mod 32 in its half) uses; the result depends on the
// slice's ids alone.  One wave per slice, eight slices per workgroup; no workgroup barrier (trip counts differ per wave).
__device__ inline int gl_lane_group(int lane) {  // the four ds_read_b128 lane groups: {0-3,12-15,20-27}, {4-11,16-19,28-31}, the same + 32
  const int l = lane & 31;
  const bool a = l < 4 || (l >= 12 && l < 16) || (l >= 20 && l < 28);
  return 2 * (lane >> 5) + (a ? 0 : 1);
}
// MAXN: super-rounds of the slices this instantiation places (those of (MAXN / 2, MAXN] super-rounds, MAXN / 2 = 0 for the smallest): the
// LDS of a wave is sized by it — 16 slots per lane for up to 4 super-rounds (most slices), 32 beyond — and eight slices share a workgroup
// whatever the items per lane of the side are, so that the latency-bound claim loop runs at 16 ... 32 waves per CU.
template <int MAXN, int MINN>
__global__ __launch_bounds__(512) void gl_place_k(uint32_t NB, int G, size_t sid0, size_t nslices, uint16_t* __restrict__ cnt,
                                                   const int64_t* __restrict__ roff, uint2* __restrict__ ids) {
  constexpr int MS = 4 * MAXN;  // slots per lane
  __shared__ uint16_t ent_s[8][MS][64], out_s[8][MS][64];
  __shared__ uint32_t taken_s[8][4][16], taken32_s[8][2][32], owner_s[8][2][16], occ_s[8][4][MS];
  const int lane = threadIdx.x & 63, wq = threadIdx.x >> 6;
  const size_t sid = sid0 + (size_t)blockIdx.x * 8 + wq;  // slice = (wave wv, band, group g) = wb * G + g
  if (sid >= nslices) return;
  const size_t wb = sid / (size_t)G;
  const int g = (int)(sid - wb * (size_t)G);
  // (bit 15 of a count is set by this kernel and gl_scale_ids_k, other waves of the same launch among them: counts are read masked)
  const volatile uint16_t* cc = cnt + wb * GL_GMAX;
  const uint32_t n = cc[g] & 0x7fffu;
  if (n <= (uint32_t)MINN || n > (uint32_t)MAXN) return;  // wave-uniform
  int64_t sr0 = roff[wb];
  for (int j = 0; j < g; ++j) sr0 += cc[j] & 0x7fffu;
  uint2* s = ids + (size_t)sr0 * 64 + lane;
  const uint32_t S4 = 4 * n;
  volatile uint16_t(*ent)[64] = ent_s[wq];
  volatile uint16_t(*out)[64] = out_s[wq];
  volatile uint32_t(*taken)[16] = taken_s[wq];
  volatile uint32_t(*taken32)[32] = taken32_s[wq];
  volatile uint32_t(*owner)[16] = owner_s[wq];
  volatile uint32_t(*occ)[MS] = occ_s[wq];
  const int grp = gl_lane_group(lane), half = lane >> 5;
  uint32_t len = 0;
  for (uint32_t r = 0; r < n; ++r) {
    const uint2 u = s[(size_t)r * 64];
    const uint32_t id[4] = {u.x & 0xffffu, u.x >> 16, u.y & 0xffffu, u.y >> 16};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      ent[4 * r + t][lane] = (uint16_t)id[t];
      len += id[t] < GL_RB ? 1u : 0u;
    }
  }
  // No lane has more than 4 n - 2 entries: the last two slots of the last super-round stay empty in EVERY lane, bit 15 of the count says so
  // and gl_apply_k does not read them (half of the slices: 8 % of the LDS reads of a pass).  The entries are then placed in S = 4 n - 2 slots.
  uint32_t mxlen = len;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mxlen = max(mxlen, (uint32_t)__shfl_xor((int)mxlen, off));
  const bool half_last = mxlen + 2 <= S4;
  const uint32_t S = half_last ? S4 - 2 : S4;
  taken[lane >> 4][lane & 15] = 0u;
  taken32[lane >> 5][lane & 31] = 0u;
  if (lane < 32) owner[lane >> 4][lane & 15] = 0xffffffffu;
  for (uint32_t i = lane; i < 4 * S4; i += 64) occ[i / S4][i % S4] = 0u;
  const uint32_t maskS = S == 32 ? 0xffffffffu : (1u << S) - 1u;
  uint32_t used = 0, j = 0, st = ((uint32_t)lane * 7u) % S;
  while (true) {
    const bool pending = j < len;
    if (__ballot(pending) == 0ull) break;
    uint32_t id = 0, rho = 0;
    if (pending) {
      id = ent[j][lane];
      rho = id & 15u;
      atomicMin(const_cast<uint32_t*>(&owner[half][rho]), (uint32_t)lane);
    }
    __builtin_amdgcn_wave_barrier();
    const bool win = pending && owner[half][rho] == (uint32_t)lane;
    __builtin_amdgcn_wave_barrier();
    if (win) {
      owner[half][rho] = 0xffffffffu;
      const uint32_t t16 = taken[grp][rho], t32 = taken32[half][id & 31u];
      uint32_t cand = ~used & ~t16 & ~t32 & maskS;
      if (!cand) cand = ~used & ~t16 & maskS;  // no slot without a b64 conflict left
      if (!cand) cand = ~used & maskS;         // the class is in every free slot already: a conflict that cannot be avoided
      const uint32_t hi = cand & ~((1u << st) - 1u);
      const uint32_t sl = (uint32_t)__builtin_ctz(hi ? hi : cand);
      taken[grp][rho] = t16 | (1u << sl);
      taken32[half][id & 31u] = t32 | (1u << sl);
      used |= 1u << sl;
      atomicOr(const_cast<uint32_t*>(&occ[grp][sl]), 1u << rho);
      out[sl][lane] = (uint16_t)id;
      ++j;
      st = sl + 1 == S ? 0u : sl + 1;
    }
    __builtin_amdgcn_wave_barrier();
  }
  // padding rows: slot = lane; per half one class that is free in both lane groups (else one per group)
  uint32_t o[4] = {0, 0, 0, 0};
  if ((uint32_t)lane < S4) {
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = occ[q][lane] & 0xffffu;
  }
  __builtin_amdgcn_wave_barrier();
  if ((uint32_t)lane < S4) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t both = ~(o[2 * h] | o[2 * h + 1]) & 0xffffu;
#pragma unroll
      for (int q = 2 * h; q < 2 * h + 2; ++q) {
        const uint32_t own = ~o[q] & 0xffffu;
        const uint32_t cl = both ? (uint32_t)__builtin_ctz(both) : own ? (uint32_t)__builtin_ctz(own) : 0u;
        occ[q][lane] = GL_RB + ((cl + 16u - GL_RB % 16u) & 15u);  // the zero row of class cl
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  for (uint32_t r = 0; r < n; ++r) {
    uint32_t v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const uint32_t sl = 4 * r + t;
      v[t] = (used >> sl) & 1u ? (uint32_t)out[sl][lane] : occ[grp][sl];
    }
    s[(size_t)r * 64] = make_uint2((v[0] | (v[1] << 16)) << 3, (v[2] | (v[3] << 16)) << 3);  // stored as byte offsets in the half plane (8 id)
  }
  if (half_last && lane == 0) cnt[wb * GL_GMAX + g] = (uint16_t)(n | 0x8000u);
}

// The stream gl_apply_k reads holds 8 * id (the row's byte offset in the float2 half plane; twice that in a float4 plane): gl_place_k
// writes that form, this kernel converts the slices it leaves alone (more than GL_PLACE_MAXN super-rounds; all of them with ISLE_GL_PLACE=0).
// ids are below 4096: the shift of a packed pair does not carry from the low id into the high one.
__global__ __launch_bounds__(512) void gl_scale_ids_k(int G, size_t sid0, size_t nslices, uint32_t min_n, uint16_t* __restrict__ cnt,
                                                       const int64_t* __restrict__ roff, uint2* __restrict__ ids) {
  const int lane = threadIdx.x & 63;
  const size_t sid = sid0 + (size_t)blockIdx.x * 8 + (threadIdx.x >> 6);
  if (sid >= nslices) return;
  const size_t wb = sid / (size_t)G;
  const int g = (int)(sid - wb * (size_t)G);
  const volatile uint16_t* cc = cnt + wb * GL_GMAX;
  const uint32_t n = cc[g] & 0x7fffu;
  if (n <= min_n) return;
  int64_t sr0 = roff[wb];
  for (int j = 0; j < g; ++j) sr0 += cc[j] & 0x7fffu;
  uint2* s = ids + (size_t)sr0 * 64 + lane;
  uint32_t len = 0;  // these slices keep their entries packed at the front: the same rule for the last two slots as in gl_place_k
  for (uint32_t r = 0; r < n; ++r) {
    const uint2 u = s[(size_t)r * 64];
    len += ((u.x & 0xffffu) < GL_RB) + ((u.x >> 16) < GL_RB) + ((u.y & 0xffffu) < GL_RB) + ((u.y >> 16) < GL_RB);
    s[(size_t)r * 64] = make_uint2(u.x << 3, u.y << 3);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) len = max(len, (uint32_t)__shfl_xor((int)len, off));
  if (len + 2 <= 4 * n && lane == 0) cnt[wb * GL_GMAX + g] = (uint16_t)(n | 0x8000u);
}

// total super-rounds of every (word block, band zone): the weights used to size the band chunks.  grid = (blocks, zones)
__global__ __launch_bounds__(256) void gl_blocktot_k(const uint32_t* __restrict__ srsum, uint32_t nwv, uint32_t wpb, uint32_t NB, uint32_t nzones,
                                                      unsigned long long* __restrict__ tot) {
  __shared__ unsigned long long sh[256];
  const size_t w0 = (size_t)blockIdx.x * wpb;
  const size_t w1 = (w0 + wpb < (size_t)nwv) ? w0 + wpb : (size_t)nwv;
  const uint32_t z = blockIdx.y;
  const uint32_t zb0 = (uint32_t)((uint64_t)z * NB / nzones), zb1 = (uint32_t)((uint64_t)(z + 1) * NB / nzones);
  const uint32_t nzb = zb1 - zb0;
  unsigned long long s = 0;
  for (size_t i = threadIdx.x; i < (w1 - w0) * nzb; i += 256) {
    const size_t w = w0 + i / nzb;
    s += srsum[w * NB + zb0 + (uint32_t)(i % nzb)];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int m = 128; m >= 1; m >>= 1) {
    if ((int)threadIdx.x < m) sh[threadIdx.x] += sh[threadIdx.x + m];
    __syncthreads();
  }
  if (threadIdx.x == 0) tot[(size_t)blockIdx.x * nzones + z] = sh[0];
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

// LPE float4 per panel row; HALF: the panel's last two columns are padding (b = 4 LPE - 2 or 4 LPE - 3), so only the low
// half of the last float4 is read and accumulated (b = 10: 40 of 48 bytes per gathered row).  G: output items per lane (the groups
// of a wave; GlSide::G): the rows of a group's super-rounds add into acc[g].
// (Rounds 1-2 also carried "merged stream" forms — a lane's items sharing one stream per band, ids tagged with the item, a static
// four-register id ring loaded by inline asm — built to cut the padded slots (2.17x -> 1.52x) and the id latency; measured no faster in
// either pass, twice, and removed in round 3.)
// the four rows of ring entry U into the accumulators of group g (inside gl_apply_k)
#define GL_ROWS(U) GL_ROWS_N(U, 4)
#define GL_ROWS_N(U, NR)                                                                                                \
  {                                                                                                                     \
    const uint32_t a8[4] = {U.x & 0xffffu, U.x >> 16, U.y & 0xffffu, U.y >> 16};                                        \
    constexpr int GL_STEP = GL_INFLIGHT < (NR) ? GL_INFLIGHT : (NR);                                                    \
    _Pragma("unroll") for (int t2 = 0; t2 < (NR); t2 += GL_STEP) {                                                     \
      float4 v[GL_STEP][NFA];                                                                                           \
      float2 h[GL_STEP];                                                                                                \
      _Pragma("unroll") for (int t = 0; t < GL_STEP; ++t) {                                                             \
        const uint32_t a8x = (GL_ABLATE & 4) ? (a8[t2 + t] & 8u) : a8[t2 + t];                                          \
        const uint32_t b16 = (a8x << 1) + p0s;                                                                          \
        if (GL_ABLATE & 16) {                                                                                           \
          const float f = __builtin_bit_cast(float, a8x);                                                               \
          _Pragma("unroll") for (int l = 0; l < NF; ++l) v[t][l] = make_float4(f, f, f, f);                             \
          h[t] = make_float2(f, f);                                                                                     \
        } else {                                                                                                        \
          _Pragma("unroll") for (int l = 0; l < NF; ++l) v[t][l] = gl_lds_f4(b16 + l * GL_PSB);                         \
          if (HALF) h[t] = gl_lds_f2(a8x);                                                                              \
        }                                                                                                               \
      }                                                                                                                 \
      _Pragma("unroll") for (int t = 0; t < GL_STEP; ++t) {                                                             \
        _Pragma("unroll") for (int l = 0; l < NF; ++l) add4(acc[g][l], v[t][l]);                                        \
        if (HALF) {                                                                                                     \
          acch[g].x += h[t].x;                                                                                          \
          acch[g].y += h[t].y;                                                                                          \
        }                                                                                                               \
      }                                                                                                                 \
      if (GL_STEP < (NR)) __builtin_amdgcn_sched_barrier(0);                                                            \
    }                                                                                                                   \
  }
// loads from an LDS byte address (the dynamic LDS of gl_apply_k starts at address 0: the kernel has no static LDS)
typedef float gl_v4f __attribute__((ext_vector_type(4)));
typedef float gl_v2f __attribute__((ext_vector_type(2)));
__device__ inline float4 gl_lds_f4(uint32_t addr) {
  const gl_v4f v = *reinterpret_cast<const __attribute__((address_space(3))) gl_v4f*>(addr);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ inline float2 gl_lds_f2(uint32_t addr) {
  const gl_v2f v = *reinterpret_cast<const __attribute__((address_space(3))) gl_v2f*>(addr);
  return make_float2(v.x, v.y);
}

// one super-round of a lane's id stream (4 ids).  GL_IDS_NT: read with the non-temporal policy — the stream is read exactly once per
// pass and should not push the staged operand (X: 4 MB, read by every workgroup; the band columns of Y) out of the XCD's L2
#ifndef GL_IDS_NT
#define GL_IDS_NT 1
#endif
#ifndef GL_ABLATE
#define GL_ABLATE 0  // timing experiments only (tools/build_variant.sh): 1 no band staging, 2 no band barriers, 4 every gather reads row 0, 8 ids re-read from one place, 16 no LDS reads, 32 half the id loads, 64 loaded ids never consumed, 256 every fifth (band, group) of a wave skipped with its ids (round 6: the ceiling of a form with 0.8x the slots)
#endif
// (a macro, not a function: with the plain load behind an inline function hipcc 7.2 allocated the kernel's registers differently and
// the 10-column kernels at 6 and 7 items per lane spilled 10 / 74 registers — pass 1 of config 3 on one GPU 1.60 -> 2.77 ms)
#if GL_IDS_NT
__device__ inline uint2 gl_ld_ids_nt(const uint2* p) {
  const unsigned long long v = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(p));
  return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}
#define gl_ld_ids(p) gl_ld_ids_nt(p)
#else
#define gl_ld_ids(p) (*(p))
#endif

// GL_STAMPS (diagnostic builds only, tools/build_variant.sh stamps "-DGL_STAMPS=1"): where a wave's cycles go, by s_memtime — waiting at
// the barrier that ends a band, staging its own pieces of the next band, waiting at the barrier behind the staging, walking the band's
// super-rounds.  Sums over all valid waves of a launch; k_gl_apply_cm prints them per pass.  No stamp executes in the regular build.
#ifndef GL_STAMPS
#define GL_STAMPS 0
#endif
#if GL_STAMPS
__device__ unsigned long long gl_stamp_acc[8];
#define GL_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define GL_STAMP(var)
#endif

template <int LPE, bool HALF, int G>
__global__ __launch_bounds__(GL_THREADS) void gl_apply_k(const float4* __restrict__ In, uint32_t n_src, const uint2* __restrict__ ids,
                                                          const int64_t* __restrict__ roff, const uint16_t* __restrict__ cnt,
                                                          const uint32_t* __restrict__ slice_of, const GlDesc* __restrict__ desc, uint32_t NB,
                                                          float4* __restrict__ Out, size_t slab_stride, uint32_t n_out,
                                                          const uint32_t* __restrict__ rowmap /*nullable: position -> output row*/,
                                                          uint32_t out_ld4 /*output row stride in float4*/,
                                                          uint32_t out_n2 /*0: whole float4 per row; GL_OUT_PLANAR: a packed operand (the input of
                                                                            pass 2); else float2 per row to store (8-byte aligned rows)*/) {
  extern __shared__ float4 xs[];  // NF planes of GL_PS float4 (+ a half plane of GL_PS float2)
  constexpr int NF = HALF ? LPE - 1 : LPE;  // whole float4 per row
  constexpr int NFA = NF > 0 ? NF : 1;
  constexpr uint32_t IMG = gl_img_bytes(LPE, HALF);
  constexpr int NPIECE = NF * 64 + (HALF ? 32 : 0);  // 1-KiB pieces of a band image: 64 per plane (the last one partial), 32 for the half plane
  // entries of a super-round whose LDS rows are in flight together: all four while the accumulators leave room (a wave has 128
  // registers at 16 waves per workgroup), two beyond — measured at a C3 shard, pass 1 with 10 columns: 5 items per lane 0.313 ms with
  // four in flight against 0.322 with two; 8 items per lane 0.50 ms (spills inside the loop) against 0.38
  constexpr int GL_INFLIGHT = G * (4 * NF + (HALF ? 2 : 0)) <= 56 ? 4 : 2;
  const GlDesc ds = desc[blockIdx.x];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform, and known to the compiler as such: what hangs on it (the wave's
                                                                          // place in the stream, its count records) is addressed from scalar registers
  const int nwaves = (int)(blockDim.x >> 6);  // 16, or the side's waves per workgroup (GlSide::wpg)
  const bool wvalid = (uint32_t)w < ds.nw;
  const size_t wv = wvalid ? (size_t)ds.wave0 + (size_t)w * ds.wstride : (size_t)ds.wave0;
  // LDS image: the half plane first (the stream holds a row's byte offset in it: 8 id), the float4 planes behind it (row at 16 id = one
  // shift-add with the plane base from the stream's value; the second plane within the 16-bit offset field of the first's address)
  constexpr uint32_t P0 = HALF ? GL_PS * 8u : 0u;  // byte offset of plane 0
  // kept in a scalar register the compiler cannot see through: folded as a constant it becomes the offset field of the plane-0 read and
  // pushes plane 1 (+ 65 504) out of the field's 16 bits — one more VALU add per gathered row
  uint32_t p0s = P0;
  asm volatile("" : "+s"(p0s));
  char* const lb = reinterpret_cast<char*>(xs);
  // the zero rows behind every plane: the staging below never lands on them
  if (threadIdx.x < GL_NZ * NF) reinterpret_cast<float4*>(lb + P0 + (threadIdx.x / GL_NZ) * GL_PSB)[GL_RB + threadIdx.x % GL_NZ] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (HALF && threadIdx.x < GL_NZ) reinterpret_cast<float2*>(lb)[GL_RB + threadIdx.x] = make_float2(0.f, 0.f);
  float4 acc[G][NFA];
  float2 acch[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int l = 0; l < NF; ++l) acc[g][l] = make_float4(0.f, 0.f, 0.f, 0.f);
    acch[g] = make_float2(0.f, 0.f);
  }
  // the wave's id stream is contiguous over bands and groups; reads run GL_PF super-rounds ahead (slack behind the array)
  const uint2* p = ids + (size_t)roff[wv * NB + ds.b0] * 64 + lane;
  uint2 q0 = gl_ld_ids(p), q1 = gl_ld_ids(p + 64), q2 = gl_ld_ids(p + 128), q3 = gl_ld_ids(p + 192);
  p += 256;
#if GL_STAMPS
  unsigned long long st_b1 = 0, st_stage = 0, st_b2 = 0, st_walk = 0, st_sr = 0;
#endif
  for (uint32_t band = ds.b0; band < ds.b1; ++band) {
    GL_STAMP(t0);
    // the band's count record (GL_GMAX u16 super-round counts of this wave): asked for before the band is staged, used behind it (it used to be
    // loaded behind the second barrier: one exposed memory round trip per band and wave)
    const uint4 cc = *reinterpret_cast<const uint4*>(cnt + (wv * NB + band) * GL_GMAX);
    if (!(GL_ABLATE & 2)) __syncthreads();  // every wave is done with the previous band
    GL_STAMP(t1);
    if (!(GL_ABLATE & 1)) {
      // stage the band: up to ten 1-KiB LDS-DMA pieces per wave (global_load_lds_dwordx4: no VGPR round trip, all in flight at once)
      const uint32_t nrow = min(GL_RB, n_src - band * GL_RB);
      const char* src = reinterpret_cast<const char*>(In) + (size_t)band * IMG;
#pragma nounroll
      for (int q = w; q < NPIECE; q += nwaves) {  // wave-uniform piece; rolled: unrolled, the address pairs stayed live beside the accumulators (spills)
          const bool hp = q >= NF * 64;  // a piece of the half plane
          const int plane = hp ? NF : q >> 6;
          const uint32_t i = (uint32_t)(hp ? q - NF * 64 : q & 63) * 64u + lane;  // float4 of the plane
          // lanes past the band's rows stay masked: they must not land on the zero rows
          if (i < (hp ? (nrow + 1) / 2 : nrow)) {
            const char* gp = src + (size_t)plane * GL_RB * 16u + (size_t)i * 16u;
            char* lp = lb + (hp ? 0u : P0 + plane * GL_PSB) + (i - lane) * 16u;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp, (__attribute__((address_space(3))) void*)lp, 16, 0, 0);
          }
      }
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's pieces have landed
    }
    GL_STAMP(t2);
    if (!(GL_ABLATE & 2)) __syncthreads();
    GL_STAMP(t3);
    uint32_t cw[4] = {(uint32_t)__builtin_amdgcn_readfirstlane(cc.x), (uint32_t)__builtin_amdgcn_readfirstlane(cc.y),
                      (uint32_t)__builtin_amdgcn_readfirstlane(cc.z), (uint32_t)__builtin_amdgcn_readfirstlane(cc.w)};
    if (!wvalid) cw[0] = cw[1] = cw[2] = cw[3] = 0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const uint32_t nf = (uint32_t)__builtin_amdgcn_readfirstlane((int)((cw[g >> 1] >> (16 * (g & 1))) & 0xffffu));
      // bit 15 (gl_place_k / gl_scale_ids_k): every lane's last two slots of the group's last super-round are padding — only its first two are read
      const bool half_last = (GL_ABLATE & 128) ? (nf & 0x7fffu) != 0u : (nf >> 15) != 0u;
      const uint32_t n = nf & 0x7fffu;  // a scalar loop count
      if ((GL_ABLATE & 256) && (band * (uint32_t)G + (uint32_t)g) % 5u == 4u) {  // timing build only: a stream with a fifth fewer slots (and ids)
        p += (size_t)64 * n;
        continue;
      }
      // two super-rounds per step: the ring moves by PAIRS (q0 = q2; q1 = q3; two loads), so the moves read entries loaded one whole step
      // = two super-rounds earlier.  (Rotating by one — q0 = q1; ... q3 = *p — the move reads the load issued the step before and every
      // step waits for it, s_waitcnt vmcnt(0) at the head of the loop: one load in flight where the ring was meant to keep four.  A C3
      // shard, pass 1 / pass 2: 0.261 / 0.381 ms by one, 0.226 / 0.316 ms by two; a flat per-band loop over a ring addressed by name with
      // the group as a switch — four in flight — 0.262 / 0.346 ms: its branches cost what the latency had.)
      uint32_t r = 0;
      for (; r + 2 <= n; r += 2) {
        const uint2 ua = q0, ub = q1;
        if (!(GL_ABLATE & 64)) {  // 64: the loaded ids are never consumed (no wait for them at all; the first super-rounds are walked again and again)
          q0 = q2;
          q1 = q3;
        }
        q2 = gl_ld_ids(p);
        if (GL_ABLATE & 32) {
          q3 = q2;
          p += 64;
        } else {
          q3 = gl_ld_ids(p + 64);
          if (!(GL_ABLATE & 8)) p += 128;
        }
        GL_ROWS(ua)
        if (half_last && r + 2 == n) {
          GL_ROWS_N(ub, 2)
        } else {
          GL_ROWS(ub)
        }
      }
      if (r < n) {
        const uint2 ua = q0;
        if (!(GL_ABLATE & 64)) {
          q0 = q1;
          q1 = q2;
          q2 = q3;
        }
        q3 = gl_ld_ids(p);
        if (!(GL_ABLATE & 8)) p += 64;
        if (half_last) {
          GL_ROWS_N(ua, 2)
        } else {
          GL_ROWS(ua)
        }
      }
#if GL_STAMPS
      st_sr += n;
#endif
    }
#if GL_STAMPS
    {
      GL_STAMP(t4);
      st_b1 += t1 - t0;
      st_stage += t2 - t1;
      st_b2 += t3 - t2;
      st_walk += t4 - t3;
    }
#endif
  }
#if GL_STAMPS
  if (wvalid && lane == 0) {
    atomicAdd(&gl_stamp_acc[0], st_b1);
    atomicAdd(&gl_stamp_acc[1], st_stage);
    atomicAdd(&gl_stamp_acc[2], st_b2);
    atomicAdd(&gl_stamp_acc[3], st_walk);
    atomicAdd(&gl_stamp_acc[4], st_sr);
    atomicAdd(&gl_stamp_acc[5], 1ull);
    atomicAdd(&gl_stamp_acc[6], (unsigned long long)(ds.b1 - ds.b0));
  }
#endif
  if (!wvalid) return;
  if ((GL_ABLATE & 64) && (q2.x ^ q3.y) == 0x12345u) acc[0][0].x += 1.f;  // keeps the loads of the ablated ring alive
  float4* out = Out + (size_t)ds.slab * slab_stride;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const uint32_t sl = slice_of[wv * G + g];
    const uint64_t pos = (uint64_t)sl * 64 + lane;
    if (sl != GL_NONE && pos < n_out) {
      if (out_n2 == GL_OUT_PLANAR) {  // pass 1 of the Gram apply writes the operand of pass 2: planar, banded by output position
        const uint32_t ob = (uint32_t)pos / GL_RB, orow = (uint32_t)pos - ob * GL_RB;
        char* o = reinterpret_cast<char*>(Out) + (size_t)ob * IMG;
#pragma unroll
        for (int l = 0; l < NF; ++l) *reinterpret_cast<float4*>(o + (size_t)l * GL_RB * 16u + (size_t)orow * 16u) = acc[g][l];
        if (HALF) *reinterpret_cast<float2*>(o + (size_t)NF * GL_RB * 16u + (size_t)orow * 8u) = acch[g];
        continue;
      }
      const size_t row = rowmap ? (size_t)rowmap[pos] : (size_t)(pos - ds.pos_base);
      if (out_n2) {  // a panel of 10-column steps inside a wider row: 8-byte aligned, exactly the panel's columns
        float2* o2 = reinterpret_cast<float2*>(out) + row * (size_t)out_ld4 * 2;
#pragma unroll
        for (int l = 0; l < NF; ++l) {
          if (2u * l < out_n2) o2[2 * l] = make_float2(acc[g][l].x, acc[g][l].y);
          if (2u * l + 1 < out_n2) o2[2 * l + 1] = make_float2(acc[g][l].z, acc[g][l].w);
        }
        if (HALF && 2u * NF < out_n2) o2[2 * NF] = acch[g];
        continue;
      }
#pragma unroll
      for (int l = 0; l < NF; ++l) out[row * out_ld4 + l] = acc[g][l];
      if (HALF) out[row * out_ld4 + NF] = make_float4(acch[g].x, acch[g].y, 0.f, 0.f);
    }
  }
}

#undef GL_ROWS
#undef GL_ROWS_N

// packed panel (planar, banded: gl_planar_off) = s_w * M[w, j0 : j0 + ncol)  (zero padded), M row-major with leading dimension ld: one panel
// of a wide operand.  One thread per (row, plane); with `half` the last plane holds two columns (float2).
__global__ __launch_bounds__(256) void gl_pack_panel_k(const float* __restrict__ M, int ld, int j0, int ncol, int LPE, int half,
                                                        const float* __restrict__ rowval, size_t n4, char* __restrict__ Xs) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const size_t w = i / LPE;
  const int l = (int)(i - w * LPE);
  const float s = rowval[w];
  const float* src = M + w * (size_t)ld + j0 + 4 * l;
  float v[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) v[t] = (4 * l + t < ncol) ? s * src[t] : 0.f;
  char* o = Xs + gl_planar_off((uint32_t)w, l, LPE, half != 0);
  if (half && l == LPE - 1) *reinterpret_cast<float2*>(o) = make_float2(v[0], v[1]);
  else *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
}

// packed panel = s_w * X[w, :] straight from the eigensolver's column-major block (V x b, leading dimension V): the packing and the
// scaling in one pass (columns >= b of the panel are zero)
__global__ __launch_bounds__(256) void gl_pack_scale_k(const float* __restrict__ Xcm, size_t V, int b, int LPE, int half,
                                                        const float* __restrict__ rowval, char* __restrict__ Xs) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // i = l * V + w: consecutive threads read consecutive rows of a column
  if (i >= V * (size_t)LPE) return;
  const int l = (int)(i / V);
  const size_t w = i - (size_t)l * V;
  const float s = rowval[w];
  float v[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) v[t] = (4 * l + t < b) ? s * Xcm[(size_t)(4 * l + t) * V + w] : 0.f;
  char* o = Xs + gl_planar_off((uint32_t)w, l, LPE, half != 0);
  if (half && l == LPE - 1) *reinterpret_cast<float2*>(o) = make_float2(v[0], v[1]);
  else *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
}

// gl_reduce_k writing the eigensolver's column-major block: Zcm[wperm[q] + j V] = s_w * sum over the slabs (fixed order), j < b
__global__ __launch_bounds__(256) void gl_reduce_cm_k(const float4* __restrict__ part, const uint32_t* __restrict__ slab0,
                                                       const uint32_t* __restrict__ nch, const uint32_t* __restrict__ wperm,
                                                       const float* __restrict__ rowval, uint32_t V, int LPE, int b, uint32_t bitems,
                                                       float* __restrict__ Zcm) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)V * LPE) return;
  const uint32_t q = (uint32_t)(i / LPE);
  const int l = (int)(i - (size_t)q * LPE);
  const uint32_t ob = q / bitems;
  const size_t stride = (size_t)bitems * LPE;
  const float4* src = part + (size_t)slab0[ob] * stride + (size_t)(q - ob * bitems) * LPE + l;
  float4 s = src[0];
  const uint32_t n = nch[ob];
  uint32_t ch = 1;
  for (; ch + 8 <= n; ch += 8) {  // eight slab loads in flight, added in slab order (a rolled loop kept one)
    float4 v8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v8[u] = src[(size_t)(ch + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) add4(s, v8[u]);
  }
  for (; ch < n; ++ch) add4(s, src[(size_t)ch * stride]);
  const uint32_t w = wperm[q];
  const float v = rowval[w];
  const float o[4] = {s.x * v, s.y * v, s.z * v, s.w * v};
#pragma unroll
  for (int t = 0; t < 4; ++t)
    if (4 * l + t < b) Zcm[(size_t)(4 * l + t) * V + w] = o[t];
}

// ---------------------------------------------------------------------------------------------------------------
// centroid update of Lloyd on B for row-constant B (lloyds_iter, src/sparseMatrix.cpp:1613-1646):
//   sum over the members d of centre c of B[w, d]  =  s_w * #{members of c that contain w}
// One workgroup per CC_CH consecutive entries of the member list (documents grouped by centre) and vocabulary part:
// word histogram in LDS (ds_add_u32, two u16 counters per dword), flushed with one global integer atomic per word
// present, once per centre the range touches.  No float atomics, no transposed copy of B; the counts are exact.
// ---------------------------------------------------------------------------------------------------------------
constexpr uint32_t CC_CH = 2048;
__global__ __launch_bounds__(GL_THREADS) void cc_hist_k(const uint32_t* __restrict__ rows, const int64_t* __restrict__ offs,
                                                         const uint32_t* __restrict__ members, const int* __restrict__ moff,
                                                         const uint32_t* __restrict__ assign, uint32_t D, uint32_t V, int ld,
                                                         uint32_t* __restrict__ cnt /* k x V, centre-major */) {
  extern __shared__ uint32_t hist[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane / GL_SUB, sl = lane % GL_SUB;
  const uint32_t w0 = blockIdx.y * GL_VP, w1 = min(V, w0 + GL_VP);
  const uint32_t nh = (w1 - w0 + 1) / 2;
  for (uint32_t j = threadIdx.x; j < nh; j += GL_THREADS) hist[j] = 0;
  __syncthreads();
  const uint32_t m0 = blockIdx.x * CC_CH, m1 = min(D, m0 + CC_CH);
  uint32_t m = m0;
  while (m < m1) {  // uniform over the workgroup
    const uint32_t cc = assign[members[m]];
    const uint32_t mend = min(m1, (uint32_t)moff[cc + 1]);  // centre cc owns members [moff[cc], moff[cc + 1])
    for (uint32_t idx = m + wave * (64 / GL_SUB) + sub; idx < mend; idx += GL_WAVES * (64 / GL_SUB)) {
      const uint32_t d = members[idx];
      const int64_t iend = offs[d + 1];
      for (int64_t i = offs[d] + sl; i < iend; i += 4 * GL_SUB) {  // four row ids in flight per lane
        uint32_t w[4];
        bool in[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int64_t iu = i + (int64_t)u * GL_SUB;
          const bool live = iu < iend;
          w[u] = rows[live ? iu : iend - 1];
          in[u] = live && w[u] >= w0 && w[u] < w1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (in[u]) atomicAdd(&hist[(w[u] - w0) >> 1], ((w[u] - w0) & 1u) ? 0x10000u : 1u);
      }
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < nh; j += GL_THREADS) {
      const uint32_t v = hist[j];
      if (v) {
        const size_t w = (size_t)w0 + 2 * j;  // (centre-major counts: a centre's words are a run — word-major, every atomic touched a line of its own)
        if (v & 0xffffu) atomicAdd(&cnt[(size_t)cc * V + w], v & 0xffffu);
        if (v >> 16) atomicAdd(&cnt[(size_t)cc * V + w + 1], v >> 16);
        hist[j] = 0;
      }
    }
    __syncthreads();
    m = mend;
  }
}
// Incremental form for the later Lloyd iterations: only documents whose centre changed touch the counts (integer counts are
// exact, so add / subtract leaves precisely what a fresh count would give).  GL_SUB lanes per document.
__global__ __launch_bounds__(256) void cc_moved_k(const uint32_t* __restrict__ rows, const int64_t* __restrict__ offs,
                                                   const uint32_t* __restrict__ assign, uint32_t* __restrict__ counted /*D: centre the counts hold*/,
                                                   uint32_t D, uint32_t V, uint32_t* __restrict__ cnt /*k x V*/) {
  const int sub = (threadIdx.x & 63) / GL_SUB, sl = threadIdx.x % GL_SUB;
  const uint32_t d = (blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / GL_SUB) + sub;
  if (d >= D) return;
  const uint32_t a = assign[d], o = counted[d];
  if (a == o) return;
  for (int64_t i = offs[d] + sl; i < offs[d + 1]; i += GL_SUB) {
    const size_t w = rows[i];
    atomicAdd(&cnt[(size_t)a * V + w], 1u);
    atomicSub(&cnt[(size_t)o * V + w], 1u);
  }
  if (sl == 0) counted[d] = a;
}
// Crm[w][c] = rowval[w] * cnt[c][w] for c < k, 0 in the padding columns: 64 x 64 tiles through LDS (the counts lie centre-major); a wave reads
// and writes 256-byte runs (round 6; 32 x 32 tiles, 128-byte runs: 321 us for 0.8 GB at k = 1000)
__global__ __launch_bounds__(256) void cc_centers_k(const uint32_t* __restrict__ cnt /*k x V*/, const float* __restrict__ rowval, uint32_t V, int k, int ld,
                                                     float* __restrict__ Crm /*V x ld*/) {
  __shared__ uint32_t t[64][65];
  const uint32_t w0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
  uint32_t v[16];  // all sixteen loads of a thread in flight
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const uint32_t cc = c0 + ty + 4 * i, w = w0 + tx;
    v[i] = (cc < (uint32_t)k && w < V) ? cnt[(size_t)cc * V + w] : 0u;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) t[ty + 4 * i][tx] = v[i];
  __syncthreads();
#pragma unroll 4
  for (int r = ty; r < 64; r += 4) {
    const uint32_t w = w0 + r, cc = c0 + tx;
    if (w < V && cc < (uint32_t)ld) Crm[(size_t)w * ld + cc] = rowval[w] * (float)t[tx][r];
  }
}

int bits_for(uint64_t n) {
  int b = 1;
  while ((1ull << b) <= n) ++b;
  return b;
}

int reserve_sort(isle_ctx* c, uint64_t n) {
  HIPCHK(c, c->gl_key_a.reserve(n));
  HIPCHK(c, c->gl_key_b.reserve(n));
  HIPCHK(c, c->gl_val_a.reserve(n));
  HIPCHK(c, c->gl_val_b.reserve(n));
  return 0;
}
// keys (maxlen - length) in gl_key_a, items in gl_val_a  ->  perm[p] = item at position p, longest first (stable)
int sort_by_key(isle_ctx* c, uint64_t n, uint64_t maxlen, uint32_t* perm) {
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
  if (nwb * 64 * GL_GMAX >= (1ull << 32))  // gl_fill1_k / gl_sort2_k take one workgroup per (wave, band): a launch of 2^32 threads does not run
    return isle_fail(c, ISLE_E_ARG, "operator build: %zu (wave, band) cells exceed one launch", nwb);
  HIPCHK(c, s.cnt.reserve(nwb * GL_GMAX));
  HIPCHK(c, s.roff.reserve(nwb + 1));
  HIPCHK(c, c->gl_srsum.reserve(nwb));
  HIPCHK(c, c->gl_scan.reserve(isle_scan::scan_scratch_elems(nwb) + 8));
  HIPCHK(c, c->gl_flag.reserve(4));
  HIPCHK(c, hipMemsetAsync(c->gl_flag.p, 0, sizeof(int), c->stream));
  hipLaunchKernelGGL((gl_cnt_k<PASS>), dim3(s.nwv), dim3(64 * s.G), 0, c->stream, s.slice_of.p, s.G, s.n_out, s.NB, c->gl_bst.p, c->gl_cellcnt.p,
                     c->wperm.p, s.cnt.p, c->gl_srsum.p, c->gl_flag.p);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, (isle_scan::exclusive_scan<uint32_t, int64_t>(c->stream, c->gl_srsum.p, nwb, s.roff.p, c->gl_scan.p)));
  int overflow = 0;
  HIPCHK(c, hipMemcpyAsync(&overflow, c->gl_flag.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&s.total_sr, s.roff.p + nwb, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (overflow) return isle_fail(c, ISLE_E_NUMERIC, "operator build: more than 65535 super-rounds in one (wave, band) cell");
  if (s.total_sr >= 0xfffffff0ll) return isle_fail(c, ISLE_E_ARG, "operator build: id stream too long (%lld super-rounds)", (long long)s.total_sr);
  HIPCHK(c, s.ids.reserve(((size_t)s.total_sr + 2 * GL_PF) * 64));
  // the prefetch ring reads up to GL_PF super-rounds past the end: that slack, and in pass 2 every slot no entry lands in,
  // holds the padding id (the zero row behind the band)
  const size_t n16_all = ((size_t)s.total_sr + 2 * GL_PF) * 64 * 4, n16_body = (size_t)s.total_sr * 64 * 4;
  uint16_t* ids16 = reinterpret_cast<uint16_t*>(s.ids.p);
  if (PASS == 1) {
    HIPCHK(c, hipMemsetD16Async((hipDeviceptr_t)(ids16 + n16_body), (unsigned short)GL_RB, n16_all - n16_body, c->stream));
    if (nwb)
      hipLaunchKernelGGL(gl_fill1_k, dim3((unsigned)(8 * cdiv((long)nwb, 8))), dim3(64 * s.G), 0, c->stream, s.slice_of.p, s.G, s.n_out, s.NB, c->gl_bst.p, c->dperm.p,
                         c->rows.p, c->offs.p, c->nnz, s.cnt.p, s.roff.p, s.ids.p, (size_t)nwb);
    HIPCHK(c, hipGetLastError());
  } else {
    HIPCHK(c, hipMemsetD16Async((hipDeviceptr_t)ids16, (unsigned short)GL_RB, n16_all, c->stream));
    HIPCHK(c, c->gl_sbase.reserve((size_t)s.nslice * s.NB));
    HIPCHK(c, c->gl_scnt.reserve((size_t)s.nslice * s.NB));
    if (nwb) hipLaunchKernelGGL(gl_sbase_k, dim3(cdiv((long)nwb, 256)), dim3(256), 0, c->stream, s.slice_of.p, s.G, s.cnt.p, s.roff.p, nwb, s.NB,
                                c->gl_sbase.p, c->gl_scnt.p);
    HIPCHK(c, hipGetLastError());
    static_assert(GL_RB <= 4096, "the packed entries of the bucketed fill hold the document in 12 bits");
    if (s.n_out <= (1u << 20) && c->nnz && !c->knob_zero(KN_GL_FILL_BUCKETS)) {  // (word positions in 20 bits)
      const uint32_t V = s.n_out;
      // buckets of about 1024 word positions, 16 ... 128 of them (W <= 8192 while V <= 2^20): at config 3 a bucket's part of a band's stream
      // region is 13 KB, and the 256 workgroups in flight on an XCD write into 3.4 MB — with 16 buckets (81 KB each) they thrashed the L2 as
      // the direct scatter does (gl_fb_fill_k 14.9 ms)
      // (W a multiple of 64: a bucket holds whole slices — gl_fb_fill_k writes a slice's super-rounds as one run)
      const uint32_t NBK0 = std::min<uint32_t>(128u, std::max<uint32_t>(16u, (V + 1023) / 1024)), W = ((V + NBK0 - 1) / NBK0 + 63) / 64 * 64, NBK = (V + W - 1) / W;
      const size_t nbb = (size_t)s.NB * NBK;
      HIPCHK(c, c->gl_fb_cnt.reserve(nbb));
      HIPCHK(c, c->gl_fb_off.reserve(nbb + 1));
      HIPCHK(c, c->gl_scan.reserve(isle_scan::scan_scratch_elems(nbb) + 8));
      HIPCHK(c, c->gl_fb_tmp.reserve(c->nnz));
      hipLaunchKernelGGL(gl_fb_count_k, dim3(s.NB), dim3(GL_THREADS), 0, c->stream, c->gl_cellcnt.p, c->wperm.p, V, W, NBK, c->gl_fb_cnt.p);
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, (isle_scan::exclusive_scan<uint32_t, int64_t>(c->stream, c->gl_fb_cnt.p, nbb, c->gl_fb_off.p, c->gl_scan.p)));
      hipLaunchKernelGGL(gl_fb_scatter_k, dim3(s.NB), dim3(GL_THREADS), 0, c->stream, c->rows.p, c->offs.p, c->dperm.p, s.n_src, c->wpos.p, W, NBK,
                         c->gl_fb_off.p, c->gl_fb_tmp.p);
      const size_t fb_head = ((size_t)W + 2 * (W / 64 + 2) + 1 + 3) / 4 * 4 * sizeof(uint32_t);  // cursors, slice bases, offsets: a multiple of 16 bytes
      const size_t fb_lds = std::max<size_t>(32u << 10, fb_head + 8 * 512);                          // 32 KB: five workgroups per CU
      ISLECHK(isle_max_lds(c, (const void*)gl_fb_fill_k, (int)fb_lds));
      hipLaunchKernelGGL(gl_fb_fill_k, dim3(s.NB, NBK), dim3(256), fb_lds, c->stream, c->gl_fb_tmp.p, c->gl_fb_off.p, V, s.NB, W, NBK, c->gl_sbase.p,
                         c->gl_scnt.p, ids16, (uint32_t)((fb_lds - fb_head) / 512));
      HIPCHK(c, hipGetLastError());
    } else {
      const uint32_t nvp = (s.n_out + GL_VP - 1) / GL_VP;
      hipLaunchKernelGGL(gl_hist_fill_k, dim3(s.NB, nvp), dim3(GL_THREADS), GL_HLDS, c->stream, c->rows.p, c->offs.p, c->dperm.p, s.n_src, s.n_out,
                         s.NB, c->wpos.p, c->gl_sbase.p, ids16);
      HIPCHK(c, hipGetLastError());
    }
    if (nwb) {
      HIPCHK(c, c->gl_biglist.reserve(nwb * GL_GMAX + 1));
      HIPCHK(c, hipMemsetAsync(c->gl_biglist.p, 0, sizeof(uint32_t), c->stream));
      hipLaunchKernelGGL(gl_sort2_k, dim3((unsigned)nwb), dim3(64 * s.G), 0, c->stream, s.NB, s.cnt.p, s.roff.p, s.ids.p, c->gl_biglist.p);
      hipLaunchKernelGGL(gl_sort2_big_k, dim3(2 * c->num_cus), dim3(128), 0, c->stream, s.NB, s.cnt.p, s.roff.p, s.ids.p, c->gl_biglist.p);
      HIPCHK(c, hipGetLastError());
    }
  }
  if (nwb) {
    const size_t nsl = nwb * (size_t)s.G;
    const bool place = !c->knob_zero(KN_GL_PLACE);
    // at most 2^22 workgroups of 512 threads per launch (a launch of 2^32 threads or more does not run, and says nothing)
    for (size_t sid0 = 0; sid0 < nsl; sid0 += (size_t)8 << 22) {
      const dim3 grid((unsigned)cdiv((long)std::min<size_t>(nsl - sid0, (size_t)8 << 22), 8));
      if (place) {
        hipLaunchKernelGGL((gl_place_k<4, 0>), grid, dim3(512), 0, c->stream, s.NB, s.G, sid0, nsl, s.cnt.p, s.roff.p, s.ids.p);
        hipLaunchKernelGGL((gl_place_k<GL_PLACE_MAXN, 4>), grid, dim3(512), 0, c->stream, s.NB, s.G, sid0, nsl, s.cnt.p, s.roff.p, s.ids.p);
      }
      hipLaunchKernelGGL(gl_scale_ids_k, grid, dim3(512), 0, c->stream, s.G, sid0, nsl, place ? (uint32_t)GL_PLACE_MAXN : 0u, s.cnt.p, s.roff.p, s.ids.p);
      HIPCHK(c, hipGetLastError());
    }
  }
  // the prefetch slack behind the stream: the padding id in the stream's final form
  HIPCHK(c, hipMemsetD16Async((hipDeviceptr_t)(ids16 + n16_body), (unsigned short)(GL_RB * 8), n16_all - n16_body, c->stream));
  return 0;
}

template <int LPE, bool HALF, int G>
int launch_apply_g(isle_ctx* c, const GlSide& s, const float4* In, float4* Out, size_t slab_stride, const uint32_t* rowmap, uint32_t out_ld4,
                   uint32_t out_n2) {
  // LDS of this panel width: its planes (+ zero rows) only
  constexpr uint32_t lds = (uint32_t)(HALF ? LPE - 1 : LPE) * GL_PSB + (HALF ? GL_PS * 8u : 0u);
  ISLECHK(isle_max_lds(c, (const void*)gl_apply_k<LPE, HALF, G>, (int)lds));
  hipLaunchKernelGGL((gl_apply_k<LPE, HALF, G>), dim3(s.ndesc), dim3(64 * s.wpg), lds, c->stream, In, s.n_src, s.ids.p, s.roff.p, s.cnt.p,
                     s.slice_of.p, s.desc.p, s.NB, Out, slab_stride, s.n_out, rowmap, out_ld4 ? out_ld4 : (uint32_t)LPE, out_n2);
  HIPCHK(c, hipGetLastError());
  return 0;
}
template <int LPE, bool HALF>
int launch_apply(isle_ctx* c, const GlSide& s, const float4* In, float4* Out, size_t slab_stride, const uint32_t* rowmap = nullptr,
                 uint32_t out_ld4 = 0, uint32_t out_n2 = 0) {
  switch (s.G) {
    case 4: return launch_apply_g<LPE, HALF, 4>(c, s, In, Out, slab_stride, rowmap, out_ld4, out_n2);
    case 5: return launch_apply_g<LPE, HALF, 5>(c, s, In, Out, slab_stride, rowmap, out_ld4, out_n2);
    case 6: return launch_apply_g<LPE, HALF, 6>(c, s, In, Out, slab_stride, rowmap, out_ld4, out_n2);
    case 7: return launch_apply_g<LPE, HALF, 7>(c, s, In, Out, slab_stride, rowmap, out_ld4, out_n2);
    case 8: return launch_apply_g<LPE, HALF, 8>(c, s, In, Out, slab_stride, rowmap, out_ld4, out_n2);
  }
  return isle_fail(c, ISLE_E_ARG, "LDS Gram apply: %d items per lane", s.G);
}
// (twelve whole columns — three full planes — do not fit the LDS: the widest panel is <3, true>, ten columns)
int launch_apply_any(isle_ctx* c, int LPE, bool half, const GlSide& s, const float4* In, float4* Out, size_t slab_stride, uint32_t out_n2) {
  if (LPE == 1) return half ? launch_apply<1, true>(c, s, In, Out, slab_stride, nullptr, 0, out_n2) : launch_apply<1, false>(c, s, In, Out, slab_stride, nullptr, 0, out_n2);
  if (LPE == 2) return half ? launch_apply<2, true>(c, s, In, Out, slab_stride, nullptr, 0, out_n2) : launch_apply<2, false>(c, s, In, Out, slab_stride, nullptr, 0, out_n2);
  if (LPE == 3 && half) return launch_apply<3, true>(c, s, In, Out, slab_stride, nullptr, 0, out_n2);
  return isle_fail(c, ISLE_E_ARG, "LDS Gram apply: a panel of %d columns", 4 * LPE);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
int k_gl_detect(isle_ctx* c) {
  if (c->gl_mode >= 0) return 0;  // decided for this B (reset by every upload / thresholding)
  c->gl_mode = 0;
  const char* e = c->knob(KN_GRAM_LDS);
  // every rank takes part in the agreement below, whatever its own shard looks like
  bool eligible = !(e && atoi(e) == 0) && c->nnz > 0 && c->D > 0 && c->V > 0 && c->D < 0xfffffff0ull && c->V < 0xfffffff0ull;
  HIPCHK(c, c->gl_flag.reserve(4));
  HIPCHK(c, hipMemsetAsync(c->gl_flag.p, 0, sizeof(int), c->stream));
  if (eligible) {
    HIPCHK(c, c->rowval.reserve(c->V));
    HIPCHK(c, hipMemsetAsync(c->rowval.p, 0, c->V * sizeof(float), c->stream));
    const dim3 g(cdiv((long)c->nnz, 256)), b(256);
    hipLaunchKernelGGL(gl_rowval_set_k, g, b, 0, c->stream, c->vals.p, c->rows.p, c->nnz, c->rowval.p);
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(gl_rowval_chk_k, g, b, 0, c->stream, c->vals.p, c->rows.p, c->nnz, c->rowval.p, c->gl_flag.p);
    HIPCHK(c, hipGetLastError());
  } else {
    const int one = 1;
    HIPCHK(c, hipMemcpyAsync(c->gl_flag.p, &one, sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  if (c->multi()) {  // one form on all ranks (the collectives per application must match)
    TimeScope ts(c, ISLE_T_COMM);
    ISLECHK(isle_allreduce(c, c->gl_flag.p, 1, ISLE_DT_I32, true));
  }
  int flag = 0;
  HIPCHK(c, hipMemcpyAsync(&flag, c->gl_flag.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->gl_mode = flag ? 0 : 1;
  return 0;
}

int k_gl_ablate_skip(isle_ctx* c);
int k_gl_build(isle_ctx* c) {
  const uint32_t D = (uint32_t)c->D, V = (uint32_t)c->V;
  GlSide& s1 = c->gl1;
  GlSide& s2 = c->gl2;
  ISLECHK(isle_max_lds(c, (const void*)gl_hist_count_k, GL_HLDS));
  ISLECHK(isle_max_lds(c, (const void*)gl_hist_fill_k, GL_HLDS));
  // ---- documents by decreasing length
  HIPCHK(c, c->dperm.reserve(D));
  HIPCHK(c, c->dpos.reserve(D));
  ISLECHK(reserve_sort(c, std::max(D, V)));
  hipLaunchKernelGGL(gl_len_key_k, dim3(cdiv(D, 256)), dim3(256), 0, c->stream, c->offs.p, (uint64_t)D, (uint64_t)1, (uint64_t)V, c->gl_key_a.p,
                     c->gl_val_a.p);
  HIPCHK(c, hipGetLastError());
  ISLECHK(sort_by_key(c, D, V, c->dperm.p));
  hipLaunchKernelGGL(gl_invert_k, dim3(cdiv(D, 256)), dim3(256), 0, c->stream, c->dperm.p, (uint64_t)D, c->dpos.p);
  HIPCHK(c, hipGetLastError());

  // ---- pass 1: outputs = documents (position order), sources = words
  s1.n_out = D;
  s1.n_src = V;
  s1.NB = (V + GL_RB - 1) / GL_RB;
  s1.nslice = (D + 63) / 64;
  // items per lane G and waves per workgroup: the makespan model  rounds x (waves x G x LDS time of a slice + staging of all bands)
  // over G = 4..8 and 1..16 waves.  Every workgroup stages every word band, so a shard whose documents need more than one round of
  // workgroups at G = 4 (a C3 shard: 306 workgroups on 256 CUs, 0.44 ms per pass) runs one fuller round at G = 5 (245 workgroups,
  // 0.31 ms), and all of config 3 on one GPU five rounds at G = 8 instead of ten (2.69 -> 2.34 ms); small matrices keep G = 4 and
  // take fewer waves per workgroup so that all CUs stay busy.  The model's figures against the measured ones, pass 1 with 10 columns:
  // C3 shard G = 5: 312 / 313 us; config 3 on one GPU G = 4 / 6 / 8: 2610 / 2530 / 2310 against 2690 / 2490 / 2336 us.
  // ISLE_GL_G1 = 4..8 forces G.
  const uint32_t maxw = GL_APPLY_WAVES;
  uint32_t wpw = maxw;
  const uint32_t cus = c->knob(KN_GL_TEST_CUS) ? (uint32_t)std::max(1, atoi(c->knob(KN_GL_TEST_CUS))) : (uint32_t)c->num_cus;  // test hook: the geometry of a larger problem on a small one
  {
    const double t_slice = 0.05 * 2.5 * (double)c->nnz / 256.0 / std::max<uint32_t>(1u, s1.nslice);  // us: ~50 ns per super-round, ~2.5x padded
    const double t_stage = 2.0 * s1.NB;                                                               // us: ~2 us per band from L2
    const char* e_g = c->knob(KN_GL_G1);
    const int g_lo = e_g ? std::max(4, std::min(GL_GMAX, atoi(e_g))) : 4, g_hi = e_g ? g_lo : GL_GMAX - 1;  // 8: the 10-column kernel spills there
    double best = 1e300;
    for (int G = g_lo; G <= g_hi; ++G) {
      const uint32_t nwv = (s1.nslice + G - 1) / G;
      for (uint32_t cand = maxw; cand >= 1; --cand) {
        const uint32_t wgs = (nwv + cand - 1) / cand;
        const double fewer = 1.0 + 0.25 * (double)(GL_WAVES - cand) / GL_WAVES;  // fewer waves hide less of the id-stream latency
        const uint32_t rounds = (wgs + cus - 1) / cus;
        // more than one round: the workgroups are spread over WHOLE rounds (below), a wave then owns nslice / (rounds x CUs x waves) slices
        const double per_wave = rounds > 1 ? (double)s1.nslice / ((double)rounds * cus * cand) : (double)G;
        const double cost = (double)rounds * (cand * per_wave * t_slice * fewer + t_stage);
        if (cost < best * 0.97) {  // prefer fewer items per lane and more waves per workgroup unless clearly worse
          best = cost;
          wpw = cand;
          s1.G = G;
        }
      }
    }
  }
  s1.nwv = (s1.nslice + s1.G - 1) / s1.G;
  s1.wpg = wpw;
  // More than one round of workgroups: their number is rounded up to whole rounds of the CUs — the waves of the last quantile range
  // then own one slice less — and a workgroup takes ADJACENT waves (slices of neighbouring lengths: its waves reach the barrier of a band
  // together; the workgroups differ by the length of their documents, the longest are launched first).  5.45 rounds of 7 slices per wave
  // ran like 6 (all of config 3 on one GPU, pass 1); 6 rounds of 6.36 do the same work with the CUs busy to the end.  One round (a C3
  // shard: 244 workgroups): the workgroups must finish together, so they take waves strided over the whole length order as before.
  bool adjacent = false;
  {
    const uint32_t nwg0 = (s1.nwv + wpw - 1) / wpw;
    if (nwg0 > cus && !c->knob_zero(KN_GL_ROUNDS)) {
      const uint32_t nwg_r = (nwg0 + cus - 1) / cus * cus;
      s1.nwv = nwg_r * wpw;
      adjacent = true;
    }
  }
  {
    // serpentine over G quantile ranges of the length-ordered slices: every wave gets long, middle and short slices alike
    const int G = s1.G;
    std::vector<uint32_t> so((size_t)s1.nwv * G);
    const uint64_t n = s1.nwv;
    for (uint64_t wv = 0; wv < n; ++wv)
      for (int g = 0; g < G; ++g) {
        const uint64_t cand = (g & 1) ? (uint64_t)(g + 1) * n - 1 - wv : (uint64_t)g * n + wv;
        so[wv * G + g] = cand < s1.nslice ? (uint32_t)cand : GL_NONE;
      }
    HIPCHK(c, c->gl_bst.reserve((size_t)D * (s1.NB + 1)));
    const uint64_t nb = (uint64_t)D * (s1.NB + 1);
    hipLaunchKernelGGL(gl_bst_k, dim3(cdiv((long)nb, 256)), dim3(256), 0, c->stream, c->rows.p, c->offs.p, c->dperm.p, (uint64_t)D, s1.NB,
                       c->gl_bst.p);
    HIPCHK(c, hipGetLastError());
    ISLECHK(build_side<1>(c, s1, so));
    // workgroup j = waves j, j + nwg, j + 2 nwg, ... : equal totals, one slab (Y itself)
    const uint32_t nwg = (s1.nwv + wpw - 1) / wpw;
    std::vector<GlDesc> ds(nwg);
    for (uint32_t j = 0; j < nwg; ++j) {
      uint32_t nw = 0;
      while (nw < wpw && (uint64_t)j + (uint64_t)nw * nwg < s1.nwv) ++nw;
      ds[j] = adjacent ? GlDesc{j * wpw, 1u, (uint32_t)std::min<uint64_t>(wpw, s1.nwv - (uint64_t)j * wpw), 0u, s1.NB, 0u, 0u, 0u} : GlDesc{j, nwg, nw, 0u, s1.NB, 0u, 0u, 0u};
    }
    s1.ndesc = nwg;
    HIPCHK(c, s1.desc.reserve(nwg));
    HIPCHK(c, hipMemcpyAsync(s1.desc.p, ds.data(), ds.size() * sizeof(GlDesc), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // ds / so are stack-owned
  }

  // ---- (word, document band) cell sizes: one LDS histogram per band and vocabulary part
  s2.n_out = V;
  s2.n_src = D;
  s2.NB = (D + GL_RB - 1) / GL_RB;
  const size_t ncell = (size_t)V * s2.NB;
  const uint32_t nvp = (V + GL_VP - 1) / GL_VP;
  HIPCHK(c, c->gl_cellcnt.reserve(ncell));
  hipLaunchKernelGGL(gl_hist_count_k, dim3(s2.NB, nvp), dim3(GL_THREADS), GL_HLDS, c->stream, c->rows.p, c->offs.p, c->dperm.p, D, V, s2.NB,
                     c->gl_cellcnt.p);
  HIPCHK(c, hipGetLastError());

  // ---- words by decreasing row length
  HIPCHK(c, c->wperm.reserve(V));
  HIPCHK(c, c->wpos.reserve(V));
  hipLaunchKernelGGL(gl_rowlen_key_k, dim3(cdiv(V, 256)), dim3(256), 0, c->stream, c->gl_cellcnt.p, V, s2.NB, (uint64_t)D, c->gl_key_a.p,
                     c->gl_val_a.p);
  HIPCHK(c, hipGetLastError());
  ISLECHK(sort_by_key(c, V, D, c->wperm.p));
  hipLaunchKernelGGL(gl_invert_k, dim3(cdiv(V, 256)), dim3(256), 0, c->stream, c->wperm.p, (uint64_t)V, c->wpos.p);
  HIPCHK(c, hipGetLastError());

  // ---- pass 2: outputs = words (position order), sources = documents (position order)
  s2.nslice = (V + 63) / 64;
  // a word block = wpb waves = 4 wpb consecutive slices; 16 waves unless the vocabulary is so small that blocks x bands would
  // leave CUs idle
  {
    const char* e_g = c->knob(KN_GL_G2);  // items per lane in pass 2 (4 ... 8)
    // 4; 6 beyond 1024 document bands (more than 4 M documents): fewer word blocks, and every block stages every band of its columns.
    // Measured with the final kernels, pass 2: config 3 on one GPU (2452 bands) 2.41 / 2.40 / 2.26 / 3.07 ms at 4 / 5 / 6 / 8; a C3 shard
    // (307 bands) 0.309 / 0.334 / 0.314 at 4 / 5 / 6
    s2.G = e_g ? std::max(4, std::min(GL_GMAX, atoi(e_g))) : (s2.NB > 1024 ? 6 : 4);
  }
  const uint32_t G2 = (uint32_t)s2.G;
  uint32_t wpb = maxw;
  while (wpb > 1 && (uint64_t)((s2.nslice + G2 * wpb - 1) / (G2 * wpb)) * s2.NB < 2ull * c->num_cus) wpb /= 2;
  const uint32_t bslices = G2 * wpb, bitems = 64 * bslices;
  const uint32_t nblk = (s2.nslice + bslices - 1) / bslices;
  s2.nwv = nblk * wpb;
  s2.wpg = wpb;
  c->gl_block_items = bitems;
  {
    // serpentine inside the block keeps its waves level
    std::vector<uint32_t> so((size_t)s2.nwv * G2);
    for (uint32_t ob = 0; ob < nblk; ++ob)
      for (uint32_t w = 0; w < wpb; ++w)
        for (uint32_t g = 0; g < G2; ++g) {
          const uint32_t cand = (g & 1u) ? (g + 1) * wpb - 1 - w : g * wpb + w;
          const uint64_t sl = (uint64_t)ob * bslices + cand;
          so[((size_t)ob * wpb + w) * G2 + g] = sl < s2.nslice ? (uint32_t)sl : GL_NONE;
        }
    ISLECHK(build_side<2>(c, s2, so));
    const double bc = GL_BAND_COST, wgs_per_cu = 2.0;  // measured by sweeps at C2
    std::vector<uint32_t> slab0(nblk), nch(nblk, 0);
    std::vector<GlDesc> ds;
    uint32_t nslab = 0;
    const char* e_col = c->knob(KN_GL_COLUMNS);
    const bool columns = !(e_col && atoi(e_col) == 0) && s2.NB >= 16;  // ISLE_GL_COLUMNS=0: per-block chunking only
    std::vector<unsigned long long> tot;
    if (columns) {
      // Band columns shared through L2.  Every word block walks every document band, so Y (48 MB at C2) is staged from HBM once
      // per word block (13 x 48 MB = 0.62 GB of the 1.2 GB pass 2 moves; 25 x 60 MB = 1.5 of 2.6 GB at a C3 shard, where pass 2 runs
      // at the HBM rate).  Here the document bands are cut into NC "columns" of equal total cost — at most GL_COL_BANDS bands, 1.3 MB
      // of Y — the SAME cut for all word blocks, and all workgroups of a column are queued back to back on ONE XCD (workgroup i
      // runs on XCD i % 8), so the column's bands are fetched from HBM once and found in that XCD's 4 MB L2 by the other word
      // blocks (measured with one workgroup per (block, column): FETCH_SIZE of pass 2 1.11 -> 0.66 GB).  Inside a column a word
      // block is cut into sub-chunks of about the same cost as everybody else's (the word blocks differ 5x in cost; with whole
      // columns per workgroup the heavy ones set the makespan: 0.301 -> 0.349 ms at C2).
      const uint32_t NB = s2.NB;
      HIPCHK(c, c->gl_blocktot.reserve((size_t)nblk * NB));
      hipLaunchKernelGGL(gl_blocktot_k, dim3(nblk, NB), dim3(256), 0, c->stream, c->gl_srsum.p, s2.nwv, wpb, NB, NB, c->gl_blocktot.p);
      HIPCHK(c, hipGetLastError());
      tot.resize((size_t)nblk * NB);
      HIPCHK(c, hipMemcpyAsync(tot.data(), c->gl_blocktot.p, tot.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      std::vector<double> bcost(NB, 0.0);
      double all = 0;
      for (uint32_t bnd = 0; bnd < NB; ++bnd) {
        for (uint32_t ob = 0; ob < nblk; ++ob) bcost[bnd] += (double)tot[(size_t)ob * NB + bnd] + bc;
        all += bcost[bnd];
      }
      // <= 12 bands (1.9 MB of Y) per column: measured at C2 / C3 shard, pass 2 in ms — per-block chunks 0.291 / 0.474, columns of
      // <= 8 bands 0.294 / 0.459, <= 12 bands 0.283 / 0.436, <= 16 bands 0.282 / 0.464
      // with the planar bands and the final kernels (round 3), all of config 3 on one GPU (2452 bands): 8 / 12 / 16 / 24 / 32 / 48 / 64 bands
      // 2.42 / 2.28 / 2.20 / 2.15 / 2.16 / 2.26 / 2.43 ms; a C3 shard and C2 do not depend on it (their column count is set by the other term)
      const uint32_t colbands = 24u;
      const double W = wgs_per_cu * c->num_cus;
      uint32_t NC = 8u * (uint32_t)std::ceil(std::max(W / (1.25 * nblk), (double)NB / colbands) / 8.0);
      NC = std::max(8u, std::min(NC, (NB / 8u) * 8u));
      std::vector<uint32_t> cut(NC + 1, 0);  // column cc = bands [cut[cc], cut[cc + 1])
      {
        double pre = 0;
        uint32_t cc = 1;
        for (uint32_t bnd = 0; bnd < NB && cc < NC; ++bnd) {
          pre += bcost[bnd];
          // close column cc - 1 behind this band once its share of the cost is reached, leaving at least one band per later column
          while (cc < NC && (pre >= all * cc / NC || NB - (bnd + 1) <= NC - cc) && bnd + 1 > cut[cc - 1]) cut[cc++] = bnd + 1;
        }
        while (cc <= NC) cut[cc++] = NB;
        cut[NC] = NB;
      }
      for (uint32_t cc = 0; cc < NC; ++cc)
        if (cut[cc + 1] <= cut[cc]) return isle_fail(c, ISLE_E_NUMERIC, "operator build: empty band column");
      const double target = std::max(1.0, all / W);
      // sub-chunks per (block, column), then the blocks' slab ranges, then the descriptors in XCD queues
      std::vector<uint32_t> nsub((size_t)nblk * NC);
      for (uint32_t ob = 0; ob < nblk; ++ob) {
        slab0[ob] = nslab;
        for (uint32_t cc = 0; cc < NC; ++cc) {
          double cost = 0;
          for (uint32_t bnd = cut[cc]; bnd < cut[cc + 1]; ++bnd) cost += (double)tot[(size_t)ob * NB + bnd] + bc;
          const uint32_t nb = cut[cc + 1] - cut[cc];
          const uint32_t n = (uint32_t)std::min<double>((double)nb, std::max(1.0, std::floor(cost / target + 0.5)));
          nsub[(size_t)ob * NC + cc] = n;
          nch[ob] += n;
          nslab += n;
        }
      }
      std::vector<std::vector<GlDesc>> xq(8);
      std::vector<std::pair<double, uint32_t>> order(nblk);
      std::vector<uint32_t> next(nblk);
      for (uint32_t ob = 0; ob < nblk; ++ob) next[ob] = slab0[ob];
      for (uint32_t cc = 0; cc < NC; ++cc) {
        const uint32_t b0 = cut[cc], b1 = cut[cc + 1], nb = b1 - b0;
        for (uint32_t ob = 0; ob < nblk; ++ob) {
          double t = 0;
          for (uint32_t bnd = b0; bnd < b1; ++bnd) t += (double)tot[(size_t)ob * NB + bnd];
          order[ob] = {-t / nsub[(size_t)ob * NC + cc], ob};
        }
        std::sort(order.begin(), order.end());  // heaviest workgroups of the column first
        for (auto& o : order) {
          const uint32_t ob = o.second, n = nsub[(size_t)ob * NC + cc];
          for (uint32_t q = 0; q < n; ++q)
            xq[cc % 8].push_back(GlDesc{ob * wpb, 1u, wpb, b0 + (uint32_t)((uint64_t)q * nb / n), b0 + (uint32_t)((uint64_t)(q + 1) * nb / n), next[ob]++,
                                        ob * bitems, 0u});
        }
      }
      size_t qmax = 0;
      for (auto& q : xq) qmax = std::max(qmax, q.size());
      ds.reserve(qmax * 8);
      for (size_t j = 0; j < qmax; ++j)
        for (uint32_t x = 0; x < 8; ++x) ds.push_back(j < xq[x].size() ? xq[x][j] : GlDesc{0u, 1u, 0u, 0u, 0u, 0u, 0u, 0u});  // empty: no wave is valid
    } else {
      // per word block: band chunks in proportion to the block's cost (round 1's form; small matrices, or ISLE_GL_COLUMNS=0).
      // Cost = super-rounds (LDS-bound, ~50 ns of CU time each) + GL_BAND_COST per band staged; without the second term a block
      // of rare words would walk all bands in a single workgroup
      HIPCHK(c, c->gl_blocktot.reserve((size_t)nblk));
      hipLaunchKernelGGL(gl_blocktot_k, dim3(nblk, 1), dim3(256), 0, c->stream, c->gl_srsum.p, s2.nwv, wpb, s2.NB, 1u, c->gl_blocktot.p);
      HIPCHK(c, hipGetLastError());
      tot.resize(nblk);
      HIPCHK(c, hipMemcpyAsync(tot.data(), c->gl_blocktot.p, tot.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      double all = 0;
      for (auto t : tot) all += (double)t;
      all += bc * (double)s2.NB * nblk;
      const double target = std::max(1.0, all / (wgs_per_cu * c->num_cus));
      for (uint32_t ob = 0; ob < nblk; ++ob) {
        slab0[ob] = nslab;
        const uint32_t nzb = s2.NB;
        const double cost = (double)tot[ob] + bc * nzb;
        const uint32_t n = (uint32_t)std::min<double>((double)nzb, std::max(1.0, std::ceil(cost / target)));
        for (uint32_t ch = 0; ch < n; ++ch) {
          const uint32_t b0 = (uint32_t)((uint64_t)ch * nzb / n), b1 = (uint32_t)((uint64_t)(ch + 1) * nzb / n);
          ds.push_back(GlDesc{ob * wpb, 1u, wpb, b0, b1, nslab, ob * bitems, 0u});
          ++nslab;
          ++nch[ob];
        }
      }
    }
    s2.ndesc = (uint32_t)ds.size();
    HIPCHK(c, s2.desc.reserve(ds.size()));
    HIPCHK(c, c->gl_slab0.reserve(nblk));
    HIPCHK(c, c->gl_nch.reserve(nblk));
    HIPCHK(c, hipMemcpyAsync(s2.desc.p, ds.data(), ds.size() * sizeof(GlDesc), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->gl_slab0.p, slab0.data(), nblk * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->gl_nch.p, nch.data(), nblk * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, c->gl_part.reserve((size_t)nslab * bitems * 12));  // sized for the widest panel (BP = 12)
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->knob_on(KN_GL_VERBOSE)) {
      fprintf(stderr, "[gram_lds] pass-2 %s, word blocks (super-rounds, chunks):", columns ? "band columns shared per XCD" : "per-block band chunks");
      const size_t per = columns ? s2.NB : 1;
      for (uint32_t ob = 0; ob < nblk; ++ob) {
        unsigned long long t = 0;
        for (size_t z = 0; z < per; ++z) t += tot[(size_t)ob * per + z];
        fprintf(stderr, " %llu/%u", t, nch[ob]);
      }
      fprintf(stderr, "\n");
    }
    if (c->knob_on(KN_GL_VERBOSE)) {  // how many (wave, band, group) slices end in a half super-round, and the slices' lengths
      for (const GlSide* sd : {&s1, &s2}) {
        const size_t ncnt = (size_t)sd->nwv * sd->NB * GL_GMAX;
        std::vector<uint16_t> h(ncnt);
        HIPCHK(c, hipMemcpy(h.data(), sd->cnt.p, ncnt * sizeof(uint16_t), hipMemcpyDeviceToHost));
        unsigned long long nsl = 0, nhalf = 0, sr = 0, hist[6] = {0, 0, 0, 0, 0, 0};
        for (size_t i = 0; i < ncnt; ++i) {
          const uint32_t n = h[i] & 0x7fffu;
          if (!n) continue;
          ++nsl;
          nhalf += h[i] >> 15;
          sr += n;
          ++hist[n < 5 ? n : 5];
        }
        fprintf(stderr, "[gram_lds] pass %d: %llu slices, %llu super-rounds, %.1f %% end in a half one; slices of 1/2/3/4/5+ super-rounds: %.1f %.1f %.1f %.1f %.1f %%\n",
                sd == &s1 ? 1 : 2, nsl, sr, 100.0 * nhalf / std::max(1ull, nsl), 100.0 * hist[1] / std::max(1ull, nsl), 100.0 * hist[2] / std::max(1ull, nsl),
                100.0 * hist[3] / std::max(1ull, nsl), 100.0 * hist[4] / std::max(1ull, nsl), 100.0 * hist[5] / std::max(1ull, nsl));
      }
    }
    if (c->knob_on(KN_GL_VERBOSE))
      fprintf(stderr,
              "[gram_lds] V=%u D=%u nnz=%llu | pass1: bands=%u items/lane=%d waves=%u wgs=%u padded=%.2fx | pass2: bands=%u items/lane=%d blocks=%u wgs=%u slabs=%u "
              "padded=%.2fx\n",
              V, D, (unsigned long long)c->nnz, s1.NB, s1.G, s1.nwv, s1.ndesc, (double)s1.total_sr * 256.0 / (double)c->nnz, s2.NB, s2.G, nblk, s2.ndesc,
              nslab, (double)s2.total_sr * 256.0 / (double)c->nnz);
  }
  return k_gl_ablate_skip(c);
}

// TIMING EXPERIMENT (ISLE_GL_ABLATE_SKIP=m, round 6): the count records of every m-th (band, group) of every wave are zeroed behind the
// build, so that the unchanged apply kernel walks (1 - 1/m) of its super-rounds and reads as much of its id stream (the waves then read
// the ids of other cells: valid rows, wrong sums — Z is wrong by construction).  Prices a form with that many fewer padded slots at no
// other cost; the in-kernel variant of the same cut (GL_ABLATE 256) moves the register allocation and spills.
__global__ __launch_bounds__(256) void gl_ablate_cnt_k(uint16_t* __restrict__ cnt, size_t n, uint32_t NB, int G, uint32_t m) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t g = (uint32_t)(i % GL_GMAX), band = (uint32_t)((i / GL_GMAX) % NB);
  if ((int)g < G && (band * (uint32_t)G + g) % m == m - 1) cnt[i] = 0;
}
int k_gl_ablate_skip(isle_ctx* c) {
  const char* e = c->knob(KN_GL_ABLATE_SKIP);
  const int m = e ? atoi(e) : 0;
  if (m < 2) return 0;
  for (GlSide* sd : {&c->gl1, &c->gl2}) {
    const size_t n = (size_t)sd->nwv * sd->NB * GL_GMAX;
    hipLaunchKernelGGL(gl_ablate_cnt_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, sd->cnt.p, n, sd->NB, sd->G, (uint32_t)m);
    HIPCHK(c, hipGetLastError());
  }
  fprintf(stderr, "[gram_lds] ISLE_GL_ABLATE_SKIP=%d: timing experiment, every %d-th (band, group) of a wave is not walked — results are WRONG\n", m, m);
  return 0;
}

// floats of a packed operand of n rows (whole bands, + slack for the last float4 of a half plane)
static size_t gl_packed_floats(size_t n_rows, int LPE, bool half) {
  return ((n_rows + GL_RB - 1) / GL_RB) * (size_t)(gl_img_bytes(LPE, half) / 4) + 8;
}

// Zcm (V x b column-major) = B (B^T Xcm): the operator application on the eigensolver's own layout, b <= 10 columns in a panel of
// BP = 4, 8 or 12 (the packing of X is fused with its scaling, the slab reduction writes the column-major block)
#if GL_STAMPS
static void gl_stamps_report(isle_ctx* c, const char* what) {
  unsigned long long h[8], z[8] = {};
  (void)hipStreamSynchronize(c->stream);
  (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(gl_stamp_acc), sizeof h);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(gl_stamp_acc), z, sizeof z);
  const double w = h[5] ? (double)h[5] : 1.0, tot = (double)(h[0] + h[1] + h[2] + h[3]);
  // s_memtime counts shader cycles (MI355X_MICROARCH.md, cycle constants): figures are kilocycles per wave
  fprintf(stderr, "[gl stamps] %s: %llu waves, %.1f bands and %.0f super-rounds per wave; per wave %.1f kcycles = end-of-band barrier %.1f (%.0f %%) + own staging %.1f (%.0f %%) + "
                  "barrier behind the staging %.1f (%.0f %%) + walking the super-rounds %.1f (%.0f %%: %.0f cycles per super-round)\n",
          what, h[5], h[6] / w, h[4] / w, tot / w * 1e-3, h[0] / w * 1e-3, 100.0 * h[0] / tot, h[1] / w * 1e-3, 100.0 * h[1] / tot, h[2] / w * 1e-3, 100.0 * h[2] / tot,
          h[3] / w * 1e-3, 100.0 * h[3] / tot, h[4] ? (double)h[3] / h[4] : 0.0);
}
#endif

int k_gl_apply_cm(isle_ctx* c, const float* Xcm, int b, int BP, float* Zcm) {
  const int LPE = BP / 4;
  if (LPE < 1 || LPE > 3 || b > BP || b <= BP - 4 || b > 10) return isle_fail(c, ISLE_E_ARG, "LDS Gram apply: b = %d in a panel of %d", b, BP);
  const bool half = b <= BP - 2;
  const uint32_t V = (uint32_t)c->V;
  const size_t nx4 = (size_t)V * LPE;
  HIPCHK(c, c->gl_Xs.reserve(gl_packed_floats(V, LPE, half)));
  HIPCHK(c, c->Yrm.reserve(gl_packed_floats(c->D, LPE, half)));
  // isle_hip_timing_enable(ctx, 2) — the bench's roofline figure, taken inside the timed region: ONE event pair around the whole
  // application (booked under pass 1) instead of one per pass; every event record sits in the queue between two kernels for ~5 us
  const bool one_pair = c->timing && c->timing_mask == ((1u << ISLE_T_GRAM_PASS1) | (1u << ISLE_T_GRAM_PASS2));
  TimeScope whole(c, one_pair ? ISLE_T_GRAM_PASS1 : -1);
  {
    TimeScope ts(c, one_pair ? -1 : ISLE_T_GRAM_PASS1);
    hipLaunchKernelGGL(gl_pack_scale_k, dim3(cdiv((long)nx4, 256)), dim3(256), 0, c->stream, Xcm, (size_t)V, b, LPE, (int)half, c->rowval.p,
                       (char*)c->gl_Xs.p);
    HIPCHK(c, hipGetLastError());
    // Y = B^T X, written as the packed operand of pass 2 (banded by document position)
    ISLECHK(launch_apply_any(c, LPE, half, c->gl1, (const float4*)c->gl_Xs.p, (float4*)c->Yrm.p, 0, GL_OUT_PLANAR));
#if GL_STAMPS
    gl_stamps_report(c, "pass 1");
#endif
  }
  {
    TimeScope ts(c, one_pair ? -1 : ISLE_T_GRAM_PASS2);
    ISLECHK(launch_apply_any(c, LPE, half, c->gl2, (const float4*)c->Yrm.p, (float4*)c->gl_part.p, (size_t)c->gl_block_items * LPE, 0));
#if GL_STAMPS
    gl_stamps_report(c, "pass 2");
#endif
    hipLaunchKernelGGL(gl_reduce_cm_k, dim3(cdiv((long)nx4, 256)), dim3(256), 0, c->stream, (const float4*)c->gl_part.p, c->gl_slab0.p,
                       c->gl_nch.p, c->wperm.p, c->rowval.p, V, LPE, b, c->gl_block_items, Zcm);
    HIPCHK(c, hipGetLastError());
  }
  return 0;
}

// Crm (V x ld row-major) = per-centre sums of the member columns (not yet divided by the cluster sizes).
// fresh: count from scratch — c->members / c->moff must hold the documents grouped by `assign` (k_member_lists);
// otherwise the counts of the previous call (same k, ld, B) are updated by the documents that changed centre.
int k_centers_counts(isle_ctx* c, const uint32_t* assign, int k, int ld, float* Crm, bool fresh) {
  TimeScope ts(c, ISLE_T_SPARSE_UPDATE);
  const uint32_t D = (uint32_t)c->D, V = (uint32_t)c->V;
  ISLECHK(isle_max_lds(c, (const void*)cc_hist_k, GL_HLDS));
  const size_t n = (size_t)V * k;
  HIPCHK(c, c->ccount.reserve(n));
  HIPCHK(c, c->ccounted.reserve(D ? D : 1));
  if (fresh) {
    HIPCHK(c, hipMemsetAsync(c->ccount.p, 0, n * sizeof(uint32_t), c->stream));
    if (D) {
      const uint32_t nvp = (V + GL_VP - 1) / GL_VP;
      hipLaunchKernelGGL(cc_hist_k, dim3(cdiv(D, CC_CH), nvp), dim3(GL_THREADS), GL_HLDS, c->stream, c->rows.p, c->offs.p, c->members.p, c->moff.p,
                         assign, D, V, ld, c->ccount.p);
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, hipMemcpyAsync(c->ccounted.p, assign, (size_t)D * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    }
  } else if (D) {
    hipLaunchKernelGGL(cc_moved_k, dim3(cdiv(D, 4 * (64 / GL_SUB))), dim3(256), 0, c->stream, c->rows.p, c->offs.p, assign, c->ccounted.p, D, V,
                       c->ccount.p);
    HIPCHK(c, hipGetLastError());
  }
  hipLaunchKernelGGL(cc_centers_k, dim3(cdiv(V, 64), cdiv(ld, 64)), dim3(256), 0, c->stream, c->ccount.p, c->rowval.p, V, k, ld, Crm);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// Columns per pass of the k-wide / thin products through the pass-1 stream: 10, the widest panel a planar band holds (40-byte rows); 8 at
// 8 output items per lane (the 10-column form spills 8 registers there).  A pass costs what its slots cost, hardly more for more columns
// (2.20 ms per 8-column pass against 2.29 per 10-column one at config 3 on one GPU): k = 1000 is walked in 100 passes.  Steps of 10 columns
// leave the panels 8-byte aligned inside the output rows: they are stored as float2.  (Up to round 3's 48-byte row-major bands 12-column
// passes were used up to 6 items per lane: 33.0 ms per C3-shard projection against 38.9 in 8-column passes.)  ISLE_GL_PANEL = 8 | 10.
static int gl_panel_width(const isle_ctx* c) {
  const char* e = c->knob(KN_GL_PANEL);
  if (e && atoi(e) == 10) return 10;
  if (e && atoi(e) == 8) return 8;
  return c->gl1.G <= 7 ? 10 : 8;
}

// Columns a panel pass stores: its own, and behind the last panel the padding columns of the row (ld = 4 ceil(k / 4); the consumers read
// whole float4 of a row and rely on zeros there — the packed panel is zero beyond its columns, so the accumulators are).
static inline int gl_panel_store_cols(int j0, int ncol, int k, int ld) { return j0 + ncol >= k ? ld - j0 : ncol; }
// columns of the panel that starts at j0: the last panel must also cover the padding columns of the row with at most 10 stored columns
static inline int gl_panel_cols(int PW, int j0, int k, int ld) { return k - j0 <= PW && ld - j0 > 10 ? 8 : std::min(PW, k - j0); }

// one pass of the pass-1 stream over a packed panel (LPE = ceil(wcols / 4) float4 per row) -> columns [j0, j0 + wcols) of Out
// (D x ld row-major, document order)
static int gl_panel_pass(isle_ctx* c, int wcols, int j0, int ld, float* Out, float* scratch = nullptr /*grouped form: this panel's D x 12 slab*/,
                         bool by_position = false /*row p of Out = the document at position p (dperm[p]) instead of document p*/) {
  const int LPE = (wcols + 3) / 4;
  const bool half = wcols <= 4 * LPE - 2;  // the last float4 holds at most two columns
  const float4* in = (const float4*)c->gl_Xs.p;
  if (scratch) {  // whole float4 rows of 48 bytes, in position order: a wave's 64 rows are one 3 kB run
    float4* so = (float4*)scratch;
    if (LPE == 1) return half ? launch_apply<1, true>(c, c->gl1, in, so, 0, nullptr, 3, 0) : launch_apply<1, false>(c, c->gl1, in, so, 0, nullptr, 3, 0);
    if (LPE == 2) return half ? launch_apply<2, true>(c, c->gl1, in, so, 0, nullptr, 3, 0) : launch_apply<2, false>(c, c->gl1, in, so, 0, nullptr, 3, 0);
    if (LPE == 3 && half) return launch_apply<3, true>(c, c->gl1, in, so, 0, nullptr, 3, 0);
    return isle_fail(c, ISLE_E_ARG, "LDS products: a panel of %d stored columns", wcols);
  }
  const uint32_t ld4 = (uint32_t)(ld / 4);
  float4* out = (float4*)(Out + j0);
  const uint32_t n2 = (j0 % 4) || half ? (uint32_t)((wcols + 1) / 2) : 0u;  // float2 stores where float4 would be misaligned or too wide
  const uint32_t* rowmap = by_position ? nullptr : c->dperm.p;
  if (LPE == 1) return half ? launch_apply<1, true>(c, c->gl1, in, out, 0, rowmap, ld4, n2) : launch_apply<1, false>(c, c->gl1, in, out, 0, rowmap, ld4, n2);
  if (LPE == 2) return half ? launch_apply<2, true>(c, c->gl1, in, out, 0, rowmap, ld4, n2) : launch_apply<2, false>(c, c->gl1, in, out, 0, rowmap, ld4, n2);
  if (LPE == 3 && half) return launch_apply<3, true>(c, c->gl1, in, out, 0, rowmap, ld4, n2);
  return isle_fail(c, ISLE_E_ARG, "LDS products: a panel of %d stored columns", wcols);
}

int k_gl_panel_width(const isle_ctx* c) { return gl_panel_width(c); }

// Out (D x ld row-major, natural document order, ld = 4 ceil(nc / 4)) = B^T W for a THIN column-major operand W (V x nc, nc <= 32):
// ceil(nc / 10) passes of the pass-1 stream.  The k-means++ round of a large shard: against the nc newest seeds the reference
// itself forms B^T (U C_new^T) (SURVEY §8d "sparse form"): 8 nnz bytes per 10 columns instead of re-reading the D x k projection.
// by_position: row p of Out belongs to the document at position p of the length order (c->dperm[p]; a consumer reads document d's row at
// c->dpos[d]): the rows of a wave's 64 documents are then one contiguous run instead of 64 pieces of 16 - 48 bytes at the documents' places —
// no partial-sector writes (round 5: k-means++ 264 -> 246 ms at config 3 in the timing-only build).
int k_gl_thin(isle_ctx* c, const float* Wcm, int nc, int ld, float* Out, bool by_position) {
  ISLECHK(k_band_build(c));
  if (c->gl_mode != 1) return isle_fail(c, ISLE_E_ARG, "k_gl_thin needs the LDS-banded form");
  if (ld % 4 || nc > ld) return isle_fail(c, ISLE_E_ARG, "k_gl_thin: bad leading dimension");
  const uint32_t V = (uint32_t)c->V;
  HIPCHK(c, c->gl_Xs.reserve(gl_packed_floats(V, 3, true)));
  const int PW = gl_panel_width(c);
  for (int j0 = 0, ncol; j0 < nc; j0 += ncol) {
    ncol = gl_panel_cols(PW, j0, nc, ld);
    const int wcols = gl_panel_store_cols(j0, ncol, nc, ld);
    const int LPE = (wcols + 3) / 4;
    const bool half = wcols <= 4 * LPE - 2;
    const size_t n4 = (size_t)V * LPE;
    hipLaunchKernelGGL(gl_pack_scale_k, dim3(cdiv((long)n4, 256)), dim3(256), 0, c->stream, Wcm + (size_t)j0 * V, (size_t)V, ncol, LPE, (int)half,
                       c->rowval.p, (char*)c->gl_Xs.p);
    HIPCHK(c, hipGetLastError());
    ISLECHK(gl_panel_pass(c, wcols, j0, ld, Out, nullptr, by_position));
  }
  return 0;
}

// Out (D x ld row-major, natural document order) = B^T M for a wide row-major operand M (V x ld, k <= ld columns): the k-wide SpMM
// of FPSparseMatrix::multiply_with (src/sparseMatrix.cpp:1749-1782) as ceil(k / 10) passes of the pass-1 stream, ten
// columns of diag(s) M staged per pass.  Per gathered nonzero the operand comes from LDS instead of the L2 / Infinity Cache
// row gather of spmm_wide_k (~2x faster at C2).
// Grouped form of the k-wide product (round 5).  A panel pass that writes its 40 bytes straight into the documents' 4 kB rows leaves every row
// a partial sector or two per pass — 36 ms of read-modify-write traffic per projection at config 3 (profiles/r05_gl_apply_item3_measurements.txt).
// Here up to GL_WG_PANELS panels write their rows whole and in POSITION order into slabs of their own (48 bytes a row: a wave's store is
// one 3 kB run), and this kernel turns a group of slabs into the document-major rows: 64 positions per workgroup, the slabs' 3 kB pieces read
// coalesced into an LDS tile, every row's columns of the group written as one run (640 bytes for sixteen 10-column panels) to row
// rowmap[position]; the rows' squared norms over the group's columns leave as one partial per (group, document), summed in group order afterwards.
constexpr int GL_WG_PANELS = 16;
struct GlWidePanels {
  int n;                  // panels of the group
  int j0[GL_WG_PANELS];   // first column of a panel, relative to the group's first column
  int w[GL_WG_PANELS];    // columns it stores
  int ncols;              // columns of the group (a multiple of 4 unless it ends the row)
};
__global__ __launch_bounds__(256) void gl_wide_assemble_k(const float* __restrict__ scratch, size_t slab /*floats per panel slab*/, uint32_t D,
                                                           const uint32_t* __restrict__ rowmap, GlWidePanels g, int jg0, int ld,
                                                           float* __restrict__ Out, float* __restrict__ normpart /*nullable: D floats of this group*/,
                                                           uint4* __restrict__ A2 /*nullable: the split copy by position (gemm_bf16x3.h a2 layout, slabs of 16)*/,
                                                           int nslab, int oct_end /*octets [jg0 / 8, oct_end) leave from this group*/) {
  extern __shared__ float tl[];  // 64 x tw, tw = ncols + 4 (rows 16-byte aligned: whole float4 leave in the second phase)
  __shared__ uint32_t rm[64];
  const int tw = g.ncols + 4;
  const uint32_t p0 = blockIdx.x * 64;
  if (p0 >= D) {  // a chunk of the last row block of 256 behind the last document: its units of the split copy are zeros
    if (A2) {
      const uint32_t r = threadIdx.x & 63;
      const uint64_t pos = (uint64_t)p0 + r, mb = pos >> 8;
      for (int u = (int)(threadIdx.x >> 6); u < 2 * (oct_end - jg0 / 8); u += 4) {
        const int oct = jg0 / 8 + (u >> 1), term = u & 1;
        A2[(((mb * nslab + (uint64_t)(oct >> 1)) * 2 + term) * 2 + (oct & 1)) * 256 + (pos & 255)] = make_uint4(0u, 0u, 0u, 0u);
      }
    }
    return;
  }
  const uint32_t np = min(64u, D - p0);
  if (threadIdx.x < 64) rm[threadIdx.x] = threadIdx.x < np ? rowmap[p0 + threadIdx.x] : 0u;
  // phase 1: thread t < 192 owns float4 t of every panel's 3 kB piece (row t / 3, float4 t % 3): all panels' loads in flight together
  if (threadIdx.x < 192) {
    const uint32_t r = threadIdx.x / 3, c4 = threadIdx.x - 3 * r;
    float4 v[GL_WG_PANELS];
#pragma unroll
    for (int pi = 0; pi < GL_WG_PANELS; ++pi)
      if (pi < g.n && r < np) v[pi] = reinterpret_cast<const float4*>(scratch + (size_t)pi * slab + (size_t)p0 * 12)[threadIdx.x];
#pragma unroll
    for (int pi = 0; pi < GL_WG_PANELS; ++pi) {
      if (pi < g.n && r < np) {
        const int cc = 4 * (int)c4;  // first column of this float4 inside the panel (panels start on even columns: 8-byte aligned pairs)
        float* t = tl + r * tw + g.j0[pi] + cc;
        if (cc + 1 < g.w[pi]) *reinterpret_cast<float2*>(t) = make_float2(v[pi].x, v[pi].y);
        else if (cc < g.w[pi]) t[0] = v[pi].x;
        if (cc + 3 < g.w[pi]) *reinterpret_cast<float2*>(t + 2) = make_float2(v[pi].z, v[pi].w);
        else if (cc + 2 < g.w[pi]) t[2] = v[pi].z;
      }
    }
  }
  __syncthreads();
  const int nq = g.ncols / 4;  // whole float4 (ld and the groups' first columns are multiples of 4)
  for (uint32_t i = threadIdx.x; i < np * (uint32_t)nq; i += 256) {
    const uint32_t r = i / nq, q = i - r * nq;
    *reinterpret_cast<float4*>(Out + (size_t)rm[r] * ld + jg0 + 4 * q) = *reinterpret_cast<const float4*>(tl + r * tw + 4 * q);
  }
  if (normpart) {  // four threads per row, then the row's four partial sums in a fixed order
    const uint32_t r = threadIdx.x >> 2, part = threadIdx.x & 3;
    float s = 0.f;
    if (r < np) {
      const int per = (nq + 3) / 4 * 4;  // columns per part: a multiple of 4
      const float* t = tl + r * tw;
      for (int j = part * per; j < min(g.ncols, (int)(part + 1) * per); ++j) s = fmaf(t[j], t[j], s);
    }
    const float s1 = __shfl_down(s, 1), s2 = __shfl_down(s, 2), s3 = __shfl_down(s, 3);
    if (part == 0 && r < np) normpart[rm[r]] = (s + s1) + (s2 + s3);
  }
  if (A2) {
    // the two bf16 terms of the group's columns, by POSITION: unit (row block, slab, term, octet, row) = 8 consecutive coordinates of one
    // position; a wave writes the 64 positions' units of one (octet, term): one 1 KiB run.  Coordinates beyond the group's columns (the row's
    // end) and positions beyond D are zeros.
    const uint32_t r = threadIdx.x & 63;
    const uint64_t pos = (uint64_t)p0 + r, mb = pos >> 8;
    const float* t = tl + r * tw;
    for (int u = (int)(threadIdx.x >> 6); u < 2 * (oct_end - jg0 / 8); u += 4) {
      const int ol = u >> 1, term = u & 1, oct = jg0 / 8 + ol;
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = (r < np && 8 * ol + j < g.ncols) ? t[8 * ol + j] : 0.f;
      unsigned short h[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const __bf16 x0 = (__bf16)x[j];
        const __bf16 v = term ? (__bf16)(x[j] - (float)x0) : x0;
        h[j] = __builtin_bit_cast(unsigned short, v);
      }
      A2[(((mb * nslab + (uint64_t)(oct >> 1)) * 2 + term) * 2 + (oct & 1)) * 256 + (pos & 255)] =
          make_uint4(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16), h[4] | ((uint32_t)h[5] << 16), h[6] | ((uint32_t)h[7] << 16));
    }
  }
}
__global__ __launch_bounds__(256) void gl_wide_norms_k(const float* __restrict__ normpart, uint32_t D, int ngroups, float* __restrict__ norms) {
  const uint32_t d = blockIdx.x * 256 + threadIdx.x;
  if (d >= D) return;
  float s = 0.f;
  for (int gi = 0; gi < ngroups; ++gi) s += normpart[(size_t)gi * D + d];  // group order: the same bits run after run
  norms[d] = s;
}

int k_gl_wide(isle_ctx* c, const float* Mrm, int k, int ld, float* Out, float* norms, void* A2pos, bool* a2_done) {
  if (a2_done) *a2_done = false;
  ISLECHK(k_band_build(c));
  if (c->gl_mode != 1) return isle_fail(c, ISLE_E_ARG, "k_gl_wide needs the LDS-banded form");
  if (ld % 4) return isle_fail(c, ISLE_E_ARG, "k_gl_wide: leading dimension %d not a multiple of 4", ld);
  const uint32_t V = (uint32_t)c->V;
  HIPCHK(c, c->gl_Xs.reserve(gl_packed_floats(V, 3, true)));
  const int PW = gl_panel_width(c);
  // grouped form: asked for together with the norms (the projection), a wide operand on a large shard, 10-column panels (groups of 16 then
  // start on multiples of 160 columns), and while the slabs fit (7.7 GB at 10 M documents)
  const uint64_t D = c->D;
  const int ngroups = (k + PW * GL_WG_PANELS - 1) / (PW * GL_WG_PANELS);
  const size_t slab = (size_t)D * 12;
  if (norms && PW == 10 && k >= 160 && D >= 16384 && !c->knob_zero(KN_GL_WIDE_GROUPED) &&
      isle_scratch_ok(c, 0, (double)(slab * GL_WG_PANELS + (size_t)ngroups * D) * sizeof(float))) {
    HIPCHK(c, c->gl_pscratch.reserve(slab * GL_WG_PANELS + (size_t)ngroups * D));
    float* normpart = c->gl_pscratch.p + slab * GL_WG_PANELS;
    int gi = 0;
    for (int jg0 = 0; jg0 < k; ++gi) {
      GlWidePanels g;
      g.n = 0;
      int j0 = jg0;
      for (int ncol; g.n < GL_WG_PANELS && j0 < k; j0 += ncol) {
        ncol = gl_panel_cols(PW, j0, k, ld);
        const int wcols = gl_panel_store_cols(j0, ncol, k, ld);
        const int LPE = (wcols + 3) / 4;
        const bool half = wcols <= 4 * LPE - 2;
        const size_t n4 = (size_t)V * LPE;
        hipLaunchKernelGGL(gl_pack_panel_k, dim3(cdiv((long)n4, 256)), dim3(256), 0, c->stream, Mrm, ld, j0, ncol, LPE, (int)half, c->rowval.p, n4,
                           (char*)c->gl_Xs.p);
        HIPCHK(c, hipGetLastError());
        ISLECHK(gl_panel_pass(c, wcols, j0, ld, Out, c->gl_pscratch.p + (size_t)g.n * slab));
        g.j0[g.n] = j0 - jg0;
        g.w[g.n] = wcols;
        ++g.n;
        g.ncols = j0 - jg0 + wcols;
      }
      if (g.ncols % 4) return isle_fail(c, ISLE_E_ARG, "k_gl_wide: a panel group of %d columns", g.ncols);
      const size_t lds = (size_t)64 * (g.ncols + 4) * sizeof(float);
      ISLECHK(isle_max_lds(c, (const void*)gl_wide_assemble_k, (int)lds));
      // the split copy (by position) leaves from the same tile; the last group also writes the zero octets up to the last slab's end, and the
      // grid covers the whole last row block of 256 positions
      const int nslab = (k + 15) / 16;
      const int oct_end = j0 >= k ? 2 * nslab : (jg0 + g.ncols) / 8;
      if (A2pos && (jg0 % 8 || (j0 < k && g.ncols % 8))) return isle_fail(c, ISLE_E_ARG, "k_gl_wide: a panel group that does not end on an octet");
      const long nwg = A2pos ? cdiv((long)D, 256) * 4 : cdiv((long)D, 64);
      hipLaunchKernelGGL(gl_wide_assemble_k, dim3((unsigned)nwg), dim3(256), lds, c->stream, c->gl_pscratch.p, slab, (uint32_t)D, c->dperm.p, g, jg0, ld, Out,
                         normpart + (size_t)gi * D, (uint4*)A2pos, nslab, oct_end);
      HIPCHK(c, hipGetLastError());
      jg0 = j0;
    }
    if (A2pos && a2_done) *a2_done = true;
    hipLaunchKernelGGL(gl_wide_norms_k, dim3(cdiv((long)D, 256)), dim3(256), 0, c->stream, normpart, (uint32_t)D, gi, norms);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  for (int j0 = 0, ncol; j0 < k; j0 += ncol) {
    ncol = gl_panel_cols(PW, j0, k, ld);
    const int wcols = gl_panel_store_cols(j0, ncol, k, ld);
    const int LPE = (wcols + 3) / 4;
    const bool half = wcols <= 4 * LPE - 2;
    const size_t n4 = (size_t)V * LPE;
    hipLaunchKernelGGL(gl_pack_panel_k, dim3(cdiv((long)n4, 256)), dim3(256), 0, c->stream, Mrm, ld, j0, ncol, LPE, (int)half, c->rowval.p, n4,
                       (char*)c->gl_Xs.p);
    HIPCHK(c, hipGetLastError());
    ISLECHK(gl_panel_pass(c, wcols, j0, ld, Out));
  }
  if (norms) return k_rownorms(c, Out, (int)c->D, k, ld, norms);
  return 0;
}
