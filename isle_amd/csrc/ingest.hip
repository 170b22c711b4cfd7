// isle_amd/csrc/ingest.hip — tdf text -> count matrix A (CSC) in HBM.
//
//   ing_nl_count_k / ing_nl_fill_k   line starts (one '\n' scan over the text)
//   ing_parse_k                      one thread per line: "<doc> <word> <count>", 1-based ids,        include/utils.h:158-228
//                                    blanks / tabs between fields, '\r' ignored                       (DocWordEntriesReader)
//   rs_hist_k / rs_scatter_k         stable LSD radix sort of the entries by (doc, word), 8 bits      src/trainer.cpp:236-241
//                                    per pass, hand-written (wave-level multisplit)
//   ing_flag_k / ing_compact_k       drop repeated (doc, word) pairs, first in file order survives    src/trainer.cpp:243-247
//   ing_offsets_k                    column offsets, empty documents included                         src/sparseMatrix.cpp:58-87
//
// Deviations from the reference parser, shared with the host parser of isle_amd/host/prestage.h: trailing blanks do not
// leak into the next line (the reference keeps its was_whitespace flag across '\n'), blank lines are skipped, a bad
// character or a line with more than three fields is an error instead of a debug assert.  The reference's
// std::sort + std::unique keeps an unspecified one of several equal (doc, word) lines; here it is the first in the file.
#include <utility>

#include "common.h"
#include "scan.h"

namespace {

constexpr int IT = 256;
constexpr int BYTES_PER_THREAD = 16;
constexpr int TILE_BYTES = IT * BYTES_PER_THREAD;  // 4096

__device__ inline int count_nl16(const unsigned char* __restrict__ text, uint64_t pos, uint64_t n) {
  int c = 0;
  if (pos + 16 <= n) {
    const uint4 v = *reinterpret_cast<const uint4*>(text + pos);  // pos is a multiple of 16; hipMalloc aligns the base
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int b = 0; b < 4; ++b) c += ((w[j] >> (8 * b)) & 0xffu) == (uint32_t)'\n';
  } else {
    for (uint64_t p = pos; p < n; ++p) c += text[p] == '\n';
  }
  return c;
}

__global__ __launch_bounds__(IT) void ing_nl_count_k(const unsigned char* __restrict__ text, uint64_t n, uint32_t* __restrict__ tile_cnt) {
  __shared__ uint32_t sh[IT];
  const uint64_t pos = (uint64_t)blockIdx.x * TILE_BYTES + (uint64_t)threadIdx.x * BYTES_PER_THREAD;
  sh[threadIdx.x] = pos < n ? (uint32_t)count_nl16(text, pos, n) : 0u;
  __syncthreads();
  for (int o = IT / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) tile_cnt[blockIdx.x] = sh[0];
}

// line_start[j + 1] = position after the j-th '\n' (line_start[0] = 0 is set by the host)
__global__ __launch_bounds__(IT) void ing_nl_fill_k(const unsigned char* __restrict__ text, uint64_t n, const int64_t* __restrict__ tile_off,
                                                     uint64_t* __restrict__ line_start) {
  __shared__ int64_t sh[isle_scan::SCAN_T];
  const uint64_t pos = (uint64_t)blockIdx.x * TILE_BYTES + (uint64_t)threadIdx.x * BYTES_PER_THREAD;
  const int mine = pos < n ? count_nl16(text, pos, n) : 0;
  int64_t tot;
  int64_t at = tile_off[blockIdx.x] + isle_scan::block_exclusive<int64_t>((int64_t)mine, sh, &tot);
  if (mine) {
    const uint64_t e = pos + 16 < n ? pos + 16 : n;
    for (uint64_t p = pos; p < e; ++p)
      if (text[p] == '\n') line_start[++at] = p + 1;
  }
}

// err[0]: 0 ok, 1 bad character, 2 too many fields, 3 fewer than three fields, 4 doc/word id 0 or out of range; err[1] = line
__global__ __launch_bounds__(IT) void ing_parse_k(const unsigned char* __restrict__ text, uint64_t n, const uint64_t* __restrict__ line_start, uint64_t nlines,
                                                   uint64_t V, uint64_t D, int wbits, uint64_t* __restrict__ key, uint32_t* __restrict__ cnt,
                                                   uint32_t* __restrict__ valid, unsigned long long* __restrict__ err) {
  const uint64_t l = (uint64_t)blockIdx.x * IT + threadIdx.x;
  if (l >= nlines) return;
  const uint64_t s = line_start[l];
  const uint64_t e = (l + 1 < nlines) ? line_start[l + 1] - 1 : ((n && text[n - 1] == '\n') ? n - 1 : n);
  unsigned long long f[3] = {0, 0, 0};
  int state = 0;
  bool was_ws = false, any = false;
  int bad = 0;
  for (uint64_t p = s; p < e; ++p) {
    const unsigned char ch = text[p];
    if (ch == '\r') continue;
    if (ch == ' ' || ch == '\t') {
      was_ws = true;
      continue;
    }
    if (ch < '0' || ch > '9') {
      bad = 1;
      break;
    }
    if (was_ws && any) ++state;
    was_ws = false;
    any = true;
    if (state > 2) {
      bad = 2;
      break;
    }
    f[state] = f[state] * 10ull + (unsigned long long)(ch - '0');
  }
  uint32_t ok = 0;
  if (!bad && any) {
    if (state != 2) bad = 3;
    else if (f[0] == 0 || f[1] == 0 || f[0] > D || f[1] > V) bad = 4;
    else if (f[2] == 0) bad = 5;  // a document made of zero counts would normalise to 0 / 0 (src/sparseMatrix.cpp:136-167)
    else {
      ok = 1;
      key[l] = ((f[0] - 1) << wbits) | (f[1] - 1);
      cnt[l] = (uint32_t)f[2];
    }
  }
  valid[l] = ok;
  if (bad && atomicCAS(&err[0], 0ull, (unsigned long long)bad) == 0ull) err[1] = l;
}

__global__ __launch_bounds__(IT) void ing_pack_k(const uint64_t* __restrict__ key, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ valid,
                                                  const int64_t* __restrict__ at, uint64_t nlines, uint64_t* __restrict__ okey, uint32_t* __restrict__ ocnt) {
  const uint64_t l = (uint64_t)blockIdx.x * IT + threadIdx.x;
  if (l < nlines && valid[l]) {
    okey[at[l]] = key[l];
    ocnt[at[l]] = cnt[l];
  }
}

// ---------------- stable LSD radix sort, 8 bits per pass ------------------------------------------------------------
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = IT * RS_ITEMS;  // 2048 keys per workgroup
constexpr int RS_WAVES = IT / ISLE_WAVE;

__global__ __launch_bounds__(IT) void rs_hist_k(const uint64_t* __restrict__ key, uint64_t n, int shift, uint32_t nblocks, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * RS_TILE;
#pragma unroll
  for (int r = 0; r < RS_ITEMS; ++r) {
    const uint64_t i = base + (uint64_t)r * IT + threadIdx.x;
    if (i < n) atomicAdd(&h[(key[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];  // digit-major: one exclusive scan orders digits first
}

__global__ __launch_bounds__(IT) void rs_scatter_k(const uint64_t* __restrict__ key, const uint32_t* __restrict__ val, uint64_t n, int shift, uint32_t nblocks,
                                                    const int64_t* __restrict__ hist_off, uint64_t* __restrict__ okey, uint32_t* __restrict__ oval) {
  __shared__ int64_t base_of[256];
  __shared__ uint32_t run[256];
  __shared__ uint32_t wcnt[RS_WAVES][256];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  base_of[t] = hist_off[(size_t)t * nblocks + blockIdx.x];
  run[t] = 0;
#pragma unroll
  for (int w = 0; w < RS_WAVES; ++w) wcnt[w][t] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * RS_TILE;
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int r = 0; r < RS_ITEMS; ++r) {
    const uint64_t i = base + (uint64_t)r * IT + t;
    const bool live = i < n;
    const uint64_t k = live ? key[i] : 0ull;
    const uint32_t v = live ? val[i] : 0u;
    const uint32_t d = (uint32_t)(k >> shift) & 255u;
    // lanes of this wave holding the same digit (dead lanes form their own class)
    unsigned long long peers = __ballot(live);
    if (!live) peers = ~peers;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long m = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    const uint32_t rank = (uint32_t)__popcll(peers & lt);
    if (live && rank == 0) wcnt[wv][d] = (uint32_t)__popcll(peers);
    __syncthreads();
    if (live) {
      uint32_t before = 0;
      for (int w = 0; w < wv; ++w) before += wcnt[w][d];
      const int64_t pos = base_of[d] + run[d] + before + rank;
      okey[pos] = k;
      oval[pos] = v;
    }
    __syncthreads();
    uint32_t tot = 0;
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) {
      tot += wcnt[w][t];
      wcnt[w][t] = 0;
    }
    run[t] += tot;
    __syncthreads();
  }
}

__global__ __launch_bounds__(IT) void ing_flag_k(const uint64_t* __restrict__ key, uint64_t n, uint32_t* __restrict__ flag) {
  const uint64_t i = (uint64_t)blockIdx.x * IT + threadIdx.x;
  if (i < n) flag[i] = (i == 0 || key[i] != key[i - 1]) ? 1u : 0u;
}

__global__ __launch_bounds__(IT) void ing_compact_k(const uint64_t* __restrict__ key, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ flag,
                                                     const int64_t* __restrict__ at, uint64_t n, int wbits, uint32_t* __restrict__ rows,
                                                     float* __restrict__ counts, uint32_t* __restrict__ docs) {
  const uint64_t i = (uint64_t)blockIdx.x * IT + threadIdx.x;
  if (i < n && flag[i]) {
    const int64_t j = at[i];
    rows[j] = (uint32_t)(key[i] & ((1ull << wbits) - 1ull));
    counts[j] = (float)cnt[i];
    docs[j] = (uint32_t)(key[i] >> wbits);
  }
}

// offs[d] = first entry of document d (entries sorted by document); offs[D] = m
__global__ __launch_bounds__(IT) void ing_offsets_k(const uint32_t* __restrict__ docs, uint64_t m, uint64_t D, int64_t* __restrict__ offs) {
  const uint64_t i = (uint64_t)blockIdx.x * IT + threadIdx.x;
  if (i > m) return;
  const uint64_t lo = (i == 0) ? 0 : (uint64_t)docs[i - 1] + 1;  // first document not yet opened
  const uint64_t hi = (i == m) ? D : (uint64_t)docs[i];         // documents lo..hi start at entry i
  for (uint64_t d = lo; d <= hi; ++d) offs[d] = (int64_t)i;
}

}  // namespace

#define LAUNCH_CHECK(c) HIPCHK(c, hipGetLastError())

// Stable LSD radix sort of n (key, payload) pairs on the low key_bits bits of the keys, ping-ponging between the caller's
// two buffer pairs; *in_a tells which pair holds the sorted sequence.  Also used by gram_lds.hip to order documents and
// words by their number of nonzeros.
int k_sort_pairs_u64(isle_ctx* c, uint64_t* key_a, uint32_t* val_a, uint64_t* key_b, uint32_t* val_b, uint64_t n, int key_bits, bool* in_a) {
  *in_a = true;
  if (n < 2) return 0;
  const uint32_t nblocks = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
  HIPCHK(c, c->rs_hist.reserve((size_t)256 * nblocks));
  HIPCHK(c, c->rs_hist_off.reserve((size_t)256 * nblocks + 1));
  HIPCHK(c, c->rs_scratch.reserve(isle_scan_scratch((uint64_t)256 * nblocks) + 8));
  uint64_t *ka = key_a, *kb = key_b;
  uint32_t *va = val_a, *vb = val_b;
  for (int shift = 0; shift < key_bits; shift += 8) {
    hipLaunchKernelGGL(rs_hist_k, dim3(nblocks), dim3(IT), 0, c->stream, ka, n, shift, nblocks, c->rs_hist.p);
    LAUNCH_CHECK(c);
    HIPCHK(c, (isle_scan::exclusive_scan<uint32_t, int64_t>(c->stream, c->rs_hist.p, (uint64_t)256 * nblocks, c->rs_hist_off.p, c->rs_scratch.p)));
    hipLaunchKernelGGL(rs_scatter_k, dim3(nblocks), dim3(IT), 0, c->stream, ka, va, n, shift, nblocks, c->rs_hist_off.p, kb, vb);
    LAUNCH_CHECK(c);
    std::swap(ka, kb);
    std::swap(va, vb);
    *in_a = !*in_a;
  }
  return 0;
}

// text_dev: n bytes on the device.  On success the context's count matrix is set (a_cnt / a_rows / a_offs, a_nnz).
int k_ingest_tdf(isle_ctx* c, const unsigned char* text_dev, uint64_t n, uint64_t V, uint64_t D, uint64_t* entries_read, uint64_t* err_out /*2*/) {
  TimeScope ts(c, ISLE_T_INGEST);
  err_out[0] = err_out[1] = 0;
  int wbits = 1;
  while ((1ull << wbits) < V) ++wbits;
  int dbits = 1;
  while ((1ull << dbits) < D) ++dbits;
  // ---- line starts
  const uint64_t ntiles = (n + TILE_BYTES - 1) / TILE_BYTES;
  DevBuf<uint32_t> tile_cnt, valid, cnt0, cnt1, flag, docs;
  DevBuf<int64_t> tile_off, at, hist_off, scratch;
  DevBuf<uint64_t> line_start, key0, key1, errd;
  DevBuf<uint32_t> hist;
  auto cleanup = [&]() {
    tile_cnt.release(); valid.release(); cnt0.release(); cnt1.release(); flag.release(); docs.release();
    tile_off.release(); at.release(); hist_off.release(); scratch.release();
    line_start.release(); key0.release(); key1.release(); errd.release(); hist.release();
  };
#define ING(call)            \
  do {                       \
    hipError_t ing_e = (call); \
    if (ing_e != hipSuccess) { \
      cleanup();               \
      HIPCHK(c, ing_e);        \
    }                          \
  } while (0)
  ING(tile_cnt.reserve(ntiles ? ntiles : 1));
  ING(tile_off.reserve(ntiles + 1));
  ING(scratch.reserve(isle_scan_scratch(n + 16) + 8));  // every scan below is over at most n elements
  if (ntiles) hipLaunchKernelGGL(ing_nl_count_k, dim3((unsigned)ntiles), dim3(IT), 0, c->stream, text_dev, n, tile_cnt.p);
  ING(hipGetLastError());
  ING((isle_scan::exclusive_scan<uint32_t, int64_t>(c->stream, tile_cnt.p, ntiles, tile_off.p, scratch.p)));
  int64_t nnl = 0;
  unsigned char last = '\n';
  ING(hipMemcpyAsync(&nnl, tile_off.p + ntiles, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
  if (n) ING(hipMemcpyAsync(&last, text_dev + n - 1, 1, hipMemcpyDeviceToHost, c->stream));
  ING(hipStreamSynchronize(c->stream));
  const uint64_t nlines = (uint64_t)nnl + ((n && last != '\n') ? 1 : 0);
  ING(line_start.reserve(nnl + 2));
  ING(hipMemsetAsync(line_start.p, 0, sizeof(uint64_t), c->stream));
  if (ntiles) hipLaunchKernelGGL(ing_nl_fill_k, dim3((unsigned)ntiles), dim3(IT), 0, c->stream, text_dev, n, tile_off.p, line_start.p);
  ING(hipGetLastError());
  // ---- parse
  ING(key0.reserve(nlines ? nlines : 1));
  ING(cnt0.reserve(nlines ? nlines : 1));
  ING(valid.reserve(nlines ? nlines : 1));
  ING(at.reserve(nlines + 1));
  ING(errd.reserve(2));
  ING(hipMemsetAsync(errd.p, 0, 2 * sizeof(uint64_t), c->stream));
  if (nlines)
    hipLaunchKernelGGL(ing_parse_k, dim3(cdiv((long)nlines, IT)), dim3(IT), 0, c->stream, text_dev, n, line_start.p, nlines, V, D, wbits, key0.p, cnt0.p,
                       valid.p, (unsigned long long*)errd.p);
  ING(hipGetLastError());
  ING((isle_scan::exclusive_scan<uint32_t, int64_t>(c->stream, valid.p, nlines, at.p, scratch.p)));
  int64_t nent = 0;
  ING(hipMemcpyAsync(&nent, at.p + nlines, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
  ING(hipMemcpyAsync(err_out, errd.p, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  ING(hipStreamSynchronize(c->stream));
  *entries_read = (uint64_t)nent;
  if (err_out[0]) {
    cleanup();
    return 0;  // the caller formats the message
  }
  const uint64_t ne = (uint64_t)nent;
  ING(key1.reserve(ne ? ne : 1));
  ING(cnt1.reserve(ne ? ne : 1));
  if (nlines) hipLaunchKernelGGL(ing_pack_k, dim3(cdiv((long)nlines, IT)), dim3(IT), 0, c->stream, key0.p, cnt0.p, valid.p, at.p, nlines, key1.p, cnt1.p);
  ING(hipGetLastError());
  // ---- sort by (doc, word): keys in key1/cnt1, ping-pong with key0/cnt0
  uint64_t *ka = key1.p, *kb = key0.p;
  uint32_t *va = cnt1.p, *vb = cnt0.p;
  if (ne > 1) {
    const uint32_t nblocks = (uint32_t)((ne + RS_TILE - 1) / RS_TILE);
    ING(hist.reserve((size_t)256 * nblocks));
    ING(hist_off.reserve((size_t)256 * nblocks + 1));
    ING(scratch.reserve(isle_scan_scratch((uint64_t)256 * nblocks) + 8));
    for (int shift = 0; shift < wbits + dbits; shift += 8) {
      hipLaunchKernelGGL(rs_hist_k, dim3(nblocks), dim3(IT), 0, c->stream, ka, ne, shift, nblocks, hist.p);
      ING(hipGetLastError());
      ING((isle_scan::exclusive_scan<uint32_t, int64_t>(c->stream, hist.p, (uint64_t)256 * nblocks, hist_off.p, scratch.p)));
      hipLaunchKernelGGL(rs_scatter_k, dim3(nblocks), dim3(IT), 0, c->stream, ka, va, ne, shift, nblocks, hist_off.p, kb, vb);
      ING(hipGetLastError());
      std::swap(ka, kb);
      std::swap(va, vb);
    }
  }
  // ---- drop repeated pairs, build the CSC
  ING(flag.reserve(ne ? ne : 1));
  ING(at.reserve(ne + 1));
  if (ne) hipLaunchKernelGGL(ing_flag_k, dim3(cdiv((long)ne, IT)), dim3(IT), 0, c->stream, ka, ne, flag.p);
  ING(hipGetLastError());
  ING((isle_scan::exclusive_scan<uint32_t, int64_t>(c->stream, flag.p, ne, at.p, scratch.p)));
  int64_t m = 0;
  ING(hipMemcpyAsync(&m, at.p + ne, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
  ING(hipStreamSynchronize(c->stream));
  ING(c->a_cnt.reserve(m ? m : 1));
  ING(c->a_rows.reserve(m ? m : 1));
  ING(c->a_offs.reserve(D + 1));
  ING(docs.reserve(m ? m : 1));
  if (ne) hipLaunchKernelGGL(ing_compact_k, dim3(cdiv((long)ne, IT)), dim3(IT), 0, c->stream, ka, va, flag.p, at.p, ne, wbits, c->a_rows.p, c->a_cnt.p, docs.p);
  hipLaunchKernelGGL(ing_offsets_k, dim3(cdiv((long)m + 1, IT)), dim3(IT), 0, c->stream, docs.p, (uint64_t)m, D, c->a_offs.p);
  ING(hipGetLastError());
  ING(hipStreamSynchronize(c->stream));
  c->a_V = V;
  c->a_D = D;
  c->a_nnz = (uint64_t)m;
  cleanup();
#undef ING
  return 0;
}
