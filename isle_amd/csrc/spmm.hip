// isle_amd/csrc/spmm.hip — sparse kernels of the hot path (gfx950 / wave64).
//
//   Gram apply  Z = B (B^T X)      replaces MKL_SpSpTrProd::multiply  include/matUtils.h:336-365
//     pass 1  Y = B^T X  : CSC gather, one wave per document column, 64/LPE nonzeros per step,
//                          each lane one float4 of an X row (X is V x BP row-major, L2 resident)
//     pass 2  Z = B Y    : chunked-CSR copy of B (the reference also builds a CSR copy in the operator
//                          ctor, matUtils.h:103-106): same gather kernel, one wave per (column chunk, row);
//                          chunks are pinned to XCDs so that a chunk's slice of Y is served from one L2.
//                          (An LDS-tile scatter with ds_add_f32 was measured first: 7.0 ms per apply at C2
//                          against 1.9 ms with plain LDS stores — LDS float atomics are ~4 clk per lane on
//                          gfx950 — so the scatter form was dropped; see DESIGN.md.)
//   k-wide SpMM  out_d = sum_i val_i * M[row_i, :]   (M = U or centres, V x ldk row-major)
//     replaces FPSparseMatrix::multiply_with  src/sparseMatrix.cpp:1749-1782  with fused epilogues:
//       PROJECT: P = B^T U and |P_d|^2   (UT_times_docs :1785, compute_projected_docs_l2sq :1888)
//       ASSIGN : argmin_c | |b_d|^2 + |C_c|^2 - 2 b_d^T C_c |   (distsq_docs_to_centers :1494,
//                closest_centers :1553; cblas_isamin semantics = first index of min |x|)
//   centroid scatter-add  (lloyds_iter :1631-1638)
#include <cstdlib>

#include "common.h"
#include "hamerly.h"
#include "scan.h"

// ------------------------------------------------------------------------------------------
// layout helpers
// ------------------------------------------------------------------------------------------
__global__ void pack_rm_k(const float* __restrict__ Xcm, uint64_t V, int b, int BP, float* __restrict__ Xrm) {
  const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= V * (uint64_t)BP) return;
  const uint64_t r = idx / BP;
  const int j = (int)(idx - r * BP);
  Xrm[idx] = (j < b) ? Xcm[(uint64_t)j * V + r] : 0.f;
}
__global__ void unpack_cm_k(const float* __restrict__ Zrm, uint64_t V, int b, int BP, float* __restrict__ Zcm) {
  const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= V * (uint64_t)b) return;
  const int j = (int)(idx / V);
  const uint64_t r = idx - (uint64_t)j * V;
  Zcm[idx] = Zrm[r * BP + j];
}
int k_pack_rm(isle_ctx* c, const float* Xcm, uint64_t V, int b, int BP, float* Xrm) {
  const uint64_t n = V * (uint64_t)BP;
  hipLaunchKernelGGL(pack_rm_k, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, Xcm, V, b, BP, Xrm);
  HIPCHK(c, hipGetLastError());
  return 0;
}
int k_unpack_cm(isle_ctx* c, const float* Zrm, uint64_t V, int b, int BP, float* Zcm) {
  const uint64_t n = V * (uint64_t)b;
  hipLaunchKernelGGL(unpack_cm_k, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, Zrm, V, b, BP, Zcm);
  HIPCHK(c, hipGetLastError());
  return 0;
}

__device__ inline float4 f4_fma(float s, float4 a, float4 acc) {
  acc.x = fmaf(s, a.x, acc.x);
  acc.y = fmaf(s, a.y, acc.y);
  acc.z = fmaf(s, a.z, acc.z);
  acc.w = fmaf(s, a.w, acc.w);
  return acc;
}

// ------------------------------------------------------------------------------------------
// Segment gather-reduce:  Out[seg, :] = sum_{i in [offs[seg], offs[seg+1])} vals[i] * In[idx[i], :]
// One wave per segment, 64/LPE nonzeros per step, each lane one float4 of an In row.
//   pass 1 (Y = B^T X): segment = document column of the CSC, idx = word id, In = X (V x BP, L2 resident)
//   pass 2 (Z = B Y)  : segment = (column chunk, word row) of the chunked-CSR copy, idx = doc id,
//                       In = Y; a chunk's slice of Y (<= ~1.5 MB) stays in ONE XCD's L2 because the
//                       workgroups of chunk c are the ones with blockIdx % 8 == c % 8 (XCD_MAP).
// No atomics anywhere: per-chunk partial rows go to a slab that reduce_chunks_k sums in chunk order.
// ------------------------------------------------------------------------------------------
template <int LPE, bool XCD_MAP>
__global__ __launch_bounds__(256) void seg_gather_k(const float* __restrict__ vals, const uint32_t* __restrict__ idx,
                                                     const int64_t* __restrict__ offs, const float4* __restrict__ In,
                                                     float4* __restrict__ Out, uint32_t nseg /*per group*/, uint32_t ngroups) {
  constexpr int EPW = 64 / LPE;
  const int lane = threadIdx.x & 63;
  const uint32_t w = threadIdx.x >> 6;
  size_t seg;
  if (!XCD_MAP) {
    const uint32_t sid = blockIdx.x * 4 + w;
    if (sid >= nseg) return;
    seg = sid;
  } else {
    const uint32_t xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const uint32_t bpc = (nseg + 3) / 4;
    const uint32_t grp = 8 * (j / bpc) + xcd;
    const uint32_t r = 4 * (j % bpc) + w;
    if (grp >= ngroups || r >= nseg) return;
    seg = (size_t)grp * nseg + r;
  }
  const int e = lane / LPE;
  const int q = lane - e * LPE;
  const bool active = e < EPW;
  const int es = active ? e : 0;
  constexpr int NB = EPW * LPE;  // nonzeros per batch: one coalesced (idx, val) load per lane, LPE gather steps
  const int64_t beg = offs[seg], end = offs[seg + 1];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t ci = 0;
  float cv = 0.f;
  if (lane < NB && beg + lane < end) {
    ci = __builtin_nontemporal_load(&idx[beg + lane]);
    cv = __builtin_nontemporal_load(&vals[beg + lane]);
  }
  for (int64_t i0 = beg; i0 < end; i0 += NB) {
    uint32_t ni = 0;
    float nv = 0.f;
    const int64_t i1 = i0 + NB + lane;  // prefetch the next batch while this one is gathered
    if (lane < NB && i1 < end) {
      ni = __builtin_nontemporal_load(&idx[i1]);
      nv = __builtin_nontemporal_load(&vals[i1]);
    }
#pragma unroll
    for (int st = 0; st < LPE; ++st) {
      if (i0 + st * EPW < end) {  // wave-uniform
        const uint32_t r = __shfl(ci, st * EPW + es);
        const float v = __shfl(cv, st * EPW + es);  // 0 beyond the segment end
        acc = f4_fma(v, In[(size_t)r * LPE + q], acc);
      }
    }
    ci = ni;
    cv = nv;
  }
  if (!active) acc = make_float4(0.f, 0.f, 0.f, 0.f);
  // reduce over the entry slots; the lowest slot of every aligned group is always exact
#pragma unroll
  for (int m = 1; m < EPW; m <<= 1) {
    const int pe = e ^ m;
    const bool ok = active && (pe < EPW);
    const int src = ok ? pe * LPE + q : lane;
    const float ox = __shfl(acc.x, src), oy = __shfl(acc.y, src), oz = __shfl(acc.z, src), ow = __shfl(acc.w, src);
    if (ok) {
      acc.x += ox;
      acc.y += oy;
      acc.z += oz;
      acc.w += ow;
    }
  }
  if (e == 0) Out[seg * LPE + q] = acc;
}

int k_gram_pass1(isle_ctx* c, int BP) {
  TimeScope ts(c, ISLE_T_GRAM_PASS1);
  const uint32_t D = (uint32_t)c->D;
  if (D == 0) return 0;
  dim3 g(cdiv(D, 4)), b(256);
  const float4* X = (const float4*)c->Xrm.p;
  float4* Y = (float4*)c->Yrm.p;
#define L1(L) hipLaunchKernelGGL((seg_gather_k<L, false>), g, b, 0, c->stream, c->vals.p, c->rows.p, c->offs.p, X, Y, D, 1u)
  switch (BP / 4) {
    case 1: L1(1); break;
    case 2: L1(2); break;
    case 3: L1(3); break;
    case 4: L1(4); break;
    case 5: L1(5); break;
    case 6: L1(6); break;
    case 7: L1(7); break;
    case 8: L1(8); break;
    default: return isle_fail(c, ISLE_E_ARG, "unsupported panel width BP=%d", BP);
  }
#undef L1
  HIPCHK(c, hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// Chunked-CSR copy of B for pass 2: cell (chunk, row) holds the row's nonzeros whose column lies in the
// chunk.  Stands in for the CSR copy the reference builds in the operator constructor
// (mkl_scsrcsc, include/matUtils.h:103-106).  Counts are exact; the order inside a cell follows the
// placement atomics (so the fp32 summation order of pass 2 may differ between runs, by rounding only).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void csr_count_k(const uint32_t* __restrict__ rows, const int64_t* __restrict__ offs, uint32_t D,
                                                    uint32_t V, uint32_t Cc, uint32_t* __restrict__ cnt) {
  const int lane = threadIdx.x & 63;
  const uint32_t d = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (d >= D) return;
  const size_t base = (size_t)(d / Cc) * V;
  for (int64_t i = offs[d] + lane; i < offs[d + 1]; i += 64) atomicAdd(&cnt[base + rows[i]], 1u);
}
__global__ __launch_bounds__(256) void csr_fill_k(const float* __restrict__ vals, const uint32_t* __restrict__ rows,
                                                   const int64_t* __restrict__ offs, uint32_t D, uint32_t V, uint32_t Cc,
                                                   const int64_t* __restrict__ seg_off, uint32_t* __restrict__ fill,
                                                   uint32_t* __restrict__ ccol, float* __restrict__ cval) {
  const int lane = threadIdx.x & 63;
  const uint32_t d = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (d >= D) return;
  const size_t base = (size_t)(d / Cc) * V;
  for (int64_t i = offs[d] + lane; i < offs[d + 1]; i += 64) {
    const size_t cell = base + rows[i];
    const int64_t pos = seg_off[cell] + atomicAdd(&fill[cell], 1u);
    ccol[pos] = d;
    cval[pos] = vals[i];
  }
}

// The operator build of a solve: decides between the LDS-banded form (gram_lds.hip; every row of B holds one value) and the
// gather form below, and builds the transposed copy of B that form needs.
int k_band_build(isle_ctx* c) {
  if (c->band_ready) return 0;
  TimeScope ts(c, ISLE_T_BAND_BUILD);
  ISLECHK(k_gl_detect(c));
  if (c->gl_mode == 1) ISLECHK(k_gl_build(c));
  else ISLECHK(k_band_build_chunked(c));
  c->band_ready = true;
  return 0;
}

int k_band_build_chunked(isle_ctx* c) {
  const uint32_t D = (uint32_t)c->D, V = (uint32_t)c->V;
  // chunk size: a chunk's slice of Y (Cc x 16 floats) should sit comfortably in a 4 MiB XCD L2
  uint32_t Cc = c->band_rows ? c->band_rows : 32768;
  uint32_t nch = (uint32_t)((D + Cc - 1) / Cc);
  nch = (nch + 7) & ~7u;  // a multiple of the 8 XCDs
  if (nch == 0) nch = 8;
  Cc = (D + nch - 1) / nch;
  if (Cc == 0) Cc = 1;
  c->nbands = nch;
  c->chunk_cols = Cc;
  const size_t ncell = (size_t)nch * V;
  HIPCHK(c, c->bcol.reserve(c->nnz ? c->nnz : 1));
  HIPCHK(c, c->bval.reserve(c->nnz ? c->nnz : 1));
  HIPCHK(c, c->seg_off.reserve(ncell + 1));
  uint32_t* cnt = nullptr;
  int64_t* blk = nullptr;
  HIPCHK(c, hipMalloc((void**)&cnt, ncell * sizeof(uint32_t)));
  HIPCHK(c, hipMalloc((void**)&blk, isle_scan::scan_scratch_elems(ncell) * sizeof(int64_t)));
  HIPCHK(c, hipMemsetAsync(cnt, 0, ncell * sizeof(uint32_t), c->stream));
  if (D) hipLaunchKernelGGL(csr_count_k, dim3(cdiv(D, 4)), dim3(256), 0, c->stream, c->rows.p, c->offs.p, D, V, Cc, cnt);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, (isle_scan::exclusive_scan<uint32_t, int64_t>(c->stream, cnt, ncell, c->seg_off.p, blk)));
  HIPCHK(c, hipMemsetAsync(cnt, 0, ncell * sizeof(uint32_t), c->stream));
  if (D)
    hipLaunchKernelGGL(csr_fill_k, dim3(cdiv(D, 4)), dim3(256), 0, c->stream, c->vals.p, c->rows.p, c->offs.p, D, V, Cc, c->seg_off.p,
                       cnt, c->bcol.p, c->bval.p);
  HIPCHK(c, hipGetLastError());
  int64_t total = 0;
  HIPCHK(c, hipMemcpyAsync(&total, c->seg_off.p + ncell, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  (void)hipFree(cnt);
  (void)hipFree(blk);
  if ((uint64_t)total != c->nnz)
    return isle_fail(c, ISLE_E_NUMERIC, "operator build: placed %lld of %llu nonzeros", (long long)total, (unsigned long long)c->nnz);
  return 0;
}

// Z[row, :] = sum_chunk part[chunk][row][:]   (fixed chunk order)
__global__ __launch_bounds__(256) void reduce_chunks_k(const float4* __restrict__ part, uint32_t nch, size_t n4 /*V*LPE*/,
                                                        float4* __restrict__ Z) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 s = part[i];
  for (uint32_t ch = 1; ch < nch; ++ch) {
    const float4 p = part[(size_t)ch * n4 + i];
    s.x += p.x;
    s.y += p.y;
    s.z += p.z;
    s.w += p.w;
  }
  Z[i] = s;
}

int k_gram_pass2(isle_ctx* c, int BP) {
  ISLECHK(k_band_build(c));
  if (c->gl_mode == 1) return isle_fail(c, ISLE_E_ARG, "gather pass 2 called on the LDS-banded form");
  TimeScope ts(c, ISLE_T_GRAM_PASS2);
  const uint32_t V = (uint32_t)c->V;
  const uint32_t nch = c->nbands;
  const int LPE = BP / 4;
  HIPCHK(c, c->Zpart.reserve((size_t)nch * V * BP));
  const uint32_t bpc = (V + 3) / 4;
  dim3 g(8 * (nch / 8) * bpc), b(256);
  const float4* Y = (const float4*)c->Yrm.p;
  float4* Pt = (float4*)c->Zpart.p;
#define L2(L) hipLaunchKernelGGL((seg_gather_k<L, true>), g, b, 0, c->stream, c->bval.p, c->bcol.p, c->seg_off.p, Y, Pt, V, nch)
  switch (LPE) {
    case 1: L2(1); break;
    case 2: L2(2); break;
    case 3: L2(3); break;
    case 4: L2(4); break;
    case 5: L2(5); break;
    case 6: L2(6); break;
    case 7: L2(7); break;
    case 8: L2(8); break;
    default: return isle_fail(c, ISLE_E_ARG, "unsupported panel width BP=%d", BP);
  }
#undef L2
  HIPCHK(c, hipGetLastError());
  const size_t n4 = (size_t)V * LPE;
  hipLaunchKernelGGL(reduce_chunks_k, dim3(cdiv(n4, 256)), dim3(256), 0, c->stream, Pt, nch, n4, (float4*)c->Zrm.p);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// Frobenius: sum of squares in double, two-stage (deterministic)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_k(const float* __restrict__ v, uint64_t n, double* __restrict__ part) {
  __shared__ double sh[256];
  double s = 0.0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    const double x = (double)v[i];
    s += x * x;
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}
int k_frobenius(isle_ctx* c, double* out_host) {
  const int nb = 1024;
  HIPCHK(c, c->part.reserve(nb));
  hipLaunchKernelGGL(sumsq_k, dim3(nb), dim3(256), 0, c->stream, c->vals.p, c->nnz, c->part.p);
  HIPCHK(c, hipGetLastError());
  std::vector<double> h(nb);
  HIPCHK(c, hipMemcpyAsync(h.data(), c->part.p, nb * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  double s = 0.0;
  for (double x : h) s += x;
  *out_host = s;
  return 0;
}

// ------------------------------------------------------------------------------------------
// Validity of an uploaded CSC matrix, checked on the device (the reference asserts the same in MKL_SpSpTrProd's constructor,
// include/matUtils.h:138-148): offsets monotone, row indices below V and strictly ascending inside a column.  err[0] = smallest failing
// column + 1 and what failed there (1 offsets, 2 range, 3 order) as ONE word, (column + 1) << 2 | kind, under atomicMin: column and kind
// always belong together (err[1] unused).  A thread per column; 1 B nonzeros in a few ms where the
// host loop it replaces took seconds.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void csc_validate_k(const uint32_t* __restrict__ rows, const int64_t* __restrict__ offs, uint64_t D, uint32_t V,
                                                       unsigned long long* __restrict__ err) {
  const uint64_t d = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (d >= D) return;
  const int64_t b = offs[d], e = offs[d + 1];
  int bad = 0;
  if (e < b) {
    bad = 1;
  } else {
    uint32_t prev = 0;
    for (int64_t i = b; i < e; ++i) {
      const uint32_t r = rows[i];
      if (r >= V) bad = bad ? bad : 2;
      else if (i > b && r <= prev) bad = bad ? bad : 3;
      prev = r;
    }
  }
  if (bad) atomicMin(&err[0], ((unsigned long long)(d + 1) << 2) | (unsigned long long)bad);
}
int k_csc_validate(isle_ctx* c, unsigned long long* err_host2) {
  const uint64_t D = c->D;
  HIPCHK(c, c->dbg_cnt.reserve(2));
  const unsigned long long init[2] = {~0ull, 0ull};
  HIPCHK(c, hipMemcpyAsync(c->dbg_cnt.p, init, sizeof init, hipMemcpyHostToDevice, c->stream));
  if (D) {
    hipLaunchKernelGGL(csc_validate_k, dim3(cdiv((long)D, 256)), dim3(256), 0, c->stream, c->rows.p, c->offs.p, D, (uint32_t)c->V, c->dbg_cnt.p);
    HIPCHK(c, hipGetLastError());
  }
  HIPCHK(c, hipMemcpyAsync(err_host2, c->dbg_cnt.p, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (err_host2[0] == ~0ull) {
    err_host2[0] = err_host2[1] = 0;
  } else {  // decode: [0] = column + 1, [1] = kind
    err_host2[1] = err_host2[0] & 3ull;
    err_host2[0] >>= 2;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------
// k-wide SpMM, one wave per document; lane owns float4 chunks {lane + 64*it} of the output row
// ------------------------------------------------------------------------------------------
enum { WIDE_PROJECT = 0, WIDE_ASSIGN = 1 };
#include "hamerly.h"
constexpr int YY_GROUP = 8;  // centres per Yinyang group: two float4 of a centre row, never straddling a 64-byte line

// The assignment epilogue shared by the fused k-wide SpMM (spmm_wide_k) and the dots-from-memory variant (dots_assign_k):
// lane `lane` holds the dot products of document d with centres 4 (lane + 64 it) .. + 3.
// argmin_c | |b_d|^2 + |C_c|^2 - 2 b_d^T C_c |  with cblas_isamin semantics (first index of the minimum), plus the Hamerly /
// Yinyang bounds of the chosen centre (src/sparseMatrix.cpp:1494-1572).
template <int NIT>
__device__ inline void wide_assign_epilogue(const float4 (&acc)[NIT], int lane, uint32_t d, int nq, int k, const float* __restrict__ cn,
                                            const float* __restrict__ dn, uint32_t* __restrict__ assign, float* __restrict__ ub,
                                            float* __restrict__ lb, int G) {
  const float dnd = dn[d];
  float best = 3.4e38f, second = 3.4e38f, cmax = 0.f;
  uint32_t bidx = 0xffffffffu;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int cidx = lane + 64 * it;
    if (cidx < nq) {
      const float a[4] = {acc[it].x, acc[it].y, acc[it].z, acc[it].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int cc = 4 * cidx + j;
        if (cc < k) {
          cmax = fmaxf(cmax, cn[cc]);
          const float dist = fabsf((-2.0f * a[j] + cn[cc]) + dnd);
          if (dist < best) {  // ascending cc per lane -> first index wins ties
            second = best;
            best = dist;
            bidx = (uint32_t)cc;
          } else {
            second = fminf(second, dist);
          }
        }
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ob = __shfl_xor(best, off);
    const float os = __shfl_xor(second, off);
    const uint32_t oi = __shfl_xor(bidx, off);
    if (ob < best || (ob == best && oi < bidx)) {
      second = fminf(best, os);
      best = ob;
      bidx = oi;
    } else {
      second = fminf(second, ob);
    }
  }
  if (ub) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cmax = fmaxf(cmax, __shfl_xor(cmax, off));
  }
  if (ub && G > 0) {
    // Yinyang group bounds: for every group of YY_GROUP consecutive centres the distance to its closest member other than
    // the assigned centre.  A lane holds 4 consecutive centres, lane ^ 1 the other half of the group.
    const float E = ISLE_SLACK_REL * (dnd + cmax), sE = sqrtf(E);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int cidx = lane + 64 * it;
      float m = 3.4e38f;
      if (cidx < nq) {
        const float a[4] = {acc[it].x, acc[it].y, acc[it].z, acc[it].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int cc = 4 * cidx + j;
          if (cc < k && (uint32_t)cc != bidx) m = fminf(m, fabsf((-2.0f * a[j] + cn[cc]) + dnd));
        }
      }
      m = fminf(m, __shfl_xor(m, 1));
      const int g = cidx >> 1;
      if (!(lane & 1) && g < G) {
        const float l = sqrtf(m);
        lb[(size_t)d * G + g] = fmaxf(l - fminf(sE, E / fmaxf(l, 1e-30f)), 0.f);
      }
    }
    if (lane == 0) {
      const float u = sqrtf(best);
      ub[d] = u + fminf(sE, E / fmaxf(u, 1e-30f));
    }
  }
  if (lane == 0) {
    assign[d] = bidx;
    if (ub && G == 0) hamerly_store_bounds(best, second, dnd + cmax, &ub[d], &lb[d]);
  }
}

template <int NIT, int MODE>
__global__ __launch_bounds__(256) void spmm_wide_k(const float* __restrict__ vals, const uint32_t* __restrict__ rows,
                                                    const int64_t* __restrict__ offs, const float4* __restrict__ M, int nq,
                                                    int k, uint32_t D, float4* __restrict__ P, float* __restrict__ norms,
                                                    const float* __restrict__ cn, const float* __restrict__ dn,
                                                    uint32_t* __restrict__ assign, const uint32_t* __restrict__ perm,
                                                    const uint32_t* __restrict__ nslots /*nullable: device-side slot count*/,
                                                    float* __restrict__ ub, float* __restrict__ lb /*nullable: Hamerly bounds out*/,
                                                    int G /*0: lb = one bound per document; > 0: lb = G group bounds per document*/) {
  const int lane = threadIdx.x & 63;
  if (nslots) D = min(D, *nslots);
  // XCD-contiguous slots: workgroups of one XCD (blockIdx % 8) walk one contiguous eighth of the slot list, so that
  // with `perm` = documents grouped by their previous centre each L2 keeps re-serving one cluster's hot vocabulary rows
  // (with a device-side slot count only a prefix of the slots is live: keep the plain interleaved map there, or all
  // live slots would land on one XCD)
  const uint32_t nb = gridDim.x, per = (nb + 7) / 8;
  const uint32_t vb = nslots ? blockIdx.x : (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  uint32_t d = vb * 4 + (threadIdx.x >> 6);
  d = __builtin_amdgcn_readfirstlane(d);
  if (vb >= nb || d >= D) return;
  if (perm) d = __builtin_amdgcn_readfirstlane(perm[d]);
  const int64_t beg = offs[d], end = offs[d + 1];
  float4 acc[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) acc[it] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t base = beg; base < end; base += 64) {
    const int cnt = (int)min((int64_t)64, end - base);
    const uint32_t myrow = (lane < cnt) ? rows[base + lane] : 0u;
    const float myval = (lane < cnt) ? vals[base + lane] : 0.f;
#pragma unroll 4
    for (int e = 0; e < cnt; ++e) {
      const uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)myrow, e);
      const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myval), e));
      const float4* mr = M + (size_t)r * nq;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int cidx = lane + 64 * it;
        if (cidx < nq) acc[it] = f4_fma(v, mr[cidx], acc[it]);
      }
    }
  }
  if (MODE == WIDE_PROJECT) {
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int cidx = lane + 64 * it;
      if (cidx < nq) {
        P[(size_t)d * nq + cidx] = acc[it];
        s += acc[it].x * acc[it].x + acc[it].y * acc[it].y + acc[it].z * acc[it].z + acc[it].w * acc[it].w;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) norms[d] = s;
  } else {
    wide_assign_epilogue<NIT>(acc, lane, d, nq, k, cn, dn, assign, ub, lb, G);
  }
}

template <int MODE>
static int launch_wide(isle_ctx* c, const float* Mrm, int k, int ldk, float* P, float* norms, const float* cn, const float* dn,
                       uint32_t* assign, const uint32_t* perm, const uint32_t* nslots = nullptr, float* ub = nullptr,
                       float* lb = nullptr, int G = 0) {
  const uint32_t D = (uint32_t)c->D;
  if (D == 0) return 0;
  const int nq = ldk / 4;
  const int nit = cdiv(nq, 64);
  dim3 g(8 * cdiv(cdiv(D, 4), 8)), b(256);  // multiple of 8 so that the XCD slot map is a bijection
#define LW(N)                                                                                                              \
  hipLaunchKernelGGL((spmm_wide_k<N, MODE>), g, b, 0, c->stream, c->vals.p, c->rows.p, c->offs.p, (const float4*)Mrm, nq, k, D, \
                     (float4*)P, norms, cn, dn, assign, perm, nslots, ub, lb, G)
  if (nit <= 1) LW(1);
  else if (nit <= 2) LW(2);
  else if (nit <= 4) LW(4);
  else if (nit <= 8) LW(8);
  else return isle_fail(c, ISLE_E_ARG, "k = %d too large (max 2048)", k);
#undef LW
  HIPCHK(c, hipGetLastError());
  return 0;
}
// The k-wide products as ceil(k / 12) passes of the LDS-banded pass-1 stream pay one staging of every word band per pass: a
// win while the vocabulary spans few bands (C2: 15 bands, 5.8 vs 7.0 ms), a loss at 30 bands (C3 shard: 68 vs 52 ms).
// ISLE_WIDE_GATHER=1 / ISLE_WIDE_LDS=1 force either form.
static bool wide_through_lds(const isle_ctx* c) {
  if (c->knob_on(KN_WIDE_GATHER)) return false;
  if (c->knob_on(KN_WIDE_LDS)) return true;
  return c->V <= 32 * 3412;  // 8-column panels (gram_lds.hip gl_panel_width): measured 46.7 against 51.7 ms at 30 word bands (V = 100k, k = 1000)
}

// Assignment from stored dot products (dots = B^T C computed by the LDS-banded wide SpMM, gram_lds.hip): one wave per document.
template <int NIT>
__global__ __launch_bounds__(256) void dots_assign_k(const float4* __restrict__ dots, int nq, int k, uint32_t D, const float* __restrict__ cn,
                                                      const float* __restrict__ dn, uint32_t* __restrict__ assign, float* __restrict__ ub,
                                                      float* __restrict__ lb, int G) {
  const int lane = threadIdx.x & 63;
  const uint32_t d = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (d >= D) return;
  float4 acc[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int cidx = lane + 64 * it;
    acc[it] = cidx < nq ? dots[(size_t)d * nq + cidx] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  wide_assign_epilogue<NIT>(acc, lane, d, nq, k, cn, dn, assign, ub, lb, G);
}

// The same assignment + Yinyang group bounds straight from the COLUMN-major dot products a GEMM leaves (dotsT[c * D + d], the first
// assignment of Lloyd on B through the projection): one thread per document walks its row — consecutive threads read consecutive
// documents of one centre, 128 coalesced bytes per half wave — so the D x k matrix is read once where the doc-major form needed a
// transposition into c->P first (40 GB written and read again for all of config 3: 50 + 24 ms per step).  Per group of YY_GROUP
// centres the smallest distance, its first index and the runner-up are final when the group's last column has been seen; the
// group that holds the assignment gets its runner-up as bound (wide_assign_epilogue's "closest member other than the assigned
// centre"), which is known once the row is done: the bounds go through an LDS tile [document][group] and leave it as whole rows.
__global__ __launch_bounds__(256) void dots_assign_cm_k(const float* __restrict__ dotsT, uint32_t D, int k, int G, const float* __restrict__ cn,
                                                        const float* __restrict__ dn, const float* __restrict__ cn_max_p, uint32_t* __restrict__ assign,
                                                        float* __restrict__ ub, float* __restrict__ lb) {
  // bounds leave through an LDS tile of DA_GCH groups per document at a time (64 contiguous bytes of a document's row per flush): a
  // tile of all 125 groups (64 KB for 128 documents) left two workgroups per CU and the kernel at 1.7 TB/s
  constexpr int DA_GCH = 16;
  __shared__ float tile[256][DA_GCH + 1];
  const uint32_t d0 = blockIdx.x * 256u;
  const uint32_t j = threadIdx.x;
  const uint32_t nd = min(256u, D - d0);
  const bool live = j < nd;
  const uint32_t d = d0 + (live ? j : 0u);
  const float dnd = dn[d];
  const float E = ISLE_SLACK_REL * (dnd + *cn_max_p), sE = sqrtf(E);
  float best = 3.4e38f, bg_m2 = 3.4e38f;
  uint32_t bidx = 0xffffffffu;
  int bg = 0;
  for (int g0 = 0; g0 < G; g0 += DA_GCH) {
    const int ng = min(DA_GCH, G - g0);
    for (int gg = 0; gg < ng; ++gg) {
      const int g = g0 + gg;
      float dot[YY_GROUP];
#pragma unroll
      for (int t = 0; t < YY_GROUP; ++t) dot[t] = dotsT[(size_t)min(YY_GROUP * g + t, k - 1) * D + d];  // eight loads in flight
      float m1 = 3.4e38f, m2 = 3.4e38f;
      uint32_t i1 = 0xffffffffu;
#pragma unroll
      for (int t = 0; t < YY_GROUP; ++t) {
        const int cc = YY_GROUP * g + t;
        if (cc < k) {
          const float dist = fabsf((-2.0f * dot[t] + cn[cc]) + dnd);
          if (dist < m1) {  // ascending index: a tie keeps the earlier centre
            m2 = m1;
            m1 = dist;
            i1 = (uint32_t)cc;
          } else {
            m2 = fminf(m2, dist);
          }
        }
      }
      tile[j][gg] = yy_slack_down_sq(m1, E, sE);
      if (m1 < best) {
        best = m1;
        bidx = i1;
        bg = g;
        bg_m2 = m2;
      }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nd * (uint32_t)ng; i += 256) {
      const uint32_t jj = i / (uint32_t)ng, gg = i - jj * (uint32_t)ng;
      lb[(size_t)(d0 + jj) * G + g0 + gg] = tile[jj][gg];
    }
    __syncthreads();
  }
  if (live) {
    lb[(size_t)d * G + bg] = yy_slack_down_sq(bg_m2, E, sE);  // the assigned centre's group: its closest OTHER member
    const float u = sqrtf(best);
    ub[d] = u + fminf(sE, E / fmaxf(u, 1e-30f));
    assign[d] = bidx;
  }
}
int k_dots_assign_cm(isle_ctx* c, const float* dotsT, int k, int G, const float* cn, const float* dn, const float* cn_max_dev, uint32_t* assign, float* ub,
                     float* lb) {
  const uint32_t D = (uint32_t)c->D;
  if (D == 0) return 0;
  hipLaunchKernelGGL(dots_assign_cm_k, dim3(cdiv(D, 256)), dim3(256), 0, c->stream, dotsT, D, k, G, cn, dn, cn_max_dev, assign, ub, lb);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// Assignment (and bounds) from dot products held doc-major in c->P (D x ldk): dist = (-2 dot + |C_c|^2) + |b_d|^2
int k_dots_assign(isle_ctx* c, int k, int ldk, const float* cn, const float* dn, uint32_t* assign, float* ub, float* lb, int G) {
  const uint32_t D = (uint32_t)c->D;
  if (D == 0) return 0;
  const int nq = ldk / 4, nit = cdiv(nq, 64);
  const dim3 g(cdiv(D, 4)), b(256);
#define DA(N) hipLaunchKernelGGL((dots_assign_k<N>), g, b, 0, c->stream, (const float4*)c->P.p, nq, k, D, cn, dn, assign, ub, lb, G)
  if (nit <= 1) DA(1);
  else if (nit <= 2) DA(2);
  else if (nit <= 4) DA(4);
  else if (nit <= 8) DA(8);
  else return isle_fail(c, ISLE_E_ARG, "assignment: k = %d too large (max 2048)", k);
#undef DA
  HIPCHK(c, hipGetLastError());
  return 0;
}

int k_spmm_wide_project(isle_ctx* c, const float* Mrm, int k, int ldk, float* P, float* norms, void* A2pos, bool* a2_done) {
  TimeScope ts(c, ISLE_T_PROJECT);
  if (a2_done) *a2_done = false;
  ISLECHK(k_gl_detect(c));
  if (c->gl_mode == 1 && wide_through_lds(c)) {  // row-constant B: panels of M through LDS (gram_lds.hip)
    return k_gl_wide(c, Mrm, k, ldk, P, norms, A2pos, a2_done);  // (the grouped form forms the norms — and the split copy — on its way; the plain one calls k_rownorms)
  }
  return launch_wide<WIDE_PROJECT>(c, Mrm, k, ldk, P, norms, nullptr, nullptr, nullptr, nullptr);
}
int k_spmm_wide_assign(isle_ctx* c, const float* Mrm, int k, int ldk, const float* cn, const float* dn, uint32_t* assign,
                       const uint32_t* perm, const uint32_t* nslots, float* ub, float* lb, int G) {
  TimeScope ts(c, ISLE_T_SPARSE_ASSIGN);
  ISLECHK(k_gl_detect(c));
  if (c->gl_mode == 1 && !nslots && c->D && wide_through_lds(c)) {
    // full assignment on a row-constant B: dots = B^T C by the LDS-banded wide SpMM into the projection buffer (P is
    // recomputed if it is needed again), then the same epilogue from memory.  `perm` is only a visiting order: ignored.
    const uint32_t D = (uint32_t)c->D;
    HIPCHK(c, c->P.reserve((size_t)D * ldk));
    c->P_ready = false;
    c->Pt_ready = false;
    c->Pt2_ready = false;
    ISLECHK(k_gl_wide(c, Mrm, k, ldk, c->P.p));
    return k_dots_assign(c, k, ldk, cn, dn, assign, ub, lb, G);
  }
  return launch_wide<WIDE_ASSIGN>(c, Mrm, k, ldk, nullptr, nullptr, cn, dn, assign, perm, nslots, ub, lb, G);
}

// |b_d|^2  (compute_docs_l2sq  src/sparseMatrix.cpp:1680-1687)
__global__ __launch_bounds__(256) void doc_norms_k(const float* __restrict__ vals, const int64_t* __restrict__ offs, uint32_t D,
                                                    float* __restrict__ dn) {
  const int lane = threadIdx.x & 63;
  const uint32_t d = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (d >= D) return;
  float s = 0.f;
  for (int64_t i = offs[d] + lane; i < offs[d + 1]; i += 64) s = fmaf(vals[i], vals[i], s);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) dn[d] = s;
}
int k_doc_norms(isle_ctx* c, float* dn) {
  const uint32_t D = (uint32_t)c->D;
  if (D == 0) return 0;
  hipLaunchKernelGGL(doc_norms_k, dim3(cdiv(D, 4)), dim3(256), 0, c->stream, c->vals.p, c->offs.p, D, dn);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// centre[c] += b_d for every member d  (src/sparseMatrix.cpp:1631-1638), without global atomics: Crm[w][c] = sum over the nonzeros (w, d) with assign[d] == c of B[w,d].
// One wave per vocabulary row w walks the row's cells of the chunked-CSR copy and keeps a k-bin histogram in LDS
// (one ds_add_f32 per nonzero); the finished row of sums is written once, coalesced.
__global__ __launch_bounds__(256) void centers_from_rows_k(const float* __restrict__ cval, const uint32_t* __restrict__ ccol,
                                                            const int64_t* __restrict__ seg_off, uint32_t V, uint32_t nch,
                                                            const uint32_t* __restrict__ assign, int ldk, float* __restrict__ Crm) {
  extern __shared__ float bins_all[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* bins = bins_all + wave * ldk;
  const uint32_t w = blockIdx.x * 4 + wave;
  if (w >= V) return;
  for (int j = lane; j < ldk; j += 64) bins[j] = 0.f;
  for (uint32_t ch = 0; ch < nch; ++ch) {
    const size_t seg = (size_t)ch * V + w;
    const int64_t beg = seg_off[seg], end = seg_off[seg + 1];
    for (int64_t i = beg + lane; i < end; i += 64) {
      const uint32_t d = __builtin_nontemporal_load(&ccol[i]);
      const float v = __builtin_nontemporal_load(&cval[i]);
      atomicAdd(&bins[assign[d]], v);
    }
  }
  // LDS ops of one wave complete in order; the wave's own atomics are visible to its later reads
  __builtin_amdgcn_s_waitcnt(0xC07F);
  for (int j = lane; j < ldk; j += 64) Crm[(size_t)w * ldk + j] = bins[j];
}

// first_of_run: first centroid update of a Lloyd run (the counting form then counts from scratch, later calls only move the
// documents that changed centre)
int k_centers_from_rows(isle_ctx* c, const uint32_t* assign, int k, int ldk, float* Crm, bool first_of_run) {
  ISLECHK(k_gl_detect(c));
  if (c->gl_mode == 1)  // row-constant B: integer counting, no transposed copy
    return k_centers_counts(c, assign, k, ldk, Crm, first_of_run || c->knob_on(KN_CENTERS_FRESH));
  ISLECHK(k_band_build(c));
  TimeScope ts(c, ISLE_T_SPARSE_UPDATE);
  const uint32_t V = (uint32_t)c->V;
  const size_t lds = 4 * (size_t)ldk * sizeof(float);
  hipLaunchKernelGGL(centers_from_rows_k, dim3(cdiv(V, 4)), dim3(256), lds, c->stream, c->bval.p, c->bcol.p, c->seg_off.p, V, c->nbands,
                     assign, ldk, Crm);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// Slot of this thread's item in an append-only list: ONE atomic per workgroup on the list's counter (ballots give the waves'
// counts, thread 0 claims the workgroup's range).  One atomic per wave put 15 600 atomics per pass on a single address at
// D = 1M — they serialise in L2 at ~10 ns each, 150 us of a 177 us kernel whose memory traffic needs 10.  Every thread of the
// workgroup must call it (it synchronises); order inside the list is arbitrary anyway.

// Hamerly's bounds for Lloyd on the sparse matrix (an EXACT acceleration: a document is skipped only when its bounds
// prove that its closest centre cannot have changed).  ub >= distance to the assigned centre, lb <= distance to the
// second-closest (both already widened by the fp32 error of the distance evaluation, see hamerly.h); after the centres
// move by delta[c], ub grows by delta[assigned] and lb shrinks by the largest movement of any other centre.  Documents
// whose bounds overlap are appended to `active` and re-evaluated against all centres.
__global__ __launch_bounds__(256) void hamerly_filter_k(const uint32_t* __restrict__ order, uint32_t D, const uint32_t* __restrict__ assign,
                                                         float* __restrict__ ub, float* __restrict__ lb, const float* __restrict__ delta,
                                                         const HamTop* __restrict__ top, uint32_t* __restrict__ active,
                                                         uint32_t* __restrict__ nactive) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const bool in = i < D;
  const uint32_t amax = top->amax;
  const float d1 = top->d1, d2 = top->d2;
  uint32_t d = 0;
  bool act = false;
  if (in) {
    d = order ? order[i] : i;
    const uint32_t a = assign[d];
    const float u = (ub[d] + delta[a]) * 1.000001f;  // the factors absorb the rounding of these two updates
    float l = lb[d] - (a == amax ? d2 : d1) * 1.000001f;
    l = l > 0.f ? l * 0.999999f : l;
    ub[d] = u;
    lb[d] = l;
    act = u >= l;
  }
  const uint32_t slot = block_append_slot(act, nactive);
  if (act) active[slot] = d;
}
// delta[i] <- rounded-up movement of centre i (from its squared movement); top <- {index of the largest, largest, second largest}.
// One workgroup; replaces a device -> host -> device round trip per Lloyd iteration.
__global__ __launch_bounds__(256) void ham_delta_k(float* __restrict__ delta, int k, HamTop* __restrict__ top) {
  __shared__ float s1[256], s2[256];
  __shared__ uint32_t si[256];
  float d1 = 0.f, d2 = 0.f;
  uint32_t a = 0;
  for (int i = threadIdx.x; i < k; i += 256) {
    const float dv = sqrtf(fmaxf(delta[i], 0.f)) * (1.0f + 1e-5f) + 1e-7f;  // rounded up
    delta[i] = dv;
    if (dv > d1) {
      d2 = d1;
      d1 = dv;
      a = (uint32_t)i;
    } else if (dv > d2) {
      d2 = dv;
    }
  }
  // merge (largest, its first index, second largest): butterfly inside the wave, then the four waves in order — the serial merge
  // of 256 entries by one thread took 25 of this kernel's 30 us, twenty times per step
  auto merge = [](float& x1, float& x2, uint32_t& xi, float y1, float y2, uint32_t yi) {
    if (y1 > x1 || (y1 == x1 && y1 > 0.f && yi < xi)) {
      x2 = fmaxf(x1, y2);
      x1 = y1;
      xi = yi;
    } else {
      x2 = fmaxf(x2, y1);
    }
  };
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float y1 = __shfl_xor(d1, off), y2 = __shfl_xor(d2, off);
    const uint32_t yi = (uint32_t)__shfl_xor((int)a, off);
    merge(d1, d2, a, y1, y2, yi);
  }
  if ((threadIdx.x & 63) == 0) {
    s1[threadIdx.x >> 6] = d1;
    s2[threadIdx.x >> 6] = d2;
    si[threadIdx.x >> 6] = a;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float b1 = s1[0], b2 = s2[0];
    uint32_t bi = si[0];
    for (int w = 1; w < 4; ++w) merge(b1, b2, bi, s1[w], s2[w], si[w]);
    top->amax = bi;
    top->d1 = b1;
    top->d2 = b2;
  }
}
int k_ham_delta(isle_ctx* c, float* delta_dev, int k, HamTop* top_dev) {
  hipLaunchKernelGGL(ham_delta_k, dim3(1), dim3(256), 0, c->stream, delta_dev, k, top_dev);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int k_hamerly_filter(isle_ctx* c, const uint32_t* order, const uint32_t* assign, float* ub, float* lb, const float* delta_dev, const HamTop* top_dev,
                     uint32_t* active, uint32_t* nactive, int fam) {
  TimeScope ts(c, fam);
  const uint32_t D = (uint32_t)c->D;
  HIPCHK(c, hipMemsetAsync(nactive, 0, sizeof(uint32_t), c->stream));
  if (D == 0) return 0;
  hipLaunchKernelGGL(hamerly_filter_k, dim3(cdiv(D, 256)), dim3(256), 0, c->stream, order, D, assign, ub, lb, delta_dev, top_dev, active,
                     nactive);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// Tile bounds for Lloyd in span(U) at large k (Yinyang's group bounds with a group = one 32-centre MFMA tile of
// proj_assign_reg_k).  Hamerly's single lower bound prunes nothing at k = 1000: it falls by the LARGEST centre movement per
// iteration.  Here tlb[d][t] <= distance from d to the closest centre of tile t (other than d's own), lowered per iteration by the
// largest movement INSIDE tile t; a document whose upper bound stays below all of them is skipped, any other re-examines only the
// tiles whose bound it reaches (plus its own centre's), by matrix-core tiles on the compacted active list.  Exact: the
// partition is the full scan's (tests: all bound modes identical).  One thread per document, rows of TL floats.
// (round 5: NQ = TL / 4 is a template parameter — the row's float4 and the movers' products are asked for together, before any of them is
// used; with a run-time trip count a thread walked its row one dependent load at a time: 1.5 -> see DESIGN 14a)
template <int NQ>
__global__ __launch_bounds__(256) void pt_filter_k(const uint32_t* __restrict__ order, uint32_t D, const uint32_t* __restrict__ assign,
                                                    float* __restrict__ ub, float* __restrict__ tlb, int T, const float* __restrict__ delta,
                                                    const float* __restrict__ tmove, uint32_t* __restrict__ need, uint32_t* __restrict__ active,
                                                    uint32_t* __restrict__ nactive, YyMovers mv, const float* __restrict__ mdots /*D x mv.ld: P_d . c_mover*/,
                                                    const float* __restrict__ cn, const float* __restrict__ pn, const uint32_t* __restrict__ dpos /*nullable: the movers' products lie by position (k_gl_thin by_position)*/) {
  constexpr int TL = 4 * NQ;
  __shared__ float tmv[32], mcn[10];
  if (threadIdx.x < 32) tmv[threadIdx.x] = (int)threadIdx.x < T ? tmove[threadIdx.x] * 1.000001f : 0.f;
  if ((int)threadIdx.x < mv.n) mcn[threadIdx.x] = cn[mv.id[threadIdx.x]];
  __syncthreads();
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const bool in = i < D;
  uint32_t d = 0;
  bool act = false;
  if (in) {
    d = order ? order[i] : i;
    float4* row = reinterpret_cast<float4*>(tlb + (size_t)d * TL);
    float4 v[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) v[q] = row[q];
    const uint32_t a = assign[d];
    float u = (ub[d] + delta[a]) * 1.000001f;  // the factors absorb the rounding of the updates
    uint32_t mask = 0u;
    // movers (see YyMovers): centres left out of their tiles' movements; their exact new distances bound their tiles instead.  The
    // bounds of their tiles are collected first (at most ten), applied while the row is walked.
    float ml[10];
    if (mv.n) {
      const float nd = pn[d];
      const float4* mrow = reinterpret_cast<const float4*>(mdots + (size_t)(dpos ? dpos[d] : d) * mv.ld);  // mv.ld <= 12 floats, a run
      float md[12];
#pragma unroll
      for (int q4 = 0; q4 < 3; ++q4) {
        const float4 m4 = 4 * q4 < mv.ld ? mrow[q4] : make_float4(0.f, 0.f, 0.f, 0.f);
        md[4 * q4] = m4.x, md[4 * q4 + 1] = m4.y, md[4 * q4 + 2] = m4.z, md[4 * q4 + 3] = m4.w;
      }
#pragma unroll
      for (int jm = 0; jm < 10; ++jm) {
        ml[jm] = 3.4e38f;
        if (jm < mv.n) {
          const uint32_t cj = mv.id[jm];
          const float dist = fabsf((-2.0f * md[jm] + mcn[jm]) + nd);
          float uu, ll;
          hamerly_store_bounds(dist, dist, nd + mcn[jm], &uu, &ll);
          if (cj != a) ml[jm] = ll;  // the assigned centre does not bound its own tile:
          else u = fminf(u, uu);     // its exact new distance replaces the upper bound grown by its movement
        }
      }
    }
    ub[d] = u;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      float l[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int t = 4 * q + e;
        if (t < T) {
          float x = l[e] - tmv[t];
          x = x > 0.f ? x * 0.999999f : x;
          if (mv.n) {
#pragma unroll
            for (int jm = 0; jm < 10; ++jm)
              if (jm < mv.n && (int)(mv.id[jm] >> 5) == t) x = fminf(x, ml[jm]);
          }
          l[e] = x;
          if (u >= x) mask |= 1u << t;
        }
      }
      row[q] = make_float4(l[0], l[1], l[2], l[3]);
    }
    act = mask != 0u;
    if (act) need[d] = mask | (1u << (a >> 5));  // the own centre's tile is always re-examined: it holds the exact new distance
  }
  const uint32_t slot = block_append_slot(act, nactive);
  if (act) active[slot] = d;
}
// Second stage of the filter: the upper bound of a candidate has grown by its centre's movement in every iteration since the
// document was last examined; here it is replaced by the exact distance to its centre (one 4k-byte dot product per candidate, a
// wave each) and the tile test is repeated.  Most candidates drop out: their centre moved, but they did not get closer to any
// other tile.  A wave takes 64 candidates in turn; lane j keeps the verdict on the j-th for the block-level append.
__global__ __launch_bounds__(256) void pt_tighten_k(const float* __restrict__ P, const float* __restrict__ pn, int ldk, const float* __restrict__ C,
                                                     const float* __restrict__ cn, const uint32_t* __restrict__ assign,
                                                     const uint32_t* __restrict__ cand, const uint32_t* __restrict__ ncand, float* __restrict__ ub,
                                                     const float* __restrict__ tlb, int T, int TL, uint32_t* __restrict__ need,
                                                     uint32_t* __restrict__ active, uint32_t* __restrict__ nactive) {
  const uint32_t n = *ncand;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t base = blockIdx.x * 256 + (uint32_t)wave * 64;
  bool flag = false;
  uint32_t mydoc = 0;
  const int nq = ldk >> 2;
  for (int j = 0; j < 64; ++j) {
    const uint32_t i = base + (uint32_t)j;
    if (i >= n) break;  // wave-uniform
    const uint32_t d = cand[i];
    const uint32_t a = assign[d];
    const float4* pr = reinterpret_cast<const float4*>(P + (size_t)d * ldk);
    const float4* cr = reinterpret_cast<const float4*>(C + (size_t)a * ldk);
    float s = 0.f;
    for (int q = lane; q < nq; q += 64) {
      const float4 x = pr[q], y = cr[q];
      s += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float nd = pn[d], cc = cn[a];
    const float dist = fabsf((-2.0f * s + cc) + nd);
    float uu, ll;
    hamerly_store_bounds(dist, dist, nd + cc, &uu, &ll);
    const float l = lane < T ? tlb[(size_t)d * TL + lane] : 3.4e38f;
    const uint32_t mask = (uint32_t)__ballot(uu >= l);
    if (lane == j) {
      flag = mask != 0u;
      mydoc = d;
    }
    if (lane == 0) {
      ub[d] = uu;
      if (mask) need[d] = mask | (1u << (a >> 5));
    }
  }
  const uint32_t slot = block_append_slot(flag, nactive);
  if (flag) active[slot] = mydoc;
}
// pt_tighten_k with its round trips taken out of the candidates' turns.  There a wave walks cand -> assign -> the two rows for each of
// its 64 candidates, three round trips a candidate (2.0 ms per call at config 3).  Here lane j fetches candidate j's document, centre and
// norms up front (one round for the 64), and the rows of the NEXT candidate are asked for before the current one's are summed: a candidate
// costs one round trip.  Rows of at most 4 x 256 floats (NQ float4 per lane).  Same sums in the same order: same bits.
template <int NQ>
__global__ __launch_bounds__(256) void pt_tighten_ahead_k(const float* __restrict__ P, const float* __restrict__ pn, int ldk, const float* __restrict__ C,
                                                           const float* __restrict__ cn, const uint32_t* __restrict__ assign,
                                                           const uint32_t* __restrict__ cand, const uint32_t* __restrict__ ncand, float* __restrict__ ub,
                                                           const float* __restrict__ tlb, int T, int TL, uint32_t* __restrict__ need,
                                                           uint32_t* __restrict__ active, uint32_t* __restrict__ nactive) {
  const uint32_t n = *ncand;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t base = blockIdx.x * 256 + (uint32_t)wave * 64;
  const int cnt = base < n ? (int)min(64u, n - base) : 0;  // wave-uniform
  bool flag = false;
  uint32_t mydoc = 0;
  const int nq = ldk >> 2;
  uint32_t md = 0, ma = 0;
  float mnd = 0.f, mcc = 0.f;
  if (lane < cnt) {
    md = cand[base + (uint32_t)lane];
    ma = assign[md];
    mnd = pn[md];
    mcc = cn[ma];
  }
  struct Rows {
    float4 x[NQ], y[NQ];
    float l;
  };
  auto fetch = [&](Rows& r, int j) {  // no conditions on the loads (a wait can then count them): past the row's end a lane reads its last float4 again
    const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)md, j), a = (uint32_t)__builtin_amdgcn_readlane((int)ma, j);
    const float4* pr = reinterpret_cast<const float4*>(P + (size_t)d * ldk);
    const float4* cr = reinterpret_cast<const float4*>(C + (size_t)a * ldk);
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
      const int q = min(lane + 64 * t, nq - 1);
      r.x[t] = pr[q];
      r.y[t] = cr[q];
    }
    r.l = tlb[(size_t)d * TL + min(lane, T - 1)];
  };
  auto step = [&](const Rows& cur, Rows& nxt, int j) {  // candidate j < cnt
    fetch(nxt, min(j + 1, cnt - 1));
    __builtin_amdgcn_sched_barrier(0);
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
      if (lane + 64 * t < nq) {
        const float4 x = cur.x[t], y = cur.y[t];
        s += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)md, j), a = (uint32_t)__builtin_amdgcn_readlane((int)ma, j);
    const float nd = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mnd), j));
    const float cc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mcc), j));
    const float dist = fabsf((-2.0f * s + cc) + nd);
    float uu, ll;
    hamerly_store_bounds(dist, dist, nd + cc, &uu, &ll);
    const float l = lane < T ? cur.l : 3.4e38f;
    const uint32_t mask = (uint32_t)__ballot(uu >= l);
    if (lane == j) {
      flag = mask != 0u;
      mydoc = d;
    }
    if (lane == 0) {
      ub[d] = uu;
      if (mask) need[d] = mask | (1u << (a >> 5));
    }
  };
  Rows r0, r1;
  if (cnt) fetch(r0, 0);
  for (int j = 0; j < cnt; j += 2) {
    step(r0, r1, j);
    if (j + 1 < cnt) step(r1, r0, j + 1);
  }
  const uint32_t slot = block_append_slot(flag, nactive);
  if (flag) active[slot] = mydoc;
}
int k_pt_tighten(isle_ctx* c, const float* P, const float* pn, int ldk, const float* C, const float* cn, const uint32_t* assign, const uint32_t* cand,
                 const uint32_t* ncand, float* ub, const float* tlb, int T, int TL, uint32_t* need, uint32_t* active, uint32_t* nactive) {
  TimeScope ts(c, ISLE_T_LLOYD_PROJ);
  const uint32_t D = (uint32_t)c->D;
  HIPCHK(c, hipMemsetAsync(nactive, 0, sizeof(uint32_t), c->stream));
  if (D == 0) return 0;
  const int nq64 = cdiv(ldk >> 2, 64);
  auto kern = nq64 == 1 ? pt_tighten_ahead_k<1> : nq64 == 2 ? pt_tighten_ahead_k<2> : nq64 == 3 ? pt_tighten_ahead_k<3> : nq64 == 4 ? pt_tighten_ahead_k<4> : pt_tighten_k;
  hipLaunchKernelGGL(kern, dim3(cdiv(D, 256)), dim3(256), 0, c->stream, P, pn, ldk, C, cn, assign, cand, ncand, ub, tlb, T, TL, need, active, nactive);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int k_pt_filter(isle_ctx* c, const uint32_t* order, const uint32_t* assign, float* ub, float* tlb, int T, int TL, const float* delta_dev,
                const float* tmove_dev, uint32_t* need, uint32_t* active, uint32_t* nactive, const YyMovers& mv, const float* mdots, const float* cn,
                const float* pn) {
  TimeScope ts(c, ISLE_T_LLOYD_PROJ);
  const uint32_t D = (uint32_t)c->D;
  HIPCHK(c, hipMemsetAsync(nactive, 0, sizeof(uint32_t), c->stream));
  if (D == 0) return 0;
  if (TL < 4 || TL > 32 || (TL & 3) || T > 32) return isle_fail(c, ISLE_E_ARG, "k_pt_filter: rows of %d tile bounds", TL);
#define PF(N) hipLaunchKernelGGL(pt_filter_k<N>, dim3(cdiv(D, 256)), dim3(256), 0, c->stream, order, D, assign, ub, tlb, T, delta_dev, tmove_dev, need, active, \
                                 nactive, mv, mdots, cn, pn, mv.n ? c->dpos.p : nullptr)
  switch (TL / 4) {
    case 1: PF(1); break;
    case 2: PF(2); break;
    case 3: PF(3); break;
    case 4: PF(4); break;
    case 5: PF(5); break;
    case 6: PF(6); break;
    case 7: PF(7); break;
    default: PF(8); break;
  }
#undef PF
  HIPCHK(c, hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// Yinyang bounds for Lloyd on the sparse matrix (Ding et al., ICML 2015) — like Hamerly's an EXACT acceleration, with one
// lower bound per GROUP of YY_GROUP centres instead of one per document: after an update a group's bound only shrinks by
// the largest movement inside that group, and a document that cannot be skipped re-examines only the groups whose bound
// overlaps its upper bound.  A group is 32 bytes of a centre row, so a group scan gathers one 32-byte piece per nonzero
// of the document instead of the whole k-wide row.
// ------------------------------------------------------------------------------------------
// The group bounds of a block of documents lowered by the groups' movements, written back and left in the LDS tile [document][group]: a wave
// takes NU documents at a time — their rows of G floats (500 bytes at k = 1000) are contiguous runs, in the documents' own order or a visiting
// order's — with ALL their loads in flight together (NU x NT per lane).  Round 5: the loop used to keep 8 (visiting order) or 1 (document
// order) load per lane in flight and ran at 1.2 - 1.6 TB/s of its 10 GB; this form is bound by the memory again.
#ifndef YY_NU2
#define YY_NU2 8  // documents a wave lowers at a time at 65 - 128 groups: 16 loads per lane in flight (timing builds: tools/build_variant.sh)
#endif
// The bounds are a stream (each row is read and written once per iteration): non-temporal accesses keep them from pushing the group-major
// centre table — which the tightening phase of the same kernel gathers from — out of the L2
#ifndef YY_NT
#define YY_NT 1
#endif
#if YY_NT
#define YY_NT_LOAD(p) __builtin_nontemporal_load(p)
#define YY_NT_STORE(v, p) __builtin_nontemporal_store(v, p)
#else
#define YY_NT_LOAD(p) (*(p))
#define YY_NT_STORE(v, p) (*(p) = (v))
#endif
template <int NT, int NU>
__device__ inline void yy_lower_block(float* __restrict__ glb, int G, const float* __restrict__ gmax, const uint32_t* __restrict__ order, uint32_t d0,
                                      uint32_t nd, float* __restrict__ tile) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
  float gm[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) gm[t] = lane + 64 * t < G ? gmax[lane + 64 * t] * 1.000001f : 0.f;
  for (uint32_t jb = (uint32_t)w * NU; jb < nd; jb += (uint32_t)nw * NU) {
    uint32_t doc[NU];  // (rows are addressed from the document again when they are written: pointers would double the registers held)
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const uint32_t j = min(jb + u, nd - 1);
      doc[u] = order ? order[d0 + j] : d0 + j;
    }
    float v[NU][NT];
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t) v[u][t] = lane + 64 * t < G ? YY_NT_LOAD(&glb[(size_t)doc[u] * G + lane + 64 * t]) : 0.f;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (jb + u < nd) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int g = lane + 64 * t;
          if (g < G) {
            float l = v[u][t] - gm[t];
            l = l > 0.f ? l * 0.999999f : l;
            YY_NT_STORE(l, &glb[(size_t)doc[u] * G + g]);
            tile[(jb + u) * (uint32_t)G + g] = l;
          }
        }
      }
    }
  }
}
__device__ inline void yy_lower_block_any(float* __restrict__ glb, int G, const float* __restrict__ gmax, const uint32_t* __restrict__ order, uint32_t d0,
                                          uint32_t nd, float* __restrict__ tile) {
  if (G <= 64) yy_lower_block<1, 2 * YY_NU2>(glb, G, gmax, order, d0, nd, tile);
  else if (G <= 128) yy_lower_block<2, YY_NU2>(glb, G, gmax, order, d0, nd, tile);
  else if (G <= 192) yy_lower_block<3, (YY_NU2 + 1) / 2>(glb, G, gmax, order, d0, nd, tile);
  else if (G <= 256) yy_lower_block<4, (YY_NU2 + 1) / 2>(glb, G, gmax, order, d0, nd, tile);
  else {  // beyond 256 groups (k > 2048, the by-document bounds only): element by element
    const uint32_t nel = nd * (uint32_t)G;
    for (uint32_t i = threadIdx.x; i < nel; i += blockDim.x) {
      const uint32_t j = i / (uint32_t)G, g = i - j * (uint32_t)G;
      float* src = glb + (size_t)(order ? order[d0 + j] : d0 + j) * G + g;
      float l = *src - gmax[g] * 1.000001f;
      l = l > 0.f ? l * 0.999999f : l;
      *src = l;
      tile[i] = l;
    }
  }
}
__global__ __launch_bounds__(256) void yy_filter_k(uint32_t D, const uint32_t* __restrict__ order /*nullable: visiting order*/,
                                                    const uint32_t* __restrict__ assign, float* __restrict__ ub, float* __restrict__ glb, int G,
                                                    const float* __restrict__ delta, const float* __restrict__ gmax, uint32_t* __restrict__ active,
                                                    uint32_t* __restrict__ nactive, int docs_per_block) {
  // The group bounds of a block of documents are lowered by the group movements and written back through an LDS tile with
  // coalesced accesses (a thread walking its own document's G floats touches a different cache line per lane: 0.30 ms per call at
  // C2), then one thread per document takes the minimum from the tile (stride G words: conflict-free for odd G, 2-way at worst).
  // Without `order` the block's bounds are one contiguous run of docs x G floats; with it (documents grouped by their centre, so that
  // the active list comes out grouped by Yinyang group: k_yy2_assign) every document's G floats are a run of their own.
  extern __shared__ float tile[];  // docs_per_block x G
  const uint32_t d0 = blockIdx.x * (uint32_t)docs_per_block;
  const uint32_t nd = min((uint32_t)docs_per_block, D - d0);
  yy_lower_block_any(glb, G, gmax, order, d0, nd, tile);
  __syncthreads();
  for (uint32_t j0 = 0; j0 < (uint32_t)docs_per_block; j0 += 256) {  // the same trip count for every thread: block_append_slot synchronises
    const uint32_t j = j0 + threadIdx.x;
    const bool in = j < nd;
    uint32_t d = d0 + (in ? j : 0u);
    if (order) d = order[d];
    float u = 0.f, lmin = 3.4e38f;
    if (in) {
      u = (ub[d] + delta[assign[d]]) * 1.000001f;
      for (int g = 0; g < G; ++g) lmin = fminf(lmin, tile[j * (uint32_t)G + g]);
      ub[d] = u;
    }
    const bool act = in && u >= lmin;
    const uint32_t slot = block_append_slot(act, nactive);
    if (act) active[slot] = d;
  }
}


// top-2 of a group from the 4 distances of this lane and the 4 of lane ^ 1
struct YyTop2 {
  float m1, m2;
  uint32_t i1;
};
// (c0 + j is a SLOT; the centre it holds is map.id(slot) — the slot itself without a regrouping.  Equal distances: the smaller id, whatever
// the slots' order.)
__device__ inline YyTop2 yy_group_top2(const float dist[4], int c0, int k, const YyMap& map = YyMap()) {
  // The four ids are asked for together and used through selects.  (Round 5: they used to be read one by one inside the comparison chain,
  // `if (slot < k) { id = map.id(slot); ... }` — four memory round trips in a row for every group scan; yy2_scan_k 6.2 -> 3.4 ms per call at
  // config 3.)  A slot past k stands as (3.4e38, 0xffffffff): it never wins and leaves m2 as it is.
  uint32_t ids[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) ids[j] = map.id((uint32_t)min(c0 + j, k - 1));
  YyTop2 t{3.4e38f, 3.4e38f, 0xffffffffu};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bool in = c0 + j < k;
    const float dj = in ? dist[j] : 3.4e38f;
    const uint32_t id = in ? ids[j] : 0xffffffffu;
    if (dj < t.m1 || (dj == t.m1 && id < t.i1)) {
      t.m2 = t.m1;
      t.m1 = dj;
      t.i1 = id;
    } else {
      t.m2 = fminf(t.m2, dj);
    }
  }
  const float om1 = __shfl_xor(t.m1, 1), om2 = __shfl_xor(t.m2, 1);
  const uint32_t oi1 = __shfl_xor(t.i1, 1);
  if (om1 < t.m1 || (om1 == t.m1 && oi1 < t.i1)) {
    t.m2 = fminf(t.m1, om2);
    t.m1 = om1;
    t.i1 = oi1;
  } else {
    t.m2 = fminf(t.m2, om1);
  }
  return t;
}

__device__ inline float yy_parity_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x124, 0xf, 0xf, true));  // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x128, 0xf, 0xf, true));  // row_ror:8
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// a document's entries held by the wave (fetched once, not once per scanned group, while it has at most 64 * YY_NCH of them)
constexpr int YY_NCH = 4;
struct YyDoc {
  int64_t beg, end;
  int len;
  bool small;  // wave-uniform
  uint32_t rrow[YY_NCH];
  float rval[YY_NCH];
};
__device__ inline YyDoc yy_load_doc(const float* __restrict__ vals, const uint32_t* __restrict__ rows, const int64_t* __restrict__ offs, uint32_t d,
                                    int lane) {
  YyDoc dc;
  dc.beg = offs[d];
  dc.end = offs[d + 1];
  dc.len = (int)(dc.end - dc.beg);
  dc.small = dc.len <= 64 * YY_NCH;
#pragma unroll
  for (int ch = 0; ch < YY_NCH; ++ch) {
    const bool in = dc.small && 64 * ch + lane < dc.len;
    dc.rrow[ch] = in ? rows[dc.beg + 64 * ch + lane] : 0u;
    dc.rval[ch] = in ? vals[dc.beg + 64 * ch + lane] : 0.f;
  }
  return dc;
}
// distances of the document to the YY_GROUP centres of one group; lane parity q holds centres 8g + 4q .. 8g + 4q + 3.  `base` = the
// group's float4 of row 0 for this parity, `rstride` = row stride in float4 (the row-major centres: ld / 4; the group-major copy: 2)
__device__ inline void yy_group_dists(const YyDoc& dc, const float* __restrict__ vals, const uint32_t* __restrict__ rows, const float4* __restrict__ base,
                                      size_t rstride, float live, int lane, int col, int k, const float* __restrict__ cn, float dnd, float dist[4]) {
  const int e = lane >> 1;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (dc.small) {
    // all gathers of the group in flight together (the loop below keeps two): same products, same order of accumulation
    float4 gv[2 * YY_NCH];
#pragma unroll
    for (int s2 = 0; s2 < 2 * YY_NCH; ++s2) {
      gv[s2] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (32 * s2 < dc.len) {  // wave-uniform
        const uint32_t r = __shfl(dc.rrow[s2 >> 1], (s2 & 1) * 32 + e);
        gv[s2] = base[(size_t)r * rstride];
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < 2 * YY_NCH; ++s2) {
      if (32 * s2 < dc.len) {
        const float v = __shfl(dc.rval[s2 >> 1], (s2 & 1) * 32 + e) * live;  // 0 beyond the document's end
        acc = f4_fma(v, gv[s2], acc);
      }
    }
  } else {
    for (int64_t b0 = dc.beg; b0 < dc.end; b0 += 64) {
      const int cnt = (int)min((int64_t)64, dc.end - b0);
      const uint32_t myrow = (lane < cnt) ? rows[b0 + lane] : 0u;
      const float myval = (lane < cnt) ? vals[b0 + lane] : 0.f;
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        if (st * 32 < cnt) {  // wave-uniform
          const uint32_t r = __shfl(myrow, st * 32 + e);
          const float v = __shfl(myval, st * 32 + e) * live;  // 0 beyond the document's end
          acc = f4_fma(v, base[(size_t)r * rstride], acc);
        }
      }
    }
  }
  // sum over the 32 lanes of equal parity: inside a DPP row by data-parallel moves that keep the parity (swap quad halves,
  // rotate by 4, rotate by 8), across the four rows by two ds_bpermute steps (five of those per value before)
  acc.x = yy_parity_sum(acc.x);
  acc.y = yy_parity_sum(acc.y);
  acc.z = yy_parity_sum(acc.z);
  acc.w = yy_parity_sum(acc.w);
  const float aa[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int cc = min(col + j, k - 1);
    dist[j] = fabsf((-2.0f * aa[j] + cn[cc]) + dnd);
  }
}
__device__ inline float yy_slack_down(float m, float E, float sE) { return yy_slack_down_sq(m, E, sE); }

__global__ __launch_bounds__(256) void yy_scan_k(const float* __restrict__ vals, const uint32_t* __restrict__ rows, const int64_t* __restrict__ offs,
                                                  const float* __restrict__ C /*V x ld row-major*/, const float4* __restrict__ Cg /*nullable: group-major copy*/,
                                                  uint32_t V, int ld, int k, int G, const float* __restrict__ cn,
                                                  const float* __restrict__ dn, const float* __restrict__ cn_max_p, const uint32_t* __restrict__ active,
                                                  const uint32_t* __restrict__ nactive, uint32_t* __restrict__ assign, float* __restrict__ ub,
                                                  float* __restrict__ glb, unsigned long long* __restrict__ dbg /*nullable: [0] group scans, [1] their nonzeros*/,
                                                  YyMap map) {
  const int lane = threadIdx.x & 63;
  uint32_t slot = blockIdx.x * 4 + (threadIdx.x >> 6);
  slot = __builtin_amdgcn_readfirstlane(slot);
  if (slot >= *nactive) return;
  const uint32_t d = __builtin_amdgcn_readfirstlane(active[slot]);
  const float dnd = dn[d];
  const uint32_t a = assign[d];
  const int ga = (int)(map.slot(a) / YY_GROUP);
  const float E = ISLE_SLACK_REL * (dnd + *cn_max_p), sE = sqrtf(E);
  const int q = lane & 1;
  float* gl = glb + (size_t)d * G;
  const YyDoc dc = yy_load_doc(vals, rows, offs, d, lane);
  auto scan_group = [&](int g, float dist[4]) {
    const int col = YY_GROUP * g + 4 * q;
    const float live = (col < ld) ? 1.f : 0.f;
    if (Cg) yy_group_dists(dc, vals, rows, Cg + (size_t)g * V * 2 + q, 2, live, lane, col, k, cn, dnd, dist);  // one 32-byte-row table per group
    else yy_group_dists(dc, vals, rows, reinterpret_cast<const float4*>(C + min(col, ld - 4)), (size_t)ld / 4, live, lane, col, k, cn, dnd, dist);
  };

  float best = 3.4e38f, best_group_second = 3.4e38f;
  uint32_t bidx = 0xffffffffu;
  auto absorb = [&](int g, const YyTop2& t) {
    // provisional bound of the group: its closest centre; the group of the final assignment is fixed up at the end
    if (lane == 0) gl[g] = yy_slack_down(t.m1, E, sE);
    if (t.m1 < best || (t.m1 == best && t.i1 < bidx)) {
      best = t.m1;
      bidx = t.i1;
      best_group_second = t.m2;
    }
  };
  float dist[4];
  int nscan = 1;
  scan_group(ga, dist);                                   // tighten: exact distance to the assigned centre (and its group)
  absorb(ga, yy_group_top2(dist, YY_GROUP * ga + 4 * q, k, map));
  for (int g = 0; g < G; ++g) {
    if (g == ga) continue;
    const float u = sqrtf(best);
    const float uhi = u + fminf(sE, E / fmaxf(u, 1e-30f));
    const float lg = gl[g];                               // already lowered by this update's movement (yy_filter_k)
    if (lg <= uhi) {                                      // wave-uniform
      ++nscan;
      scan_group(g, dist);
      absorb(g, yy_group_top2(dist, YY_GROUP * g + 4 * q, k, map));
    }
  }
  if (dbg && lane == 0) {
    atomicAdd(dbg, (unsigned long long)nscan);
    atomicAdd(dbg + 1, (unsigned long long)nscan * (unsigned long long)dc.len);
  }
  if (lane == 0) {
    const float u = sqrtf(best);
    ub[d] = u + fminf(sE, E / fmaxf(u, 1e-30f));
    assign[d] = bidx;
    gl[map.slot(bidx) / YY_GROUP] = yy_slack_down(best_group_second, E, sE);  // the assigned centre does not bound its own group
  }
}

// ------------------------------------------------------------------------------------------
// The same Yinyang iteration ordered by GROUP instead of by document (large k).  yy_scan_k walks a document's groups one after the
// other, each scan a gather of 32-byte pieces of 4 KB centre rows: at k = 1000 the 400 MB of centres sit in HBM, every piece costs a
// DRAM access and every scan of a document waits for the one before it.  Here
//   (1) the centres are copied group-major (Cg[g][w][8]: one 32-byte-row table of V x 32 bytes per group, 3.2 MB at V = 100k);
//   (2) the active documents arrive grouped by their own group (yy_filter_k over the member lists) and every one scans its OWN group
//       (the tightening step): waves running together gather from one table, which stays in L2;
//   (3) with the upper bound that step leaves, the other groups whose lower bound it reaches become (group, document) pairs — the
//       test Ding et al. make, not narrowed group by group as yy_scan_k does (a superset: exact either way);
//   (4) the pairs are sorted by group (one radix pass) and scanned in that order: again one table at a time in L2;
//   (5) a last pass folds a document's scans together in ascending group order — the assignment (first index among equal minima),
//       the upper bound and the bounds of the scanned groups exactly as yy_scan_k leaves them.
// Per (document, group) the distances are the same sums in the same order as yy_scan_k's (yy_group_dists).
// ------------------------------------------------------------------------------------------
// Cg[(g * V + w) * 2 + q] = the centres of slots 8 g + 4 q .. + 3 at word w (regrouped: centre id_of_slot[slot]; else column = slot, zero beyond ld).
// Round 6: a workgroup takes YYP_W = 16 words — their rows (4 kB each) go to LDS by coalesced float4 loads, and the group-major image leaves as
// runs of 512 bytes (a group's 16 words x 32 bytes).  The thread-per-float4 form wrote every 16-byte piece to a line of its own (consecutive
// threads = consecutive groups, V x 32 bytes apart): 456 us for 0.8 GB at k = 1000.
constexpr int YYP_W = 16;
__global__ __launch_bounds__(256) void yy2_pack_k(const float* __restrict__ Crm, uint32_t V, int ld, int G, float4* __restrict__ Cg, int k,
                                                   const uint32_t* __restrict__ id_of_slot /*nullable*/) {
  extern __shared__ float yyp_rows[];  // YYP_W x ld
  const uint32_t w0 = blockIdx.x * YYP_W;
  const uint32_t nw = min((uint32_t)YYP_W, V - w0);
  const int q4 = ld / 4;  // ld is a multiple of 4 (launcher)
  for (int i = threadIdx.x; i < (int)nw * q4; i += 256) {
    const int r = i / q4, c4 = i - r * q4;
    reinterpret_cast<float4*>(yyp_rows)[(size_t)r * q4 + c4] = reinterpret_cast<const float4*>(Crm + (size_t)(w0 + r) * ld)[c4];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < G * 2 * YYP_W; i += 256) {
    const int g = i / (2 * YYP_W), rem = i - g * (2 * YYP_W), wl = rem >> 1, q = rem & 1;
    if ((uint32_t)wl >= nw) continue;
    const int col = 8 * g + 4 * q;
    const float* row = yyp_rows + (size_t)wl * ld;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (id_of_slot) {
      if (col < k) v.x = row[id_of_slot[col]];
      if (col + 1 < k) v.y = row[id_of_slot[col + 1]];
      if (col + 2 < k) v.z = row[id_of_slot[col + 2]];
      if (col + 3 < k) v.w = row[id_of_slot[col + 3]];
    } else if (col < ld) {
      v = *reinterpret_cast<const float4*>(row + col);
    }
    Cg[((size_t)g * V + w0 + wl) * 2 + q] = v;
  }
}

struct YyRes {  // top two of one group scan
  float m1, m2;
  uint32_t i1;
};

// (2) + (3): wave per active slot.  own[e] = the scan of the document's own group; need[e * NW + j] = bit mask of the other groups to
// scan; cnt[e] = their number
__global__ __launch_bounds__(256) void yy2_tighten_k(const float* __restrict__ vals, const uint32_t* __restrict__ rows, const int64_t* __restrict__ offs,
                                                      const float4* __restrict__ Cg, uint32_t V, int ld, int k, int G, int NW, const float* __restrict__ cn,
                                                      const float* __restrict__ dn, const float* __restrict__ cn_max_p,
                                                      const uint32_t* __restrict__ active, const uint32_t* __restrict__ nactive,
                                                      const uint32_t* __restrict__ assign, const float* __restrict__ glb, YyRes* __restrict__ own,
                                                      unsigned long long* __restrict__ need, uint32_t* __restrict__ cnt, YyMap map) {
  const int lane = threadIdx.x & 63;
  uint32_t slot = blockIdx.x * 4 + (threadIdx.x >> 6);
  slot = __builtin_amdgcn_readfirstlane(slot);
  if (slot >= *nactive) return;
  const uint32_t d = __builtin_amdgcn_readfirstlane(active[slot]);
  const float dnd = dn[d];
  const int ga = (int)(map.slot(assign[d]) / YY_GROUP);
  const float E = ISLE_SLACK_REL * (dnd + *cn_max_p), sE = sqrtf(E);
  const int q = lane & 1;
  const YyDoc dc = yy_load_doc(vals, rows, offs, d, lane);
  const int col = YY_GROUP * ga + 4 * q;
  float dist[4];
  yy_group_dists(dc, vals, rows, Cg + (size_t)ga * V * 2 + q, 2, (col < ld) ? 1.f : 0.f, lane, col, k, cn, dnd, dist);
  const YyTop2 t = yy_group_top2(dist, col, k, map);
  const float u = sqrtf(t.m1);
  const float uhi = u + fminf(sE, E / fmaxf(u, 1e-30f));
  const float* gl = glb + (size_t)d * G;
  uint32_t total = 0;
  for (int j = 0; j < NW; ++j) {
    const int g = 64 * j + lane;
    const bool nd = g < G && g != ga && gl[g] <= uhi;
    const unsigned long long m = __ballot(nd);
    total += (uint32_t)__popcll(m);
    if (lane == 0) need[(size_t)slot * NW + j] = m;
  }
  if (lane == 0) {
    own[slot] = YyRes{t.m1, t.m2, t.i1};
    cnt[slot] = total;
  }
}

// V x n column-major block of n columns of a row-major V x ld matrix (the movers' centres as the thin operand of k_gl_thin)
__global__ __launch_bounds__(256) void yy_gather_cols_k(const float* __restrict__ Crm, uint32_t V, int ld, YyMovers mv, float* __restrict__ Wcm) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)V * mv.n) return;
  const int j = (int)(i / V);
  const uint32_t w = (uint32_t)(i - (size_t)j * V);
  Wcm[i] = Crm[(size_t)w * ld + mv.id[j]];
}

// yy_filter_k and yy2_tighten_k in one launch (the by-group iteration): a workgroup lowers the group bounds of its block of documents
// through the LDS tile as yy_filter_k does — same arithmetic, same outputs: glb, ub, the active list — and then its waves take the block's
// ACTIVE documents in turn and run yy2_tighten_k's step on them with the bounds still in the tile: the D x G bound array (5 GB at 10 M
// documents and k = 1000) is read once per iteration instead of twice.  own / need / cnt are indexed by the active slot as before.
__global__ __launch_bounds__(256) void yy2_filter_tighten_k(uint32_t D, const uint32_t* __restrict__ order /*nullable*/, const uint32_t* __restrict__ assign,
                                                             float* __restrict__ ub, float* __restrict__ glb, int G, const float* __restrict__ delta,
                                                             const float* __restrict__ gmax, uint32_t* __restrict__ active, uint32_t* __restrict__ nactive,
                                                             int docs_per_block, const float* __restrict__ vals, const uint32_t* __restrict__ rows,
                                                             const int64_t* __restrict__ offs, const float4* __restrict__ Cg, uint32_t V, int ld, int k, int NW,
                                                             const float* __restrict__ cn, const float* __restrict__ dn, const float* __restrict__ cn_max_p,
                                                             YyRes* __restrict__ own, unsigned long long* __restrict__ need, uint32_t* __restrict__ cnt,
                                                             YyMovers mv, const float* __restrict__ mdots /*D x mv.ld: b_d . c_mover*/, YyMap map,
                                                             const float* __restrict__ cn_by_id /*the movers' norms (cn is indexed by slot)*/,
                                                             const uint32_t* __restrict__ dpos /*document -> row of mdots (k_gl_thin by_position)*/) {
  extern __shared__ float tile[];  // docs_per_block x G bounds, then docs_per_block local indices of the active documents
  uint32_t* lact = reinterpret_cast<uint32_t*>(tile + (size_t)docs_per_block * G);
  const uint32_t d0 = blockIdx.x * (uint32_t)docs_per_block;
  const uint32_t nd = min((uint32_t)docs_per_block, D - d0);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  yy_lower_block_any(glb, G, gmax, order, d0, nd, tile);
  __syncthreads();
  uint32_t base = 0, total = 0;
  for (uint32_t j0 = 0; j0 < (uint32_t)docs_per_block; j0 += 256) {  // one trip (docs_per_block <= 256): the append synchronises
    const uint32_t j = j0 + threadIdx.x;
    const bool in = j < nd;
    uint32_t d = d0 + (in ? j : 0u);
    if (order) d = order[d];
    float u = 0.f, lmin = 3.4e38f;
    if (in) {
      const uint32_t a = assign[d];
      u = (ub[d] + delta[a]) * 1.000001f;
      if (mv.n) {
        // the few centres that moved far were left out of their groups' movements (gmax): their groups' bounds take the exact new
        // distances to them instead — min(bound lowered by the others' movement, distance to the mover) bounds the group as before
        const float dnd = dn[d];
        const float E = ISLE_SLACK_REL * (dnd + *cn_max_p), sE = sqrtf(E);
        for (int jm = 0; jm < mv.n; ++jm) {
          const uint32_t cj = mv.id[jm];
          const float dist = fabsf((-2.0f * mdots[(size_t)dpos[d] * mv.ld + jm] + cn_by_id[cj]) + dnd);
          if (cj == a) {
            // the assigned centre does not bound its own group; its exact new distance replaces the upper bound grown by its movement
            // (a cluster of 1.5 M documents whose centre moves by 0.02 per iteration made every one of them active: config 3)
            const float x = sqrtf(dist);
            u = fminf(u, x + fminf(sE, E / fmaxf(x, 1e-30f)));
            continue;
          }
          const float l = yy_slack_down_sq(dist, E, sE);
          const uint32_t gj = map.slot(cj) >> 3;
          const uint32_t at = j * (uint32_t)G + gj;
          if (l < tile[at]) {
            tile[at] = l;
            glb[(size_t)d * G + gj] = l;
          }
        }
      }
      for (int g = 0; g < G; ++g) lmin = fminf(lmin, tile[j * (uint32_t)G + g]);
      ub[d] = u;
    }
    const bool act = in && u >= lmin;
    const uint32_t slot = block_append_slot_range(act, nactive, &base, &total);
    if (act) {
      active[slot] = d;
      lact[slot - base] = j;
    }
  }
  __syncthreads();
  // second phase: the tightening step of the block's active documents, a wave each in turn (yy2_tighten_k with the bounds read from the tile)
  for (uint32_t i = (uint32_t)w; i < total; i += 4) {
    const uint32_t j = lact[i], slot = base + i;
    const uint32_t d = __builtin_amdgcn_readfirstlane(order ? order[d0 + j] : d0 + j);
    const float dnd = dn[d];
    const int ga = (int)(map.slot(assign[d]) / YY_GROUP);
    const float E = ISLE_SLACK_REL * (dnd + *cn_max_p), sE = sqrtf(E);
    const int q = lane & 1;
    const YyDoc dc = yy_load_doc(vals, rows, offs, d, lane);
    const int col = YY_GROUP * ga + 4 * q;
    float dist[4];
    yy_group_dists(dc, vals, rows, Cg + (size_t)ga * V * 2 + q, 2, (col < ld) ? 1.f : 0.f, lane, col, k, cn, dnd, dist);
    const YyTop2 t = yy_group_top2(dist, col, k, map);
    const float u = sqrtf(t.m1);
    const float uhi = u + fminf(sE, E / fmaxf(u, 1e-30f));
    uint32_t tot = 0;
    for (int jj = 0; jj < NW; ++jj) {
      const int g = 64 * jj + lane;
      const bool ndd = g < G && g != ga && tile[j * (uint32_t)G + min(g, G - 1)] <= uhi;
      const unsigned long long m = __ballot(ndd);
      tot += (uint32_t)__popcll(m);
      if (lane == 0) need[(size_t)slot * NW + jj] = m;
    }
    if (lane == 0) {
      own[slot] = YyRes{t.m1, t.m2, t.i1};
      cnt[slot] = tot;
    }
  }
}

// pairs of slot e at off[e] ..: key = document << 8 | group (sorted on the low 8 bits only: stable, so a group's pairs keep the slot order),
// val = pair index, pgrp = group (ascending inside a slot)
__global__ __launch_bounds__(256) void yy2_emit_k(const uint32_t* __restrict__ nactive, const uint32_t* __restrict__ active,
                                                   const unsigned long long* __restrict__ need, int NW,
                                                   const uint32_t* __restrict__ off, uint64_t* __restrict__ key, uint32_t* __restrict__ val,
                                                   uint8_t* __restrict__ pgrp) {
  const uint32_t e = blockIdx.x * 256 + threadIdx.x;
  if (e >= *nactive) return;
  const uint64_t d = active[e];
  uint32_t p = off[e];
  for (int j = 0; j < NW; ++j) {
    unsigned long long m = need[(size_t)e * NW + j];
    while (m) {
      const int b = __ffsll((long long)m) - 1;
      m &= m - 1;
      key[p] = (d << 8) | (uint64_t)(64 * j + b);
      val[p] = p;
      pgrp[p] = (uint8_t)(64 * j + b);
      ++p;
    }
  }
}

// (4): wave per pair, in group order
__global__ __launch_bounds__(256) void yy2_scan_k(const float* __restrict__ vals, const uint32_t* __restrict__ rows, const int64_t* __restrict__ offs,
                                                   const float4* __restrict__ Cg, uint32_t V, int ld, int k, const float* __restrict__ cn,
                                                   const float* __restrict__ dn, uint32_t npairs, const uint64_t* __restrict__ key,
                                                   const uint32_t* __restrict__ val, YyRes* __restrict__ res, YyMap map) {
  const int lane = threadIdx.x & 63;
  // the grid is capped (a launch of more than 2^32 threads does not run: 2^26 pairs at four per workgroup): waves stride over the pairs
  for (uint64_t i64 = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i64 < npairs; i64 += (uint64_t)gridDim.x * 4) {
    const uint32_t i = __builtin_amdgcn_readfirstlane((uint32_t)i64);
    const uint64_t kd = key[i];
    const int g = (int)__builtin_amdgcn_readfirstlane((uint32_t)(kd & 0xffu));
    const uint32_t d = __builtin_amdgcn_readfirstlane((uint32_t)(kd >> 8));
    const uint32_t p = __builtin_amdgcn_readfirstlane(val[i]);
    const float dnd = dn[d];
    const int q = lane & 1;
    const YyDoc dc = yy_load_doc(vals, rows, offs, d, lane);
    const int col = YY_GROUP * g + 4 * q;
    float dist[4];
    yy_group_dists(dc, vals, rows, Cg + (size_t)g * V * 2 + q, 2, (col < ld) ? 1.f : 0.f, lane, col, k, cn, dnd, dist);
    const YyTop2 t = yy_group_top2(dist, col, k, map);
    if (lane == 0) res[p] = YyRes{t.m1, t.m2, t.i1};
  }
}

// (5): thread per active slot
__global__ __launch_bounds__(256) void yy2_combine_k(const uint32_t* __restrict__ nactive, const uint32_t* __restrict__ active, const YyRes* __restrict__ own,
                                                      const uint32_t* __restrict__ off, const uint8_t* __restrict__ pgrp,
                                                      const YyRes* __restrict__ res, const float* __restrict__ dn, const float* __restrict__ cn_max_p, int G,
                                                      uint32_t* __restrict__ assign, float* __restrict__ ub, float* __restrict__ glb, YyMap map) {
  const uint32_t e = blockIdx.x * 256 + threadIdx.x;
  if (e >= *nactive) return;
  const uint32_t d = active[e];
  const float E = ISLE_SLACK_REL * (dn[d] + *cn_max_p), sE = sqrtf(E);
  float* gl = glb + (size_t)d * G;
  const int ga = (int)(map.slot(assign[d]) / YY_GROUP);
  float best = 3.4e38f, best_group_second = 3.4e38f;
  uint32_t bidx = 0xffffffffu;
  auto absorb = [&](int g, const YyRes& t) {
    gl[g] = yy_slack_down(t.m1, E, sE);
    if (t.m1 < best || (t.m1 == best && t.i1 < bidx)) {
      best = t.m1;
      bidx = t.i1;
      best_group_second = t.m2;
    }
  };
  absorb(ga, own[e]);
  for (uint32_t p = off[e]; p < off[e + 1]; ++p) absorb((int)pgrp[p], res[p]);
  const float u = sqrtf(best);
  ub[d] = u + fminf(sE, E / fmaxf(u, 1e-30f));
  assign[d] = bidx;
  gl[map.slot(bidx) / YY_GROUP] = yy_slack_down(best_group_second, E, sE);  // the assigned centre does not bound its own group
}

// ---- regrouped Yinyang groups (YyMap, common.h): maps on the device, values and rows by slot, labels back to ids
__global__ __launch_bounds__(256) void yy_gather_by_slot_k(const float* __restrict__ v, const uint32_t* __restrict__ id_of_slot, int k, int n, float* __restrict__ out) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s < n) out[s] = s < k ? v[id_of_slot[s]] : 0.f;
}
__global__ __launch_bounds__(256) void yy_rows_by_slot_k(const float* __restrict__ in, int ld, int k, const uint32_t* __restrict__ id_of_slot, float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)k * ld) return;
  const int s = (int)(i / ld), j = (int)(i - (size_t)s * ld);
  out[i] = in[(size_t)id_of_slot[s] * ld + j];
}
__global__ __launch_bounds__(256) void yy_labels_to_ids_k(uint32_t* __restrict__ assign, uint64_t D, const uint32_t* __restrict__ id_of_slot) {
  for (uint64_t d = (uint64_t)blockIdx.x * 256 + threadIdx.x; d < D; d += (uint64_t)gridDim.x * 256) assign[d] = id_of_slot[assign[d]];
}
int k_yy_map_upload(isle_ctx* c, const uint32_t* id_of_slot_host, const uint32_t* slot_of_id_host, int k, int G, YyMap* map) {
  HIPCHK(c, c->yy_map.reserve((size_t)8 * G + k));
  HIPCHK(c, hipMemcpyAsync(c->yy_map.p, id_of_slot_host, (size_t)8 * G * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->yy_map.p + 8 * G, slot_of_id_host, (size_t)k * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));  // the host arrays are the caller's
  map->id_of_slot = c->yy_map.p;
  map->slot_of_id = c->yy_map.p + 8 * G;
  return 0;
}
int k_yy_gather_by_slot(isle_ctx* c, const YyMap& map, int k, int G, const float* cn, float* cn_slot) {
  hipLaunchKernelGGL(yy_gather_by_slot_k, dim3(cdiv(8 * G, 256)), dim3(256), 0, c->stream, cn, map.id_of_slot, k, 8 * G, cn_slot);
  HIPCHK(c, hipGetLastError());
  return 0;
}
int k_yy_rows_by_slot(isle_ctx* c, const YyMap& map, int k, const float* in, int ld, float* out) {
  hipLaunchKernelGGL(yy_rows_by_slot_k, dim3(cdiv((long)((size_t)k * ld), 256)), dim3(256), 0, c->stream, in, ld, k, map.id_of_slot, out);
  HIPCHK(c, hipGetLastError());
  return 0;
}
int k_yy_labels_to_ids(isle_ctx* c, const YyMap& map, uint32_t* assign, uint64_t D) {
  if (!D) return 0;
  hipLaunchKernelGGL(yy_labels_to_ids_k, dim3((unsigned)std::min<uint64_t>((D + 255) / 256, 1u << 20)), dim3(256), 0, c->stream, assign, D, map.id_of_slot);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int k_yy_filter(isle_ctx* c, const uint32_t* order, const uint32_t* assign, float* ub, float* glb, int G, const float* delta_dev, const float* gmax_dev,
                uint32_t* active, uint32_t* nactive) {
  TimeScope ts(c, ISLE_T_SPARSE_ASSIGN);
  const uint32_t D = (uint32_t)c->D;
  HIPCHK(c, hipMemsetAsync(nactive, 0, sizeof(uint32_t), c->stream));
  if (D == 0) return 0;
  int dpb = 256;
  while (dpb > 32 && (size_t)dpb * G * sizeof(float) > 32 * 1024) dpb /= 2;  // LDS tile of at most 32 KB: five workgroups per CU
  hipLaunchKernelGGL(yy_filter_k, dim3(cdiv(D, dpb)), dim3(256), (size_t)dpb * G * sizeof(float), c->stream, D, order, assign, ub, glb, G, delta_dev,
                     gmax_dev, active, nactive, dpb);
  HIPCHK(c, hipGetLastError());
  return 0;
}
// k_yy_filter and the tightening step of k_yy2_assign in one launch (yy2_filter_tighten_k); k_yy2_assign(..., pre_tightened = true) follows
int k_yy_filter_tighten(isle_ctx* c, const uint32_t* order, const uint32_t* assign, float* ub, float* glb, int G, const float* delta_dev,
                        const float* gmax_dev, uint32_t* active, uint32_t* nactive, const float* Cg, int k, int ld, const float* cn, const float* dn,
                        const float* cn_max, const YyMovers& mv, const float* Crm, const YyMap& map, const float* cn_by_id) {
  TimeScope ts(c, ISLE_T_SPARSE_ASSIGN);
  const uint32_t D = (uint32_t)c->D, V = (uint32_t)c->V;
  if (mv.n && D) {  // b_d . c_mover for every document: one thin pass of the pass-1 stream (k_gl_thin), ten columns at most
    HIPCHK(c, c->Tmp.reserve((size_t)V * mv.n));
    HIPCHK(c, c->yy_mdots.reserve((size_t)D * mv.ld));
    hipLaunchKernelGGL(yy_gather_cols_k, dim3(cdiv((long)((size_t)V * mv.n), 256)), dim3(256), 0, c->stream, Crm, V, ld, mv, c->Tmp.p);
    HIPCHK(c, hipGetLastError());
    ISLECHK(k_gl_thin(c, c->Tmp.p, mv.n, mv.ld, c->yy_mdots.p, true));  // rows by position: read through dpos
  }
  HIPCHK(c, hipMemsetAsync(nactive, 0, sizeof(uint32_t), c->stream));
  if (D == 0) return 0;
  const int NW = cdiv(G, 64);
  HIPCHK(c, c->yy_own.reserve((size_t)D * 3));
  HIPCHK(c, c->yy_need.reserve((size_t)D * NW));
  HIPCHK(c, c->yy_cnt.reserve((size_t)D + 1));
  HIPCHK(c, hipMemsetAsync(c->yy_cnt.p, 0, ((size_t)D + 1) * sizeof(uint32_t), c->stream));
  // blocks of 64 documents at k = 1000 (a 32 KB tile): five workgroups = twenty waves per CU for the gathers of the second phase
  int dpb = 256;
  while (dpb > 32 && (size_t)dpb * (G + 1) * sizeof(float) > 32 * 1024) dpb /= 2;
  const size_t lds = (size_t)dpb * (G + 1) * sizeof(float);
  ISLECHK(isle_max_lds(c, (const void*)yy2_filter_tighten_k, (int)lds));
  hipLaunchKernelGGL(yy2_filter_tighten_k, dim3(cdiv(D, dpb)), dim3(256), lds, c->stream, D, order, assign, ub, glb, G, delta_dev, gmax_dev, active, nactive, dpb,
                     c->vals.p, c->rows.p, c->offs.p, (const float4*)Cg, V, ld, k, NW, cn, dn, cn_max, (YyRes*)c->yy_own.p, (unsigned long long*)c->yy_need.p,
                     c->yy_cnt.p, mv, c->yy_mdots.p, map, cn_by_id ? cn_by_id : cn, c->dpos.p);
  HIPCHK(c, hipGetLastError());
  return 0;
}
// diagnostic (ISLE_DEBUG_HAMERLY): why the active documents are active — the group whose bound the upper bound reaches first (own or another)
// and by how much: hist[own ? 0 : 1][b], b = decade of (ub - smallest bound): < 1e-6, < 1e-5, ... , < 1e-1, >= 1e-1
__global__ __launch_bounds__(256) void yy_dbg_margins_k(const uint32_t* __restrict__ active, const uint32_t* __restrict__ nactive,
                                                         const uint32_t* __restrict__ assign, const float* __restrict__ ub, const float* __restrict__ glb,
                                                         int G, YyMap map, unsigned long long* __restrict__ hist /*2 x 8*/) {
  const uint32_t e = blockIdx.x * 256 + threadIdx.x;
  if (e >= *nactive) return;
  const uint32_t d = active[e];
  const float* row = glb + (size_t)d * G;
  float lmin = 3.4e38f;
  int gmin = 0;
  for (int g = 0; g < G; ++g)
    if (row[g] < lmin) lmin = row[g], gmin = g;
  const int ga = (int)(map.slot(assign[d]) / YY_GROUP);
  const float margin = ub[d] - lmin;
  int b = 0;
  for (float t = 1e-6f; b < 7 && margin >= t; t *= 10.f) ++b;
  atomicAdd(&hist[(gmin == ga ? 0 : 8) + b], 1ull);
}
int k_yy_dbg_margins(isle_ctx* c, const uint32_t* active, const uint32_t* nactive, const uint32_t* assign, const float* ub, const float* glb, int G,
                     const YyMap& map, unsigned long long* hist_host /*16*/) {
  HIPCHK(c, c->dbg_cnt.reserve(18));
  HIPCHK(c, hipMemsetAsync(c->dbg_cnt.p + 2, 0, 16 * sizeof(unsigned long long), c->stream));
  hipLaunchKernelGGL(yy_dbg_margins_k, dim3(cdiv((long)c->D, 256)), dim3(256), 0, c->stream, active, nactive, assign, ub, glb, G, map, c->dbg_cnt.p + 2);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpy(hist_host, c->dbg_cnt.p + 2, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return 0;
}
int k_yy_pack_groups(isle_ctx* c, const float* Crm, int ld, int G, const YyMap& map) {
  TimeScope ts(c, ISLE_T_SPARSE_ASSIGN);
  const uint32_t V = (uint32_t)c->V;
  HIPCHK(c, c->yy_cg.reserve((size_t)V * 8 * G));
  if ((ld & 3) != 0 || ((uintptr_t)Crm & 15) != 0) return isle_fail(c, ISLE_E_ARG, "yy_pack_groups: leading dimension %d not a multiple of 4", ld);
  const size_t lds = (size_t)YYP_W * ld * sizeof(float);
  ISLECHK(isle_max_lds(c, (const void*)yy2_pack_k, (int)lds));
  hipLaunchKernelGGL(yy2_pack_k, dim3(cdiv(V, YYP_W)), dim3(256), lds, c->stream, Crm, V, ld, G, (float4*)c->yy_cg.p, c->centers_k, map.id_of_slot);
  HIPCHK(c, hipGetLastError());
  return 0;
}
int k_yy_scan(isle_ctx* c, const float* Crm, const float* Cg, int k, int ld, int G, const float* cn, const float* dn, const float* cn_max, const uint32_t* active,
              const uint32_t* nactive, uint32_t* assign, float* ub, float* glb, unsigned long long* dbg, const YyMap& map) {
  TimeScope ts(c, ISLE_T_SPARSE_ASSIGN);
  const uint32_t D = (uint32_t)c->D;
  if (D == 0) return 0;
  hipLaunchKernelGGL(yy_scan_k, dim3(cdiv(D, 4)), dim3(256), 0, c->stream, c->vals.p, c->rows.p, c->offs.p, Crm, (const float4*)Cg, (uint32_t)c->V, ld, k, G,
                     cn, dn, cn_max, active, nactive, assign, ub, glb, dbg, map);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// The Yinyang iteration ordered by group (see the kernels): active -> assign / ub / glb.  *done = false if the pair list would be too
// long (more than 48 pairs per document) — nothing has been changed then and the caller runs yy_scan_k instead.
int k_yy2_assign(isle_ctx* c, const float* Cg, int k, int ld, int G, const float* cn, const float* dn, const float* cn_max, const uint32_t* active,
                 const uint32_t* nactive, uint32_t* assign, float* ub, float* glb, bool* done, unsigned long long* pairs_out, bool pre_tightened,
                 const YyMap& map) {
  TimeScope ts(c, ISLE_T_SPARSE_ASSIGN);
  *done = false;
  const uint32_t D = (uint32_t)c->D, V = (uint32_t)c->V;
  if (D == 0 || G > 256) return 0;
  if ((uint64_t)D * (uint64_t)(G - 1) >= (1ull << 32)) return 0;  // pair counts and offsets are 32-bit: the by-document form takes over
  const int NW = cdiv(G, 64);
  HIPCHK(c, c->yy_own.reserve((size_t)D * 3));
  HIPCHK(c, c->yy_need.reserve((size_t)D * NW));
  HIPCHK(c, c->yy_cnt.reserve((size_t)D + 1));
  HIPCHK(c, c->yy_off.reserve((size_t)D + 2));
  HIPCHK(c, c->gl_scan.reserve(isle_scan::scan_scratch_elems(D) + 8));
  if (!pre_tightened) {  // k_yy_filter_tighten has done this step with the bounds it had in LDS
    HIPCHK(c, hipMemsetAsync(c->yy_cnt.p, 0, ((size_t)D + 1) * sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(yy2_tighten_k, dim3(cdiv(D, 4)), dim3(256), 0, c->stream, c->vals.p, c->rows.p, c->offs.p, (const float4*)Cg, V, ld, k, G, NW, cn, dn,
                       cn_max, active, nactive, assign, glb, (YyRes*)c->yy_own.p, (unsigned long long*)c->yy_need.p, c->yy_cnt.p, map);
    HIPCHK(c, hipGetLastError());
  }
  HIPCHK(c, (isle_scan::exclusive_scan<uint32_t, uint32_t>(c->stream, c->yy_cnt.p, D, c->yy_off.p, reinterpret_cast<uint32_t*>(c->gl_scan.p))));
  uint32_t npairs = 0;
  HIPCHK(c, hipMemcpyAsync(&npairs, c->yy_off.p + D, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (pairs_out) *pairs_out = npairs;
  if ((uint64_t)npairs > 48ull * D) return 0;
  if (npairs) {
    const size_t np = npairs;
    // the pair count of the first iteration follows the seeds (± 20 % from step to step at config 3): grown with half as much again, or
    // every step that sets a new maximum frees and allocates several GB in the middle of the loop (82 ms of idle GPU in a profiled step)
    if (np > c->yy_pgrp.cap) {
      const size_t cap = np + np / 2;
      HIPCHK(c, c->gl_key_a.reserve(cap));
      HIPCHK(c, c->gl_key_b.reserve(cap));
      HIPCHK(c, c->gl_val_a.reserve(cap));
      HIPCHK(c, c->gl_val_b.reserve(cap));
      HIPCHK(c, c->yy_pgrp.reserve(cap));
      HIPCHK(c, c->yy_res.reserve(cap * 3));
    }
    hipLaunchKernelGGL(yy2_emit_k, dim3(cdiv(D, 256)), dim3(256), 0, c->stream, nactive, active, (const unsigned long long*)c->yy_need.p, NW, c->yy_off.p,
                       c->gl_key_a.p, c->gl_val_a.p, c->yy_pgrp.p);
    HIPCHK(c, hipGetLastError());
    bool in_a = true;
    ISLECHK(k_sort_pairs_u64(c, c->gl_key_a.p, c->gl_val_a.p, c->gl_key_b.p, c->gl_val_b.p, np, 8, &in_a));
    hipLaunchKernelGGL(yy2_scan_k, dim3((unsigned)std::min<size_t>(cdiv((long)np, 4), (size_t)1 << 22)), dim3(256), 0, c->stream, c->vals.p, c->rows.p, c->offs.p, (const float4*)Cg, V, ld, k, cn, dn,
                       npairs, in_a ? c->gl_key_a.p : c->gl_key_b.p, in_a ? c->gl_val_a.p : c->gl_val_b.p, (YyRes*)c->yy_res.p, map);
    HIPCHK(c, hipGetLastError());
  }
  hipLaunchKernelGGL(yy2_combine_k, dim3(cdiv(D, 256)), dim3(256), 0, c->stream, nactive, active, (const YyRes*)c->yy_own.p, c->yy_off.p, c->yy_pgrp.p,
                     (const YyRes*)c->yy_res.p, dn, cn_max, G, assign, ub, glb, map);
  HIPCHK(c, hipGetLastError());
  *done = true;
  return 0;
}
