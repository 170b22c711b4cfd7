// isle_amd/csrc/dense.hip — tall-skinny panel kernels, f32 MFMA GEMM, small symmetric EVD (gfx950).
//
//   k_vtf / k_update      H = V^T F ; F -= V H        block-ks/restarted_block_ks.h:83-91 (CGS + 2x DGKS)
//   k_panel_qr            panel QR as CholQR2 on an fp64 Gram matrix, all on the device; stands in for the fp64 MGS of
//                         utils::compute_qr  block-ks/ks_utils.h:43-127  (rank test kept: see solver.cpp)
//   k_gemm_nn             C = A B, exact-f32 MFMA (v_mfma_f32_32x32x2_f32): Ritz rotation
//                         block-ks/restarted_block_ks.h:166-167 and lift src/sparseMatrix.cpp:1446-1449
//   k_jacobi_eig          one-sided (Hestenes) Jacobi in fp64: arma::eig_sym -> ssyevd
//                         block-ks/restarted_block_ks.h:150-161
#include <algorithm>
#include <cstdlib>

#include "gemm_f32.h"
#include "gemm_bf16x3.h"

#include "common.h"
#include "gridbar.h"
#include "hamerly.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------
// V^T F : partial products per (row chunk, column group) in double, then a fixed-order reduce.
// ------------------------------------------------------------------------------------------
constexpr int VTF_CG = 32;    // basis columns per workgroup (8 per wave)
// rows per chunk: the F panel chunk [BT][RC] must fit the 64 KB static LDS limit
constexpr int vtf_rc(int BT) { return BT <= 12 ? 1024 : (BT <= 16 ? 512 : 256); }

template <int BT>
__global__ __launch_bounds__(256) void vtf_partial_k(const float* __restrict__ Vb, uint64_t n, uint64_t ld, int m, const float* __restrict__ F,
                                                      int b, double* __restrict__ part /*[chunk][m][BT]*/) {
  constexpr int VTF_RC = vtf_rc(BT);
  __shared__ float Fs[BT][VTF_RC];
  const uint64_t r0 = (uint64_t)blockIdx.x * VTF_RC;
  const int rc = (int)min((uint64_t)VTF_RC, n - r0);
  for (int idx = threadIdx.x; idx < BT * VTF_RC; idx += 256) {
    const int j = idx / VTF_RC, r = idx - j * VTF_RC;
    Fs[j][r] = (j < b && r < rc) ? F[(uint64_t)j * ld + r0 + r] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int ci = wave; ci < VTF_CG; ci += 4) {
    const int i = blockIdx.y * VTF_CG + ci;
    if (i >= m) break;
    const float* v = Vb + (uint64_t)i * ld + r0;
    double acc[BT];
#pragma unroll
    for (int j = 0; j < BT; ++j) acc[j] = 0.0;
    for (int r = lane; r < rc; r += 64) {
      const double x = (double)v[r];
#pragma unroll
      for (int j = 0; j < BT; ++j) acc[j] = fma(x, (double)Fs[j][r], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < BT; ++j) {
      double s = acc[j];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
      acc[j] = s;
    }
    if (lane == 0) {
      double* o = part + ((size_t)blockIdx.x * m + i) * BT;
#pragma unroll
      for (int j = 0; j < BT; ++j) o[j] = acc[j];
    }
  }
}

// Matrix-core variant for the orthogonalisation coefficients (b <= 16): v_mfma_f32_16x16x4_f32 with basis columns
// on the MFMA row index, panel columns on the MFMA column index and the V rows as the contraction index.  A wave owns
// 16 basis columns of a 1024-row chunk; per 64 rows every lane loads four float4 of its basis column from memory and four of its
// panel column from LDS (MFMA step s pairs element s of both operands, so any row order inside a group is consistent) and issues
// 16 MFMAs on two alternating accumulators.  fp32 inside a chunk, fp64 across chunks (vtf_reduce_k).  The fp64 Gram
// matrix of the panel QR keeps the fp64 VALU kernel above.
typedef float floatx4 __attribute__((ext_vector_type(4)));
// What bounds it (round 6, profiles/r06_u_ortho_kernels.txt): not the loads in flight per wave (2, 8 or 16 float4: the same time) and not the
// occupancy — the vector L1's tag path.  A load of this shape is 16 column pieces of 64 bytes = 16 tag look-ups for 1 KB.
constexpr int VM_RC = 1024;           // most rows of a chunk; a short panel (a rank's row slice: V / N rows) takes 512 or 256 (vm_rows_per_chunk)
constexpr int VM_WAVES = 8;           // waves (16 basis columns each) of a workgroup: they share the staged panel chunk
#ifndef VM_NG_V
#define VM_NG_V 2
#endif
constexpr int VM_NG = VM_NG_V;        // 64-row groups per trip of the main loop
// floats between the panel's columns in LDS = rows per chunk + 4: the 16 columns of a ds_read_b128 quarter lie 16 bytes apart mod 256
// The panel chunk (b columns x 1024 rows, 41 KB at b = 10) is staged in LDS once per workgroup (round 6).  Read from global memory by every wave it
// doubled the wave's load instructions, and a load instruction of this shape — 16 columns x 64 bytes — is 16 tag look-ups in the vector L1
// (≈ 4 clk each: 64 clk for 1 KB, ≈ 4.9 TB/s chip-wide for the basis when the panel takes half of them; the kernel ran at 4.6).  Same MFMAs on the same
// operands in the same order as before: same bits.
__global__ __launch_bounds__(64 * VM_WAVES) void vtf_mfma_k(const float* __restrict__ Vb, uint64_t n, uint64_t ld, int m, const float* __restrict__ F, int b,
                                                            int BT, double* __restrict__ part /*[chunk][m][BT]*/, int rc /*rows per chunk*/) {
  extern __shared__ float vm_fs[];  // [b][rc + 4], rows beyond the chunk's end are zeros
  const int VM_FS = rc + 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const uint64_t r0 = (uint64_t)blockIdx.x * rc;
  const uint64_t r1 = min(n, r0 + rc);
  const bool aligned = (ld & 3) == 0 && (((uintptr_t)Vb | (uintptr_t)F) & 15) == 0;  // columns are 16-byte aligned
  for (int idx = threadIdx.x; idx < b * (rc / 4); idx += 64 * VM_WAVES) {
    const int j = idx / (rc / 4), q4 = idx - j * (rc / 4);
    const uint64_t rr = r0 + 4 * (uint64_t)q4;
    const float* src = F + (uint64_t)j * ld + rr;
    float4 v;
    if (aligned && rr + 4 <= r1) v = *(const float4*)src;
    else {
      v.x = rr < r1 ? src[0] : 0.f;
      v.y = rr + 1 < r1 ? src[1] : 0.f;
      v.z = rr + 2 < r1 ? src[2] : 0.f;
      v.w = rr + 3 < r1 ? src[3] : 0.f;
    }
    *(float4*)(vm_fs + (size_t)j * VM_FS + 4 * q4) = v;
  }
  __syncthreads();
  const int col0 = (blockIdx.y * VM_WAVES + wave) * 16;
  if (col0 >= m) return;
  const float* va = Vb + (uint64_t)min(col0 + l15, m - 1) * ld;  // clamped: rows of H beyond m are not written
  // (lanes l15 >= b read column b - 1 again: column j of the B operand only reaches column j of the tile, and those are never read)
  const float* fs = vm_fs + (size_t)min(l15, b - 1) * VM_FS;
  floatx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  uint64_t r = r0;
  if (aligned) {
    // VM_NG groups of 64 rows per trip: all their basis loads are issued before the first MFMA (the compiler, left alone, keeps two in flight
    // per wave: with 24 waves a CU that is 12 MB in flight chip-wide, one memory latency at 5.4 TB/s)
    for (; r + 64 * VM_NG <= r1; r += 64 * VM_NG) {
      float4 av[4 * VM_NG];
#pragma unroll
      for (int u = 0; u < 4 * VM_NG; ++u) av[u] = *(const float4*)(va + r + 16 * u + 4 * g);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4 * VM_NG; ++u) {
        const float4 fv = *(const float4*)(fs + (r - r0) + 16 * u + 4 * g);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, fv.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, fv.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, fv.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, fv.w, acc1, 0, 0, 0);
      }
    }
    for (; r + 64 <= r1; r += 64) {  // the groups a trip does not fill
      float4 av[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) av[u] = *(const float4*)(va + r + 16 * u + 4 * g);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 fv = *(const float4*)(fs + (r - r0) + 16 * u + 4 * g);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, fv.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, fv.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, fv.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, fv.w, acc1, 0, 0, 0);
      }
    }
  }
  for (; r < r1; r += 4) {  // tail / unaligned: 4 rows per MFMA, rows beyond the end contribute zeros
    const uint64_t rr = r + g;
    const float a = rr < r1 ? va[rr] : 0.f;
    const float f = fs[min(rr, r1 - 1) - r0];
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, f, acc0, 0, 0, 0);
  }
  // C/D layout of the 16x16 tile: column j = lane & 15, row i = 4 * (lane >> 4) + reg
  if (l15 < BT) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int i = col0 + 4 * g + reg;
      if (i < m) part[((size_t)blockIdx.x * m + i) * BT + l15] = (double)(acc0[reg] + acc1[reg]);
    }
  }
}

// coef[j][i] = sum over chunks (fixed order) of part[chunk][i][j]; thread = one (i, j) of the slab, chunk slabs are read coalesced
template <class Tout>
__global__ __launch_bounds__(256) void vtf_reduce_k(const double* __restrict__ part, int nchunks, int m, int BT, int b, Tout* __restrict__ coef) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int slab = m * BT;
  if (idx >= slab) return;
  double s = 0.0;
  // 32 chunks' loads in flight per batch (98 chunks at V = 100 000: four dependent batches instead of thirteen of 8; the additions keep their order)
#pragma unroll 32
  for (int ch = 0; ch < nchunks; ++ch) s += part[(size_t)ch * slab + idx];
  const int i = idx / BT, j = idx - i * BT;
  if (j < b) coef[(size_t)j * m + i] = (Tout)s;
}

static int bt_of(int b) { return b <= 4 ? 4 : b <= 8 ? 8 : b <= 12 ? 12 : b <= 16 ? 16 : 32; }

template <class Tout>
static int vtf_impl(isle_ctx* c, const float* Vb, uint64_t n, uint64_t ld, int m, const float* F, int b, Tout* out_dev) {
  if (b > 32 || b < 1) return isle_fail(c, ISLE_E_ARG, "block width %d not in [1,32]", b);
  const int BT = bt_of(b);
  const int nchunks = cdiv(n, vtf_rc(BT));
  HIPCHK(c, c->part.reserve((size_t)nchunks * m * BT));
  dim3 g(nchunks, cdiv(m, VTF_CG)), blk(256);
  switch (BT) {
    case 4: hipLaunchKernelGGL(vtf_partial_k<4>, g, blk, 0, c->stream, Vb, n, ld, m, F, b, c->part.p); break;
    case 8: hipLaunchKernelGGL(vtf_partial_k<8>, g, blk, 0, c->stream, Vb, n, ld, m, F, b, c->part.p); break;
    case 12: hipLaunchKernelGGL(vtf_partial_k<12>, g, blk, 0, c->stream, Vb, n, ld, m, F, b, c->part.p); break;
    case 16: hipLaunchKernelGGL(vtf_partial_k<16>, g, blk, 0, c->stream, Vb, n, ld, m, F, b, c->part.p); break;
    default: hipLaunchKernelGGL(vtf_partial_k<32>, g, blk, 0, c->stream, Vb, n, ld, m, F, b, c->part.p); break;
  }
  HIPCHK(c, hipGetLastError());
  hipLaunchKernelGGL(vtf_reduce_k<Tout>, dim3(cdiv((long)m * BT, 256)), dim3(256), 0, c->stream, c->part.p, nchunks, m, BT, b, out_dev);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// Vb, F: column-major with leading dimension ld (0 = n); n rows take part (a row slice when the step is sharded over ranks)
int k_vtf(isle_ctx* c, const float* Vb, uint64_t n, int m, const float* F, int b, float* coef, uint64_t ld) {
  TimeScope ts(c, ISLE_T_ORTHO);
  if (ld == 0) ld = n;
  if (n == 0) {  // an empty slice contributes zeros
    HIPCHK(c, hipMemsetAsync(coef, 0, (size_t)m * b * sizeof(float), c->stream));
    return 0;
  }
  if (b <= 16 && m >= 64) {  // matrix-core path (enough columns to fill the chip)
    const int BT = bt_of(b);
    // Rows per chunk: 1024, or — a rank's row slice under the row-sharded step, V / N rows — 512 / 256 while that leaves at most 128 chunks
    // (vtf_reduce_k walks them one after another) and the grid would not reach two workgroups per CU otherwise.  At n = 12 500 (N = 8) the
    // 1024-row chunks gave 13 x 16 workgroups of eight dependent trips each: the call took as long as at n = 100 000 / 3 (round 6,
    // tools/ortho_slice_probe.py).  One GPU at V = 100 000: 1024 at every basis width, as before.
    int rc = VM_RC;
    while (rc > 256 && cdiv(n, rc / 2) <= 128 && (long)cdiv(n, rc) * cdiv(m, 16 * VM_WAVES) < 2L * c->num_cus) rc /= 2;
    const int nch = cdiv(n, rc);
    HIPCHK(c, c->part.reserve((size_t)nch * m * BT));
    const size_t lds = (size_t)b * (rc + 4) * sizeof(float);
    ISLECHK(isle_max_lds(c, (const void*)vtf_mfma_k, (int)lds));
    hipLaunchKernelGGL(vtf_mfma_k, dim3(nch, cdiv(m, 16 * VM_WAVES)), dim3(64 * VM_WAVES), lds, c->stream, Vb, n, ld, m, F, b, BT, c->part.p, rc);
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(vtf_reduce_k<float>, dim3(cdiv((long)m * BT, 256)), dim3(256), 0, c->stream, c->part.p, nch, m, BT, b, coef);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  return vtf_impl<float>(c, Vb, n, ld, m, F, b, coef);
}

// ------------------------------------------------------------------------------------------
// F[r, :] -= sum_i Vb[r, i] * coef[i, :]        one thread per row, coef tiles staged in LDS
// ------------------------------------------------------------------------------------------
// Workgroup = 64 rows x 4 waves; wave q takes the basis columns i = q (mod 4) blocks of UPD_TILE / 4, so a 50k-row panel
// spreads over ~3000 waves instead of ~800 (the loop is a chain of dependent-address loads: latency-bound at low occupancy).
// The four partial sums of a row are added in fixed order.
constexpr int UPD_TILE = 128;
template <int BT>
__global__ __launch_bounds__(256) void update_k(float* __restrict__ F, uint64_t n, uint64_t ld, int b, const float* __restrict__ Vb, int m,
                                                 const float* __restrict__ coef /*m x b col-major*/) {
  __shared__ float Cs[UPD_TILE][BT];
  __shared__ float red[3][64][BT + 1];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const uint64_t r = (uint64_t)blockIdx.x * 64 + lane;
  const bool live = r < n;
  float acc[BT];
#pragma unroll
  for (int j = 0; j < BT; ++j) acc[j] = 0.f;
  for (int i0 = 0; i0 < m; i0 += UPD_TILE) {
    const int cnt = min(UPD_TILE, m - i0);
    __syncthreads();
    for (int idx = threadIdx.x; idx < UPD_TILE * BT; idx += 256) {
      const int ii = idx / BT, j = idx - ii * BT;
      Cs[ii][j] = (ii < cnt && j < b) ? coef[(size_t)j * m + i0 + ii] : 0.f;
    }
    __syncthreads();
    if (live) {
      const int ib = q * (UPD_TILE / 4), ie = min(cnt, ib + UPD_TILE / 4);
      const float* v = Vb + (uint64_t)i0 * ld + r;
#pragma unroll 8
      for (int ii = ib; ii < ie; ++ii) {
        const float x = v[(uint64_t)ii * ld];
#pragma unroll
        for (int j = 0; j < BT; ++j) acc[j] = fmaf(x, Cs[ii][j], acc[j]);
      }
    }
  }
  if (q > 0) {
#pragma unroll
    for (int j = 0; j < BT; ++j) red[q - 1][lane][j] = acc[j];
  }
  __syncthreads();
  if (q == 0 && live) {
#pragma unroll
    for (int j = 0; j < BT; ++j)
      if (j < b) F[(uint64_t)j * ld + r] -= ((acc[j] + red[0][lane][j]) + red[1][lane][j]) + red[2][lane][j];
  }
}

// The same update on the matrix cores (round 5; b <= 16, columns 16-byte aligned).  update_k reads three ds_read_b128 of coefficients for
// every float of the basis it loads (137 us a call at config 3); here (94 us) a wave takes 64 rows: a lane loads FOUR consecutive rows of a
// basis column as one float4 (a wave instruction = 4 columns x 256 bytes), component c of the 16 lanes of a quarter is the A operand of
// tile c (rows r0 + 4 i + c, i < 16), the coefficients are the B operand (one cached 16-byte load per sixteen MFMAs since round 6), v_mfma_f32_16x16x4_f32 (exact
// f32 products and sums).  The four waves of a workgroup take the basis columns i = 32 q .. 32 q + 31 of every 128-column tile, eight
// float4 loads in flight each (UM_SETS = 1: four; the same time), and their sums are added in the order q = 0 .. 3: deterministic, other
// rounding than update_k's chains.

#ifndef UM_SETS_V
#define UM_SETS_V 2
#endif
constexpr int UM_SETS = UM_SETS_V;  // sets of 16 columns whose loads a wave issues together (1 or 2)
template <int NQ>  // waves of a workgroup (4; 8 / 16 on a short panel: a rank's row slice), each takes 32 of every 32 NQ basis columns
__global__ __launch_bounds__(64 * NQ) void update_mfma_k(float* __restrict__ F, uint64_t n, uint64_t ld, int b, const float* __restrict__ Vb, int m,
                                                         const float* __restrict__ coef /*m x b col-major*/) {
  extern __shared__ float um_red[];  // [NQ][64][17]
  float (*red)[64][17] = reinterpret_cast<float (*)[64][17]>(um_red);
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const uint64_t r0 = (uint64_t)blockIdx.x * 64;
  // (ld is a multiple of 4 and n <= ld: a float4 that starts on a row below n stays inside its column's stride)
  const uint64_t rb = min(r0 + 4 * (uint64_t)l15, (n - 1) & ~(uint64_t)3);
  // the coefficients are the B operand, one float per lane and MFMA group (row k0 + g of column l15): read straight from the cached
  // m x b block — no staging in LDS, no barrier in the loop, the waves run free like vtf_mfma_k's (with the coefficients staged per 128 columns behind two barriers:
  // 128 us a call at config 3; this form: 94)
  const float* cf = coef + (size_t)min(l15, b - 1) * m;
  // (lanes l15 >= b read column b - 1 again: column j of the B operand only reaches column j of the tiles, and those are never stored)
  floatx4 acc[4];
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4) acc[c4] = floatx4{0.f, 0.f, 0.f, 0.f};
  // A trip = the wave's 32 columns of a 128-column tile, two sets of 16; in a set, MFMA step s takes the columns base + 4 g + s (g = the lane's quarter):
  // a lane's four coefficients of a set are CONSECUTIVE in its column of coef, one 16-byte load per 16 MFMAs (round 6; until then the step
  // took the columns base + 4 s + g and every MFMA quadruple had a dword load of its own — a wave instruction over 16 cache lines, 64 clk of
  // tag look-ups in the vector L1 beside the 32 clk of the basis load it fed: 4.25 TB/s).  All loads of a trip are issued before its first MFMA.
  for (int i0 = 0; i0 < m; i0 += 32 * NQ) {
    const int kb = i0 + 32 * q;
    if (kb < m) {  // wave-uniform
#pragma unroll
      for (int h0 = 0; h0 < 2; h0 += UM_SETS) {  // UM_SETS sets' loads in flight together
        float4 av[4 * UM_SETS], bq[UM_SETS];
#pragma unroll
        for (int h = 0; h < UM_SETS; ++h) {
          const int k0 = kb + 16 * (h0 + h) + 4 * g;  // the lane's four columns of this set
#pragma unroll
          for (int s = 0; s < 4; ++s) av[4 * h + s] = *reinterpret_cast<const float4*>(Vb + (uint64_t)min(k0 + s, m - 1) * ld + rb);  // clamped: meets a zero
          if (k0 + 3 < m) __builtin_memcpy(&bq[h], cf + k0, 16);  // (4-byte aligned: m need not be a multiple of 4)
          else bq[h] = float4{k0 < m ? cf[k0] : 0.f, k0 + 1 < m ? cf[k0 + 1] : 0.f, k0 + 2 < m ? cf[k0 + 2] : 0.f, 0.f};
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < UM_SETS; ++h) {
          const float bb[4] = {bq[h].x, bq[h].y, bq[h].z, bq[h].w};
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[4 * h + s].x, bb[s], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[4 * h + s].y, bb[s], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[4 * h + s].z, bb[s], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[4 * h + s].w, bb[s], acc[3], 0, 0, 0);
          }
        }
      }
    }
  }
  // C/D layout of a 16 x 16 tile: column j = lane & 15, row i = 4 (lane >> 4) + reg; tile c's row i is row r0 + 4 i + c of the block
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) red[q][16 * g + 4 * reg + c4][l15] = acc[c4][reg];
  __syncthreads();
  const uint64_t r = r0 + lane;
  if (r < n)
    for (int j = q; j < b; j += NQ) {
      float s = red[0][lane][j];
#pragma unroll
      for (int qq = 1; qq < NQ; ++qq) s += red[qq][lane][j];  // q = 0, 1, ...: the order of the four-wave form's ((r0 + r1) + r2) + r3
      F[(uint64_t)j * ld + r] -= s;
    }
}

int k_update(isle_ctx* c, float* F, uint64_t n, int b, const float* Vb, int m, const float* coef, uint64_t ld) {
  TimeScope ts(c, ISLE_T_ORTHO);
  if (ld == 0) ld = n;
  if (n == 0) return 0;
  const int BT = bt_of(b);
  dim3 g(cdiv(n, 64)), blk(256);
  if (b <= 16 && m >= 32 && (ld & 3) == 0 && (((uintptr_t)Vb | (uintptr_t)F) & 15) == 0 && !c->knob_zero(KN_UPDATE_MFMA)) {
    // waves per workgroup: a workgroup is 64 rows, so a short panel (a rank's row slice) leaves CUs empty and every wave a long chain of
    // dependent trips (n = 12 500: 196 workgroups x 16 trips at m = 2010) — there the columns are dealt over 8 or 16 waves instead of 4
    const long wgs = cdiv(n, 64);
    const int nq = wgs <= c->num_cus ? 16 : wgs <= 2L * c->num_cus ? 8 : 4;
    const size_t lds = (size_t)nq * 64 * 17 * sizeof(float);
    switch (nq) {
      case 16:
        ISLECHK(isle_max_lds(c, (const void*)update_mfma_k<16>, (int)lds));
        hipLaunchKernelGGL(update_mfma_k<16>, g, dim3(1024), lds, c->stream, F, n, ld, b, Vb, m, coef);
        break;
      case 8: hipLaunchKernelGGL(update_mfma_k<8>, g, dim3(512), lds, c->stream, F, n, ld, b, Vb, m, coef); break;
      default: hipLaunchKernelGGL(update_mfma_k<4>, g, blk, lds, c->stream, F, n, ld, b, Vb, m, coef); break;
    }
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  switch (BT) {
    case 4: hipLaunchKernelGGL(update_k<4>, g, blk, 0, c->stream, F, n, ld, b, Vb, m, coef); break;
    case 8: hipLaunchKernelGGL(update_k<8>, g, blk, 0, c->stream, F, n, ld, b, Vb, m, coef); break;
    case 12: hipLaunchKernelGGL(update_k<12>, g, blk, 0, c->stream, F, n, ld, b, Vb, m, coef); break;
    case 16: hipLaunchKernelGGL(update_k<16>, g, blk, 0, c->stream, F, n, ld, b, Vb, m, coef); break;
    default: hipLaunchKernelGGL(update_k<32>, g, blk, 0, c->stream, F, n, ld, b, Vb, m, coef); break;
  }
  HIPCHK(c, hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// Panel QR entirely on the device: rank-revealing CholQR2 on an fp64 Gram matrix (stands in for the fp64 MGS of
// utils::compute_qr, block-ks/ks_utils.h:43-127, same drop rule :66-69).  Per pass: pqr_gram_k (partial Gram matrices
// of 256-row slabs, fp64) -> pqr_factor_k (fixed-order reduction, Cholesky with column dropping, triangular inverse;
// pass 2 also forms R = R2 R1) -> pqr_apply_k (Q = F T).  The rank lives in device memory between the kernels; the host
// reads rank, status and R once at the end.
// ------------------------------------------------------------------------------------------
constexpr int PQ_ROWS = 256;
constexpr int PQ_W = 32;  // widest panel
constexpr int PQ_SUB = 2;  // 256-row slabs per workgroup of pqr_gram_k

__global__ __launch_bounds__(256) void pqr_gram_k(const float* __restrict__ F, uint64_t n, int wmax, const int* __restrict__ w_dev,
                                                   double* __restrict__ part /*[block][PQ_W*PQ_W]*/) {
  __shared__ float Ft[PQ_W][PQ_ROWS + 1];
  const int w = w_dev ? *w_dev : wmax;
  // this thread's outputs (upper triangle only); at most 4 for w = 32
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int sub = 0; sub < PQ_SUB; ++sub) {
    const uint64_t r0 = ((uint64_t)blockIdx.x * PQ_SUB + sub) * PQ_ROWS;
    if (r0 >= n) break;
    const int rc = (int)min((uint64_t)PQ_ROWS, n - r0);
    __syncthreads();
    for (int idx = threadIdx.x; idx < w * PQ_ROWS; idx += 256) {
      const int j = idx / PQ_ROWS, r = idx - j * PQ_ROWS;
      Ft[j][r] = (r < rc) ? F[(uint64_t)j * n + r0 + r] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int o = threadIdx.x + 256 * q;
      if (o >= w * w) break;
      const int i = o / w, j = o - i * w;
      if (j < i) continue;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;  // four independent chains, combined in a fixed order
#pragma unroll 4
      for (int r = 0; r < PQ_ROWS; r += 4) {
        s0 = fma((double)Ft[i][r], (double)Ft[j][r], s0);
        s1 = fma((double)Ft[i][r + 1], (double)Ft[j][r + 1], s1);
        s2 = fma((double)Ft[i][r + 2], (double)Ft[j][r + 2], s2);
        s3 = fma((double)Ft[i][r + 3], (double)Ft[j][r + 3], s3);
      }
      acc[q] += (s0 + s1) + (s2 + s3);
    }
  }
  double* out = part + (size_t)blockIdx.x * (PQ_W * PQ_W);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int o = threadIdx.x + 256 * q;
    if (o >= w * w) break;
    const int i = o / w, j = o - i * w;
    if (j >= i) out[j * w + i] = acc[q];
  }
}

// meta: [0] rank, [1] status (0 ok, 1 second Gram matrix not positive definite), [2..2+PQ_W) pivots
constexpr int PQ_NSEG = 4;                      // the partial Gram matrices are summed in four runs of consecutive slabs, combined (0 + 1) + (2 + 3)
constexpr int PQ_NE = PQ_W * (PQ_W + 1) / 2;   // entries of the upper triangle at the widest panel
struct PqFactorLds {
  double G[PQ_W][PQ_W + 1];
  union {
    struct {
      double Rm[PQ_W][PQ_W + 1];  // Rm[r][col]: row r of the triangular factor
      double X[PQ_W][PQ_W + 1];
    };
    double S[PQ_NSEG][PQ_NE];     // the four runs' sums, before Rm and X are in use
  };
  int piv[PQ_W];
  int sh_rk, sh_bad;
  double sh_nrm;
};
// The factor step of one CholQR pass for a workgroup of 256 threads.
template <bool COH>  // COH: the partial sums were stored by other workgroups of this launch (agent-scope loads, see gridbar.h); unused since round 6
__device__ inline double pq_ld(const double* p) {
  return COH ? gb_ld(p) : *p;
}
// the LDS stores of this wave's earlier instructions are visible to its later loads (one wave's LDS accesses execute in order); the
// fences keep the compiler from moving accesses across the point
__device__ inline void pq_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <bool COH>
__device__ inline void pq_factor(PqFactorLds& L, const double* part, int nparts, int wmax, int pass, double* R1g /*PQ_W*PQ_W*/, int* meta,
                                 float* T /*PQ_W*PQ_W*/, float* Rout) {
  auto& G = L.G;
  auto& Rm = L.Rm;
  auto& X = L.X;
  auto& piv = L.piv;
  int& sh_rk = L.sh_rk;
  int& sh_bad = L.sh_bad;
  const int t = threadIdx.x;
  const int w = (pass == 1) ? wmax : meta[0];
  {
    // Sum of the partial Gram matrices: only the upper triangle (the slabs store nothing else), the slabs cut into PQ_NSEG runs of
    // consecutive indices, a thread per (run, entry) with up to 48 loads in flight, each run summed in index order and the runs combined
    // in a fixed order — a function of (nparts, w) alone, so every workgroup of the fused form and every rank gets the same bits.  (Until
    // round 6: all w x w entries, two runs, 16 loads in flight: a chain of seven dependent load batches at 196 slabs, most of the kernel.)
    const int ne = w * (w + 1) / 2;
    for (int it = t; it < ne * PQ_NSEG; it += 256) {
      const int seg = it / ne, en = it - seg * ne;
      // entry en of the upper triangle in column-major order: column j holds rows 0 .. j
      int j = 0;
      while ((j + 1) * (j + 2) / 2 <= en) ++j;
      const int i = en - j * (j + 1) / 2;
      const int u = j * w + i;
      const int p0 = (int)((long)nparts * seg / PQ_NSEG), p1 = (int)((long)nparts * (seg + 1) / PQ_NSEG);
      double sacc = 0.0;
      int p = p0;
      for (; p + 48 <= p1; p += 48) {  // a run of the 196 slabs of V = 100 000 is 49 long: one batch and one straggler, not four batches
        double v[48];
#pragma unroll
        for (int q = 0; q < 48; ++q) v[q] = pq_ld<COH>(&part[(size_t)(p + q) * (PQ_W * PQ_W) + u]);
#pragma unroll
        for (int q = 0; q < 48; ++q) sacc += v[q];
      }
      for (; p + 32 <= p1; p += 32) {
        double v[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) v[q] = pq_ld<COH>(&part[(size_t)(p + q) * (PQ_W * PQ_W) + u]);
#pragma unroll
        for (int q = 0; q < 32; ++q) sacc += v[q];
      }
      for (; p + 8 <= p1; p += 8) {
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = pq_ld<COH>(&part[(size_t)(p + q) * (PQ_W * PQ_W) + u]);
#pragma unroll
        for (int q = 0; q < 8; ++q) sacc += v[q];
      }
      for (; p < p1; ++p) sacc += pq_ld<COH>(&part[(size_t)p * (PQ_W * PQ_W) + u]);
      L.S[seg][en] = sacc;
    }
    __syncthreads();
    for (int en = t; en < ne; en += 256) {
      int j = 0;
      while ((j + 1) * (j + 2) / 2 <= en) ++j;
      const int i = en - j * (j + 1) / 2;
      const double g = (L.S[0][en] + L.S[1][en]) + (L.S[2][en] + L.S[3][en]);
      G[i][j] = g;
      G[j][i] = g;
    }
    __syncthreads();
  }
  for (int o = t; o < PQ_W * PQ_W; o += 256) {
    Rm[o / PQ_W][o % PQ_W] = 0.0;
    X[o / PQ_W][o % PQ_W] = 0.0;
    T[o] = 0.f;
  }
  if (t == 0) {
    sh_rk = 0;
    sh_bad = 0;
  }
  __syncthreads();
  // Cholesky with column dropping and the triangular inverse: 2 w dependent steps, by wave 0 ALONE (w <= 32 columns = lanes) — the steps
  // hand their values on through readlane and wave-ordered LDS accesses instead of two workgroup barriers each (round 6; the arithmetic and
  // its order are those of rounds 1 - 5)
  if (t < 64) {
    int rk = 0, bad = 0;
    for (int i = 0; i < w; ++i) {
      double tj = 0.0;
      if (t < w) {
        tj = G[i][t];
        for (int r = 0; r < rk; ++r) tj -= Rm[r][i] * Rm[r][t];
      }
      const double s = __shfl(tj, i);  // the pivot candidate: lane i's value, known to every lane
      double nrm = s > 0.0 ? sqrt(s) : 0.0;
      bool drop;
      if (pass == 1) {
        // ks_utils.h:66-69 absolute test, plus a relative guard for what an fp64 Gram matrix can resolve
        drop = ((float)nrm < 1e-6f) || (s <= 1e-13 * G[i][i] * (double)w);
      } else {
        drop = false;
        if (!(s > 0.0)) {
          bad = 1;
          nrm = 1.0;
        }
      }
      if (!drop && nrm != 0.0) {
        if (t < w) Rm[rk][t] = (t >= i) ? tj / nrm : 0.0;
        if (t == 0) piv[rk] = i;
        ++rk;
      }
      pq_wave_sync();
    }
    if (t == 0) {
      sh_rk = rk;
      sh_bad = bad;
    }
    // X = inverse of the rk x rk upper-triangular U(a, b) = Rm[a][piv[b]]; lane = column
    if (t < rk) {
      const int j = t;
      X[j][j] = 1.0 / Rm[j][piv[j]];
      for (int i = j - 1; i >= 0; --i) {
        double sacc = 0.0;
        for (int q = i + 1; q <= j; ++q) sacc += Rm[i][piv[q]] * X[q][j];
        X[i][j] = -sacc / Rm[i][piv[i]];
      }
    }
  }
  __syncthreads();
  const int rk = sh_rk;
  // T (w x rk, col-major): rows at the pivot columns hold X
  for (int o = t; o < rk * rk; o += 256) {
    const int cc = o / rk, a = o - cc * rk;
    T[cc * w + piv[a]] = (float)X[a][cc];
  }
  if (pass == 1) {
    for (int o = t; o < PQ_W * PQ_W; o += 256) R1g[o] = Rm[o / PQ_W][o % PQ_W];  // R1g[r * PQ_W + col]
    if (t == 0) {
      meta[0] = rk;
      meta[1] = 0;
    }
    if (t < PQ_W) meta[2 + t] = (t < rk) ? piv[t] : -1;
  } else {
    // R (rk x w1, [j*rk + r]) = R2 R1;  w1 = wmax
    for (int o = t; o < rk * wmax; o += 256) {
      const int j = o / rk, r = o - j * rk;
      double s = 0.0;
      for (int q = r; q < rk; ++q) s += Rm[r][q] * R1g[q * PQ_W + j];
      Rout[o] = (float)s;
    }
    if (t == 0) meta[1] = sh_bad;
  }
}
__global__ __launch_bounds__(256) void pqr_factor_k(const double* __restrict__ part, int nparts, int wmax, int pass, double* __restrict__ R1g /*PQ_W*PQ_W*/,
                                                     int* __restrict__ meta, float* __restrict__ T /*PQ_W*PQ_W*/, float* __restrict__ Rout) {
  __shared__ PqFactorLds L;
  pq_factor<false>(L, part, nparts, wmax, pass, R1g, meta, T, Rout);
}

// Q[r, 0:rk] = F[r, 0:b] * T (b x rk col-major), rk (and b in pass 2) read from device memory.  Q may alias F.
__global__ __launch_bounds__(256) void pqr_apply_k(const float* F, uint64_t n, int bmax, int pass, const float* __restrict__ T, const int* __restrict__ meta,
                                                    float* Q) {
  __shared__ float Ts[PQ_W * PQ_W];
  const int rk = meta[0];
  const int b = (pass == 1) ? bmax : rk;
  for (int idx = threadIdx.x; idx < b * rk; idx += 256) Ts[idx] = T[idx];
  __syncthreads();
  const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  float f[PQ_W];
#pragma unroll
  for (int j = 0; j < PQ_W; ++j) f[j] = (j < b) ? F[(uint64_t)j * n + r] : 0.f;
  for (int cc = 0; cc < rk; ++cc) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < PQ_W; ++j)
      if (j < b) s = fmaf(f[j], Ts[cc * b + j], s);
    Q[(uint64_t)cc * n + r] = s;
  }
}

// pqr_apply_k of pass 1 fused with pqr_gram_k of pass 2: a workgroup forms Q1 = F T for its PQ_SUB slabs of 256 rows, writes them, and
// accumulates their partial Gram matrix from the copy it keeps in LDS — one launch and one sweep over Q1 fewer per panel QR
// (61 per step at C2, 300 at a C3 shard; the chain is latency-bound).  Same arithmetic and summation order as the two kernels.
__global__ __launch_bounds__(256) void pqr_apply_gram_k(const float* F, uint64_t n, int b, const float* __restrict__ T, const int* __restrict__ meta,
                                                         float* Q, double* __restrict__ part /*[block][PQ_W*PQ_W]*/) {
  __shared__ float Ts[PQ_W * PQ_W];
  __shared__ float Ft[PQ_W][PQ_ROWS + 1];
  const int rk = meta[0];
  const int w = rk;
  for (int idx = threadIdx.x; idx < b * rk; idx += 256) Ts[idx] = T[idx];
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int sub = 0; sub < PQ_SUB; ++sub) {
    const uint64_t r0 = ((uint64_t)blockIdx.x * PQ_SUB + sub) * PQ_ROWS;
    if (r0 >= n) break;
    __syncthreads();  // Ts is staged; the previous slab's Ft has been consumed
    const uint64_t r = r0 + threadIdx.x;
    const bool live = r < n;
    float f[PQ_W];
#pragma unroll
    for (int j = 0; j < PQ_W; ++j) f[j] = (j < b && live) ? F[(uint64_t)j * n + r] : 0.f;
    for (int cc = 0; cc < rk; ++cc) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < PQ_W; ++j)
        if (j < b) s = fmaf(f[j], Ts[cc * b + j], s);
      if (live) Q[(uint64_t)cc * n + r] = s;
      Ft[cc][threadIdx.x] = live ? s : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int o = threadIdx.x + 256 * q;
      if (o >= w * w) break;
      const int i = o / w, j = o - i * w;
      if (j < i) continue;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;  // four independent chains, combined in a fixed order (as pqr_gram_k)
#pragma unroll 4
      for (int rr = 0; rr < PQ_ROWS; rr += 4) {
        s0 = fma((double)Ft[i][rr], (double)Ft[j][rr], s0);
        s1 = fma((double)Ft[i][rr + 1], (double)Ft[j][rr + 1], s1);
        s2 = fma((double)Ft[i][rr + 2], (double)Ft[j][rr + 2], s2);
        s3 = fma((double)Ft[i][rr + 3], (double)Ft[j][rr + 3], s3);
      }
      acc[q] += (s0 + s1) + (s2 + s3);
    }
  }
  double* out = part + (size_t)blockIdx.x * (PQ_W * PQ_W);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int o = threadIdx.x + 256 * q;
    if (o >= w * w) break;
    const int i = o / w, j = o - i * w;
    if (j >= i) out[j * w + i] = acc[q];
  }
}

// F: n x w (device, destroyed).  Q: n x rank at Qdst.  R_host: room for w*w floats, rank x w as [j*rank + r].
// Kernels only.  meta_dev (2 + PQ_W ints: rank, status, pivots) and Rout_dev (rank x w, leading dimension rank) are device
// buffers of the caller's choice, so that a pipelined caller can fetch them, together with whatever else it needs from the
// step, in ONE copy (every small copy costs ~20 us of queue time).
int k_panel_qr_kernels(isle_ctx* c, float* F, uint64_t n, int w, float* Qdst, int* meta_dev, float* Rout_dev) {
  TimeScope ts(c, ISLE_T_QR);
  if (w < 1 || w > PQ_W) return isle_fail(c, ISLE_E_ARG, "panel QR: width %d not in [1, %d]", w, PQ_W);
  const int nparts = cdiv((long)n, PQ_ROWS * PQ_SUB);
  HIPCHK(c, c->pq_part.reserve((size_t)2 * nparts * PQ_W * PQ_W));
  HIPCHK(c, c->pq_R1.reserve(PQ_W * PQ_W));
  HIPCHK(c, c->pq_T.reserve(2 * PQ_W * PQ_W));
  // (A one-launch form of the whole CholQR2 — rows resident in LDS, two grid barriers, every workgroup repeating the factor step — was
  // measured in rounds 1, 2 and 6: 83 / 79 / 91 us per QR against 78 / 75 / 76 for this chain, whose launches follow each other without
  // gaps; removed in round 6.)
  const dim3 rows(cdiv((long)n, 256));
  hipLaunchKernelGGL(pqr_gram_k, dim3(nparts), dim3(256), 0, c->stream, F, n, w, (const int*)nullptr, c->pq_part.p);
  hipLaunchKernelGGL(pqr_factor_k, dim3(1), dim3(256), 0, c->stream, c->pq_part.p, nparts, w, 1, c->pq_R1.p, meta_dev, c->pq_T.p, Rout_dev);
  hipLaunchKernelGGL(pqr_apply_gram_k, dim3(nparts), dim3(256), 0, c->stream, F, n, w, c->pq_T.p, meta_dev, Qdst, c->pq_part.p);
  hipLaunchKernelGGL(pqr_factor_k, dim3(1), dim3(256), 0, c->stream, c->pq_part.p, nparts, w, 2, c->pq_R1.p, meta_dev, c->pq_T.p, Rout_dev);
  hipLaunchKernelGGL(pqr_apply_k, rows, dim3(256), 0, c->stream, Qdst, n, w, 2, c->pq_T.p, meta_dev, Qdst);
  HIPCHK(c, hipGetLastError());
  return 0;
}
int k_panel_qr(isle_ctx* c, float* F, uint64_t n, int w, float* Qdst, float* R_host, int* rank_out) {
  HIPCHK(c, c->pq_T.reserve(2 * PQ_W * PQ_W));
  HIPCHK(c, c->pq_meta.reserve(2 + PQ_W));
  float* Rout = c->pq_T.p + PQ_W * PQ_W;
  ISLECHK(k_panel_qr_kernels(c, F, n, w, Qdst, c->pq_meta.p, Rout));
  int meta[2] = {0, 0};
  HIPCHK(c, hipMemcpyAsync(meta, c->pq_meta.p, sizeof(meta), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(R_host, Rout, (size_t)w * w * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (meta[1]) return isle_fail(c, ISLE_E_NUMERIC, "CholQR2: second Gram matrix not positive definite");
  *rank_out = meta[0];
  return 0;
}

// uniform [0,1) fill (arma::randu stand-in, block-ks/restarted_block_ks.h:212)
__global__ void randu_k(float* __restrict__ F, uint64_t count, uint64_t seed) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  uint64_t z = seed * 0x9E3779B97F4A7C15ull + (i + 1) * 0xD1342543DE82EF95ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  F[i] = (float)(z >> 40) * (1.0f / 16777216.0f);
}
int k_randu(isle_ctx* c, float* F, uint64_t count, uint64_t seed) {
  hipLaunchKernelGGL(randu_k, dim3(cdiv(count, 256)), dim3(256), 0, c->stream, F, count, seed);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------
// C (M x N) = A (M x K, lda = M) * B (K x N, ldb), col-major, exact f32 on the matrix cores: gemm_f32.h (v_mfma_f32_32x32x2_f32,
// 16-byte LDS operand units, two-stage LDS ring, XCD-aware tile order).  Every plain product of the path goes through it — the Ritz
// rotation, the lift, the D x k x k products of the projected / first word-space assignment, the thin k-means++ products, the dense
// test operator.  The tile shape is a function of the problem's shape alone (so the result is a function of the operands alone):
//   N <= 32: 256 x 32, N <= 64: 256 x 64 (thin products);  >= 1024 tiles of 256 x 256: that shape with 16 waves;
//   >= 512 tiles of 256 x 128: 8 waves;  else 128 x 128 with 4 waves.
// tools/microbench/gemm_probe.hip (random operands, one MI355X): 123 TFLOP/s at 1.25 M x 1000 x 1000, 105 - 109 at 100 000 x 1000 x 2000,
// 97 - 101 at 100 000 x 1000 x 1000, 44 at 1.25 M x 33 x 1000; rocBLAS sgemm on the same operands 121 / 116 / 105 / 33.
// ------------------------------------------------------------------------------------------
using GemmHuge = isle_gemm::Cfg<2, 2, 4, 4, 16, 4>;   // 256 x 256 x 16, 1024 threads
using GemmBig = isle_gemm::Cfg<2, 2, 4, 2, 16, 2>;    // 256 x 128 x 16, 512 threads
using GemmSmall = isle_gemm::Cfg<2, 2, 2, 2, 16, 2>;  // 128 x 128 x 16, 256 threads
using GemmThin64 = isle_gemm::Cfg<2, 2, 4, 1, 16, 2>;  // 256 x 64 x 16, 256 threads
using GemmThin32 = isle_gemm::Cfg<2, 1, 4, 1, 16, 2>;  // 256 x 32 x 16, 256 threads

template <class Epi>
static hipError_t gemm_dispatch(hipStream_t st, const float* A, uint64_t M, int K, const float* B, int ldb, int N, Epi epi) {
  const uint64_t mb256 = (M + 255) / 256;
  if (N <= 32) return isle_gemm::launch<GemmThin32>(st, A, M, K, B, ldb, N, epi);
  if (N <= 64) return isle_gemm::launch<GemmThin64>(st, A, M, K, B, ldb, N, epi);
  if (mb256 * (uint64_t)((N + 255) / 256) >= 1024) return isle_gemm::launch<GemmHuge>(st, A, M, K, B, ldb, N, epi);
  if (mb256 * (uint64_t)((N + 127) / 128) >= 512) return isle_gemm::launch<GemmBig>(st, A, M, K, B, ldb, N, epi);
  return isle_gemm::launch<GemmSmall>(st, A, M, K, B, ldb, N, epi);
}

int k_gemm_nn(isle_ctx* c, const float* A, uint64_t M, int K, const float* B, int ldb, int N, float* C, int family) {
  TimeScope ts(c, family);
  if (M == 0 || N == 0) return 0;
  if (M >= (1ull << 32) || (uint64_t)N * (uint64_t)ldb >= (1ull << 32)) return isle_fail(c, ISLE_E_ARG, "gemm: operand too large for 32-bit row / column offsets");
  HIPCHK(c, gemm_dispatch(c->stream, A, M, K, B, ldb, N, isle_gemm::StoreC{C, M}));
  return 0;
}

// The D x k x k dot products of the assignment steps (projected full pass, first assignment of Lloyd on B): the bf16 matrix cores with
// both operands split into three bf16 terms (gemm_bf16x3.h) — the six partial products down to 2^-16 of the leading one, each exact, summed
// in f32: error against fp64 5.6e-8 of sum |a b| where gemm_f32.h has 8.6e-8 (tools/microbench/gemm3_probe.hip, random operands), 161
// against 123 TFLOP/s at 1.25 M x 1000 x 1000, 102 against 67 at 1 M x 200 x 200.  Small products and ISLE_GEMM_BF16X3=0: gemm_f32.h.
using Gemm3Huge = isle_gemm3::Cfg<2, 2, 4, 4, 4>;  // 256 x 256 x 16, 1024 threads
int k_gemm_nn_assign(isle_ctx* c, const float* A, uint64_t M, int K, const float* B, int ldb, int N, float* C, int family) {
  if (c->knob_zero(KN_GEMM_BF16X3) || N < 64 || K < 32 || (M + 255) / 256 * (uint64_t)((N + 255) / 256) < 512)
    return k_gemm_nn(c, A, M, K, B, ldb, N, C, family);
  TimeScope ts(c, family);
  if (M >= (1ull << 32)) return isle_fail(c, ISLE_E_ARG, "gemm: operand too large for 32-bit row offsets");
  HIPCHK(c, c->gemm_b3.reserve((size_t)3 * isle_gemm3::kp8_of(K) * isle_gemm3::np_of<Gemm3Huge>(N)));
  HIPCHK(c, isle_gemm3::launch<Gemm3Huge>(c->stream, A, M, K, B, ldb, N, c->gemm_b3.p, isle_gemm3::StoreC{C, M}));
  return 0;
}

// ------------------------------------------------------------------------------------------
// The two assignment steps that rest on a D x k x k product — the first assignment of Lloyd on B through the projection (distsq_docs_to_centers /
// closest_centers, src/sparseMatrix.cpp:1494-1572) and the full pass of Lloyd in span(U) (distsq_projected_docs_to_projected_centers /
// projected_closest_centers, :1794-1871) — with their epilogues INSIDE the product (gemm_bf16x3.h, group epilogues): distances, group / tile
// bounds and the per-slot candidates are formed from the accumulators; the D x k matrix (40 GB at config 3) is neither written nor read
// again, and dots_assign_cm_k / proj_dots_tiles_k with their 12 + 10 ms per step are replaced by a pass over 16 candidates per document.
// Same arithmetic on the same dot products as those kernels: assignments and bounds are bit-identical to the two-kernel route.
// ------------------------------------------------------------------------------------------
// Two passes since round 4 (ISLE_GEMM_TERMS=3: one): the product first runs with TWO bf16 terms per operand (three partial products, half
// the matrix-core work: 58 against 108 ms for 10 M x 1000 x 1000) and every distance it forms is within  eps = ga_eta(K) (|row|^2 + max |c|^2)
// of the three-term value: bf16 keeps 8 significand bits (rounding error <= 2^-8 |x|), so |x1| <= 2^-8 |x| and the remainder x - x0 - x1 is
// <= 2^-16 |x|; the dropped a1 b1 and the two remainders are <= 3 * 2^-16 |a_k b_k| per term, the dot product is within 4.6e-5 |a| |b| <=
// 2.3e-5 (|a|^2 + |b|^2), the distance within twice that = 4.6e-5 (|a|^2 + |b|^2) = GA_ETA_TRUNC (tests/test_two_term_bound_cpu.py checks the
// chain in NumPy).  The f32 accumulation of the two routes is bounded the worst-case way as well: an accumulator takes 3 (6) MFMA results per
// 16 k, each addition within 2^-24 of the running sum <= sum |a_k b_k|, i.e. 9 ceil(K / 16) 2^-24 for both routes together: ga_eta(1000) =
// 8.2e-5.  (On data the two routes differ by 1e-6; the margin costs a few thousand more rows in the second pass.)  Lower bounds are taken from d - eps, the upper bound from d + eps; a row whose two smallest
// distances are closer than 2 eps (its arg-min is not decided, exact ties included) goes on a list, and the listed rows are gathered and
// run through the three-term product with the same epilogue: the assignment is, row by row, the three-term route's.
constexpr float GA_ETA_TRUNC = 4.65e-5f;
static float ga_eta(int K) { return GA_ETA_TRUNC + 1.05f * 9.f * (float)((K + 15) / 16) * 5.9604645e-8f; }
struct AssignRec {  // best of one 64-column slot of one document
  float m1;
  uint32_t i1;
  float m2, ax, r2;  // runner-up of the best's group, largest aux of that group, smallest distance among the slot's other columns
};
struct YyGroupEpi {  // Lloyd on B: Yinyang groups of 8 centres (dots_assign_cm_k's rules)
  static constexpr int kGroup = 8;
  const float* __restrict__ cn;
  const float* __restrict__ dn;
  const float* __restrict__ cn_max;
  float* __restrict__ lb;
  int G, nslot;
  AssignRec* __restrict__ part;
  const uint32_t* __restrict__ map;  // row of the product -> document (null: identity)
  float eta;                         // 0: three-term product
  const float* __restrict__ an;      // squared norms of the product's rows (the projections) and the largest squared norm of its columns (the
  const float* __restrict__ bmax;    // centres' coordinates): the operands the two-term error is relative to (dn, cn are word-space norms)
  __device__ inline uint64_t doc(uint64_t m) const { return map ? map[m] : m; }
  __device__ inline float rowdata(uint64_t m) const { return dn[doc(m)]; }
  __device__ inline float dist(float dot, int col, float dnd) const { return fabsf((-2.0f * dot + cn[col]) + dnd); }
  __device__ inline float aux(int) const { return 0.f; }
  __device__ inline void group(uint64_t m, int g, float dnd, float m1, uint32_t, float, float) const {
#ifdef GA_ABLATE_LB  // timing-only experiment: what the 4-byte stores of the group bounds cost
    if (g >= 0) return;
#endif
    const float E = ISLE_SLACK_REL * (dnd + *cn_max), sE = sqrtf(E);
    const float eps = eta > 0.f ? eta * (an[doc(m)] + *bmax) : 0.f;
    lb[doc(m) * (uint64_t)G + g] = yy_slack_down_sq(fmaxf(m1 - eps, 0.f), E, sE);
  }
  __device__ inline void slot(uint64_t m, int sl, float, float m1, uint32_t i1, float m2, float ax, float r2) const {
    part[m * (uint64_t)nslot + sl] = AssignRec{m1, i1, m2, ax, r2};
  }
};
// the row's winner over its slots, and whether the runner-up leaves it open (two-term pass only)
struct AssignPick {
  float best, m2, ax, run2;
  uint32_t bidx;
};
__device__ inline AssignPick assign_pick(const AssignRec* __restrict__ part, size_t row, int nslot) {
  AssignPick p{3.4e38f, 3.4e38f, 0.f, 3.4e38f, 0xffffffffu};
  for (int sl0 = 0; sl0 < nslot; sl0 += 8) {  // eight records asked for together (one at a time, the comparison chain waited for every load)
    AssignRec rr[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) rr[u] = part[row * nslot + min(sl0 + u, nslot - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) {  // ascending columns: a tie keeps the earlier centre
      if (sl0 + u < nslot) {
        const AssignRec r = rr[u];
        if (r.m1 < p.best) {
          p.run2 = fminf(fminf(p.run2, p.best), r.r2);
          p.best = r.m1;
          p.bidx = r.i1;
          p.m2 = r.m2;
          p.ax = r.ax;
        } else {
          p.run2 = fminf(p.run2, r.m1);
        }
      }
    }
  }
  return p;
}
__global__ __launch_bounds__(256) void yy_first_combine_k(const AssignRec* __restrict__ part, int nslot, uint32_t n, int G, const float* __restrict__ dn,
                                                          const float* __restrict__ cn_max, uint32_t* __restrict__ assign, float* __restrict__ ub,
                                                          float* __restrict__ lb, const uint32_t* __restrict__ map, float eta,
                                                          const float* __restrict__ an, const float* __restrict__ bmax,
                                                          uint32_t* __restrict__ redo, uint32_t* __restrict__ nredo) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const bool in = i < n;
  bool open = false;
  if (in) {
    const uint32_t d = map ? map[i] : i;
    const AssignPick p = assign_pick(part, i, nslot);
    const float eps = eta > 0.f ? eta * (an[d] + *bmax) : 0.f;
    const float E = ISLE_SLACK_REL * (dn[d] + *cn_max), sE = sqrtf(E);
    lb[(size_t)d * G + p.bidx / 8] = yy_slack_down_sq(fmaxf(p.m2 - eps, 0.f), E, sE);  // the assigned centre's group: its closest OTHER member
    const float u = sqrtf(p.best + eps);
    ub[d] = u + fminf(sE, E / fmaxf(u, 1e-30f));
    assign[d] = p.bidx;
    open = eta > 0.f && p.run2 - p.best <= 2.f * eps;
  }
  if (eta > 0.f) {  // uniform over the launch
    const uint32_t at = block_append_slot(open, nredo);
    if (open) redo[at] = map ? map[i] : i;
  }
}
struct TileEpi {  // Lloyd in span(U): tiles of 32 centres (proj_dots_tiles_k's rules)
  static constexpr int kGroup = 32;
  const float* __restrict__ cn;
  const float* __restrict__ pn;
  const float* __restrict__ cmax;
  float* __restrict__ lb;
  int TL, nslot;
  AssignRec* __restrict__ part;
  const uint32_t* __restrict__ map;
  float eta;
  __device__ inline uint64_t doc(uint64_t m) const { return map ? map[m] : m; }
  __device__ inline float rowdata(uint64_t m) const { return pn[doc(m)]; }
  __device__ inline float dist(float dot, int col, float nd) const { return fabsf((-2.0f * dot + cn[col]) + nd); }
  __device__ inline float aux(int col) const { return cn[col]; }
  __device__ inline void group(uint64_t m, int T, float nd, float m1, uint32_t, float, float tc) const {
    float uu, ll;
    const float lo = fmaxf(m1 - eta * (nd + *cmax), 0.f);
    hamerly_store_bounds(lo, lo, nd + tc, &uu, &ll);
    lb[doc(m) * (uint64_t)TL + T] = ll;
  }
  __device__ inline void slot(uint64_t m, int sl, float, float m1, uint32_t i1, float m2, float ax, float r2) const {
    part[m * (uint64_t)nslot + sl] = AssignRec{m1, i1, m2, ax, r2};
  }
};
__global__ __launch_bounds__(256) void tiles_combine_k(const AssignRec* __restrict__ part, int nslot, uint32_t n, int TL, const float* __restrict__ pn,
                                                       const float* __restrict__ cmax_p, uint32_t* __restrict__ assign, float* __restrict__ ub,
                                                       float* __restrict__ lb, const uint32_t* __restrict__ map, float eta,
                                                       uint32_t* __restrict__ redo, uint32_t* __restrict__ nredo) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const bool in = i < n;
  bool open = false;
  if (in) {
    const uint32_t d = map ? map[i] : i;
    const AssignPick p = assign_pick(part, i, nslot);
    const float nd = pn[d];
    const float eps = eta * (nd + *cmax_p);
    assign[d] = p.bidx;
    float uu, ll, l2;
    hamerly_store_bounds(p.best + eps, fmaxf(p.m2 - eps, 0.f), nd + p.ax, &uu, &l2);
    hamerly_store_bounds(p.best + eps, p.best + eps, nd + *cmax_p, &uu, &ll);
    ub[d] = uu;
    lb[(size_t)d * TL + (p.bidx >> 5)] = l2;  // the assigned centre's tile: closest OTHER centre in it
    open = eta > 0.f && p.run2 - p.best <= 2.f * eps;
  }
  if (eta > 0.f) {
    const uint32_t at = block_append_slot(open, nredo);
    if (open) redo[at] = map ? map[i] : i;
  }
}
// may the fused route be taken?  (the bf16 product with 256 x 256 tiles; small products keep the two-kernel route)
bool k_gemm_assign_fused_ok(isle_ctx* c, uint64_t M, int K, int N) {
  return !c->knob_zero(KN_GEMM_BF16X3) && !c->knob_zero(KN_GEMM_EPILOGUE) && N >= 64 && K >= 32 && (M + 255) / 256 * (uint64_t)((N + 255) / 256) >= 512 &&
         M < (1ull << 32);
}
using Gemm2Huge = isle_gemm3::Cfg<2, 2, 4, 4, 4, 16, 2>;  // the same tile with two bf16 terms per operand
// Both assignment steps: first pass (two terms unless ISLE_GEMM_TERMS=3) over all rows, then the rows it left open through the three-term
// product.  make(map, eta) builds the epilogue, combine(part, n, map, eta, redo, nredo) launches the step's combine kernel.
using Gemm2Dma = isle_gemm3::CfgDma<2, 16>;  // the two-term product with A pre-split and staged by LDS-DMA (gemm_bf16x2_dma_k)
int k_gemm_split_a(isle_ctx* c, const float* A, uint64_t M, int K, void* A2) {
  HIPCHK(c, isle_gemm3::split_a(c->stream, A, M, K, A2, Gemm2Dma::TK));
  return 0;
}
size_t k_gemm_split_a_bytes(uint64_t M, int K) { return isle_gemm3::a2_units(M, K, Gemm2Dma::TK) * 16; }
template <class MakeEpi, class Combine>
static int gemm_assign_two_pass(isle_ctx* c, const float* A, const float* Arm, int lda_rm, const float* rown, uint64_t M, int K, const float* B, int ldb, int N,
                                MakeEpi make, Combine combine, const uint32_t* map0 = nullptr /*row of A -> document (null: identity)*/,
                                const void* A2 = nullptr /*A split beforehand (k_gemm_split_a): the first pass reads it by LDS-DMA*/,
                                const uint32_t* map2 = nullptr /*row of A2 -> document (the split copy by position: c->dperm)*/) {
  // A == nullptr: the operand is the context's projection and its f32 coordinate-major copy has not been made (the default routes read A2):
  // the passes that need it make it
  auto need_a = [&]() -> int {
    if (A) return 0;
    ISLECHK(k_ensure_pt(c));
    A = c->Pt.p;
    return 0;
  };
  const int nslot = (N + 63) / 64;
  static_assert(sizeof(AssignRec) == 20, "");
  const char* gt = c->knob(KN_GEMM_TERMS);
  const bool two = !(gt && atoi(gt) == 3) && Arm != nullptr;
  HIPCHK(c, c->gemm_b3.reserve((size_t)3 * isle_gemm3::kp8_of(K) * isle_gemm3::np_of<Gemm3Huge>(N)));
  HIPCHK(c, c->assign_part.reserve((size_t)M * nslot * 5));
  AssignRec* part = reinterpret_cast<AssignRec*>(c->assign_part.p);
  if (!two) {
    ISLECHK(need_a());
    HIPCHK(c, isle_gemm3::launch<Gemm3Huge>(c->stream, A, M, K, B, ldb, N, c->gemm_b3.p, make(part, map0, 0.f)));
    combine(part, (uint32_t)M, map0, 0.f, nullptr, nullptr);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  const float eta2 = ga_eta(K);
  HIPCHK(c, c->ga_redo.reserve(M + 1));
  uint32_t* nredo = c->ga_redo.p + M;
  HIPCHK(c, hipMemsetAsync(nredo, 0, sizeof(uint32_t), c->stream));
  if (A2 && !c->knob_zero(KN_GEMM_DMA)) {
    HIPCHK(c, isle_gemm3::launch_dma<Gemm2Dma>(c->stream, A2, M, K, B, ldb, N, c->gemm_b3.p, make(part, map2 ? map2 : map0, eta2)));
    combine(part, (uint32_t)M, map2 ? map2 : map0, eta2, c->ga_redo.p, nredo);
  } else {
    ISLECHK(need_a());
    HIPCHK(c, isle_gemm3::launch<Gemm2Huge>(c->stream, A, M, K, B, ldb, N, c->gemm_b3.p, make(part, map0, eta2)));
    combine(part, (uint32_t)M, map0, eta2, c->ga_redo.p, nredo);
  }
  HIPCHK(c, hipGetLastError());
  uint32_t* n_pin = reinterpret_cast<uint32_t*>(c->pin + isle_ctx::PIN_SMALL + (192u << 10) + 192);  // page-locked
  HIPCHK(c, hipMemcpyAsync(n_pin, nredo, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const uint32_t n = *n_pin;
  c->ga_last_redo = n;
  if (c->knob_on(KN_DEBUG_HAMERLY)) fprintf(stderr, "[assignment product] the two-term pass left %u of %llu rows open\n", n, (unsigned long long)M);
  if (n == 0) return 0;
  if ((uint64_t)n * 4 > M) {  // hardly a saving left: the whole product again with three terms
    ISLECHK(need_a());
    HIPCHK(c, isle_gemm3::launch<Gemm3Huge>(c->stream, A, M, K, B, ldb, N, c->gemm_b3.p, make(part, map0, 0.f)));
    combine(part, (uint32_t)M, map0, 0.f, nullptr, nullptr);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  // the open rows, gathered coordinate-major (the product's A layout); the list is in arrival order — every row's result is its own
  HIPCHK(c, c->ga_rows.reserve((size_t)n * lda_rm));
  HIPCHK(c, c->ga_rown.reserve(n));
  ISLECHK(k_compact_rows(c, Arm, rown, lda_rm, c->ga_redo.p, n, c->ga_rows.p, c->ga_rown.p));
  HIPCHK(c, isle_gemm3::launch<Gemm3Huge>(c->stream, c->ga_rows.p, (uint64_t)n, K, B, ldb, N, c->gemm_b3.p, make(part, c->ga_redo.p, 0.f)));
  combine(part, n, c->ga_redo.p, 0.f, nullptr, nullptr);
  HIPCHK(c, hipGetLastError());
  return 0;
}
// assign / ub / lb (D x G Yinyang group bounds) of the first assignment of Lloyd on B from A (D x k projection, coordinate-major; Arm the
// same rows row-major with leading dimension lda_rm, for the second pass) and the k lifted centres' coordinates B (k x k, leading dimension
// ldb): dots_assign_cm_k's outputs without the D x k product in memory
int k_gemm_assign_yy(isle_ctx* c, const float* A, const float* Arm, int lda_rm, const float* an, uint64_t M, int K, const float* B, int ldb, int N, int G,
                     const float* cn, const float* dn, const float* cn_max, uint32_t* assign, float* ub, float* lb, int family, const void* A2,
                     const uint32_t* map2) {
  TimeScope ts(c, family);
  const int nslot = (N + 63) / 64;
  // largest squared norm of the product's columns (rows of B as stored: centre n's K coordinates)
  HIPCHK(c, c->ga_bn.reserve((size_t)N + 4));
  float* bmax = c->ga_bn.p + N;
  if (an) {
    ISLECHK(k_rownorms(c, B, N, K, ldb, c->ga_bn.p));
    ISLECHK(k_max_f32(c, c->ga_bn.p, N, bmax));
  }
  return gemm_assign_two_pass(
      c, A, an ? Arm : nullptr, lda_rm, an, M, K, B, ldb, N,
      [&](AssignRec* part, const uint32_t* map, float eta) { return YyGroupEpi{cn, dn, cn_max, lb, G, nslot, part, map, eta, an, bmax}; },
      [&](const AssignRec* part, uint32_t n, const uint32_t* map, float eta, uint32_t* redo, uint32_t* nredo) {
        hipLaunchKernelGGL(yy_first_combine_k, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, part, nslot, n, G, dn, cn_max, assign, ub, lb, map, eta, an, bmax, redo,
                           nredo);
      },
      nullptr, A2, map2);
}
// the same for the full pass of Lloyd in span(U): assign / ub / one lower bound per tile of 32 centres (row stride TL); cmax = max |c|^2 on the device
int k_gemm_assign_tiles(isle_ctx* c, const float* A, const float* Arm, int lda_rm, uint64_t M, int K, const float* B, int ldb, int N, int TL, const float* cn,
                        const float* pn, const float* cmax, uint32_t* assign, float* ub, float* tlb, int family, const uint32_t* map0, const void* A2,
                        const uint32_t* map2) {
  TimeScope ts(c, family);
  const int nslot = (N + 63) / 64;
  return gemm_assign_two_pass(
      c, A, Arm, lda_rm, pn, M, K, B, ldb, N,
      [&](AssignRec* part, const uint32_t* map, float eta) { return TileEpi{cn, pn, cmax, tlb, TL, nslot, part, map, eta}; },
      [&](const AssignRec* part, uint32_t n, const uint32_t* map, float eta, uint32_t* redo, uint32_t* nredo) {
        hipLaunchKernelGGL(tiles_combine_k, dim3(cdiv(n, 256)), dim3(256), 0, c->stream, part, nslot, n, TL, pn, cmax, assign, ub, tlb, map, eta, redo, nredo);
      },
      map0, A2, map2);
}

// in: element (r, cidx) at in[cidx*ld_in + r], r < rows, cidx < cols.  out[r*ld_out + cidx] = in(r, cidx).
__global__ __launch_bounds__(256) void transpose_k(const float* __restrict__ in, uint64_t rows, uint64_t cols, uint64_t ld_in,
                                                    float* __restrict__ out, uint64_t ld_out) {
  __shared__ float t[32][33];
  const uint64_t r0 = (uint64_t)blockIdx.x * 32, c0 = (uint64_t)blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int yy = ty; yy < 32; yy += 8) {
    const uint64_t r = r0 + tx, cc = c0 + yy;
    t[yy][tx] = (r < rows && cc < cols) ? in[cc * ld_in + r] : 0.f;
  }
  __syncthreads();
  for (int yy = ty; yy < 32; yy += 8) {
    const uint64_t r = r0 + yy, cc = c0 + tx;
    if (r < rows && cc < cols) out[r * ld_out + cc] = t[tx][yy];
  }
}
int k_transpose(isle_ctx* c, const float* in, uint64_t rows, uint64_t cols, uint64_t ld_in, float* out, uint64_t ld_out) {
  if (rows == 0 || cols == 0) return 0;
  // row chunks of at most 2^31 threads per launch (10 M x 1000 is 2.5e9 threads in one: a launch of 2^32 or more does not run)
  const uint64_t cb = cdiv(cols, 32);
  const uint64_t rows_per = std::max<uint64_t>(32, ((1ull << 31) / 256 / cb) * 32);
  for (uint64_t r0 = 0; r0 < rows; r0 += rows_per) {
    const uint64_t nr = std::min(rows_per, rows - r0);
    dim3 g((unsigned)cdiv(nr, 32), (unsigned)cb), blk(256);
    hipLaunchKernelGGL(transpose_k, g, blk, 0, c->stream, in + r0, nr, cols, ld_in, out + r0 * ld_out, ld_out);
    HIPCHK(c, hipGetLastError());
  }
  return 0;
}

// The coordinate-major f32 copy of the projection (Pt[j * D + d] = P[d][j]).  Until round 5 every projection was followed by this
// transposition; the default routes of a large shard now read the split copy the grouped projection leaves (c->Pt2, by position), and the
// routes that still want f32 columns — the register kernels, the three-term whole-product fallbacks, the non-fused products, the small-shape
// k-means++ pass — ask for it here.
int k_ensure_pt(isle_ctx* c) {
  if (c->Pt_ready) return 0;
  if (!c->P_ready || !c->D) return isle_fail(c, ISLE_E_ARG, "no projection to transpose");
  HIPCHK(c, c->Pt.reserve((size_t)c->D * c->ldk));
  ISLECHK(k_transpose(c, c->P.p, c->ldk, c->D, c->ldk, c->Pt.p, c->D));
  c->Pt_ready = true;
  return 0;
}

// ------------------------------------------------------------------------------------------
// Block one-sided Jacobi (Hestenes) on W = (S + mu I), V = I, fp64.
// Columns are grouped in blocks of 16; a round pairs the blocks round-robin and one workgroup owns a pair:
//   (1) G = [Wa Wb]^T [Wa Wb]  (32 x 32, rows streamed through LDS)
//   (2) ONE cyclic sweep of two-sided Jacobi rotations on G in LDS (31 inner rounds x 16 disjoint pairs),
//       accumulating the 32 x 32 rotation product Q  ("cyclic by blocks" ordering of the scalar method)
//   (3) [Wa Wb] <- [Wa Wb] Q,  [Va Vb] <- [Va Vb] Q
// n/16 - 1 launches per sweep instead of n - 1 for the scalar method (the scalar kernel was launch-latency
// bound: 9.4k launches and 58 ms per solve pair at n = 400).
// ------------------------------------------------------------------------------------------
__device__ inline double block_sum(double v, double* sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += sh[w];
  return s;
}

constexpr int BJ_W = 16;          // block width
constexpr int BJ_P = 2 * BJ_W;    // columns per pair

__device__ inline void rr_pair(int nplayers, int round, int slot, int* p, int* q) {
  const int m = nplayers - 1;
  if (slot == 0) {
    *p = nplayers - 1;
    *q = round % m;
  } else {
    *p = (round + slot) % m;
    *q = (round - slot + m) % m;
  }
}

// per-pair scratch in global memory: Q (32x32 doubles) and a rotation count; the Gram partials (one 32x32 slab per
// row chunk, summed in fixed order by bj_inner_k: bitwise reproducible, so replicated solves on several GPUs agree)
// live in a second buffer
constexpr int BJ_SCR = BJ_P * BJ_P + 8;
constexpr int BJ_ROWS = 64;   // rows per workgroup in the streaming kernels (one LDS tile)

__device__ inline void bj_blocks(int nblk2, int round, int slot, int* A, int* Bk) {
  rr_pair(nblk2, round, slot, A, Bk);
  if (*A > *Bk) {
    const int x = *A;
    *A = *Bk;
    *Bk = x;
  }
}
__device__ inline int bj_col(int A, int Bk, int cc) { return cc < BJ_W ? A * BJ_W + cc : Bk * BJ_W + (cc - BJ_W); }

// (1) G[pair] += [Wa Wb]^T [Wa Wb] over this workgroup's rows.  grid = (pairs, row chunks)
__global__ __launch_bounds__(256) void bj_gram_k(const double* __restrict__ W, int n, int nblk2, int round, double* __restrict__ gpart) {
  __shared__ double T[64][BJ_P + 1];
  int A, Bk;
  bj_blocks(nblk2, round, blockIdx.x, &A, &Bk);
  if (A * BJ_W >= n) return;
  const int t = threadIdx.x;
  const int gi = t >> 3, gj0 = (t & 7) * 4;
  double g0 = 0.0, g1 = 0.0, g2 = 0.0, g3 = 0.0;
  const int rbeg = blockIdx.y * BJ_ROWS, rend = min(n, rbeg + BJ_ROWS);
  for (int r0 = rbeg; r0 < rend; r0 += 64) {
    __syncthreads();
    for (int idx = t; idx < 64 * BJ_P; idx += 256) {
      const int rr = idx & 63, cc = idx >> 6;
      T[rr][cc] = W[(size_t)bj_col(A, Bk, cc) * n + r0 + rr];  // n is padded: no guards, no branches around loads
    }
    __syncthreads();
#pragma unroll 8
    for (int rr = 0; rr < 64; ++rr) {
      const double a = T[rr][gi];
      g0 = fma(a, T[rr][gj0 + 0], g0);
      g1 = fma(a, T[rr][gj0 + 1], g1);
      g2 = fma(a, T[rr][gj0 + 2], g2);
      g3 = fma(a, T[rr][gj0 + 3], g3);
    }
  }
  double* G = gpart + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * (BJ_P * BJ_P) + gi * BJ_P + gj0;
  G[0] = g0;
  G[1] = g1;
  G[2] = g2;
  G[3] = g3;
}

// (2) one cyclic sweep of two-sided rotations on G in LDS, Q accumulated; grid = pairs
// cross_only: rotate only the 16 x 16 pairs (column of block A, column of block B), 16 rounds instead of the 31 of the full
// tournament.  The host asks for the full tournament in round 0 of every sweep (each block is in exactly one pair there),
// so every column pair of the matrix is rotated exactly once per sweep — the classical cyclic Jacobi in a block ordering —
// instead of the intra-block pairs being revisited in every round.
__global__ __launch_bounds__(256) void bj_inner_k(int n, int nblk2, int round, double tol, double abs_tol, const double* __restrict__ gpart,
                                                   int nrowch, double* __restrict__ scr, unsigned int* __restrict__ rotated, int cross_only) {
  __shared__ double G[BJ_P][BJ_P + 1];
  __shared__ double Q[BJ_P][BJ_P + 1];
  __shared__ unsigned int nrot;
  int A, Bk;
  bj_blocks(nblk2, round, blockIdx.x, &A, &Bk);
  if (A * BJ_W >= n) return;
  const int t = threadIdx.x;
  double* S = scr + (size_t)blockIdx.x * BJ_SCR;
  const double* gp = gpart + (size_t)blockIdx.x * nrowch * (BJ_P * BJ_P);
  for (int idx = t; idx < BJ_P * BJ_P; idx += 256) {
    const int i = idx / BJ_P, j = idx % BJ_P;
    const int u = (i <= j) ? i * BJ_P + j : j * BJ_P + i;  // upper triangle defines G (exactly symmetric)
    double sum = 0.0;
    for (int ch = 0; ch < nrowch; ++ch) sum += gp[(size_t)ch * (BJ_P * BJ_P) + u];  // fixed order
    G[i][j] = sum;
    Q[i][j] = (i == j) ? 1.0 : 0.0;
  }
  if (t == 0) nrot = 0;
  __syncthreads();
  {
    // Already diagonal to tolerance (every pair in the late sweeps, all of them in the last one)?  Then no rotation below
    // would fire: leave without the 31 rounds.  Same test as the rotation rule, applied to the whole block at once.
    int live = 0;
    for (int idx = t; idx < BJ_P * BJ_P; idx += 256) {
      const int i = idx / BJ_P, j = idx % BJ_P;
      if (i < j && (!cross_only || (i < BJ_W && j >= BJ_W))) {
        const double a = G[i][i], b = G[j][j], g = G[i][j];
        live |= (a > 0.0 && b > 0.0 && g * g > tol * tol * a * b && fabs(g) > abs_tol) ? 1 : 0;
      }
    }
    if (!__syncthreads_or(live)) {
      if (t == 0) S[BJ_P * BJ_P] = 0.0;
      return;
    }
  }
  // Thread (bi, bj) owns the 2x2 block G[{p_i,q_i}][{p_j,q_j}] of rotation pairs i and j, and two rows of Q's column
  // pair i.  Every thread derives the two rotations it needs from G's diagonal blocks itself (redundant arithmetic, but
  // no separate parameter phase): a round is read -> barrier -> write -> barrier.
  const int bi = t >> 4, bj = t & 15;
  unsigned int my_rot = 0;
  const int nrounds = cross_only ? BJ_W : BJ_P - 1;
  for (int ir = 0; ir < nrounds; ++ir) {
    int pi, qi, pj, qj;
    if (cross_only) {
      pi = bi;
      qi = BJ_W + ((bi + ir) & (BJ_W - 1));
      pj = bj;
      qj = BJ_W + ((bj + ir) & (BJ_W - 1));
    } else {
      rr_pair(BJ_P, ir, bi, &pi, &qi);
      rr_pair(BJ_P, ir, bj, &pj, &qj);
    }
    if (pi > qi) {
      const int x = pi;
      pi = qi;
      qi = x;
    }
    if (pj > qj) {
      const int x = pj;
      pj = qj;
      qj = x;
    }
    double cs[2], sn[2];
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const int p = w ? pj : pi, q = w ? qj : qi;
      const double a = G[p][p], b = G[q][q], gpq = G[p][q];
      cs[w] = 1.0;
      sn[w] = 0.0;
      // relative test between live columns; absolute floor for (numerically) null columns of a singular S + mu I
      if (a > 0.0 && b > 0.0 && gpq * gpq > tol * tol * a * b && fabs(gpq) > abs_tol) {
        // tan(theta) only steers the annihilation: fp32 is enough (the sweep repeats); cos must make the rotation
        // orthogonal to fp64 accuracy, so it gets an fp64 reciprocal square root with Newton refinement
        const float zf = (float)(b - a) / (float)(2.0 * gpq);
        const float tf = (zf >= 0.f ? 1.f : -1.f) / (fabsf(zf) + sqrtf(1.f + zf * zf));
        const double tt = (double)tf;
        const double u = 1.0 + tt * tt;
        double y = (double)rsqrtf((float)u);
        y = y * (1.5 - 0.5 * u * y * y);
        y = y * (1.5 - 0.5 * u * y * y);
        y = y * (1.5 - 0.5 * u * y * y);
        cs[w] = y;
        sn[w] = y * tt;
        if (w == 0 && bj == 0) ++my_rot;
      }
    }
    const double x00 = G[pi][pj], x01 = G[pi][qj], x10 = G[qi][pj], x11 = G[qi][qj];
    const int r0 = bj * 2;
    const double q0p = Q[r0][pi], q0q = Q[r0][qi], q1p = Q[r0 + 1][pi], q1q = Q[r0 + 1][qi];
    // columns (pair j), then rows (pair i):  G <- J_i^T (G J_j)
    const double y00 = cs[1] * x00 - sn[1] * x01, y01 = sn[1] * x00 + cs[1] * x01;
    const double y10 = cs[1] * x10 - sn[1] * x11, y11 = sn[1] * x10 + cs[1] * x11;
    __syncthreads();
    G[pi][pj] = cs[0] * y00 - sn[0] * y10;
    G[qi][pj] = sn[0] * y00 + cs[0] * y10;
    G[pi][qj] = cs[0] * y01 - sn[0] * y11;
    G[qi][qj] = sn[0] * y01 + cs[0] * y11;
    Q[r0][pi] = cs[0] * q0p - sn[0] * q0q;
    Q[r0][qi] = sn[0] * q0p + cs[0] * q0q;
    Q[r0 + 1][pi] = cs[0] * q1p - sn[0] * q1q;
    Q[r0 + 1][qi] = sn[0] * q1p + cs[0] * q1q;
    __syncthreads();
  }
  if (my_rot) atomicAdd(&nrot, my_rot);
  __syncthreads();
  for (int idx = t; idx < BJ_P * BJ_P; idx += 256) S[idx] = Q[idx / BJ_P][idx % BJ_P];
  if (t == 0) {
    // store the COUNT (not `nrot ? 1.0 : 0.0`): hipcc 7.2 lowered that select to s_cselect on a stale SCC here and
    // always wrote 0.0 (seen in the ISA; the rotations were then never applied)
    S[BJ_P * BJ_P] = (double)nrot;
    if (nrot) atomicAdd(rotated, nrot);
  }
}

// (3) [Ma Mb] <- [Ma Mb] Q for M = W and V, one thread per row; grid = (pairs, row chunks)
__global__ __launch_bounds__(256) void bj_apply_k(double* __restrict__ W, double* __restrict__ Vv, int n, int nblk2, int round,
                                                   const double* __restrict__ scr, int nmat) {
  __shared__ double Q[BJ_P][BJ_P + 1];
  int A, Bk;
  bj_blocks(nblk2, round, blockIdx.x, &A, &Bk);
  if (A * BJ_W >= n) return;
  const double* S = scr + (size_t)blockIdx.x * BJ_SCR;
  const int t = threadIdx.x;
  if (S[BJ_P * BJ_P] == 0.0) return;  // no rotation in this pair (uniform)
  for (int idx = t; idx < BJ_P * BJ_P; idx += 256) Q[idx / BJ_P][idx % BJ_P] = S[idx];
  __syncthreads();
  // thread = (row, group of 8 output columns): Q is read 8x less often from LDS than with one thread per row.
  // The matrices are zero-padded to a multiple of 64 rows/columns, so no load or store needs a guard (guarded loads
  // made hipcc branch around every load, wait for each separately and spill: 33 us per launch).
  const int r = blockIdx.y * BJ_ROWS + (t & 63);
  const int c0 = (t >> 6) * 8;
  for (int mat = 0; mat < nmat; ++mat) {
    double* M = mat ? Vv : W;
    double x[BJ_P];
#pragma unroll
    for (int j = 0; j < BJ_P; ++j) x[j] = M[(size_t)bj_col(A, Bk, j) * n + r];
    double o0 = 0.0, o1 = 0.0, o2 = 0.0, o3 = 0.0, o4 = 0.0, o5 = 0.0, o6 = 0.0, o7 = 0.0;
#pragma unroll
    for (int j = 0; j < BJ_P; ++j) {
      const double xv = x[j];
      const double* qr = &Q[j][c0];
      o0 = fma(xv, qr[0], o0);
      o1 = fma(xv, qr[1], o1);
      o2 = fma(xv, qr[2], o2);
      o3 = fma(xv, qr[3], o3);
      o4 = fma(xv, qr[4], o4);
      o5 = fma(xv, qr[5], o5);
      o6 = fma(xv, qr[6], o6);
      o7 = fma(xv, qr[7], o7);
    }
    __syncthreads();  // all four column groups of a row have read x before anyone overwrites it
    M[(size_t)bj_col(A, Bk, c0 + 0) * n + r] = o0;
    M[(size_t)bj_col(A, Bk, c0 + 1) * n + r] = o1;
    M[(size_t)bj_col(A, Bk, c0 + 2) * n + r] = o2;
    M[(size_t)bj_col(A, Bk, c0 + 3) * n + r] = o3;
    M[(size_t)bj_col(A, Bk, c0 + 4) * n + r] = o4;
    M[(size_t)bj_col(A, Bk, c0 + 5) * n + r] = o5;
    M[(size_t)bj_col(A, Bk, c0 + 6) * n + r] = o6;
    M[(size_t)bj_col(A, Bk, c0 + 7) * n + r] = o7;
  }
}

// lambda_i = sign(v_i . w_i) * |w_i| on the padded (ld = np) matrices
__global__ __launch_bounds__(256) void bj_evals_k(const double* __restrict__ W, const double* __restrict__ Vv, int n, int np,
                                                   double* __restrict__ ev) {
  __shared__ double sh[8];
  const int i = blockIdx.x;
  double nn = 0.0, dd = 0.0;
  for (int r = threadIdx.x; r < n; r += 256) {
    const double w = W[(size_t)i * np + r];
    nn = fma(w, w, nn);
    dd = fma(w, Vv[(size_t)i * np + r], dd);
  }
  nn = block_sum(nn, sh);
  dd = block_sum(dd, sh);
  if (threadIdx.x == 0) ev[i] = (dd < 0.0 ? -1.0 : 1.0) * sqrt(nn);
}
__global__ void bj_gather_k(const double* __restrict__ Vv, int n, int np, const int* __restrict__ order, float* __restrict__ out) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)n * n) return;
  const int cidx = (int)(idx / n), r = (int)(idx - (size_t)cidx * n);
  out[idx] = (float)Vv[(size_t)order[cidx] * np + r];
}

int k_jacobi_eig(isle_ctx* c, const float* S_host, int n, float* evals_host, float* vecs_dev) {
  TimeScope ts(c, ISLE_T_EVD);
  if (n <= 0) return 0;
  const int np = (n + 63) & ~63;  // zero-padded to whole 64-row tiles / 16-column blocks
  const size_t nn = (size_t)n * n, npp = (size_t)np * np;
  std::vector<double> Wh(npp, 0.0), Vh(npp, 0.0);
  // LAPACK 'U' semantics: the upper triangle defines the matrix.
  double mu = 0.0;
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) Wh[(size_t)j * np + i] = (i <= j) ? (double)S_host[(size_t)j * n + i] : (double)S_host[(size_t)i * n + j];
  for (int i = 0; i < n; ++i) {  // Gershgorin: shift so that the matrix is positive semi-definite
    double off = 0.0;
    for (int j = 0; j < n; ++j)
      if (j != i) off += fabs(Wh[(size_t)j * np + i]);
    mu = std::max(mu, off - Wh[(size_t)i * np + i]);
  }
  double gscale = 0.0;  // largest squared column norm of W = S + mu I  (<= lambda_max^2)
  for (int i = 0; i < n; ++i) {
    Wh[(size_t)i * np + i] += mu;
    Vh[(size_t)i * np + i] = 1.0;
  }
  for (int j = 0; j < n; ++j) {
    double s2 = 0.0;
    for (int i = 0; i < n; ++i) s2 += Wh[(size_t)j * np + i] * Wh[(size_t)j * np + i];
    gscale = std::max(gscale, s2);
  }
  HIPCHK(c, c->jacW.reserve(npp));
  HIPCHK(c, c->jacV.reserve(npp));
  HIPCHK(c, c->small.reserve((size_t)std::max(4096, n + 64)));
  HIPCHK(c, c->part.reserve((size_t)n));
  HIPCHK(c, hipMemcpyAsync(c->jacW.p, Wh.data(), npp * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->jacV.p, Vh.data(), npp * sizeof(double), hipMemcpyHostToDevice, c->stream));
  unsigned int* rot = (unsigned int*)c->small.p;
  const int nblk = np / BJ_W;  // even (np is a multiple of 64)
  // off-diagonal tolerance: eigenvalue errors are quadratic in it, eigenvector errors linear; the results are
  // rounded to fp32 (1.2e-7) by the caller, so 1e-12 leaves five digits of slack
  const double tol = 1e-12;
  const double abs_tol = 1e-14 * (double)n * gscale;
  const int npairs = nblk / 2;
  const int nrowch = np / BJ_ROWS;
  HIPCHK(c, c->jacS.reserve((size_t)npairs * BJ_SCR + (size_t)npairs * nrowch * BJ_P * BJ_P));
  double* gpart = c->jacS.p + (size_t)npairs * BJ_SCR;
  bool converged = (n == 1);
  for (int sweep = 0; sweep < 60 && !converged; ++sweep) {
    HIPCHK(c, hipMemsetAsync(rot, 0, sizeof(unsigned int), c->stream));
    for (int round = 0; round < nblk - 1; ++round) {
      hipLaunchKernelGGL(bj_gram_k, dim3(npairs, nrowch), dim3(256), 0, c->stream, c->jacW.p, np, nblk, round, gpart);
      hipLaunchKernelGGL(bj_inner_k, dim3(npairs), dim3(256), 0, c->stream, np, nblk, round, tol, abs_tol, gpart, nrowch, c->jacS.p, rot,
                         round > 0 ? 1 : 0);
      hipLaunchKernelGGL(bj_apply_k, dim3(npairs, nrowch), dim3(256), 0, c->stream, c->jacW.p, c->jacV.p, np, nblk, round, c->jacS.p,
                         2);
    }
    HIPCHK(c, hipGetLastError());
    unsigned int nrot = 0;
    HIPCHK(c, hipMemcpyAsync(&nrot, rot, sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->knob_on(KN_DEBUG_EVD)) fprintf(stderr, "[evd n=%d] sweep %d rotations %u\n", n, sweep, nrot);
    if (nrot == 0) converged = true;
  }
  if (!converged) return isle_fail(c, ISLE_E_NUMERIC, "Jacobi EVD (n=%d) did not converge in 60 sweeps", n);
  hipLaunchKernelGGL(bj_evals_k, dim3(n), dim3(256), 0, c->stream, c->jacW.p, c->jacV.p, n, np, c->part.p);
  HIPCHK(c, hipGetLastError());
  std::vector<double> ev(n);
  HIPCHK(c, hipMemcpyAsync(ev.data(), c->part.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return ev[x] > ev[y]; });
  for (int i = 0; i < n; ++i) evals_host[i] = (float)(ev[order[i]] - mu);
  int* ord_dev = (int*)(c->small.p + 16);
  HIPCHK(c, hipMemcpyAsync(ord_dev, order.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(bj_gather_k, dim3(cdiv(nn, 256)), dim3(256), 0, c->stream, c->jacV.p, n, np, ord_dev, vecs_dev);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));  // `order` (pageable) must outlive the copy
  return 0;
}

// ------------------------------------------------------------------------------------------
// column sums of squares of a row-major matrix (centre norms, src/sparseMatrix.cpp:1575-1584)
// ------------------------------------------------------------------------------------------
constexpr int CN_ROWS = 128;
// 256 threads = 4 row groups x 64 column lanes: coalesced 256-B row segments, 32 rows per thread, fp64 partial sums
__global__ __launch_bounds__(256) void colnorm_partial_k(const float* __restrict__ Mrm, const float* __restrict__ Sub /*nullable*/,
                                                          uint64_t rows, int k, int ldk, double* __restrict__ part) {
  __shared__ double sh[4][64];
  const uint64_t r0 = (uint64_t)blockIdx.x * CN_ROWS;
  const uint64_t r1 = min(rows, r0 + CN_ROWS);
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  for (int c0 = 0; c0 < k; c0 += 64) {
    const int cc = c0 + cl;
    double s = 0.0;
    if (cc < k) {
      uint64_t r = r0 + rg;
      for (; r + 28 < r1; r += 32) {  // eight rows in flight per thread, accumulated in row order
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          a[u] = Mrm[(r + 4 * u) * ldk + cc];
          b[u] = Sub ? Sub[(r + 4 * u) * ldk + cc] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const double x = (double)a[u] - (double)b[u];
          s = fma(x, x, s);
        }
      }
      for (; r < r1; r += 4) {
        const double x = (double)Mrm[r * ldk + cc] - (Sub ? (double)Sub[r * ldk + cc] : 0.0);
        s = fma(x, x, s);
      }
    }
    sh[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && cc < k) part[(size_t)blockIdx.x * k + cc] = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
    __syncthreads();
  }
}
// out[cc] = sum over chunks, fixed order: 4 chunk groups x 64 columns per workgroup
__global__ __launch_bounds__(256) void colnorm_reduce_k(const double* __restrict__ part, int nchunks, int k, float* __restrict__ out) {
  __shared__ double sh[4][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int cc = blockIdx.x * 64 + cl;
  double s = 0.0;
  if (cc < k) {
    int ch = rg;
    for (; ch + 28 < nchunks; ch += 32) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(ch + 4 * u) * k + cc];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; ch < nchunks; ch += 4) s += part[(size_t)ch * k + cc];
  }
  sh[rg][cl] = s;
  __syncthreads();
  if (rg == 0 && cc < k) out[cc] = (float)((sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]));
}
// out[c] = sum_r (M[r][c] - Sub[r][c])^2   (Sub nullable: plain column norms)
int k_colnorms_rm(isle_ctx* c, const float* Mrm, uint64_t rows, int k, int ldk, float* out, const float* Sub) {
  const int nchunks = cdiv(rows, CN_ROWS);
  HIPCHK(c, c->part.reserve((size_t)nchunks * k));
  hipLaunchKernelGGL(colnorm_partial_k, dim3(nchunks), dim3(256), 0, c->stream, Mrm, Sub, rows, k, ldk, c->part.p);
  hipLaunchKernelGGL(colnorm_reduce_k, dim3(cdiv(k, 64)), dim3(256), 0, c->stream, c->part.p, nchunks, k, out);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// centre /= cluster size, true division, empty clusters stay zero (src/sparseMatrix.cpp:1641-1646)
// (round 6: four rows per workgroup, a float4 per thread and trip — the flat form spent a 64-bit modulo per element: 316 us for 0.8 GB at k = 1000)
__global__ __launch_bounds__(256) void scale_centers_k(float* __restrict__ Crm, uint64_t rows, int k, int ldk, const int* __restrict__ counts) {
  const uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* r = Crm + row * (uint64_t)ldk;
  for (int c4 = (int)(threadIdx.x & 63) * 4; c4 < ldk; c4 += 256) {  // ldk is a multiple of 4
    float4 v = *reinterpret_cast<float4*>(r + c4);
    float* e = reinterpret_cast<float*>(&v);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (c4 + j < k) {
        const float div = (float)counts[c4 + j];
        if (div > 0.0f) e[j] /= div;
      }
    *reinterpret_cast<float4*>(r + c4) = v;
  }
}
int k_scale_centers(isle_ctx* c, float* Crm, uint64_t rows, int k, int ldk, const int* counts) {
  TimeScope ts(c, ISLE_T_SPARSE_UPDATE);
  if (rows == 0) return 0;
  if ((ldk & 3) != 0 || ((uintptr_t)Crm & 15) != 0) return isle_fail(c, ISLE_E_ARG, "scale_centers: leading dimension %d not a multiple of 4", ldk);
  hipLaunchKernelGGL(scale_centers_k, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, c->stream, Crm, rows, k, ldk, counts);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// arma::eig_sym stand-in: eigenvalues (all n, descending) on the host, the nvec leading eigenvectors on the device
// (n x nvec col-major).  Tridiagonalisation + bisection + twisted factorisation (evd_tridiag.hip) first; the block Jacobi
// solver above is the fallback for sizes it does not take and for spectra whose eigenvectors fail its orthogonality check
// (clusters far tighter than Ritz matrices show).  ISLE_EVD_JACOBI=1 forces the Jacobi solver.
int k_eig_small(isle_ctx* c, const float* S_host, int n, float* evals_host, float* vecs_dev, int nvec) {
  if (n >= 16 && !c->knob_on(KN_EVD_JACOBI)) {
    int rc;
    {
      TimeScope ts(c, ISLE_T_EVD);
      rc = k_tridiag_eig(c, S_host, n, evals_host, vecs_dev, nvec);
    }
    if (rc <= 0) return rc;
  }
  return k_jacobi_eig(c, S_host, n, evals_host, vecs_dev);
}
