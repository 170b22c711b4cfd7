// isle_amd/csrc/gemm_bf16x3.h — C (M x N) = A (M x K) * B (K x N), all column-major f32, on the bf16 matrix cores of gfx950 with both
// operands split into THREE bf16 terms:  x = x0 + x1 + x2  (x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1): 3 x 8 significand
// bits, the subtractions are exact), and
//     a b  ~  a0 b0 + (a0 b1 + a1 b0) + (a1 b1 + a0 b2 + a2 b0)
// — the six products down to 2^-16 of the leading one; the three dropped ones (a1 b2, a2 b1, a2 b2) are below 2^-23 |a b|, the
// rounding of the f32 product itself.  Every partial product is exact in the MFMA (8 x 8 bits) and summed in f32, as
// v_mfma_f32_32x32x2_f32 sums its own.  Six v_mfma_f32_32x32x16_bf16 (32 cycles each) do the work of eight v_mfma_f32_32x32x2_f32
// (64 cycles each): 2.7 x the rate of gemm_f32.h at the same accuracy class (tools/microbench/gemm3_probe.hip: error against fp64
// relative to sum |a_k b_k|, both kernels on the same random operands).
//
// Used for the D x k x k products of the assignment steps only (the projected full pass of Lloyd in span(U), the first assignment of
// Lloyd on B: src/sparseMatrix.cpp:1819-1826 in its P * C^T form) — distances that feed an argmin, evaluated in another summation order
// than the reference's anyway.  The eigen-solver's products (Ritz rotation, lift) stay on gemm_f32.h.
//
// Shape: as gemm_f32.h — a workgroup of WAVES_M x WAVES_N waves owns a TM x TN tile, a wave WMT x WNT tiles of 32 x 32; K is walked
// in slabs of 16 (one MFMA k-step) through a two-stage LDS ring.  A is split on the fly between its global load and the LDS store
// (three planes of 16-byte units = 8 consecutive k of one row: one ds_read_b128 is a lane's whole operand); B (K x N, a few MB) is
// split once per call into the same units in global memory (gemm3_split_b_k), zero-padded to whole slabs and tiles, and copied.
// The MFMA is issued "transposed" (first operand = B fragment), so a lane owns one ROW m of C: same epilogue as gemm_f32.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace isle_gemm3 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline void split3(float x, __bf16& x0, __bf16& x1, __bf16& x2) {
  x0 = (__bf16)x;
  const float r1 = x - (float)x0;
  x1 = (__bf16)r1;
  const float r2 = r1 - (float)x1;
  x2 = (__bf16)r2;
}

// B3[(p * Kp8 + q) * Np + n] = the p-th bf16 term of B[8 q .. 8 q + 7][n]  (zero beyond K and N)
__global__ __launch_bounds__(256) void gemm3_split_b_k(const float* __restrict__ B, int ldb, int K, int N, int Kp8, int Np, bf16x8* __restrict__ B3) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)Kp8 * Np) return;
  const int q = (int)(i / Np), n = (int)(i - (size_t)q * Np);
  bf16x8 v[3];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * q + j;
    const float x = (k < K && n < N) ? B[(size_t)n * ldb + k] : 0.f;
    __bf16 t0, t1, t2;
    split3(x, t0, t1, t2);
    v[0][j] = t0;
    v[1][j] = t1;
    v[2][j] = t2;
  }
#pragma unroll
  for (int p = 0; p < 3; ++p) B3[((size_t)p * Kp8 + q) * Np + n] = v[p];
}

// NP_ = 3: the six products above.  NP_ = 2: x ~ x0 + x1 (the remainder is at most 2^-16 |x|: bf16 rounds to 8 significand bits) and
// a b ~ a0 b0 + (a0 b1 + a1 b0): three products, |error| <= 3 * 2^-16 |a b| per term (a1 b1 and the two remainders, 2^-16 each; 1e-6 on
// data) — for callers that carry that bound along
// (dense.hip: the assignment steps widen their bounds by it and recompute the rows whose arg-min it leaves open with NP_ = 3).
template <int WMT_, int WNT_, int WAVES_M_, int WAVES_N_, int OCC_, int TK_ = 16, int NP_ = 3>
struct Cfg {
  static constexpr int WMT = WMT_, WNT = WNT_, WAVES_M = WAVES_M_, WAVES_N = WAVES_N_, OCC = OCC_, TK = TK_;  // TK: 16 or 32 (one or two MFMA k-steps per slab)
  static constexpr int NP = NP_;     // bf16 terms per operand
  static constexpr int KO = TK / 8;  // k-octets per slab
  static constexpr int TM = 32 * WMT * WAVES_M, TN = 32 * WNT * WAVES_N;
  static constexpr int NT = 64 * WAVES_M * WAVES_N;
  static constexpr int A_STAGE = NP * KO * TM, B_STAGE = NP * KO * TN;  // 16-byte units per stage: [plane NP][octet KO][row]
  static constexpr size_t LDS_BYTES = (size_t)2 * (A_STAGE + B_STAGE) * 16;
  static constexpr int A_UNITS = TM * (TK / 4) / NT;        // k-quads (4 floats of one row) per thread and slab
  static constexpr int B_UNITS = (B_STAGE + NT - 1) / NT;   // 16-byte units of the split B per thread and slab
  static_assert(TM * (TK / 4) % NT == 0 && (TK == 16 || TK == 32) && (NP == 2 || NP == 3), "tile / thread counts");
};

struct StoreC {
  static constexpr int kGroup = 0;  // element epilogue: operator()(row, column, value)
  float* __restrict__ C;
  uint64_t ldc;
  __device__ inline void operator()(uint64_t m, int n, float v) const { C[(uint64_t)n * ldc + m] = v; }
};

// GROUP epilogues (kGroup = 8 or 32): the product is never stored.  Per row and per group of kGroup consecutive columns the kernel forms
// from its accumulators  dist(value, column, rowdata(row))  for every column, the smallest distance with its first column, the runner-up
// and the largest aux(column), and hands them to  group(row, group index, rowdata, m1, i1, m2, auxmax)  (one lane per row and group); per
// row and per 64-column slot (the columns one wave owns) the best of the slot's groups goes to  slot(row, slot index, rowdata, m1, i1,
// m2 of that group, auxmax of that group, smallest distance among the slot's other columns).  The assignment steps of k-means use this (dense.hip): group bounds and the per-slot
// candidates leave the kernel, a small kernel picks the row's winner — instead of D x k x 4 bytes written and read again.
// A lane holds, of a row of a 32 x 32 tile, columns (r & 3) + 8 (r >> 2) + 4 h in register r (h = lane / 32): a group of 8 is registers
// 4 q .. 4 q + 3 of the lanes l and l + 32, a group of 32 all sixteen of both; the lane pair is merged by __shfl_xor(., 32).
struct Top2 {
  float m1, m2, ax;
  uint32_t i1;
};
__device__ inline void top2_take(Top2& t, float d, uint32_t idx, float ax) {  // columns arrive in ascending order: a tie keeps the earlier one
  if (d < t.m1) {
    t.m2 = t.m1;
    t.m1 = d;
    t.i1 = idx;
  } else {
    t.m2 = fminf(t.m2, d);
  }
  t.ax = fmaxf(t.ax, ax);
}
__device__ inline void top2_merge(Top2& t, float om1, float om2, uint32_t oi1, float oax) {
  if (om1 < t.m1 || (om1 == t.m1 && oi1 < t.i1)) {
    t.m2 = fminf(t.m1, om2);
    t.m1 = om1;
    t.i1 = oi1;
  } else {
    t.m2 = fminf(t.m2, om1);
  }
  t.ax = fmaxf(t.ax, oax);
}
__device__ inline void top2_pair(Top2& t) {  // with the lane that holds the other half of the row's columns
  const float om1 = __shfl_xor(t.m1, 32), om2 = __shfl_xor(t.m2, 32), oax = __shfl_xor(t.ax, 32);
  const uint32_t oi1 = (uint32_t)__shfl_xor((int)t.i1, 32);
  top2_merge(t, om1, om2, oi1, oax);
}

// The epilogue of a workgroup's TM x TN tile from the waves' accumulators (shared by gemm_bf16x3_k and gemm_bf16x2_dma_k)
template <class CF, class Epi>
__device__ inline void tile_epilogue(f32x16 (&acc)[CF::WNT][CF::WMT], uint64_t m0, int n0, int wm, int wn, int l31, int h, uint64_t M, int N, const Epi& epi) {
  constexpr int TN = CF::TN, WMT = CF::WMT, WNT = CF::WNT;
  if constexpr (Epi::kGroup != 0) {
    static_assert(Epi::kGroup == 8 || Epi::kGroup == 32, "group epilogue: groups of 8 or 32 columns");
#pragma unroll
    for (int i = 0; i < WMT; ++i) {
      const uint64_t m = m0 + wm * (32 * WMT) + i * 32 + l31;
      const bool live = m < M;
      const float rd = epi.rowdata(live ? m : M - 1);
      Top2 best{3.4e38f, 3.4e38f, 0.f, 0xffffffffu};  // of this wave's columns (one 64-column slot per 32 WNT columns ... WNT = 2)
      float run2 = 3.4e38f;                           // smallest distance of the slot's columns other than best.i1
      auto fold = [&](const Top2& t) {                // groups in ascending order: a tie keeps the earlier group's (lower) column
        if (t.m1 < best.m1) {
          run2 = fminf(fminf(run2, best.m1), t.m2);
          best = t;
        } else {
          run2 = fminf(run2, t.m1);
        }
      };
#pragma unroll
      for (int j = 0; j < WNT; ++j) {
        const int cb = n0 + wn * (32 * WNT) + j * 32;  // first column of the 32 x 32 tile
        if (cb >= N) continue;                          // wave-uniform
        float dist[16], ax[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int col = cb + 4 * h + (r & 3) + 8 * (r >> 2);
          const bool in = col < N;
          dist[r] = in ? epi.dist(acc[j][i][r], col, rd) : 3.4e38f;
          ax[r] = in ? epi.aux(col) : 0.f;
        }
        if constexpr (Epi::kGroup == 8) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            Top2 t{3.4e38f, 3.4e38f, 0.f, 0xffffffffu};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int col = cb + 4 * h + u + 8 * q;
              if (col < N) top2_take(t, dist[4 * q + u], (uint32_t)col, ax[4 * q + u]);
            }
            top2_pair(t);
            if (cb + 8 * q < N) {  // wave-uniform
              if (live && h == (q & 1)) epi.group(m, (cb >> 3) + q, rd, t.m1, t.i1, t.m2, t.ax);
              fold(t);
            }
          }
        } else {
          Top2 t{3.4e38f, 3.4e38f, 0.f, 0xffffffffu};
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int col = cb + 4 * h + (r & 3) + 8 * (r >> 2);
            if (col < N) top2_take(t, dist[r], (uint32_t)col, ax[r]);
          }
          top2_pair(t);
          if (live && h == (j & 1)) epi.group(m, cb >> 5, rd, t.m1, t.i1, t.m2, t.ax);
          fold(t);
        }
      }
      if (live && h == 0 && n0 + wn * (32 * WNT) < N) epi.slot(m, (n0 + wn * (32 * WNT)) / (32 * WNT), rd, best.m1, best.i1, best.m2, best.ax, run2);
    }
  } else {
  // lane owns row m = l31 of each 32 x 32 tile; register r holds column (r & 3) + 8 (r >> 2) + 4 h
  const bool full_n = n0 + TN <= N;
#pragma unroll
  for (int i = 0; i < WMT; ++i) {
    const uint64_t m = m0 + wm * (32 * WMT) + i * 32 + l31;
    if (m < M) {
#pragma unroll
      for (int j = 0; j < WNT; ++j) {
        const int nb = n0 + wn * (32 * WNT) + j * 32 + 4 * h;
        if (full_n) {
#pragma unroll
          for (int r = 0; r < 16; ++r) epi(m, nb + (r & 3) + 8 * (r >> 2), acc[j][i][r]);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int n = nb + (r & 3) + 8 * (r >> 2);
            if (n < N) epi(m, n, acc[j][i][r]);
          }
        }
      }
    }
  }
  }
}

template <class CF, class Epi>
__global__ __launch_bounds__(CF::NT, CF::OCC) void gemm_bf16x3_k(const float* __restrict__ A, uint64_t M, int K, const bf16x8* __restrict__ B3, int Kp8,
                                                                  int Np, int N, uint32_t nMB, uint32_t nNT, Epi epi) {
  constexpr int TM = CF::TM, TN = CF::TN, NT = CF::NT, WMT = CF::WMT, WNT = CF::WNT, TK = CF::TK, KO = CF::KO, NP = CF::NP;
  extern __shared__ bf16x8 lds3[];
  bf16x8* As = lds3;                    // [2][NP][KO][TM]
  bf16x8* Bs = lds3 + 2 * CF::A_STAGE;  // [2][NP][KO][TN]
  // tile of this workgroup: xcd = id % 8 owns row blocks xcd, xcd + 8, ...; its N-tiles are consecutive slots (gemm_f32.h)
  const uint32_t wg = blockIdx.x, xcd = wg & 7u, slot = wg >> 3;
  const uint32_t mb = (slot / nNT) * 8u + xcd, nt = slot % nNT;
  if (mb >= nMB) return;
  const uint64_t m0 = (uint64_t)mb * TM;
  const int n0 = (int)nt * TN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave % CF::WAVES_M, wn = wave / CF::WAVES_M;

  uint32_t arow[CF::A_UNITS];
  int akq[CF::A_UNITS];
#pragma unroll
  for (int u = 0; u < CF::A_UNITS; ++u) {
    const int unit = tid + u * NT;
    akq[u] = unit / TM;
    const uint64_t m = m0 + (uint32_t)(unit % TM);
    arow[u] = (uint32_t)(m < M ? m : M - 1);  // clamped: rows past the end are computed on valid data, never stored
  }
  float ra[4 * CF::A_UNITS];
  uint4 rb[CF::B_UNITS];
  auto load_slab = [&](int k0) {
#pragma unroll
    for (int u = 0; u < CF::A_UNITS; ++u) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int k = min(k0 + 4 * akq[u] + t, K - 1);  // the k tail is zero on the B side, A stays finite data
        ra[4 * u + t] = A[(uint64_t)k * M + arow[u]];
      }
    }
    const int q0 = k0 >> 3;
#pragma unroll
    for (int u = 0; u < CF::B_UNITS; ++u) {
      const int unit = min(tid + u * NT, CF::B_STAGE - 1);  // [plane][octet][n]
      const int p = unit / (KO * TN), r = unit - p * KO * TN, q = r / TN, n = r - q * TN;
      rb[u] = *reinterpret_cast<const uint4*>(B3 + ((size_t)p * Kp8 + q0 + q) * Np + n0 + n);
    }
  };
  auto store_slab = [&](int stage) {
    char* a = reinterpret_cast<char*>(As + stage * CF::A_STAGE);
#pragma unroll
    for (int u = 0; u < CF::A_UNITS; ++u) {
      const int unit = tid + u * NT;
      const int row = unit % TM;
      bf16x4 v[3];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        __bf16 t0, t1, t2;
        split3(ra[4 * u + t], t0, t1, t2);
        v[0][t] = t0;
        v[1][t] = t1;
        v[2][t] = t2;
      }
#pragma unroll
      for (int p = 0; p < NP; ++p) *reinterpret_cast<bf16x4*>(a + ((size_t)((p * KO + (akq[u] >> 1)) * TM + row)) * 16 + (akq[u] & 1) * 8) = v[p];
    }
    uint4* b = reinterpret_cast<uint4*>(Bs + stage * CF::B_STAGE);
#pragma unroll
    for (int u = 0; u < CF::B_UNITS; ++u) {
      const int unit = tid + u * NT;
      if (unit < CF::B_STAGE) b[unit] = rb[u];
    }
  };

  f32x16 acc[WNT][WMT];
#pragma unroll
  for (int j = 0; j < WNT; ++j)
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  // (staging two slabs ahead through a second register set was measured: 161 -> 120 TFLOP/s where it did not spill, 43 where it did)
  const int nslab = (K + TK - 1) / TK;
  load_slab(0);
  store_slab(0);
  __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    const int cur = s & 1;
    if (s + 1 < nslab) load_slab((s + 1) * TK);
    const bf16x8* a = As + cur * CF::A_STAGE + wm * (32 * WMT) + l31;
    const bf16x8* b = Bs + cur * CF::B_STAGE + wn * (32 * WNT) + l31;
#pragma unroll
    for (int ks = 0; ks < TK / 16; ++ks) {
      const int oc = 2 * ks + h;  // the lane's octet of this k-step
      bf16x8 bv[WNT][NP];
#pragma unroll
      for (int j = 0; j < WNT; ++j)
#pragma unroll
        for (int p = 0; p < NP; ++p) bv[j][p] = b[(p * KO + oc) * TN + 32 * j];
      // the small terms first: a2 b0; a1 b1, a1 b0; a0 b2, a0 b1, a0 b0 — and the wave's WMT x WNT accumulators in turn inside a term, so
      // that an MFMA never waits for the one before it
#pragma unroll
      for (int pa = NP - 1; pa >= 0; --pa) {
        bf16x8 av[WMT];
#pragma unroll
        for (int i = 0; i < WMT; ++i) av[i] = a[(pa * KO + oc) * TM + 32 * i];
#pragma unroll
        for (int pb = NP - 1 - pa; pb >= 0; --pb)
#pragma unroll
          for (int i = 0; i < WMT; ++i)
#pragma unroll
            for (int j = 0; j < WNT; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[j][pb], av[i], acc[j][i], 0, 0, 0);
      }
    }
    if (s + 1 < nslab) store_slab(cur ^ 1);
    __syncthreads();
  }

  tile_epilogue<CF, Epi>(acc, m0, n0, wm, wn, l31, h, M, N, epi);
}

// padded extents of the split B for a tile width TN
inline int kp8_of(int K) { return 4 * ((K + 31) / 32); }  // whole slabs of 16 or 32
template <class CF>
inline int np_of(int N) { return (N + CF::TN - 1) / CF::TN * CF::TN; }

// B3 must hold 3 * kp8_of(K) * np_of<CF>(N) units of 16 bytes
template <class CF, class Epi>
inline hipError_t launch(hipStream_t stream, const float* A, uint64_t M, int K, const float* B, int ldb, int N, void* B3, Epi epi) {
  const int Kp8 = kp8_of(K), Np = np_of<CF>(N);
  const size_t nb = (size_t)Kp8 * Np;
  hipLaunchKernelGGL(gemm3_split_b_k, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, stream, B, ldb, K, N, Kp8, Np, reinterpret_cast<bf16x8*>(B3));
  const uint32_t nMB = (uint32_t)((M + CF::TM - 1) / CF::TM), nNT = (uint32_t)(Np / CF::TN);
  const uint64_t slots = (uint64_t)((nMB + 7) / 8) * nNT;  // per XCD
  static bool lds_attr_set[64] = {};  // per device
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64 || !lds_attr_set[dev]) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16x3_k<CF, Epi>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CF::LDS_BYTES);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) lds_attr_set[dev] = true;
  }
  hipLaunchKernelGGL((gemm_bf16x3_k<CF, Epi>), dim3((uint32_t)(slots * 8)), dim3(CF::NT), CF::LDS_BYTES, stream, A, M, K, reinterpret_cast<const bf16x8*>(B3), Kp8,
                     Np, N, nMB, nNT, epi);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------------------------
// The two-term product with BOTH operands split beforehand and staged by LDS-DMA through a ring of NS stages (round 5).
// gemm_bf16x3_k<Cfg<..., 16, 2>> stages the next slab through registers (global load -> split -> ds_write) with ONE slab of lookahead: a
// slab is 12 MFMAs per wave = 1536 matrix-core cycles per SIMD, a loaded slab arrives after an HBM latency of 2000 - 4000 — the kernel
// runs at the latency, not at the matrix cores (312 of 830 TFLOP/s f32-equivalent; 4100 cycles per slab measured).  Here A (the D x k
// projection, the same for every call of a k-means phase) is split ONCE into its two bf16 terms, stored as the LDS image of each
// (row block, slab) — 16 KB contiguous: [term][octet][row] 16-byte units — and a slab is fetched by two global_load_lds_dwordx4 per wave
// (no VGPRs, no VALU), three slabs ahead: counted s_waitcnt vmcnt(N) and a raw s_barrier (cdna_hip_programming.md, "Pipelining across
// barriers": __syncthreads() would drain the ring with vmcnt(0)).  Same products in the same order as the register-staged kernel: the
// accumulators, and with them every distance and bound of the epilogues, are bit-identical.
// ------------------------------------------------------------------------------------------------------------------------------------
template <int NS_, int TK_ = 16>
struct CfgDma {
  static constexpr int NS = NS_;  // stages of the ring
  static constexpr int WMT = 2, WNT = 2, WAVES_M = 4, WAVES_N = 4, TK = TK_, KO = TK_ / 8, NP = 2, OCC = 4;
  static constexpr int TM = 256, TN = 256, NT = 1024;
  static constexpr int A_STAGE = NP * KO * TM, B_STAGE = NP * KO * TN, STAGE = A_STAGE + B_STAGE;  // 16-byte units
  static constexpr int PPW = STAGE / 64 / 16;  // 1-KiB pieces of a slab per wave (A: waves 0 .. 7, B: waves 8 .. 15)
  static constexpr size_t LDS_BYTES = (size_t)NS * STAGE * 16;
  static_assert(TK_ == 16 || TK_ == 32, "slabs of 16 or 32");
};
inline int nslab_of(int K, int TK) { return (K + TK - 1) / TK; }
// units of 16 bytes of the split A of an M x K operand (whole row blocks of 256, whole slabs of TK)
inline size_t a2_units(uint64_t M, int K, int TK = 16) { return (size_t)((M + 255) / 256) * nslab_of(K, TK) * (4 * (TK / 8)) * 256; }

// A (M x K column-major: element (m, k) at A[k * M + m]) -> A2: per (row block mb, slab s) the LDS image [term p][octet oc][row r] of 16-byte
// units, unit = 8 bf16 = term p of A[256 mb + r][TK s + 8 oc .. + 7] (zero beyond M and K).  grid (row blocks, octets); a thread per row:
// every read and write of a wave is one contiguous run.
__global__ __launch_bounds__(256) void gemm2_split_a_k(const float* __restrict__ A, uint64_t M, int K, int nslab, int KO, bf16x8* __restrict__ A2) {
  const uint64_t mb = blockIdx.x, m = mb * 256 + threadIdx.x;
  const int q = blockIdx.y;
  bf16x8 v0, v1;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * q + j;
    const float x = (m < M && k < K) ? A[(uint64_t)k * M + m] : 0.f;
    const __bf16 x0 = (__bf16)x;
    v0[j] = x0;
    v1[j] = (__bf16)(x - (float)x0);
  }
  const size_t base = ((mb * nslab + (size_t)(q / KO)) * (2 * KO) + (size_t)(q % KO)) * 256 + threadIdx.x;  // term 0: [p = 0][oc]
  A2[base] = v0;
  A2[base + (size_t)KO * 256] = v1;  // term 1: [p = 1][oc]
}
inline hipError_t split_a(hipStream_t stream, const float* A, uint64_t M, int K, void* A2, int TK = 16) {
  const int nslab = nslab_of(K, TK);
  const uint64_t nMB = (M + 255) / 256;
  if (nMB == 0) return hipSuccess;
  if (nMB >= (1ull << 31)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gemm2_split_a_k, dim3((uint32_t)nMB, (uint32_t)(nslab * (TK / 8))), dim3(256), 0, stream, A, M, K, nslab, TK / 8, reinterpret_cast<bf16x8*>(A2));
  return hipGetLastError();
}

template <class CF, class Epi>
__global__ __launch_bounds__(CF::NT) void gemm_bf16x2_dma_k(const bf16x8* __restrict__ A2, uint64_t M, int K, const bf16x8* __restrict__ B3, int Kp8, int Np,
                                                             int N, uint32_t nMB, uint32_t nNT, Epi epi) {
  constexpr int TM = CF::TM, TN = CF::TN, WMT = CF::WMT, WNT = CF::WNT, KO = CF::KO, NP = CF::NP, NS = CF::NS, TK = CF::TK, PPW = CF::PPW;
  static_assert((NS & (NS - 1)) == 0 && NS >= 2 && NS <= 4, "ring of 2 or 4 stages");
  extern __shared__ bf16x8 ldsr[];  // [NS][A: NP x KO x TM | B: NP x KO x TN] — ONE array: the DMA ring and the fragment reads share it
  const uint32_t wg = blockIdx.x, xcd = wg & 7u, slot = wg >> 3;
  const uint32_t mb = (slot / nNT) * 8u + xcd, nt = slot % nNT;
  if (mb >= nMB) return;
  const uint64_t m0 = (uint64_t)mb * TM;
  const int n0 = (int)nt * TN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave % CF::WAVES_M, wn = wave / CF::WAVES_M;
  const int nslab = (K + TK - 1) / TK;
  // this wave's PPW 1-KiB pieces of a slab: the A stage is walked by waves 0 .. 7, the B stage by waves 8 .. 15
  const bool mine_a = wave < 8;
  const int w8 = mine_a ? wave : wave - 8;
  const bf16x8* ga = A2 + ((size_t)mb * nslab) * CF::A_STAGE + (size_t)(w8 * PPW) * 64 + lane;  // + s * A_STAGE (+ 64 per further piece)
  // B stage unit u = (w8 * PPW + i) * 64 + lane = [p][oc][n]: p = u / (KO * 256), oc = (u / 256) % KO, n = u % 256
  const bf16x8* gb[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int u = (w8 * PPW + i) * 64 + lane;
    gb[i] = B3 + ((size_t)((u / (KO * 256)) * Kp8 + ((u >> 8) % KO))) * Np + n0 + (u & 255);  // + KO s * Np
  }
  const uint32_t lpiece = (uint32_t)(((mine_a ? 0 : CF::A_STAGE) + w8 * PPW * 64) * 16);  // byte offset of the wave's first piece in a stage
  auto issue = [&](int s) {
    char* l = reinterpret_cast<char*>(ldsr) + (size_t)(s & (NS - 1)) * CF::STAGE * 16 + lpiece;
    if (mine_a) {
      const bf16x8* g = ga + (size_t)s * CF::A_STAGE;
#pragma unroll
      for (int i = 0; i < PPW; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 64 * i), (__attribute__((address_space(3))) void*)(l + 1024 * i), 16, 0, 0);
    } else {
      const size_t o = (size_t)(KO * s) * Np;
#pragma unroll
      for (int i = 0; i < PPW; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb[i] + o), (__attribute__((address_space(3))) void*)(l + 1024 * i), 16, 0, 0);
    }
  };

  f32x16 acc[WNT][WMT];
#pragma unroll
  for (int j = 0; j < WNT; ++j)
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nslab) issue(s);
  for (int s = 0; s < nslab; ++s) {
    // this wave's pieces of slab s have landed once at most the pieces of the NS - 2 slabs behind it are outstanding (PPW per slab)
    if (s + NS - 2 < nslab) __builtin_amdgcn_s_waitcnt(0x0f70 | (PPW * (NS - 2)));
    else __builtin_amdgcn_s_waitcnt(0x0f70);
    __builtin_amdgcn_s_barrier();  // everybody's pieces of slab s are in LDS, and everybody is done with slab s - 1: its stage is free
    if (s + NS - 1 < nslab) issue(s + NS - 1);
    const bf16x8* st = ldsr + (size_t)(s & (NS - 1)) * CF::STAGE;
    const bf16x8* a = st + wm * (32 * WMT) + l31;
    const bf16x8* b = st + CF::A_STAGE + wn * (32 * WNT) + l31;
#pragma unroll
    for (int ks = 0; ks < TK / 16; ++ks) {
      const int oc = 2 * ks + h;  // the lane's octet of this k-step
      bf16x8 bv[WNT][NP];
#pragma unroll
      for (int j = 0; j < WNT; ++j)
#pragma unroll
        for (int p = 0; p < NP; ++p) bv[j][p] = b[(p * KO + oc) * TN + 32 * j];
#pragma unroll
      for (int pa = NP - 1; pa >= 0; --pa) {  // the order of gemm_bf16x3_k: a1 b0; a0 b1, a0 b0
        bf16x8 av[WMT];
#pragma unroll
        for (int i = 0; i < WMT; ++i) av[i] = a[(pa * KO + oc) * TM + 32 * i];
#pragma unroll
        for (int pb = NP - 1 - pa; pb >= 0; --pb)
#pragma unroll
          for (int i = 0; i < WMT; ++i)
#pragma unroll
            for (int j = 0; j < WNT; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[j][pb], av[i], acc[j][i], 0, 0, 0);
      }
    }
  }
  tile_epilogue<CF, Epi>(acc, m0, n0, wm, wn, l31, h, M, N, epi);
}

// B3 as for launch<> (three planes' worth of room; the two-term kernel reads planes 0 and 1); A2 from split_a with the same TK
template <class CF, class Epi>
inline hipError_t launch_dma(hipStream_t stream, const void* A2, uint64_t M, int K, const float* B, int ldb, int N, void* B3, Epi epi) {
  const int Kp8 = kp8_of(K), Np = (N + CF::TN - 1) / CF::TN * CF::TN;
  const size_t nb = (size_t)Kp8 * Np;
  hipLaunchKernelGGL(gemm3_split_b_k, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, stream, B, ldb, K, N, Kp8, Np, reinterpret_cast<bf16x8*>(B3));
  const uint32_t nMB = (uint32_t)((M + CF::TM - 1) / CF::TM), nNT = (uint32_t)(Np / CF::TN);
  const uint64_t slots = (uint64_t)((nMB + 7) / 8) * nNT;
  static bool lds_attr_set[64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64 || !lds_attr_set[dev]) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16x2_dma_k<CF, Epi>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CF::LDS_BYTES);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) lds_attr_set[dev] = true;
  }
  hipLaunchKernelGGL((gemm_bf16x2_dma_k<CF, Epi>), dim3((uint32_t)(slots * 8)), dim3(CF::NT), CF::LDS_BYTES, stream, reinterpret_cast<const bf16x8*>(A2), M, K,
                     reinterpret_cast<const bf16x8*>(B3), Kp8, Np, N, nMB, nNT, epi);
  return hipGetLastError();
}

}  // namespace isle_gemm3
