// isle_amd/csrc/gemm_f32.h — C (M x N) = A (M x K) * B (K x N), all column-major, exact f32 on the matrix cores of gfx950.
//
// Stands for the plain sgemm calls of the reference's hot path: the Ritz rotation of BlockKs::truncate
// (block-ks/restarted_block_ks.h:166-167), the lift of the projected centres (src/sparseMatrix.cpp:1446-1449) and the
// D x k x k products of the projected / first word-space assignment (src/sparseMatrix.cpp:1819-1826 in its P * C^T form).
// M is the long dimension (vocabulary or documents: 1e5 ... 1e7), N and K are of the order of the topic count.
//
// Shape (template Cfg): a workgroup of WAVES_M x WAVES_N waves owns a TM x TN tile of C, every wave WMT x WNT tiles of
// v_mfma_f32_32x32x2_f32 (16 accumulator registers each).  K is walked in slabs of TK through a two-stage LDS ring: the global loads of
// slab s + 1 are issued before the MFMAs of slab s and written to the other stage behind them, one barrier per slab.
//
// Operand layout in LDS: [k / 4][row][k % 4] — four consecutive k of one row are one 16-byte unit, so that ONE ds_read_b128 feeds a
// lane's operand of FOUR consecutive MFMAs.  The MFMA takes k = h (h = lane / 32) from a lane; inside a group of 8 k the t-th MFMA is
// given k = 4 h + t instead of 2 t + h — the sum over k is the same set of products in another order, and both operands agree on it.
// Lanes of one 16-lane LDS group read consecutive 16-byte units: conflict-free.  The MFMA is issued "transposed" (first operand = B
// fragment), so a lane owns one ROW m of C and every store of a half-wave is 128 contiguous bytes of a column.
//
// Workgroup -> tile map: consecutive workgroup ids go to different XCDs, so tiles are dealt such that the N-tiles of one row block run on
// ONE XCD back to back: the TM x K block of A is fetched from HBM once and found in that XCD's L2 by the other N-tiles.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace isle_gemm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WMT_, int WNT_, int WAVES_M_, int WAVES_N_, int TK_, int OCC_>
struct Cfg {
  static constexpr int WMT = WMT_, WNT = WNT_, WAVES_M = WAVES_M_, WAVES_N = WAVES_N_, TK = TK_, OCC = OCC_;
  static constexpr int TM = 32 * WMT * WAVES_M, TN = 32 * WNT * WAVES_N;
  static constexpr int NT = 64 * WAVES_M * WAVES_N;  // threads
  static constexpr int KQ = TK / 4;                  // 16-byte k-units per slab
  static constexpr int A_STRIDE = TM;                // float4 units per k-unit of A
  static constexpr int B_STRIDE = TN + 1;            // padded: the lanes that write one column's k land in different banks
  static constexpr int A_STAGE = KQ * A_STRIDE, B_STAGE = KQ * B_STRIDE;  // float4 units per stage
  static constexpr size_t LDS_BYTES = (size_t)2 * (A_STAGE + B_STAGE) * 16;
  static constexpr int A_UNITS = TM * KQ / NT;       // float4 units of A per thread and slab
  static constexpr int B_ELEMS = TN * TK / NT;       // floats of B per thread and slab
  static_assert(TM * KQ % NT == 0 && TN * TK % NT == 0 && NT % TK == 0, "tile / thread counts");
};

// Epilogue: called once per (lane, accumulator register) with the element's row, column and value
struct StoreC {
  float* __restrict__ C;
  uint64_t ldc;
  __device__ inline void operator()(uint64_t m, int n, float v) const { C[(uint64_t)n * ldc + m] = v; }
};

template <class CF, class Epi>
__global__ __launch_bounds__(CF::NT, CF::OCC) void gemm_f32_k(const float* __restrict__ A, uint64_t M, int K, const float* __restrict__ B, int ldb, int N,
                                                               uint32_t nMB, uint32_t nNT, Epi epi) {
  constexpr int TM = CF::TM, TN = CF::TN, TK = CF::TK, NT = CF::NT, WMT = CF::WMT, WNT = CF::WNT;
  extern __shared__ f32x4 lds[];
  f32x4* As = lds;                          // [2][KQ][A_STRIDE]
  f32x4* Bs = lds + 2 * CF::A_STAGE;        // [2][KQ][B_STRIDE]
  // tile of this workgroup (see the header): xcd = id % 8 owns row blocks xcd, xcd + 8, ...; its N-tiles are consecutive slots
  const uint32_t wg = blockIdx.x, xcd = wg & 7u, slot = wg >> 3;
  const uint32_t mb = (slot / nNT) * 8u + xcd, nt = slot % nNT;
  if (mb >= nMB) return;
  const uint64_t m0 = (uint64_t)mb * TM;
  const int n0 = (int)nt * TN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave % CF::WAVES_M, wn = wave / CF::WAVES_M;

  // global -> register staging.  A: lanes along m (256 contiguous bytes per wave instruction); B: lanes along k (TK contiguous floats per
  // column); the B column offsets of a thread are kept for the whole K loop.
  uint32_t arow[CF::A_UNITS];
  int akq[CF::A_UNITS];
#pragma unroll
  for (int u = 0; u < CF::A_UNITS; ++u) {
    const int unit = tid + u * NT;
    akq[u] = unit / TM;
    const uint64_t m = m0 + (uint32_t)(unit % TM);
    arow[u] = (uint32_t)(m < M ? m : M - 1);  // clamped: rows past the end are computed on valid data, never stored
  }
  const int bk = tid % TK, bn0 = tid / TK;
  uint32_t bcol[CF::B_ELEMS];
#pragma unroll
  for (int u = 0; u < CF::B_ELEMS; ++u) bcol[u] = (uint32_t)min(n0 + bn0 + (NT / TK) * u, N - 1) * (uint32_t)ldb;
  float ra[4 * CF::A_UNITS], rb[CF::B_ELEMS];
  auto load_slab = [&](int k0) {
#pragma unroll
    for (int u = 0; u < CF::A_UNITS; ++u) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int k = min(k0 + 4 * akq[u] + t, K - 1);  // the k tail is zeroed on the B side, A stays finite data
        ra[4 * u + t] = A[(uint64_t)k * M + arow[u]];
      }
    }
    const uint32_t kl = (uint32_t)min(bk, K - 1 - k0);  // = bk except in a ragged last slab
    const float* bb = B + k0;
#pragma unroll
    for (int u = 0; u < CF::B_ELEMS; ++u) rb[u] = bb[bcol[u] + kl];
  };
  auto store_slab = [&](int stage, int k0) {
    f32x4* a = As + stage * CF::A_STAGE;
#pragma unroll
    for (int u = 0; u < CF::A_UNITS; ++u) {
      const int unit = tid + u * NT;
      f32x4 v = {ra[4 * u], ra[4 * u + 1], ra[4 * u + 2], ra[4 * u + 3]};
      a[akq[u] * CF::A_STRIDE + unit % TM] = v;
    }
    const float mask = k0 + bk < K ? 1.f : 0.f;  // applied here, behind the MFMA block: the loads stay in flight across it
    float* b = reinterpret_cast<float*>(Bs + stage * CF::B_STAGE);
#pragma unroll
    for (int u = 0; u < CF::B_ELEMS; ++u) b[((bk >> 2) * CF::B_STRIDE + bn0 + (NT / TK) * u) * 4 + (bk & 3)] = rb[u] * mask;
  };

  f32x16 acc[WNT][WMT];  // [j: n sub-tile][i: m sub-tile]
#pragma unroll
  for (int j = 0; j < WNT; ++j)
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  const int nslab = (K + TK - 1) / TK;
  load_slab(0);
  store_slab(0, 0);
  __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    const int cur = s & 1;
    if (s + 1 < nslab) load_slab((s + 1) * TK);
    const f32x4* a = As + cur * CF::A_STAGE + wm * (32 * WMT) + l31;
    const f32x4* b = Bs + cur * CF::B_STAGE + wn * (32 * WNT) + l31;
#ifdef ISLE_GEMM_PRIO
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int g = 0; g < TK / 8; ++g) {
      const int kq = 2 * g + h;
      f32x4 av[WMT], bv[WNT];
#pragma unroll
      for (int i = 0; i < WMT; ++i) av[i] = a[kq * CF::A_STRIDE + 32 * i];
#pragma unroll
      for (int j = 0; j < WNT; ++j) bv[j] = b[kq * CF::B_STRIDE + 32 * j];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < WNT; ++j)
#pragma unroll
          for (int i = 0; i < WMT; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[j][t], av[i][t], acc[j][i], 0, 0, 0);
    }
#ifdef ISLE_GEMM_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    if (s + 1 < nslab) store_slab(cur ^ 1, (s + 1) * TK);
    __syncthreads();
  }

  // lane owns row m = l31 of each 32 x 32 tile; register r holds column (r & 3) + 8 (r >> 2) + 4 h
  const bool full_n = n0 + TN <= N;  // workgroup-uniform: interior tiles store without per-element tests
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

template <class CF, class Epi>
inline hipError_t launch(hipStream_t stream, const float* A, uint64_t M, int K, const float* B, int ldb, int N, Epi epi) {
  const uint32_t nMB = (uint32_t)((M + CF::TM - 1) / CF::TM), nNT = (uint32_t)((N + CF::TN - 1) / CF::TN);
  const uint64_t slots = (uint64_t)((nMB + 7) / 8) * nNT;  // per XCD
  static bool lds_attr_set[64] = {};  // per device: the attribute belongs to the device's copy of the kernel
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64 || !lds_attr_set[dev]) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_k<CF, Epi>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CF::LDS_BYTES);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) lds_attr_set[dev] = true;
  }
  hipLaunchKernelGGL((gemm_f32_k<CF, Epi>), dim3((uint32_t)(slots * 8)), dim3(CF::NT), CF::LDS_BYTES, stream, A, M, K, B, ldb, N, nMB, nNT, epi);
  return hipGetLastError();
}

}  // namespace isle_gemm
