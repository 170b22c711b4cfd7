// isle_amd/csrc/api_internal.h — what the translation units behind the C ABI share (api.cpp: context, transport, upload, measurement;
// api_ks.cpp: the block Krylov-Schur solver; api_kmeans.cpp: k-means++ and the two Lloyd loops; api_stages.cpp: the stages either side of the
// path).  Nothing here is part of the boundary: include/isle_hip.h is.
#pragma once
#include <cstdint>
#include <vector>

#include "common.h"

template <class T>
struct DtOf;
template <>
struct DtOf<float> { static constexpr int v = ISLE_DT_F32; };
template <>
struct DtOf<double> { static constexpr int v = ISLE_DT_F64; };
template <>
struct DtOf<int> { static constexpr int v = ISLE_DT_I32; };
template <>
struct DtOf<uint32_t> { static constexpr int v = ISLE_DT_U32; };
template <>
struct DtOf<uint64_t> { static constexpr int v = ISLE_DT_U64; };

template <class T>
inline int allreduce_sum(isle_ctx* c, T* buf, size_t count) {
  if (!c->multi()) return 0;
  TimeScope ts(c, ISLE_T_COMM);
  return isle_allreduce(c, buf, count, DtOf<T>::v);
}

int agree_i32(isle_ctx* c, int v, const char* what);  // all ranks hold the same control value, or all return ISLE_E_COMM (api.cpp)
int drain_events(isle_ctx* c);
inline int round4(int k) { return (k + 3) & ~3; }
inline int panel_width(int b) { return 4 * ((b + 3) / 4); }  // BP in {4, 8, ..., 32}: one float4 lane per 4 columns
// rows [r0, r0 + nl) of a column-major n x w matrix <-> a packed nloc x w block (api.cpp)
int k_slice_rows(isle_ctx* c, float* M, uint64_t n, int w, uint64_t r0, uint64_t nl, uint64_t nloc, float* blk, bool pack);
// Zcm (V x b col-major, device) = B (B^T Xcm) with the all-reduce over the shards (api.cpp)
int gram_apply_dev(isle_ctx* c, const float* Xcm, int b, float* Zcm);
// U (V x k col-major on the device) becomes the context's basis: row-major copy, validity flags (api_ks.cpp)
int install_U(isle_ctx* c, const float* Ucm_dev, int k);

// ------------------------------------------------------------------------------------------
// host RNG (rand() stand-in)
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// host RNG (rand() stand-in)
// ------------------------------------------------------------------------------------------
// glibc's rand() (the TYPE_3 additive feedback generator of random_r.c: r[i] = r[i - 31] + r[i - 3], 31 state words seeded by the
// Lehmer generator 16807 mod 2^31 - 1, the first 310 outputs discarded, results shifted right by one).  The reference calls rand()
// without ever calling srand(), i.e. with seed 1: rng_seed = 1 therefore draws the numbers a reference binary linked against glibc
// draws (tests/test_abi_cpu.py checks the sequence against this machine's libc).  What still separates un-injected seeds from a
// reference run is the prefix sum they index (fp32 and sequential there, :2170-2172; fp64 and parallel here).
struct HostRng {
  uint32_t r[34];
  int k = 0;  // next output is the k-th
  explicit HostRng(uint64_t seed64) {
    uint32_t seed = (uint32_t)seed64;
    if (seed == 0) seed = 1;
    int32_t w[34];
    w[0] = (int32_t)seed;
    for (int i = 1; i < 31; ++i) {
      int64_t v = (16807LL * w[i - 1]) % 2147483647LL;
      if (v < 0) v += 2147483647LL;
      w[i] = (int32_t)v;
    }
    for (int i = 31; i < 34; ++i) w[i] = w[i - 31];
    for (int i = 0; i < 34; ++i) r[i] = (uint32_t)w[i];
    for (int i = 34; i < 344; ++i) step();  // discarded
  }
  uint32_t step() {  // the ring holds the last 34 words; word i lives at i % 34
    const uint32_t v = r[(k + 34 - 31) % 34] + r[(k + 34 - 3) % 34];
    r[k % 34] = v;
    k = (k + 1) % 34;
    return v;
  }
  uint32_t next31() { return step() >> 1; }  // rand(): 0 .. RAND_MAX = 2^31 - 1
  // include/matUtils.h:473-477: (double)rand() + (double)rand() * (RAND_MAX + 1), over (RAND_MAX + 1)^2.  The two rand() calls of that
  // expression are UNSEQUENCED in C++: which of them supplies the low word is the reference compiler's choice.  Assumed here: the left
  // operand is evaluated first (what g++ does for this expression at -O3 — the only arrangement under which rng_seed = 1 reproduces an
  // unseeded reference binary's dice); with the other order the low and high words swap.  Un-injected seeds are not promised equal to
  // a reference run's in any case, which is why the parity tests inject them.
  double fraction() {
    const double R1 = 2147483648.0;
    const double lo = (double)next31();
    const double hi = (double)next31();
    return (lo + hi * R1) / (R1 * R1);
  }
};
