// Timing + correctness probe (not part of the product) for isle_amd/csrc/gemm_bf16x3.h beside gemm_f32.h at the shapes of the hot path, on RANDOM data
// (zero-filled operands let the chip hold a higher clock: MI355X_MICROARCH.md "DVFS give-back").
//   hipcc -O3 --offload-arch=gfx950 -o gemm3_probe gemm3_probe.hip     usage: gemm3_probe [big]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef WITH_ROCBLAS
#include <rocblas/rocblas.h>
#endif
#include "../../isle_amd/csrc/gemm_f32.h"
#include "../../isle_amd/csrc/gemm_bf16x3.h"

__global__ void fill_k(float* p, size_t n, uint64_t seed) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint64_t z = seed * 0x9E3779B97F4A7C15ull + (i + 1) * 0xD1342543DE82EF95ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  p[i] = (float)(int64_t)(z >> 40) * (1.0f / 8388608.0f) - 1.0f;
}


struct S { long M, N, K, ldb; const char* what; };

template <class CF>
static float run_cfg(hipStream_t st, const S& s, const float* A, const float* B, float* C, hipEvent_t e0, hipEvent_t e1) {
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0, st);
    hipError_t e = isle_gemm::launch<CF>(st, A, (uint64_t)s.M, (int)s.K, B, (int)s.ldb, (int)s.N, isle_gemm::StoreC{C, (uint64_t)s.M});
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    if (e != hipSuccess) {
      printf("launch failed: %s\n", hipGetErrorString(e));
      return -1.f;
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  return best;
}


template <class CF>
static float run_cfg3(hipStream_t st, const S& s, const float* A, const float* B, float* C, void* B3, hipEvent_t e0, hipEvent_t e1) {
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0, st);
    hipError_t e = isle_gemm3::launch<CF>(st, A, (uint64_t)s.M, (int)s.K, B, (int)s.ldb, (int)s.N, B3, isle_gemm3::StoreC{C, (uint64_t)s.M});
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    if (e != hipSuccess) {
      printf("launch failed: %s\n", hipGetErrorString(e));
      return -1.f;
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  return best;
}

template <class CF>
static float run_dma(hipStream_t st, const S& s, const float* A, const float* B, float* C, void* B3, void* A2, hipEvent_t e0, hipEvent_t e1, float* split_ms) {
  float best = 1e30f;
  hipEventRecord(e0, st);
  hipError_t e = isle_gemm3::split_a(st, A, (uint64_t)s.M, (int)s.K, A2, CF::TK);
  hipEventRecord(e1, st);
  hipEventSynchronize(e1);
  hipEventElapsedTime(split_ms, e0, e1);
  if (e != hipSuccess) {
    printf("split failed: %s\n", hipGetErrorString(e));
    return -1.f;
  }
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0, st);
    e = isle_gemm3::launch_dma<CF>(st, A2, (uint64_t)s.M, (int)s.K, B, (int)s.ldb, (int)s.N, B3, isle_gemm3::StoreC{C, (uint64_t)s.M});
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    if (e != hipSuccess) {
      printf("launch failed: %s\n", hipGetErrorString(e));
      return -1.f;
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  return best;
}

static double check(const S& s, const float* A, const float* B, const float* C) {  // fp64 on 64 random entries + the four corners
  std::vector<float> a(s.K), b(s.K);
  double worst = 0;
  for (int c = 0; c < 68; ++c) {
    long m = c < 64 ? (long)((uint64_t)(c * 2654435761u + 12345) % (uint64_t)s.M) : (c & 1 ? s.M - 1 : 0);
    long n = c < 64 ? (long)((uint64_t)(c * 40503u + 7) % (uint64_t)s.N) : (c & 2 ? s.N - 1 : 0);
    hipMemcpy2D(a.data(), 4, A + m, (size_t)s.M * 4, 4, s.K, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), B + (size_t)n * s.ldb, s.K * 4, hipMemcpyDeviceToHost);
    float got;
    hipMemcpy(&got, C + (size_t)n * s.M + m, 4, hipMemcpyDeviceToHost);
    double ref = 0, mag = 0;
    for (long k = 0; k < s.K; ++k) ref += (double)a[k] * b[k], mag += fabs((double)a[k] * b[k]);
    worst = fmax(worst, fabs(got - ref) / mag);
  }
  return worst;
}

int main(int argc, char** argv) {
  const bool big = argc > 1 && !strcmp(argv[1], "big");
  const bool zeros = argc > 1 && !strcmp(argv[1], "zeros");
  std::vector<S> shapes = {{100000, 1000, 2000, 2000, "rotation, C3"},        {1250000, 1000, 1000, 1000, "D x k x k, C3 shard"},
                           {100000, 1000, 1000, 1000, "lift, C3"},            {50000, 200, 400, 400, "rotation, C2"},
                           {1000000, 200, 200, 200, "D x k x k, C2"},         {100003, 999, 1997, 1999, "ragged everything"},
                           {4096, 4096, 4096, 4096, "4096^3"},                {1250000, 33, 1000, 1000, "k-means++ round, 33 seeds"}};
  if (big) shapes = {{10000000, 1000, 1000, 1000, "D x k x k, all of C3"}};
  hipStream_t st;
  hipStreamCreate(&st);
#ifdef WITH_ROCBLAS
  rocblas_handle h;
  rocblas_create_handle(&h);
  rocblas_set_stream(h, st);
  rocblas_set_atomics_mode(h, rocblas_atomics_not_allowed);
#endif
  for (auto& s : shapes) {
    float *A, *B, *C;
    if (hipMalloc(&A, (size_t)s.M * s.K * 4) != hipSuccess || hipMalloc(&B, (size_t)s.ldb * s.N * 4) != hipSuccess ||
        hipMalloc(&C, (size_t)s.M * s.N * 4) != hipSuccess) {
      printf("%s: allocation failed\n", s.what);
      return 1;
    }
    if (zeros) {
      hipMemsetAsync(A, 0, (size_t)s.M * s.K * 4, st);
      hipMemsetAsync(B, 0, (size_t)s.ldb * s.N * 4, st);
    } else {
      fill_k<<<(unsigned)(((size_t)s.M * s.K + 255) / 256), 256, 0, st>>>(A, (size_t)s.M * s.K, 1);
      fill_k<<<(unsigned)(((size_t)s.ldb * s.N + 255) / 256), 256, 0, st>>>(B, (size_t)s.ldb * s.N, 2);
    }
    void* B3;
    hipMalloc(&B3, (size_t)3 * (4 * ((s.K + 31) / 32)) * ((s.N + 255) / 256 * 256) * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("%-26s M=%ld N=%ld K=%ld ldb=%ld%s\n", s.what, s.M, s.N, s.K, s.ldb, zeros ? "  (ZERO operands)" : "");
#define RUN(NAME, ...)                                                                                                     \
  {                                                                                                                        \
    hipMemsetAsync(C, 0xff, (size_t)s.M * s.N * 4, st);                                                                     \
    const float ms = run_cfg<isle_gemm::Cfg<__VA_ARGS__>>(st, s, A, B, C, e0, e1);                                         \
    const double w = zeros ? 0.0 : check(s, A, B, C);                                                                       \
    printf("    %-34s %8.3f ms %6.1f TFLOP/s  err %.1e%s\n", NAME, ms, 2.0 * s.M * s.N * s.K / ms / 1e9, w, w < 1e-6 ? "" : "  WRONG"); \
    fflush(stdout);                                                                                                        \
  }
    RUN("f32 MFMA 256x256x16 1024thr", 2, 2, 4, 4, 16, 4)
#define RUN3(NAME, ...)                                                                                                    \
  {                                                                                                                        \
    hipMemsetAsync(C, 0xff, (size_t)s.M * s.N * 4, st);                                                                     \
    const float ms = run_cfg3<isle_gemm3::Cfg<__VA_ARGS__>>(st, s, A, B, C, B3, e0, e1);                                   \
    const double w = zeros ? 0.0 : check(s, A, B, C);                                                                       \
    printf("    %-34s %8.3f ms %6.1f TFLOP/s  err %.1e%s\n", NAME, ms, 2.0 * s.M * s.N * s.K / ms / 1e9, w, w < 2e-5 ? "" : "  WRONG"); \
    fflush(stdout);                                                                                                        \
  }
    RUN3("bf16 x 3 256x256 1024thr occ4", 2, 2, 4, 4, 4)
    RUN3("bf16 x 2 (3 products) 256x256", 2, 2, 4, 4, 4, 16, 2)
#define RUND(NAME, NS, TKK)                                                                                                    \
  {                                                                                                                        \
    void* A2 = nullptr;                                                                                                    \
    if (hipMalloc(&A2, isle_gemm3::a2_units((uint64_t)s.M, (int)s.K, TKK) * 16) == hipSuccess) {                               \
      hipMemsetAsync(C, 0xff, (size_t)s.M * s.N * 4, st);                                                                   \
      float sp = 0.f;                                                                                                      \
      const float ms = run_dma<isle_gemm3::CfgDma<NS, TKK>>(st, s, A, B, C, B3, A2, e0, e1, &sp);                                \
      const double w = zeros ? 0.0 : check(s, A, B, C);                                                                     \
      printf("    %-34s %8.3f ms %6.1f TFLOP/s  err %.1e%s   (split of A: %.3f ms)\n", NAME, ms, 2.0 * s.M * s.N * s.K / ms / 1e9, w, w < 2e-5 ? "" : "  WRONG", sp); \
      fflush(stdout);                                                                                                      \
      hipFree(A2);                                                                                                         \
    }                                                                                                                      \
  }
    RUND("bf16 x 2 LDS-DMA ring of 4", 4, 16)
    RUND("bf16 x 2 LDS-DMA ring of 2", 2, 16)
    RUND("bf16 x 2 LDS-DMA ring of 2, TK 32", 2, 32)
    RUN3("bf16 x 2 (3 products) 256x128 occ2", 2, 2, 4, 2, 2, 16, 2)
    RUN3("bf16 x 2 (3 products) 256x256 TK32", 2, 2, 4, 4, 4, 32, 2)
    RUN3("bf16 x 2 (3 products) w128x64 TK32", 4, 2, 2, 4, 2, 32, 2)
    RUN3("bf16 x 2 (3 products) w128x64", 4, 2, 2, 4, 2, 16, 2)
    RUN3("bf16 x 3 256x128 512thr occ2", 2, 2, 4, 2, 2)
    RUN3("bf16 x 3 256x128 TK32 512thr occ2", 2, 2, 4, 2, 2, 32)
    RUN3("bf16 x 3 256x128 TK32 1024thr", 2, 1, 4, 4, 4, 32)
    RUN3("bf16 x 3 128x256 512thr occ2", 2, 2, 2, 4, 2)
    RUN3("bf16 x 3 256x256 w128x64 512thr", 4, 2, 2, 4, 2)
#ifdef WITH_ROCBLAS
    const float one = 1.f, zero = 0.f;
    float bl = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0, st);
      rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_none, (int)s.M, (int)s.N, (int)s.K, &one, A, (int)s.M, B, (int)s.ldb, &zero, C, (int)s.M);
      hipEventRecord(e1, st);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < bl) bl = ms;
    }
    printf("    %-34s %8.3f ms %6.1f TFLOP/s\n", "rocBLAS sgemm, same data", bl, 2.0 * s.M * s.N * s.K / bl / 1e9);
#endif
    fflush(stdout);
    hipFree(A);
    hipFree(B);
    hipFree(C);
    hipFree(B3);
  }
  return 0;
}
