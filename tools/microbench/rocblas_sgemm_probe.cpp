// Timing probe (not part of the product): rocBLAS sgemm at the shapes of gemm_nn_k (C = A B, all column-major, no transposes):
// Ritz rotation, lift, first assignment of Lloyd on B through the projection.  hipcc -O2 rocblas_sgemm_probe.cpp -lrocblas
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <vector>
int main() {
  rocblas_handle h;
  rocblas_create_handle(&h);
  struct S { long M, N, K; const char* what; } shapes[] = {{100000, 1000, 2010, "rotation, C3"}, {1250000, 1000, 1000, "first assignment, C3 shard"},
                                                            {50000, 200, 410, "rotation, C2"}, {1000000, 200, 200, "first assignment, C2"},
                                                            {100000, 1000, 1000, "lift, C3"}, {1250000, 33, 1000, "k-means++ round, 33 seeds, C3 shard"}, {1250000, 16, 1000, "k-means++ round, 16 seeds"}, {1000000, 15, 200, "k-means++ round, 15 seeds, C2"}};
  for (auto& s : shapes) {
    float *A, *B, *C;
    hipMalloc(&A, s.M * s.K * 4); hipMalloc(&B, s.K * s.N * 4); hipMalloc(&C, s.M * s.N * 4);
    hipMemset(A, 0, s.M * s.K * 4); hipMemset(B, 0, s.K * s.N * 4);
    const float one = 1.f, zero = 0.f;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      rocblas_status st = rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_none, (int)s.M, (int)s.N, (int)s.K, &one, A, (int)s.M, B, (int)s.K, &zero, C, (int)s.M);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("%-28s M=%ld N=%ld K=%ld: %.3f ms  %.1f TFLOP/s (status %d)\n", s.what, s.M, s.N, s.K, ms, 2.0 * s.M * s.N * s.K / ms / 1e9, (int)st);
    }
    hipFree(A); hipFree(B); hipFree(C);
  }
  return 0;
}
