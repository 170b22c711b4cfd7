// Timing probe (not part of the product): rocSOLVER's symmetric eigensolvers at the sizes of the Ritz problem (n = 410 at C2, 2010 at a
// C3 shard), fp64, vectors wanted — against evd_tridiag.hip (persistent tridiagonalisation + bisection + twisted factorisation + back-
// transformation).  hipcc -O2 rocsolver_syevd_probe.cpp -lrocsolver -lrocblas
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>
#include <cstdio>
#include <vector>
int main() {
  rocblas_handle h;
  rocblas_create_handle(&h);
  for (int n : {410, 2010}) {
    std::vector<double> A((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) {  // banded symmetric test matrix with a decaying diagonal
      A[(size_t)i * n + i] = 1000.0 / (1.0 + i);
      for (int j = i + 1; j < n && j <= i + 10; ++j) A[(size_t)i * n + j] = A[(size_t)j * n + i] = 0.3 / (1.0 + (j - i));
    }
    double *dA, *dD, *dE;
    rocblas_int* info;
    hipMalloc(&dA, sizeof(double) * n * n); hipMalloc(&dD, sizeof(double) * n); hipMalloc(&dE, sizeof(double) * n); hipMalloc(&info, sizeof(rocblas_int));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int algo = 0; algo < 2; ++algo)
      for (int rep = 0; rep < 3; ++rep) {
        hipMemcpy(dA, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
        hipEventRecord(e0);
        rocblas_status st = algo == 0 ? rocsolver_dsyevd(h, rocblas_evect_original, rocblas_fill_upper, n, dA, n, dD, dE, info)
                                      : rocsolver_dsyev(h, rocblas_evect_original, rocblas_fill_upper, n, dA, n, dD, dE, info);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) printf("n = %4d  %s: %.3f ms (status %d)\n", n, algo == 0 ? "dsyevd (divide and conquer)" : "dsyev  (QR iteration)", ms, (int)st);
      }
    hipFree(dA); hipFree(dD); hipFree(dE); hipFree(info);
  }
  return 0;
}
