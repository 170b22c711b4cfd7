// Timing probe (not part of the product): rocSPARSE's SpMM at the shape of the Gram apply of BASELINE config 2 — B^T as a CSR matrix of
// 1M documents x 50k words with ~99 Zipf-distributed words per document, a 10-column panel (row-major) — Y = B^T X and Z = B Y (the same
// CSR matrix transposed), against the LDS-banded kernels (0.20 + 0.28 ms).  hipcc -O2 rocsparse_spmm_probe.cpp -lrocsparse
#include <hip/hip_runtime.h>
#include <rocsparse/rocsparse.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#define CHK(x) do { auto s_ = (x); if (s_ != 0) { printf("%s -> %d\n", #x, (int)s_); return 1; } } while (0)
int main() {
  const int64_t D = 1000000, V = 50000, b = 10;
  std::mt19937_64 rng(1);
  std::vector<double> cdf(V);
  double acc = 0;
  for (int w = 0; w < V; ++w) { acc += 1.0 / std::pow(w + 10.0, 0.9); cdf[w] = acc; }
  std::vector<int64_t> ptr(D + 1, 0);
  std::vector<int32_t> col;
  col.reserve(D * 100);
  std::uniform_real_distribution<double> U(0.0, acc);
  std::vector<int32_t> row;
  for (int64_t d = 0; d < D; ++d) {
    row.clear();
    for (int t = 0; t < 110; ++t) row.push_back((int32_t)(std::lower_bound(cdf.begin(), cdf.end(), U(rng)) - cdf.begin()));
    std::sort(row.begin(), row.end());
    row.erase(std::unique(row.begin(), row.end()), row.end());
    col.insert(col.end(), row.begin(), row.end());
    ptr[d + 1] = (int64_t)col.size();
  }
  const int64_t nnz = (int64_t)col.size();
  std::vector<float> val(nnz, 1.0f);
  printf("B^T: %lld x %lld, %lld nonzeros (%.1f per document)\n", (long long)D, (long long)V, (long long)nnz, (double)nnz / D);
  int64_t* dptr; int32_t* dcol; float *dval, *X, *Y, *Z;
  hipMalloc(&dptr, (D + 1) * 8); hipMalloc(&dcol, nnz * 4); hipMalloc(&dval, nnz * 4);
  hipMalloc(&X, V * b * 4); hipMalloc(&Y, D * b * 4); hipMalloc(&Z, V * b * 4);
  hipMemcpy(dptr, ptr.data(), (D + 1) * 8, hipMemcpyHostToDevice); hipMemcpy(dcol, col.data(), nnz * 4, hipMemcpyHostToDevice);
  hipMemcpy(dval, val.data(), nnz * 4, hipMemcpyHostToDevice); hipMemset(X, 0, V * b * 4); hipMemset(Y, 0, D * b * 4);
  rocsparse_handle h; CHK(rocsparse_create_handle(&h));
  rocsparse_spmat_descr A; rocsparse_dnmat_descr dX, dY, dZ;
  CHK(rocsparse_create_csr_descr(&A, D, V, nnz, dptr, dcol, dval, rocsparse_indextype_i64, rocsparse_indextype_i32, rocsparse_index_base_zero, rocsparse_datatype_f32_r));
  CHK(rocsparse_create_dnmat_descr(&dX, V, b, b, X, rocsparse_datatype_f32_r, rocsparse_order_row));
  CHK(rocsparse_create_dnmat_descr(&dY, D, b, b, Y, rocsparse_datatype_f32_r, rocsparse_order_row));
  CHK(rocsparse_create_dnmat_descr(&dZ, V, b, b, Z, rocsparse_datatype_f32_r, rocsparse_order_row));
  const float one = 1.f, zero = 0.f;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct { rocsparse_spmm_alg a; const char* n; } algs[] = {{rocsparse_spmm_alg_default, "default"}, {rocsparse_spmm_alg_csr_row_split, "csr_row_split"},
                                                             {rocsparse_spmm_alg_csr_merge_path, "csr_merge_path"}, {rocsparse_spmm_alg_csr_nnz_split, "csr_nnz_split"}};
  for (int pass = 0; pass < 2; ++pass)
    for (auto& al : algs) {
      const rocsparse_operation tA = pass == 0 ? rocsparse_operation_none : rocsparse_operation_transpose;
      rocsparse_dnmat_descr in = pass == 0 ? dX : dY, out = pass == 0 ? dY : dZ;
      size_t bs = 0;
      if (rocsparse_spmm(h, tA, rocsparse_operation_none, &one, A, in, &zero, out, rocsparse_datatype_f32_r, al.a, rocsparse_spmm_stage_buffer_size, &bs, nullptr) != 0) {
        printf("pass %d %-16s: not supported\n", pass + 1, al.n);
        continue;
      }
      void* buf = nullptr;
      hipMalloc(&buf, bs ? bs : 4);
      if (rocsparse_spmm(h, tA, rocsparse_operation_none, &one, A, in, &zero, out, rocsparse_datatype_f32_r, al.a, rocsparse_spmm_stage_preprocess, &bs, buf) != 0) {
        printf("pass %d %-16s: preprocess failed\n", pass + 1, al.n);
        hipFree(buf);
        continue;
      }
      float best = 1e30f;
      int st = 0;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        st = (int)rocsparse_spmm(h, tA, rocsparse_operation_none, &one, A, in, &zero, out, rocsparse_datatype_f32_r, al.a, rocsparse_spmm_stage_compute, &bs, buf);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) best = std::min(best, ms);
      }
      printf("pass %d (%s) %-16s: %.3f ms (status %d, buffer %zu B)\n", pass + 1, pass == 0 ? "Y = B^T X" : "Z = B Y  ", al.n, best, st, bs);
      hipFree(buf);
    }
  return 0;
}
