// oracle/ref_spectra_driver.cpp — TEST INFRASTRUCTURE (checker), not product.
//
// Runs the reference's OWN vendored eigensolver on a matrix B: Spectra::SymEigsSolver<float, LARGEST_ALGE, Op>(&op, k, 2k + 1),
// init(), compute() with the default arguments — exactly the calls of FPSparseMatrix::compute_Spectra
// (/root/reference/src/sparseMatrix.cpp:1161-1190; selected by EIGENSOLVER == SPECTRA, src/trainer.cpp:494-495,
// include/hyperparams.h:24-31).  Spectra and Eigen are header-only and vendored in the reference tree
// (/root/reference/spectra-master/include, /root/reference/Eigen): they are compiled where they lie, nothing is copied.
//
// What this is NOT: the reference's operator (MKL_SpSpTrProd, include/matUtils.h:52-365, needs <mkl.h>, absent from this
// image) — the operator below is this file's own plain loop  y = B (B^T x)  in double accumulation, plugged into
// Spectra's documented operator interface (rows(), perform_op(x, y); spectra-master/include/SymEigsSolver.h:123-131).
// Nor is it the default solver (BlockKs includes <mkl.h> through block-ks/ks_types.h:7).  It pins the spectrum and the
// invariant subspace that any of the reference's eigensolvers must deliver for B B^T.
//
// usage: spectra_eigs <B.bin> <k> <out.bin>
//   B.bin  : u64 V, u64 D, u64 nnz, f32 vals[nnz], u32 rows[nnz], i64 offs[D+1]
//   out.bin: i32 nconv, i32 info, f32 evalues[k], f32 U[V*k] column-major
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <Eigen/Core>
#include <SymEigsSolver.h>

struct GramOp {
  uint64_t V, D, nnz;
  std::vector<float> vals;
  std::vector<uint32_t> rowsv;
  std::vector<int64_t> offs;
  int rows() const { return (int)V; }
  int cols() const { return (int)V; }
  void perform_op(const float* x, float* y) const {
    std::vector<double> acc(V, 0.0);
    for (uint64_t d = 0; d < D; ++d) {
      double t = 0.0;
      for (int64_t e = offs[d]; e < offs[d + 1]; ++e) t += (double)vals[e] * x[rowsv[e]];
      for (int64_t e = offs[d]; e < offs[d + 1]; ++e) acc[rowsv[e]] += (double)vals[e] * t;
    }
    for (uint64_t w = 0; w < V; ++w) y[w] = (float)acc[w];
  }
};

static bool rd(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n; }

int main(int argc, char** argv) {
  if (argc != 4) {
    fprintf(stderr, "usage: %s <B.bin> <k> <out.bin>\n", argv[0]);
    return 2;
  }
  GramOp op;
  FILE* f = fopen(argv[1], "rb");
  if (!f || !rd(f, &op.V, 8) || !rd(f, &op.D, 8) || !rd(f, &op.nnz, 8)) return 3;
  op.vals.resize(op.nnz);
  op.rowsv.resize(op.nnz);
  op.offs.resize(op.D + 1);
  if (!rd(f, op.vals.data(), 4 * op.nnz) || !rd(f, op.rowsv.data(), 4 * op.nnz) || !rd(f, op.offs.data(), 8 * (op.D + 1))) return 3;
  fclose(f);
  const int k = atoi(argv[2]);
  // the calls of compute_Spectra (src/sparseMatrix.cpp:1168-1175)
  Spectra::SymEigsSolver<float, Spectra::LARGEST_ALGE, GramOp> eigs(&op, k, 2 * k + 1);
  eigs.init();
  const int nconv = eigs.compute();
  const int info = eigs.info();
  Eigen::Matrix<float, Eigen::Dynamic, 1> ev = eigs.eigenvalues();
  Eigen::Matrix<float, Eigen::Dynamic, Eigen::Dynamic> U = eigs.eigenvectors(k);
  fprintf(stderr, "spectra_eigs: V=%llu D=%llu nnz=%llu k=%d nconv=%d info=%d ev0=%.9g evk=%.9g\n", (unsigned long long)op.V,
          (unsigned long long)op.D, (unsigned long long)op.nnz, k, nconv, info, nconv ? ev(0) : 0.f, nconv ? ev(nconv - 1) : 0.f);
  FILE* o = fopen(argv[3], "wb");
  if (!o) return 4;
  fwrite(&nconv, 4, 1, o);
  fwrite(&info, 4, 1, o);
  std::vector<float> evk(k, 0.f);
  for (int i = 0; i < k && i < ev.size(); ++i) evk[i] = ev(i);
  fwrite(evk.data(), 4, k, o);
  if (U.cols() == k) fwrite(U.data(), 4, (size_t)op.V * k, o);
  fclose(o);
  return 0;
}
