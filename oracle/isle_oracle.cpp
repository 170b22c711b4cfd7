// oracle/isle_oracle.cpp
//
// TEST INFRASTRUCTURE ONLY — NOT PART OF THE PRODUCT.
// CPU restatement (fp32 arithmetic, OpenMP) of the ISLE training hot path, used
// (1) as the parity checker for the HIP path in tests/, __graft_entry__.smoke()
// and (2) as the "port" CPU baseline timed by bench.py's cpu_baseline leg.
// Nothing under isle_amd/ may include, link, load or call this file.
//
// PINNING.  Eigensolver results (block_ks: sigma, U): pinned by a run of reference code — oracle/_ref/spectra_eigs is the
// reference's vendored Spectra::SymEigsSolver + Eigen compiled where they lie and called as compute_Spectra does
// (src/sparseMatrix.cpp:1161-1190); its outputs are tests/golden/ref_spectra.npz (tests/test_reference_golden_cpu.py).
// Everything else — the MKL operator, BlockKs, k-means — is PARITY UNPINNED: the reference ships no golden vectors or
// known-answer tests for this path, and those sources cannot be compiled in this image (they include Intel MKL's
// <mkl.h>, which is absent: include/types.h:9, block-ks/ks_types.h:7).  Those parts are pinned only by (a) fp64 NumPy
// ground truth (dense eigvalsh / brute-force k-means steps) and (b) the reference's own known-spectrum recipe
// utils::get_seed_eigs (block-ks/ks_utils.h:136-165) — see tests/test_oracle_*.py.
//
// Every function cites the reference file:line it restates (paths relative to the
// reference root).  The restatement follows the reference's ALGORITHM and operation
// order; it shares no code with it (MKL/Armadillo calls are replaced by plain loops).
//
// Build: see oracle/Makefile  (g++ -O3 -fopenmp -shared -fPIC).

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

typedef std::vector<float> fvec;
typedef std::vector<double> dvec;

// ---------------------------------------------------------------------------------
// Deterministic host RNG standing in for glibc rand() (the reference never seeds it:
// SURVEY App. C #11).  Same interface shape: next31() in [0, 2^31-1].
// ---------------------------------------------------------------------------------
struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) {}
  uint64_t next64() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  uint32_t next31() { return (uint32_t)(next64() >> 33); }  // like rand(): [0, RAND_MAX]
  // include/matUtils.h:473-477  rand_fraction(): (rand() + rand()*(RAND_MAX+1)) / (RAND_MAX+1)^2
  double fraction() {
    const double R1 = 2147483648.0;  // RAND_MAX + 1
    double lo = (double)next31();
    double hi = (double)next31();
    return (lo + hi * R1) / (R1 * R1);
  }
  // arma::randu element: uniform [0,1)  (armadillo_bits/arma_rng_cxx98.hpp:60-69 uses rand()/RAND_MAX)
  float randu() { return (float)((double)next31() / 2147483648.0); }
};

// ---------------------------------------------------------------------------------
// CSC container + CSR copy.  include/sparseMatrix.h:23-56 (vals_CSC, rows_CSC,
// offsets_CSC); CSR copy = include/matUtils.h:52-136 (mkl_scsrcsc in the operator ctor).
// ---------------------------------------------------------------------------------
struct Csc {
  uint64_t V = 0, D = 0, nnz = 0;
  fvec vals;
  std::vector<uint32_t> rows;
  std::vector<int64_t> offs;
  // CSR copy (row-major view of B), columns ascending within a row
  fvec rvals;
  std::vector<uint32_t> rcols;
  std::vector<int64_t> roffs;

  void build_csr() {
    roffs.assign(V + 1, 0);
    for (uint64_t i = 0; i < nnz; ++i) roffs[rows[i] + 1]++;
    for (uint64_t r = 0; r < V; ++r) roffs[r + 1] += roffs[r];
    rvals.resize(nnz);
    rcols.resize(nnz);
    std::vector<int64_t> cur(roffs.begin(), roffs.end() - 1);
    for (uint64_t d = 0; d < D; ++d)
      for (int64_t i = offs[d]; i < offs[d + 1]; ++i) {
        int64_t p = cur[rows[i]]++;
        rvals[p] = vals[i];
        rcols[p] = (uint32_t)d;
      }
  }
};

// Abstract symmetric operator: Z (n x b, col-major) = A * X (n x b, col-major).
struct Op {
  virtual ~Op() {}
  virtual uint64_t rows() const = 0;
  virtual void multiply(const float* X, int b, float* Z) = 0;
  long napplies = 0;
};

// include/matUtils.h:336-365  MKL_SpSpTrProd::multiply:  Z = B * (B^T * X)
//   first csrmm: rows of B^T (= CSC columns) against X (row-major there; col-major here)
//   second csrmm: CSR rows of B against the intermediate.
struct GramOp : Op {
  const Csc* m;
  fvec Y;  // D x b, row-major
  explicit GramOp(const Csc* m_) : m(m_) {}
  uint64_t rows() const override { return m->V; }
  void multiply(const float* X, int b, float* Z) override {
    napplies++;
    const uint64_t V = m->V, D = m->D;
    Y.resize((size_t)D * b);
    // row-major copy of X (include/matUtils.h:338  rm_in = trans(m_in))
    fvec Xr((size_t)V * b);
    for (int j = 0; j < b; ++j)
      for (uint64_t r = 0; r < V; ++r) Xr[r * b + j] = X[(size_t)j * V + r];
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t d = 0; d < (int64_t)D; ++d) {
      float acc[64];
      for (int j = 0; j < b; ++j) acc[j] = 0.f;
      for (int64_t i = m->offs[d]; i < m->offs[d + 1]; ++i) {
        const float v = m->vals[i];
        const float* xr = &Xr[(size_t)m->rows[i] * b];
        for (int j = 0; j < b; ++j) acc[j] += v * xr[j];
      }
      for (int j = 0; j < b; ++j) Y[(size_t)d * b + j] = acc[j];
    }
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t r = 0; r < (int64_t)V; ++r) {
      float acc[64];
      for (int j = 0; j < b; ++j) acc[j] = 0.f;
      for (int64_t i = m->roffs[r]; i < m->roffs[r + 1]; ++i) {
        const float v = m->rvals[i];
        const float* yr = &Y[(size_t)m->rcols[i] * b];
        for (int j = 0; j < b; ++j) acc[j] += v * yr[j];
      }
      for (int j = 0; j < b; ++j) Z[(size_t)j * V + r] = acc[j];
    }
  }
};

// block-ks/ks_utils.h:167-182  utils::ArmaMatProdOp — dense symmetric test operator.
struct DenseOp : Op {
  const float* A;
  uint64_t n;
  DenseOp(const float* A_, uint64_t n_) : A(A_), n(n_) {}
  uint64_t rows() const override { return n; }
  void multiply(const float* X, int b, float* Z) override {
    napplies++;
    for (int j = 0; j < b; ++j) {
#pragma omp parallel for
      for (int64_t r = 0; r < (int64_t)n; ++r) Z[(size_t)j * n + r] = 0.f;
      for (uint64_t c = 0; c < n; ++c) {
        const float x = X[(size_t)j * n + c];
        const float* a = A + c * n;
        float* z = Z + (size_t)j * n;
        for (uint64_t r = 0; r < n; ++r) z[r] += a[r] * x;
      }
    }
  }
};

// ---------------------------------------------------------------------------------
// Small dense helpers on col-major float matrices (what Armadillo's fmat ops lower to:
// sgemm).  C(m x n) = A^T(k x m)^T ... explicit loops, fp32 accumulate like sgemm.
// ---------------------------------------------------------------------------------
// H (m x b) = Vb(:, 0:m)^T * F (n x b)           block-ks/restarted_block_ks.h:83-84
static void gemm_tn(const float* Vb, size_t n, size_t m, const float* F, size_t b, float* H) {
#pragma omp parallel for schedule(static) collapse(2)
  for (int64_t j = 0; j < (int64_t)b; ++j)
    for (int64_t i = 0; i < (int64_t)m; ++i) {
      const float* v = Vb + (size_t)i * n;
      const float* f = F + (size_t)j * n;
      float s = 0.f;
      for (size_t r = 0; r < n; ++r) s += v[r] * f[r];
      H[(size_t)j * m + i] = s;
    }
}
// F (n x b) -= Vb(:, 0:m) * H (m x b)             block-ks/restarted_block_ks.h:85
static void gemm_sub(float* F, size_t n, size_t b, const float* Vb, size_t m, const float* H) {
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < (int64_t)b; ++j) {
    float* f = F + (size_t)j * n;
    for (size_t i = 0; i < m; ++i) {
      const float h = H[(size_t)j * m + i];
      const float* v = Vb + i * n;
      for (size_t r = 0; r < n; ++r) f[r] -= v[r] * h;
    }
  }
}
// C (n x q) = A (n x p) * W (p x q), all col-major  (Ritz rotation, lift)
static void gemm_nn(const float* A, size_t n, size_t p, const float* W, size_t ldw, size_t q, float* C) {
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < (int64_t)q; ++j) {
    float* c = C + (size_t)j * n;
    for (size_t r = 0; r < n; ++r) c[r] = 0.f;
    for (size_t i = 0; i < p; ++i) {
      const float w = W[(size_t)j * ldw + i];
      const float* a = A + i * n;
      for (size_t r = 0; r < n; ++r) c[r] += a[r] * w;
    }
  }
}

// ---------------------------------------------------------------------------------
// block-ks/ks_utils.h:43-127  utils::compute_qr — MGS with one DGKS correction per
// pivot, IN DOUBLE, dropping columns whose residual norm < 1e-6.
// A: n x c (col-major float).  Out: Q (n x rank, float), R (rank x c, col-major float).
// ---------------------------------------------------------------------------------
static int compute_qr(const float* A, size_t n, size_t c, fvec& Q, fvec& R) {
  dvec a((size_t)n * c);
  for (size_t i = 0; i < n * c; ++i) a[i] = (double)A[i];
  dvec Qd((size_t)n * c, 0.0), Rd((size_t)c * c, 0.0);  // Rd is (row, col) -> Rd[col*c + row]
  size_t rank = 0;
  dvec bb(c), cc(c);
  for (size_t i = 0; i < c; ++i) {
    double* v = &a[i * n];
    double dot = 0.0;
#pragma omp parallel for reduction(+ : dot)
    for (int64_t r = 0; r < (int64_t)n; ++r) dot += v[r] * v[r];
    const float v_norm = (float)std::sqrt(dot);  // ks_utils.h:66 (ARMA_FPTYPE)
    if (v_norm < 1e-6) continue;                 // ks_utils.h:69
    double* q = &Qd[rank * n];
    for (size_t r = 0; r < n; ++r) q[r] = v[r] / (double)v_norm;
    // b = q^T a(:, i:)   ;  a(:, i:) -= q b   ;  c = q^T a(:, i:)  ;  a(:, i:) -= q c
    for (int pass = 0; pass < 2; ++pass) {
      dvec& co = pass == 0 ? bb : cc;
#pragma omp parallel for schedule(static)
      for (int64_t j = (int64_t)i; j < (int64_t)c; ++j) {
        double* aj = &a[(size_t)j * n];
        double s = 0.0;
        for (size_t r = 0; r < n; ++r) s += q[r] * aj[r];
        co[j] = s;
      }
      // NB column i itself is updated too (a.tail_cols includes column i); q is a copy.
#pragma omp parallel for schedule(static)
      for (int64_t j = (int64_t)i; j < (int64_t)c; ++j) {
        double* aj = &a[(size_t)j * n];
        const double s = co[j];
        for (size_t r = 0; r < n; ++r) aj[r] -= q[r] * s;
      }
    }
    for (size_t j = i; j < c; ++j) Rd[j * c + rank] = bb[j] + cc[j];  // ks_utils.h:79
    rank++;
  }
  Q.resize((size_t)n * rank);
  for (size_t i = 0; i < n * rank; ++i) Q[i] = (float)Qd[i];
  R.assign((size_t)rank * c, 0.f);
  for (size_t j = 0; j < c; ++j)
    for (size_t r = 0; r < rank; ++r) R[j * rank + r] = (float)Rd[j * c + r];
  return (int)rank;
}

// ---------------------------------------------------------------------------------
// Symmetric eigendecomposition of a small dense matrix — stands in for LAPACK ssyevd
// reached through arma::eig_sym (block-ks/restarted_block_ks.h:150-151;
// armadillo_bits/auxlib_meat.hpp:1681-1682).  Householder tridiagonalisation +
// implicit QL (the published EISPACK tred2/tql2 algorithm), run in double.
// S: n x n col-major (upper triangle is what LAPACK 'U' reads; we symmetrise from upper).
// Out: e ascending, Z col-major eigenvectors.
// ---------------------------------------------------------------------------------
#define VV(r, c) v[(size_t)(c) * n + (r)]
static bool eig_sym_d(std::vector<double>& v, size_t n, dvec& d) {
  dvec e(n);
  d.resize(n);
  if (n == 0) return true;
  // tred2 — operates on the transposed access pattern of JAMA; matrix is symmetric so
  // V[i][j] == VV(j, i).  We index JAMA's V[i][j] as VV(j, i) to keep k-loops contiguous.
#define M(i, j) VV(j, i)
  for (size_t j = 0; j < n; ++j) d[j] = M(n - 1, j);
  for (size_t i = n - 1; i > 0; --i) {
    double scale = 0.0, h = 0.0;
    for (size_t k = 0; k < i; ++k) scale += std::fabs(d[k]);
    if (scale == 0.0) {
      e[i] = d[i - 1];
      for (size_t j = 0; j < i; ++j) {
        d[j] = M(i - 1, j);
        M(i, j) = 0.0;
        M(j, i) = 0.0;
      }
    } else {
      for (size_t k = 0; k < i; ++k) {
        d[k] /= scale;
        h += d[k] * d[k];
      }
      double f = d[i - 1];
      double g = std::sqrt(h);
      if (f > 0) g = -g;
      e[i] = scale * g;
      h = h - f * g;
      d[i - 1] = f - g;
      for (size_t j = 0; j < i; ++j) e[j] = 0.0;
      for (size_t j = 0; j < i; ++j) {
        f = d[j];
        M(j, i) = f;
        g = e[j] + M(j, j) * f;
        for (size_t k = j + 1; k <= i - 1; ++k) {
          g += M(k, j) * d[k];
          e[k] += M(k, j) * f;
        }
        e[j] = g;
      }
      f = 0.0;
      for (size_t j = 0; j < i; ++j) {
        e[j] /= h;
        f += e[j] * d[j];
      }
      const double hh = f / (h + h);
      for (size_t j = 0; j < i; ++j) e[j] -= hh * d[j];
      for (size_t j = 0; j < i; ++j) {
        f = d[j];
        g = e[j];
        for (size_t k = j; k <= i - 1; ++k) M(k, j) -= (f * e[k] + g * d[k]);
        d[j] = M(i - 1, j);
        M(i, j) = 0.0;
      }
    }
    d[i] = h;
  }
  for (size_t i = 0; i + 1 < n; ++i) {
    M(n - 1, i) = M(i, i);
    M(i, i) = 1.0;
    const double h = d[i + 1];
    if (h != 0.0) {
      for (size_t k = 0; k <= i; ++k) d[k] = M(k, i + 1) / h;
      for (size_t j = 0; j <= i; ++j) {
        double g = 0.0;
        for (size_t k = 0; k <= i; ++k) g += M(k, i + 1) * M(k, j);
        for (size_t k = 0; k <= i; ++k) M(k, j) -= g * d[k];
      }
    }
    for (size_t k = 0; k <= i; ++k) M(k, i + 1) = 0.0;
  }
  for (size_t j = 0; j < n; ++j) {
    d[j] = M(n - 1, j);
    M(n - 1, j) = 0.0;
  }
  M(n - 1, n - 1) = 1.0;
  e[0] = 0.0;
  // tql2
  for (size_t i = 1; i < n; ++i) e[i - 1] = e[i];
  e[n - 1] = 0.0;
  double f = 0.0, tst1 = 0.0;
  const double eps = std::pow(2.0, -52.0);
  for (size_t l = 0; l < n; ++l) {
    tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
    size_t m = l;
    while (m < n) {
      if (std::fabs(e[m]) <= eps * tst1) break;
      m++;
    }
    if (m > l) {
      int iter = 0;
      do {
        if (++iter > 200) return false;
        double g = d[l];
        double p = (d[l + 1] - g) / (2.0 * e[l]);
        double r = std::hypot(p, 1.0);
        if (p < 0) r = -r;
        d[l] = e[l] / (p + r);
        d[l + 1] = e[l] * (p + r);
        const double dl1 = d[l + 1];
        double h = g - d[l];
        for (size_t i = l + 2; i < n; ++i) d[i] -= h;
        f += h;
        p = d[m];
        double c = 1.0, c2 = c, c3 = c;
        const double el1 = e[l + 1];
        double s = 0.0, s2 = 0.0;
        for (size_t ii = m; ii-- > l;) {
          const size_t i = ii;
          c3 = c2;
          c2 = c;
          s2 = s;
          g = c * e[i];
          h = c * p;
          r = std::hypot(p, e[i]);
          e[i + 1] = s * r;
          s = e[i] / r;
          c = p / r;
          p = c * d[i] - s * g;
          d[i + 1] = h + s * (c * g + s * d[i]);
          // JAMA: V[k][i+1], V[k][i]  ->  M(k, i+1) = VV(i+1, k)?  No: eigenvector k-loop runs over
          // ROWS of V for COLUMNS i, i+1.  M(k, i) = VV(i, k) is strided; so the final matrix is
          // returned transposed (see caller) and here we rotate M(k,i)/M(k,i+1) as written.
          for (size_t k = 0; k < n; ++k) {
            h = M(k, i + 1);
            M(k, i + 1) = s * M(k, i) + c * h;
            M(k, i) = c * M(k, i) - s * h;
          }
        }
        p = -s * s2 * c3 * el1 * e[l] / dl1;
        e[l] = s * p;
        d[l] = c * p;
      } while (std::fabs(e[l]) > eps * tst1);
    }
    d[l] = d[l] + f;
    e[l] = 0.0;
  }
#undef M
  return true;
}
#undef VV

// eig_sym + descending sort (block-ks/restarted_block_ks.h:150-161).
// S col-major n x n float (upper triangle used).  evals desc, vecs col-major (column i = i-th vector).
static bool eig_sym_desc(const float* S, size_t n, fvec& evals, fvec& vecs) {
  std::vector<double> v((size_t)n * n);
  for (size_t c = 0; c < n; ++c)
    for (size_t r = 0; r < n; ++r) {
      const double x = (r <= c) ? (double)S[c * n + r] : (double)S[r * n + c];
      v[c * n + r] = x;
    }
  dvec d;
  if (!eig_sym_d(v, n, d)) return false;
  // After eig_sym_d, JAMA's V[k][i] (row k of eigenvector i) lives at M(k,i) = v[k*n + i]:
  // eigenvector i is v[k*n + i] for k = 0..n-1  (i.e. v is "row-major" eigenvector matrix).
  std::vector<size_t> idx(n);
  std::iota(idx.begin(), idx.end(), 0);
  std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return d[a] > d[b]; });
  evals.resize(n);
  vecs.resize((size_t)n * n);
  for (size_t i = 0; i < n; ++i) {
    evals[i] = (float)d[idx[i]];
    for (size_t k = 0; k < n; ++k) vecs[i * n + k] = (float)v[k * n + idx[i]];
  }
  return true;
}

// ---------------------------------------------------------------------------------
// block-ks/restarted_block_ks.h — BlockKs<ProdOp>, restated.  H is kept col-major
// float with explicit (rows, cols); V col-major float n x ncols.
// ---------------------------------------------------------------------------------
struct Mat {  // col-major float
  size_t r = 0, c = 0;
  fvec a;
  Mat() {}
  Mat(size_t r_, size_t c_) : r(r_), c(c_), a(r_ * c_, 0.f) {}
  float& operator()(size_t i, size_t j) { return a[j * r + i]; }
  float operator()(size_t i, size_t j) const { return a[j * r + i]; }
};
static Mat sub(const Mat& m, size_t r0, size_t c0, size_t r1, size_t c1) {  // inclusive like arma submat
  Mat o(r1 - r0 + 1, c1 - c0 + 1);
  for (size_t j = c0; j <= c1; ++j)
    for (size_t i = r0; i <= r1; ++i) o(i - r0, j - c0) = m(i, j);
  return o;
}
static Mat matmul(const Mat& A, const Mat& B) {
  Mat C(A.r, B.c);
  for (size_t j = 0; j < B.c; ++j)
    for (size_t k = 0; k < A.c; ++k) {
      const float b = B(k, j);
      for (size_t i = 0; i < A.r; ++i) C(i, j) += A(i, k) * b;
    }
  return C;
}

struct BlockKs {
  Op* op;
  size_t nev, ncv, maxit, blk, dim;
  float tol;
  Mat H;
  fvec V;        // dim x vcols
  size_t vcols;  // number of valid columns in V
  size_t nconv = 0, n_restarts = 0;
  Rng rng;
  bool ok = true;

  BlockKs(Op* op_, size_t nev_, size_t ncv_, size_t maxit_, size_t blk_, float tol_, uint64_t seed)
      : op(op_), nev(nev_), ncv(ncv_), maxit(maxit_), blk(blk_ < nev_ ? blk_ : 1),  // :198
        dim(op_->rows()), tol(tol_), vcols(0), rng(seed) {}

  void randu(fvec& F, size_t cols) {
    F.resize(dim * cols);
    for (size_t i = 0; i < dim * cols; ++i) F[i] = rng.randu();
  }

  // rank repair shared by init() (:238-258) and expand() (:106-132):
  // fill V columns [nvecs, target) with random vectors orthogonalised twice against V(:, :nvecs).
  void repair(size_t& nvecs, size_t target, size_t width) {
    size_t tries = 0;
    while (nvecs < target && tries < 100) {
      tries++;
      fvec F2, H2(nvecs * width), Q2, R2;
      randu(F2, width);
      for (int pass = 0; pass < 2; ++pass) {
        gemm_tn(V.data(), dim, nvecs, F2.data(), width, H2.data());
        gemm_sub(F2.data(), dim, width, V.data(), nvecs, H2.data());
      }
      int rk2 = compute_qr(F2.data(), dim, width, Q2, R2);
      for (int l = 0; l < rk2 && nvecs < target; ++l) {
        std::memcpy(&V[nvecs * dim], &Q2[(size_t)l * dim], dim * sizeof(float));
        nvecs++;
      }
    }
    if (nvecs < target) ok = false;  // reference constructs but never throws (:129-131, :255-257)
  }

  void init() {  // :203-259
    V.assign(dim * (ncv + blk), 0.f);
    fvec F, Q, R;
    int rank;
    do {
      randu(F, blk);
      rank = compute_qr(F.data(), dim, blk, Q, R);
    } while ((size_t)rank < blk);
    std::memcpy(V.data(), Q.data(), dim * blk * sizeof(float));
    fvec V1(dim * blk);
    op->multiply(V.data(), (int)blk, V1.data());
    Mat Hh(blk, blk), C(blk, blk);
    gemm_tn(V.data(), dim, blk, V1.data(), blk, Hh.a.data());       // H = V^T V1
    gemm_sub(V1.data(), dim, blk, V.data(), blk, Hh.a.data());      // V1 -= V H
    gemm_tn(V.data(), dim, blk, V1.data(), blk, C.a.data());        // C = V^T V1
    for (size_t i = 0; i < blk * blk; ++i) Hh.a[i] += C.a[i];       // H += C
    gemm_sub(V1.data(), dim, blk, V.data(), blk, C.a.data());       // V1 -= V C
    rank = compute_qr(V1.data(), dim, blk, Q, R);
    H = Mat(2 * blk, blk);
    for (size_t j = 0; j < blk; ++j) {
      for (size_t i = 0; i < blk; ++i) H(i, j) = Hh(i, j);
      for (size_t i = 0; i < (size_t)rank; ++i) H(blk + i, j) = R[j * rank + i];
    }
    std::memcpy(&V[blk * dim], Q.data(), dim * rank * sizeof(float));
    vcols = blk + rank;
    if ((size_t)rank < blk) repair(vcols, 2 * blk, blk - rank);
    vcols = 2 * blk;
  }

  void expand() {  // :62-136
    while (H.r < ncv) {
      const size_t m = H.r;  // current basis width
      const float* Vk = &V[H.c * dim];
      fvec F(dim * blk);
      op->multiply(Vk, (int)blk, F.data());
      Mat Hk(m, blk), Ck(m, blk);
      gemm_tn(V.data(), dim, m, F.data(), blk, Hk.a.data());
      gemm_sub(F.data(), dim, blk, V.data(), m, Hk.a.data());
      for (int j = 0; j < 2; ++j) {
        gemm_tn(V.data(), dim, m, F.data(), blk, Ck.a.data());
        gemm_sub(F.data(), dim, blk, V.data(), m, Ck.a.data());
        for (size_t i = 0; i < Hk.a.size(); ++i) Hk.a[i] += Ck.a[i];
      }
      // H = [H Hk ; 0 R]
      Mat Hn(m + blk, H.c + blk);
      for (size_t j = 0; j < H.c; ++j)
        for (size_t i = 0; i < m; ++i) Hn(i, j) = H(i, j);
      for (size_t j = 0; j < blk; ++j)
        for (size_t i = 0; i < m; ++i) Hn(i, H.c + j) = Hk(i, j);
      fvec Q, R;
      const int rk = compute_qr(F.data(), dim, blk, Q, R);
      for (int j = 0; j < rk; ++j) std::memcpy(&V[(Hn.c + j) * dim], &Q[(size_t)j * dim], dim * sizeof(float));
      for (size_t j = 0; j < blk; ++j)
        for (int i = 0; i < rk; ++i) Hn(m + i, H.c + j) = R[j * rk + i];
      H = Hn;
      if ((size_t)rk < blk) {
        size_t nvecs = H.c + rk;
        repair(nvecs, H.r, blk - rk);
      }
    }
    vcols = H.r;
  }

  void truncate() {  // :138-187
    const size_t n = H.c - nconv;
    Mat subH = sub(H, nconv, nconv, H.c - 1, H.c - 1);
    fvec eH, vH;
    if (!eig_sym_desc(subH.a.data(), n, eH, vH)) {
      ok = false;
      return;
    }
    // V = [ V(:, :nconv) | V(:, nconv:ncols-blk) * vH(:, :nev-nconv) | V(:, tail blk) ]
    const size_t keep = nev - nconv;
    fvec Vnew(dim * keep);
    gemm_nn(&V[nconv * dim], dim, n, vH.data(), n, keep, Vnew.data());
    fvec tail(&V[(vcols - blk) * dim], &V[(vcols - blk) * dim] + dim * blk);
    std::memcpy(&V[nconv * dim], Vnew.data(), dim * keep * sizeof(float));
    std::memcpy(&V[nev * dim], tail.data(), dim * blk * sizeof(float));
    vcols = nev + blk;
    // Transform H (:169-184)
    Mat last = sub(H, H.r - blk, H.c - blk, H.r - 1, H.c - 1);  // blk x blk
    Mat vHm(n, n);
    vHm.a = vH;
    Mat vtail = sub(vHm, n - blk, 0, n - 1, n - 1);  // blk x n
    Mat newrows = matmul(last, vtail);               // blk x n
    Mat top;
    if (nconv > 0) top = matmul(sub(H, 0, nconv, nconv - 1, H.c - 1), vHm);  // nconv x n
    for (size_t j = nconv; j < nev; ++j)
      for (size_t i = nconv; i < nev; ++i) H(i, j) = (i == j) ? eH[i - nconv] : 0.f;
    for (size_t j = 0; j < n; ++j)
      for (size_t i = 0; i < blk; ++i) H(nev + i, nconv + j) = newrows(i, j);
    if (nconv > 0)
      for (size_t j = 0; j < n; ++j)
        for (size_t i = 0; i < nconv; ++i) H(i, nconv + j) = top(i, j);
    H = sub(H, 0, 0, nev + blk - 1, nev - 1);
  }

  // residual rule :278-293 ; returns index of first non-converged column or ncols
  size_t first_unconverged(bool divide) const {
    for (size_t j = 0; j < H.c; ++j) {
      float s = 0.f;
      for (size_t i = H.r - blk; i < H.r; ++i) s += H(i, j) * H(i, j);
      float nrm = std::sqrt(s);
      if (divide) nrm = nrm / H(j, j);
      if (nrm >= tol) return j;
    }
    return H.c;
  }

  void compute() {  // :261-321
    n_restarts = 0;
    nconv = 0;
    expand();
    while (n_restarts < maxit && ok) {
      truncate();
      if (!ok) break;
      const size_t j = first_unconverged(true);
      if (j == H.c) {
        nconv = H.c;
        break;
      }
      nconv = j;
      ++n_restarts;
      expand();
    }
    if (n_restarts == maxit) {  // :303-317 (quirk: expanded H, no division)
      const size_t j = first_unconverged(false);
      nconv = j;
    }
    nconv = nconv >= nev ? nev : nconv;
  }
};

// ---------------------------------------------------------------------------------
// k-means on the projected / word space.  src/sparseMatrix.cpp:1494-2238.
// ---------------------------------------------------------------------------------
// src/sparseMatrix.cpp:1223-1231 compute_U_rowmajor
static void to_rowmajor(const float* Ucm, size_t V, size_t k, fvec& Urm) {
  Urm.resize(V * k);
#pragma omp parallel for
  for (int64_t r = 0; r < (int64_t)V; ++r)
    for (size_t c = 0; c < k; ++c) Urm[r * k + c] = Ucm[c * V + r];
}

// src/sparseMatrix.cpp:1749-1782 multiply_with: out (docs x cols, row-major) = B[:, d0:d1]^T * in (V x cols row-major)
static void multiply_with(const Csc& m, uint64_t d0, uint64_t d1, const float* in, float* out, size_t cols, float alpha = 1.f) {
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t d = (int64_t)d0; d < (int64_t)d1; ++d) {
    float* o = out + (size_t)(d - d0) * cols;
    for (size_t j = 0; j < cols; ++j) o[j] = 0.f;
    for (int64_t i = m.offs[d]; i < m.offs[d + 1]; ++i) {
      const float v = alpha * m.vals[i];
      const float* r = in + (size_t)m.rows[i] * cols;
      for (size_t j = 0; j < cols; ++j) o[j] += v * r[j];
    }
  }
}

// src/sparseMatrix.cpp:1888-1918 compute_projected_docs_l2sq
static void projected_docs_l2sq(const Csc& m, const fvec& Urm, size_t k, fvec& out) {
  out.resize(m.D);
  const uint64_t Db = 1 << 14;
  fvec blockbuf((size_t)Db * k);
  for (uint64_t d0 = 0; d0 < m.D; d0 += Db) {
    const uint64_t d1 = std::min<uint64_t>(m.D, d0 + Db);
    multiply_with(m, d0, d1, Urm.data(), blockbuf.data(), k);
#pragma omp parallel for
    for (int64_t d = (int64_t)d0; d < (int64_t)d1; ++d) {
      const float* p = &blockbuf[(size_t)(d - d0) * k];
      float s = 0.f;
      for (size_t j = 0; j < k; ++j) s += p[j] * p[j];
      out[d] = s;
    }
  }
}

// src/sparseMatrix.cpp:1819-1826: UUTrC (V x n, row-major) = -2 * U_rowmajor (V x k) * C^T,
// where C holds centre c at offset c*k.
static void make_UUTrC(const fvec& Urm, size_t V, size_t k, const float* C, size_t n, fvec& out) {
  out.resize(V * n);
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < (int64_t)V; ++r) {
    const float* u = &Urm[(size_t)r * k];
    float* o = &out[(size_t)r * n];
    for (size_t c = 0; c < n; ++c) {
      const float* cc = C + c * k;
      float s = 0.f;
      for (size_t j = 0; j < k; ++j) s += u[j] * cc[j];
      o[c] = -2.0f * s;
    }
  }
}

// cblas_isamin semantics: index of the first element with minimum |x|  
static inline uint32_t isamin(const float* x, size_t n) {
  uint32_t best = 0;
  float bv = std::fabs(x[0]);
  for (size_t i = 1; i < n; ++i) {
    const float a = std::fabs(x[i]);
    if (a < bv) {
      bv = a;
      best = (uint32_t)i;
    }
  }
  return best;
}

// dist block (docs x n row-major) = B_block^T * M (V x n row-major) + cn[c] + dn[d]
// src/sparseMatrix.cpp:1794-1849 and :1494-1550 share this shape.
static void dist_block(const Csc& m, uint64_t d0, uint64_t d1, const float* M, size_t n, const float* cn,
                       const float* dn, float alpha, float* dist) {
  multiply_with(m, d0, d1, M, dist, n, alpha);
#pragma omp parallel for
  for (int64_t d = (int64_t)d0; d < (int64_t)d1; ++d) {
    float* o = dist + (size_t)(d - d0) * n;
    for (size_t c = 0; c < n; ++c) o[c] = (o[c] + cn[c]) + dn[d];  // :1838 then :1843 (two rank-1 updates in this order)
  }
}

struct KmState {
  const Csc* m;
  fvec Urm;
  size_t k;
};

// src/sparseMatrix.cpp:2133-2209 kmeanspp_on_projected_space (+ seed injection hook).
// inject: if non-null, k doc ids that replace the draws (the seeding kernel is then
// exercised only for its min_dist updates); rounds follow the same schedule.
static float kmeanspp(const Csc& m, const fvec& Urm, size_t k, const uint64_t* inject, Rng& rng,
                      std::vector<uint64_t>& centers, int* rounds_out, fvec* min_dist_out, int max_rounds = 0) {
  const uint64_t D = m.D;
  fvec pl2;
  projected_docs_l2sq(m, Urm, k, pl2);
  fvec min_dist(D, FLT_MAX);
  fvec coords(k * k, 0.f);
  std::vector<float> cum(D + 1);
  centers.clear();
  uint64_t first = inject ? inject[0] : (uint64_t)(((size_t)rng.next31() * (size_t)84619573) % (size_t)D);  // :2150
  centers.push_back(first);
  multiply_with(m, first, first + 1, Urm.data(), coords.data(), k);
  int new_added = 1, rounds = 0;
  const uint64_t Db = 1 << 14;
  fvec dist;
  while (centers.size() < k) {
    if (max_rounds > 0 && rounds >= max_rounds) break;  // bench-only: bounded sample of rounds
    rounds++;
    const size_t n = (size_t)new_added;
    const float* newC = &coords[(centers.size() - n) * k];
    fvec cn(n);
    for (size_t c = 0; c < n; ++c) {
      float s = 0.f;
      for (size_t j = 0; j < k; ++j) s += newC[c * k + j] * newC[c * k + j];
      cn[c] = s;
    }
    fvec UU;
    make_UUTrC(Urm, m.V, k, newC, n, UU);
    dist.resize((size_t)Db * n);
    for (uint64_t d0 = 0; d0 < D; d0 += Db) {  // :2165-2169 -> :2075-2130
      const uint64_t d1 = std::min<uint64_t>(D, d0 + Db);
      dist_block(m, d0, d1, UU.data(), n, cn.data(), pl2.data(), 1.f, dist.data());
#pragma omp parallel for
      for (int64_t d = (int64_t)d0; d < (int64_t)d1; ++d)
        for (size_t c = 0; c < n; ++c) {
          float t = std::max(dist[(size_t)(d - d0) * n + c], 0.f);
          min_dist[d] = std::min(min_dist[d], t);
        }
    }
    cum[0] = 0.f;  // :2170-2172 sequential fp32 prefix sum
    for (uint64_t d = 0; d < D; ++d) cum[d + 1] = cum[d] + min_dist[d];
    const int s = (int)centers.size();
    new_added = 0;
    for (int c = 0; (c < 1 + std::sqrt((double)(s - 5 > 0 ? s - 5 : 0))) && centers.size() < k; ++c) {  // :2183
      uint64_t nc;
      if (inject) {
        nc = inject[centers.size()];
      } else {
        // :2184  auto dice_throw = dist_cumul[num_docs()] * rand_fraction();   (float * double -> double key)
        const double dice = (double)cum[D] * rng.fraction();
        nc = (uint64_t)(std::upper_bound(cum.begin(), cum.end(), dice) - 1 - cum.begin());
      }
      if (std::find(centers.begin(), centers.end(), nc) == centers.end()) {
        multiply_with(m, nc, nc + 1, Urm.data(), &coords[centers.size() * k], k);
        centers.push_back(nc);
        new_added++;
      }
    }
    if (inject && new_added == 0) break;  // malformed injection (duplicates): avoid infinite loop
  }
  if (rounds_out) *rounds_out = rounds;
  if (min_dist_out) *min_dist_out = min_dist;
  return cum[D - 1];  // :2208 (sic: omits the last doc)
}

// Reference stopping rule shared by both Lloyd drivers (src/sparseMatrix.cpp:2044-2064 / :1718-1738).
struct StopRule {
  std::vector<size_t> prev_sizes;
  std::vector<uint32_t> prev_assign;  // partition snapshot, only refreshed on iterations whose sizes matched
  bool have_prev = false;
  explicit StopRule(size_t k) : prev_sizes(k, 0) {}
  bool converged(const std::vector<uint32_t>& assign, size_t k) {
    std::vector<size_t> sizes(k, 0);
    for (uint32_t a : assign) sizes[a]++;
    bool changed = false;
    for (size_t c = 0; c < k; ++c)
      if (prev_sizes[c] != sizes[c]) changed = true;
    prev_sizes = sizes;
    if (!changed) {
      // prev_closest_docs starts as k empty lists: equal to the current partition only if it is all-empty
      if (!have_prev) {
        changed = !assign.empty();
      } else {
        changed = (prev_assign != assign);
      }
      prev_assign = assign;
      have_prev = true;
    }
    return !changed;
  }
};

// src/sparseMatrix.cpp:1921-2013 + :2016-2072
static int lloyds_projected(const Csc& m, const fvec& Urm, size_t k, float* C, int max_reps, std::vector<uint32_t>& assign) {
  const uint64_t D = m.D;
  fvec pl2;
  projected_docs_l2sq(m, Urm, k, pl2);  // :2032
  assign.assign(D, 0);
  StopRule stop(k);
  const uint64_t Db = 1 << 14;
  fvec dist((size_t)Db * k), proj((size_t)Db * k), UU;
  int it = 0;
  for (; it < max_reps; ++it) {
    fvec cn(k);
    for (size_t c = 0; c < k; ++c) {
      float s = 0.f;
      for (size_t j = 0; j < k; ++j) s += C[c * k + j] * C[c * k + j];
      cn[c] = s;
    }
    make_UUTrC(Urm, m.V, k, C, k, UU);
    for (uint64_t d0 = 0; d0 < D; d0 += Db) {
      const uint64_t d1 = std::min<uint64_t>(D, d0 + Db);
      dist_block(m, d0, d1, UU.data(), k, cn.data(), pl2.data(), 1.f, dist.data());
#pragma omp parallel for
      for (int64_t d = (int64_t)d0; d < (int64_t)d1; ++d) assign[d] = isamin(&dist[(size_t)(d - d0) * k], k);
    }
    std::memset(C, 0, sizeof(float) * k * k);  // :1957
    std::vector<size_t> sizes(k, 0);
    for (uint64_t d0 = 0; d0 < D; d0 += Db) {
      const uint64_t d1 = std::min<uint64_t>(D, d0 + Db);
      multiply_with(m, d0, d1, Urm.data(), proj.data(), k);  // :1975 UT_times_docs
      for (uint64_t d = d0; d < d1; ++d) {                   // doc order within a cluster = ascending (push_back order)
        const uint32_t c = assign[d];
        sizes[c]++;
        float* cc = C + (size_t)c * k;
        const float* p = &proj[(size_t)(d - d0) * k];
        for (size_t j = 0; j < k; ++j) cc[j] += p[j];
      }
    }
    for (size_t c = 0; c < k; ++c)
      if (sizes[c] > 0) {
        const float inv = 1.0f / (float)sizes[c];  // :1990-1991 FPscal(1/div)
        for (size_t j = 0; j < k; ++j) C[c * k + j] *= inv;
      }
    if (stop.converged(assign, k)) {
      ++it;
      break;
    }
  }
  return it;
}

// src/sparseMatrix.cpp:1587-1677 + :1690-1746.  centers: V x k col-major (centre c at c*V), in/out.
static int lloyds_sparse(const Csc& m, size_t k, float* centers, int max_reps, std::vector<uint32_t>& assign) {
  const uint64_t D = m.D, V = m.V;
  fvec dl2(D);
#pragma omp parallel for
  for (int64_t d = 0; d < (int64_t)D; ++d) {  // :1680-1687
    float s = 0.f;
    for (int64_t i = m.offs[d]; i < m.offs[d + 1]; ++i) s += m.vals[i] * m.vals[i];
    dl2[d] = s;
  }
  assign.assign(D, 0);
  StopRule stop(k);
  const uint64_t Db = 1 << 14;
  fvec dist((size_t)Db * k), ctr((size_t)V * k);
  int it = 0;
  for (; it < max_reps; ++it) {
    fvec cn(k);
#pragma omp parallel for
    for (int64_t c = 0; c < (int64_t)k; ++c) {  // :1575-1584
      float s = 0.f;
      const float* cc = centers + (size_t)c * V;
      for (size_t r = 0; r < V; ++r) s += cc[r] * cc[r];
      cn[c] = s;
    }
    // :1513-1514 transpose to row-major (V x k)
#pragma omp parallel for
    for (int64_t r = 0; r < (int64_t)V; ++r)
      for (size_t c = 0; c < k; ++c) ctr[(size_t)r * k + c] = centers[c * V + r];
    for (uint64_t d0 = 0; d0 < D; d0 += Db) {
      const uint64_t d1 = std::min<uint64_t>(D, d0 + Db);
      dist_block(m, d0, d1, ctr.data(), k, cn.data(), dl2.data(), -2.0f, dist.data());  // alpha = -2 (:1524)
#pragma omp parallel for
      for (int64_t d = (int64_t)d0; d < (int64_t)d1; ++d) assign[d] = isamin(&dist[(size_t)(d - d0) * k], k);
    }
    std::memset(centers, 0, sizeof(float) * (size_t)k * V);  // :1613
    std::vector<size_t> sizes(k, 0);
    std::vector<std::vector<uint32_t>> members(k);
    for (uint64_t d = 0; d < D; ++d) {
      members[assign[d]].push_back((uint32_t)d);
      sizes[assign[d]]++;
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t c = 0; c < (int64_t)k; ++c) {  // :1631-1638
      float* cc = centers + (size_t)c * V;
      for (uint32_t d : members[c])
        for (int64_t i = m.offs[d]; i < m.offs[d + 1]; ++i) cc[m.rows[i]] += m.vals[i];
    }
#pragma omp parallel for
    for (int64_t c = 0; c < (int64_t)k; ++c) {  // :1641-1646 (true division)
      const float div = (float)sizes[c];
      if (div > 0.0f) {
        float* cc = centers + (size_t)c * V;
        for (size_t r = 0; r < V; ++r) cc[r] /= div;
      }
    }
    if (stop.converged(assign, k)) {
      ++it;
      break;
    }
  }
  return it;
}

}  // namespace

// =================================================================================
// C ABI for the test-suite / bench (ctypes).
// =================================================================================
extern "C" {

void* orc_csc_create(uint64_t V, uint64_t D, uint64_t nnz, const float* vals, const uint32_t* rows, const int64_t* offs) {
  Csc* m = new Csc;
  m->V = V;
  m->D = D;
  m->nnz = nnz;
  m->vals.assign(vals, vals + nnz);
  m->rows.assign(rows, rows + nnz);
  m->offs.assign(offs, offs + D + 1);
  m->build_csr();
  return m;
}
void orc_csc_destroy(void* h) { delete (Csc*)h; }

int orc_num_threads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

// src/sparseMatrix.cpp:1096-1100
float orc_frobenius(void* h) {
  Csc* m = (Csc*)h;
  float s = 0.f;
  for (uint64_t i = 0; i < m->nnz; ++i) s += m->vals[i] * m->vals[i];
  return s;
}

int orc_gram_apply(void* h, const float* X, int b, float* Z) {
  if (b > 64) return -1;
  GramOp op((Csc*)h);
  op.multiply(X, b, Z);
  return 0;
}

static int run_ks(Op& op, int nev, int ncv, int maxit, int blk, float tol, uint64_t seed, float* evals, float* U,
                  int* nconv, int* restarts, int* napplies) {
  BlockKs ks(&op, nev, ncv, maxit, blk, tol, seed);
  ks.init();
  ks.compute();
  for (int i = 0; i < nev; ++i) evals[i] = ks.H(i, i);  // src/sparseMatrix.cpp:1212-1213
  if (U) std::memcpy(U, ks.V.data(), sizeof(float) * op.rows() * nev);
  if (nconv) *nconv = (int)ks.nconv;
  if (restarts) *restarts = (int)ks.n_restarts;
  if (napplies) *napplies = (int)op.napplies;
  return ks.ok ? 0 : 1;
}

// src/sparseMatrix.cpp:1195-1220 compute_block_ks (caller passes ncv = 2*nev + BLOCK_KS_BLOCK_SIZE etc.)
int orc_block_ks(void* h, int nev, int ncv, int maxit, int blk, float tol, uint64_t seed, float* evals, float* U,
                 int* nconv, int* restarts, int* napplies) {
  GramOp op((Csc*)h);
  return run_ks(op, nev, ncv, maxit, blk, tol, seed, evals, U, nconv, restarts, napplies);
}

// BlockKs on a dense symmetric operator (block-ks/ks_utils.h:167-182 ArmaMatProdOp): known-spectrum tests.
int orc_block_ks_dense(const float* A, uint64_t n, int nev, int ncv, int maxit, int blk, float tol, uint64_t seed,
                       float* evals, float* U, int* nconv, int* restarts, int* napplies) {
  DenseOp op(A, n);
  return run_ks(op, nev, ncv, maxit, blk, tol, seed, evals, U, nconv, restarts, napplies);
}

int orc_eig_sym(const float* S, uint64_t n, float* evals_desc, float* vecs) {
  fvec e, v;
  if (!eig_sym_desc(S, n, e, v)) return 1;
  std::memcpy(evals_desc, e.data(), n * sizeof(float));
  std::memcpy(vecs, v.data(), n * n * sizeof(float));
  return 0;
}

int orc_qr(const float* A, uint64_t n, uint64_t c, float* Q, float* R, int* rank) {
  fvec q, r;
  const int rk = compute_qr(A, n, c, q, r);
  std::memcpy(Q, q.data(), q.size() * sizeof(float));
  std::memcpy(R, r.data(), r.size() * sizeof(float));
  *rank = rk;
  return 0;
}

// P (D x k row-major) = B^T U ; norms[d] = |P_d|^2   (src/sparseMatrix.cpp:1785-1791, :1888-1918)
int orc_project(void* h, const float* Ucm, int k, float* P, float* norms) {
  Csc* m = (Csc*)h;
  fvec Urm;
  to_rowmajor(Ucm, m->V, k, Urm);
  if (P) multiply_with(*m, 0, m->D, Urm.data(), P, k);
  if (norms) {
    fvec n2;
    projected_docs_l2sq(*m, Urm, k, n2);
    std::memcpy(norms, n2.data(), m->D * sizeof(float));
  }
  return 0;
}

// src/sparseMatrix.cpp:2212-2238 kmeans_init_on_projected_space (KMEANS_INIT_REPS = 1)
int orc_kmeanspp(void* h, const float* Ucm, int k, const uint64_t* inject, uint64_t seed, uint64_t* seeds_out,
                 float* C_lowd, float* residual, int* rounds, float* min_dist_out, int max_rounds) {
  Csc* m = (Csc*)h;
  fvec Urm;
  to_rowmajor(Ucm, m->V, k, Urm);
  Rng rng(seed);
  std::vector<uint64_t> centers;
  fvec md;
  const float res = kmeanspp(*m, Urm, k, inject, rng, centers, rounds, &md, max_rounds);
  if (centers.size() != (size_t)k) return max_rounds > 0 ? 2 : 1;
  for (int c = 0; c < k; ++c) {
    seeds_out[c] = centers[c];
    multiply_with(*m, centers[c], centers[c] + 1, Urm.data(), C_lowd + (size_t)c * k, k);  // :2232-2234
  }
  if (residual) *residual = res;
  if (min_dist_out) std::memcpy(min_dist_out, md.data(), m->D * sizeof(float));
  return 0;
}

int orc_lloyds_projected(void* h, const float* Ucm, int k, float* C_lowd, int max_reps, int* iters, uint32_t* assign_out) {
  Csc* m = (Csc*)h;
  fvec Urm;
  to_rowmajor(Ucm, m->V, k, Urm);
  std::vector<uint32_t> assign;
  const int it = lloyds_projected(*m, Urm, k, C_lowd, max_reps, assign);
  if (iters) *iters = it;
  if (assign_out) std::memcpy(assign_out, assign.data(), m->D * sizeof(uint32_t));
  return 0;
}

// src/sparseMatrix.cpp:1438-1450 left_multiply_by_U_Spectra: centers (V x ncols) = U (V x k) * in (ld_in x ncols)
int orc_lift(const float* Ucm, uint64_t V, int k, const float* C_lowd, int ld_in, int ncols, float* centers) {
  gemm_nn(Ucm, V, k, C_lowd, ld_in, ncols, centers);
  return 0;
}

int orc_lloyds_sparse(void* h, int k, float* centers, uint32_t* assign_out, int max_reps, int* iters) {
  Csc* m = (Csc*)h;
  std::vector<uint32_t> assign;
  const int it = lloyds_sparse(*m, k, centers, max_reps, assign);
  if (iters) *iters = it;
  if (assign_out) std::memcpy(assign_out, assign.data(), m->D * sizeof(uint32_t));
  return 0;
}

}  // extern "C"
