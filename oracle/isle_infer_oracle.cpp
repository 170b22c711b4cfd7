// oracle/isle_infer_oracle.cpp
//
// TEST INFRASTRUCTURE ONLY — NOT PART OF THE PRODUCT.  PARITY UNPINNED (see isle_oracle.cpp: the reference ships no golden
// vectors for inference either and cannot be built here; its gemv calls go to closed-source MKL).
//
// CPU restatement of ISLEInfer: multiplicative-weights inference of the topic weights of a document
// given a topic model.  Each function cites the reference file:line it follows; FPgemv calls are plain fp32 loops.
//
// model_by_word: V x k row-major (element (word, topic) at word * k + topic), the layout load_model_from_sparse_file builds
// (src/infer.cpp:32-70).  The documents arrive as a count matrix in CSC; normalize_docs(true, true)
// (src/sparseMatrix.cpp:136-167, drivers/ISLEInfer.cpp:60) divides each count by the document's sum.
#include <cmath>
#include <cstdint>
#include <numeric>
#include <vector>

namespace {

// ISLEInfer::grad  src/infer.cpp:443-465
void grad(std::vector<float>& gradw, std::vector<float>& z, const float* a, const float* M, const float* w, int nnzs, int k) {
  for (int r = 0; r < nnzs; ++r) {  // FPgemv(RowMajor, NoTrans): z = M w
    float s = 0.f;
    for (int c = 0; c < k; ++c) s += M[(size_t)r * k + c] * w[c];
    z[r] = s;
  }
  for (int d = 0; d < nnzs; ++d) z[d] = a[d] / z[d];
  for (int c = 0; c < k; ++c) gradw[c] = 0.f;  // FPgemv(RowMajor, Trans): gradw = M^T z
  for (int r = 0; r < nnzs; ++r)
    for (int c = 0; c < k; ++c) gradw[c] += M[(size_t)r * k + c] * z[r];
}

// ISLEInfer::mwu  src/infer.cpp:394-441
bool mwu(const float* a, const float* M, float* w, int nnzs, int iters, float Lf, int k) {
  bool converged = false;
  for (int t = 0; t < k; ++t) w[t] = 1.0f / (float)k;
  if (nnzs == 0) return converged;
  std::vector<float> gradw(k), z(nnzs);
  for (int guessLf = 0; guessLf < 10; guessLf++) {
    for (int t = 0; t < k; ++t) w[t] = 1.0f / (float)k;
    for (int iter = 0; iter < iters; ++iter) {
      grad(gradw, z, a, M, w, nnzs, k);
      const double eta = std::sqrt(2.0 * std::log((float)k) / (float)(iter + 1)) / Lf;  // :415
      for (int t = 0; t < k; ++t) w[t] *= std::exp(eta * gradw[t]);                      // :418 (double exp, float store)
      const float normalizer = std::accumulate(w, w + k, 0.0f);                          // :420 (fp32, in order)
      for (int t = 0; t < k; ++t) w[t] /= normalizer;
    }
    const double sumw = std::accumulate(w, w + k, 0.0);  // :425
    if (std::isnormal(sumw)) {
      if (!(std::abs(1 - sumw) > 0.01)) {  // :427-432 (the other arm only prints and retries with the same Lf)
        converged = true;
        break;
      }
    } else {
      Lf *= 2.0f;
    }
  }
  return converged;
}

}  // namespace

extern "C" {

// drivers/ISLEInfer.cpp:60-100 + ISLEInfer::infer_doc_in_file src/infer.cpp:361-391 + calculate_llh :467-492 for every document.
// weights: D x k (uniform 1/k where inference did not converge, as the dense writer prints them :136); llh: D x 2
// (first = sum * avg_doc_sz, second = sum * words_in_doc; {0, 0} when not converged); returns the number of converged documents.
uint64_t orc_infer(uint64_t V, int k, const float* model_by_word, uint64_t D, const int64_t* offs, const uint32_t* rows, const float* counts,
                   int iters, float Lfguess, float avg_doc_sz, float* weights, float* llh) {
  (void)V;
  uint64_t nconv = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : nconv)
  for (int64_t doc = 0; doc < (int64_t)D; ++doc) {
    const int64_t beg = offs[doc], end = offs[doc + 1];
    float doc_sum = 0.f;  // normalize_docs: std::accumulate(..., (T)0.0)
    for (int64_t p = beg; p < end; ++p) doc_sum += counts[p];
    std::vector<float> a, M;
    int words_in_doc = 0;
    for (int64_t p = beg; p < end; ++p) {
      const uint32_t word = rows[p];
      words_in_doc += 1;
      const float* row = model_by_word + (size_t)word * k;
      if (std::accumulate(row, row + k, 0.0) > 1.0e-10) {  // :376 (double accumulation)
        a.push_back(counts[p] / doc_sum);
        M.insert(M.end(), row, row + k);
      }
    }
    const int nnzs = (int)a.size();
    float* w = weights + (size_t)doc * k;
    float first = 0.f, second = 0.f;
    if (mwu(a.data(), M.data(), w, nnzs, iters, Lfguess, k)) {
      // calculate_llh
      float s = 0.f;
      for (int r = 0; r < nnzs; ++r) {
        float zr = 0.f;
        for (int c = 0; c < k; ++c) zr += M[(size_t)r * k + c] * w[c];
        s += a[r] * std::log(zr);
      }
      second = s * words_in_doc;
      first = s * avg_doc_sz;
    }
    llh[2 * doc] = first;
    llh[2 * doc + 1] = second;
    if (first != 0.0f) nconv++;  // drivers/ISLEInfer.cpp:93
    else
      for (int t = 0; t < k; ++t) w[t] = 1.0f / (float)k;
  }
  return nconv;
}

}  // extern "C"
