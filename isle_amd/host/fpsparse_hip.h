// isle_amd/host/fpsparse_hip.h — C++ host side above the C ABI: the hot-path subset of the reference's
// ISLE::FPSparseMatrix<float> (include/sparseMatrix.h:204-467) with the SAME method names, argument meaning
// and error behaviour, each forwarding to libisle_hip.so.  Header-only; link with -lisle_hip.
//
// A maintainer of the reference would paste these bodies into src/sparseMatrix.cpp;
// this class exists so that the call sequence of ISLETrainer::train() (src/trainer.cpp:490-571) can be
// compiled and run against the GPU library without the reference's MKL-dependent sources.
#pragma once
#include <cassert>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/isle_hip.h"

namespace ISLE {

// include/types.h:24-36 under -DMKL_ILP64 -DSINGLE
typedef uint64_t word_id_t;
typedef uint64_t doc_id_t;
typedef int64_t offset_t;
typedef float FPTYPE;

// include/hyperparams.h
#define ISLE_BLOCK_KS_MAX_ITERS 100
#define ISLE_BLOCK_KS_BLOCK_SIZE 10
#define ISLE_BLOCK_KS_TOLERANCE 1e-4f

class FPSparseMatrixHip {
  word_id_t vocab_size_;
  doc_id_t num_docs_;
  offset_t nnzs_ = 0;
  isle_ctx* ctx_ = nullptr;
  bool uploaded_ = false;
  doc_id_t U_cols_ = 0;

  void check(int rc, const char* what) const {
    if (rc != 0) throw std::runtime_error(std::string(what) + ": " + isle_hip_last_error(ctx_));
  }
  void upload() {
    if (uploaded_) return;
    check(isle_hip_upload_csc_u64(ctx_, vocab_size_, num_docs_, (uint64_t)nnzs_, vals_CSC, rows_CSC, offsets_CSC, 0, num_docs_),
          "upload_csc");
    uploaded_ = true;
  }
  void fill_partition(const uint32_t* assign, std::vector<doc_id_t>* closest_docs, doc_id_t num_centers) const {
    if (!closest_docs) return;
    for (doc_id_t c = 0; c < num_centers; ++c) closest_docs[c].clear();
    for (doc_id_t d = 0; d < num_docs_; ++d) closest_docs[assign[d]].push_back(d);  // ascending, as :1669-1672
  }

 public:
  // the reference re-exports these as public (include/sparseMatrix.h:216,227-230)
  FPTYPE* vals_CSC = nullptr;
  word_id_t* rows_CSC = nullptr;
  offset_t* offsets_CSC = nullptr;

  FPSparseMatrixHip(word_id_t d, doc_id_t s, int device = 0) : vocab_size_(d), num_docs_(s) {
    ctx_ = isle_hip_create(device);
    if (!ctx_) throw std::runtime_error("isle_hip_create failed: no MI355X device (there is no CPU fallback)");
    offsets_CSC = new offset_t[s + 1]();
  }
  ~FPSparseMatrixHip() {
    delete[] vals_CSC;
    delete[] rows_CSC;
    delete[] offsets_CSC;
    isle_hip_destroy(ctx_);
  }
  FPSparseMatrixHip(const FPSparseMatrixHip&) = delete;

  // Device-side equivalent of SparseMatrix::normalize_docs + compute_thresholds (src/sparseMatrix.cpp:136-167, :357-485)
  // followed by FPSparseMatrix(A_sp, true) + threshold_and_copy / sampled_threshold_and_copy (:1285-1435), i.e. what
  // ISLETrainer::train does at src/trainer.cpp:430-485.  `counts`/`rows`/`offsets` are A_sp's CSC (populate_CSC layout).
  // B is built in device memory (the host CSC pointers of the returned object stay NULL); original_cols maps its columns
  // back to A's, zetas (optional) receives the per-word thresholds.
  static FPSparseMatrixHip* from_counts(word_id_t vocab_size, doc_id_t num_docs, const float* counts, const uint32_t* rows,
                                        const offset_t* offsets, doc_id_t num_topics, double sample_rate,
                                        std::vector<doc_id_t>& original_cols, std::vector<FPTYPE>* zetas = nullptr,
                                        uint64_t* entries_above_threshold = nullptr, float* avg_doc_sz = nullptr, int device = 0) {
    FPSparseMatrixHip* B = new FPSparseMatrixHip(vocab_size, 0, device);
    try {
      B->check(isle_hip_upload_counts_u32(B->ctx_, vocab_size, num_docs, (uint64_t)offsets[num_docs], counts, rows, offsets, 0, num_docs),
               "upload_counts");
      uint64_t docs_kept = 0, nnz_kept = 0;
      B->check(isle_hip_threshold(B->ctx_, num_topics, sample_rate, 0, &docs_kept, &nnz_kept, entries_above_threshold, avg_doc_sz),
               "threshold");
      B->num_docs_ = docs_kept;
      B->nnzs_ = (offset_t)nnz_kept;
      B->uploaded_ = true;
      original_cols.resize(docs_kept);
      static_assert(sizeof(doc_id_t) == sizeof(uint64_t), "doc_id_t is 8 bytes (include/types.h:25)");
      if (zetas) zetas->resize(vocab_size);
      B->check(isle_hip_get_B(B->ctx_, nullptr, nullptr, nullptr, (uint64_t*)original_cols.data(), zetas ? zetas->data() : nullptr), "get_B");
    } catch (...) {
      delete B;
      throw;
    }
    return B;
  }

  void allocate(offset_t nnzs) {  // SparseMatrix::allocate
    delete[] vals_CSC;
    delete[] rows_CSC;
    nnzs_ = nnzs;
    vals_CSC = new FPTYPE[nnzs];
    rows_CSC = new word_id_t[nnzs];
    uploaded_ = false;
  }
  word_id_t vocab_size() const { return vocab_size_; }
  doc_id_t num_docs() const { return num_docs_; }
  offset_t get_nnzs() const { return nnzs_; }

  FPTYPE frobenius() {  // src/sparseMatrix.cpp:1096-1100
    assert(offsets_CSC[0] == 0);
    upload();
    float f = 0.f;
    check(isle_hip_frobenius(ctx_, &f), "frobenius");
    return f;
  }
  void initialize_for_eigensolver(const doc_id_t num_topics) {  // :1150-1158 (device buffers are sized on demand)
    U_cols_ = num_topics;
    upload();
  }
  void compute_block_ks(const doc_id_t num_topics, std::vector<FPTYPE>& evalues) {  // :1195-1220
    upload();
    std::vector<float> ev(num_topics);
    int nconv = 0, restarts = 0, napplies = 0;
    const int rc = isle_hip_block_ks(ctx_, (int)num_topics, (int)(2 * num_topics + ISLE_BLOCK_KS_BLOCK_SIZE), ISLE_BLOCK_KS_MAX_ITERS,
                                     ISLE_BLOCK_KS_BLOCK_SIZE, ISLE_BLOCK_KS_TOLERANCE, 1, ev.data(), &nconv, &restarts, &napplies);
    // the reference reports nconv = nev even when maxit is exhausted and asserts on it (:1207)
    if (rc != 0 && rc != ISLE_E_NOCONV) check(rc, "compute_block_ks");
    std::printf("Completed with %d restarts, nconv = %d\n", restarts, rc == ISLE_E_NOCONV ? (int)num_topics : nconv);
    for (doc_id_t i = 0; i < num_topics; ++i) evalues.push_back(ev[i]);
    U_cols_ = num_topics;
  }
  void cleanup_after_eigensolver() {}  // :1264-1275 (U stays device-resident until the context dies)

  FPTYPE kmeans_init_on_projected_space(const int num_centers, const int max_reps, std::vector<doc_id_t>& best_seed,
                                        FPTYPE* const best_centers_coords) {  // :2212-2238
    FPTYPE best = 3.402823466e+38f;
    std::vector<uint64_t> seeds(num_centers), best_s;
    std::vector<float> coords((size_t)num_centers * num_centers);
    for (int rep = 0; rep < max_reps; ++rep) {
      float dist = 0.f;
      check(isle_hip_kmeanspp_projected(ctx_, num_centers, nullptr, 1 + rep, seeds.data(), coords.data(), &dist, nullptr),
            "kmeans_init_on_projected_space");
      std::cout << "k-means init residual: " << dist << std::endl;
      if (dist < best) {
        best = dist;
        best_s = seeds;
        if (best_centers_coords) std::memcpy(best_centers_coords, coords.data(), coords.size() * sizeof(float));
      }
    }
    best_seed.assign(best_s.begin(), best_s.end());
    return best;
  }
  FPTYPE run_lloyds_on_projected_space(const doc_id_t num_centers, FPTYPE* projected_centers, std::vector<doc_id_t>* closest_docs,
                                       const int max_reps) {  // :2016-2072
    if (closest_docs)
      for (doc_id_t c = 0; c < num_centers; ++c) assert(closest_docs[c].size() == 0);
    std::vector<uint32_t> assign(num_docs_);
    int iters = 0;
    check(isle_hip_lloyds_projected(ctx_, (int)num_centers, projected_centers, max_reps, &iters, assign.data()),
          "run_lloyds_on_projected_space");
    if (iters < max_reps) std::cout << "Lloyds converged\n";
    fill_partition(assign.data(), closest_docs, num_centers);
    return 0.0f;  // the reference returns the (disabled) residual: always 0 (:1995-1998)
  }
  void left_multiply_by_U_Spectra(FPTYPE* const out, const FPTYPE* in, const doc_id_t ld_in, const doc_id_t ncols) {  // :1438-1450
    assert(ld_in >= U_cols_);
    check(isle_hip_lift_centers(ctx_, in, (int)ld_in, (int)ncols, out), "left_multiply_by_U_Spectra");
  }
  FPTYPE run_lloyds(const doc_id_t num_centers, FPTYPE* centers, std::vector<doc_id_t>* closest_docs, const int max_reps) {  // :1690-1746
    if (closest_docs)
      for (doc_id_t c = 0; c < num_centers; ++c) assert(closest_docs[c].size() == 0);
    upload();
    std::vector<uint32_t> assign(num_docs_);
    int iters = 0;
    check(isle_hip_lloyds_sparse(ctx_, (int)num_centers, centers, centers, assign.data(), max_reps, &iters), "run_lloyds");
    for (int i = 0; i < iters; ++i) std::cout << "Lloyd's iter " << i << "  dist_sq residual: " << 0 << "\n";  // :1714 (residual disabled)
    if (iters < max_reps) std::cout << "Lloyds converged\n";
    fill_partition(assign.data(), closest_docs, num_centers);
    return 0.0f;
  }
};

}  // namespace ISLE
