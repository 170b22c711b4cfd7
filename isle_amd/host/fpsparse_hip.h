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
#include <algorithm>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/isle_hip.h"

namespace ISLE {

// include/types.h:24-36 under -DMKL_ILP64 -DSINGLE
typedef uint64_t word_id_t;
typedef uint64_t doc_id_t;
typedef int64_t offset_t;
typedef float FPTYPE;
typedef uint32_t count_t;  // include/types.h:29

// include/hyperparams.h
#define ISLE_BLOCK_KS_MAX_ITERS 100
#define ISLE_BLOCK_KS_BLOCK_SIZE 10
#define ISLE_BLOCK_KS_TOLERANCE 1e-4f
#define ISLE_W0_C (1.0)            // :8
#define ISLE_EPS2_C (1.0 / 3.0)    // :10
#define ISLE_RHO_C (1.1)           // :11
#define ISLE_EPS3_C (5.0)          // :12
#define ISLE_EDGE_TOPIC_MIN_DOCS 1          // :77
#define ISLE_EDGE_TOPIC_PRIMARY_RATIO 0.7   // :79

class FPSparseMatrixHip {
  word_id_t vocab_size_;
  doc_id_t num_docs_;
  offset_t nnzs_ = 0;
  isle_ctx* ctx_ = nullptr;
  bool uploaded_ = false;
  doc_id_t U_cols_ = 0;
  int last_nconv_ = 0;  // Ritz pairs that really passed the residual test in the last compute_block_ks

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

  void threshold_on_device(doc_id_t num_topics, double sample_rate, std::vector<doc_id_t>& original_cols, std::vector<FPTYPE>* zetas,
                           uint64_t* entries_above_threshold, float* avg_doc_sz) {
    uint64_t docs_kept = 0, nnz_kept = 0;
    check(isle_hip_threshold(ctx_, num_topics, sample_rate, 0, &docs_kept, &nnz_kept, entries_above_threshold, avg_doc_sz), "threshold");
    num_docs_ = docs_kept;
    nnzs_ = (offset_t)nnz_kept;
    uploaded_ = true;
    original_cols.resize(docs_kept);
    static_assert(sizeof(doc_id_t) == sizeof(uint64_t), "doc_id_t is 8 bytes (include/types.h:25)");
    if (zetas) zetas->resize(vocab_size_);
    check(isle_hip_get_B(ctx_, nullptr, nullptr, nullptr, (uint64_t*)original_cols.data(), zetas ? zetas->data() : nullptr), "get_B");
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
      B->threshold_on_device(num_topics, sample_rate, original_cols, zetas, entries_above_threshold, avg_doc_sz);
    } catch (...) {
      delete B;
      throw;
    }
    return B;
  }

  // The same with the ingest on the device as well: `text` holds the bytes of the tdf file
  // (DocWordEntriesReader::fill_doc_word_entries include/utils.h:158-228, the sort / de-duplication of
  // ISLETrainer::finalize_data src/trainer.cpp:236-247 and SparseMatrix::populate_CSC src/sparseMatrix.cpp:58-87).
  static FPSparseMatrixHip* from_tdf(word_id_t vocab_size, doc_id_t num_docs, const char* text, uint64_t nbytes, offset_t max_entries,
                                     doc_id_t num_topics, double sample_rate, std::vector<doc_id_t>& original_cols,
                                     uint64_t* entries_in_A = nullptr, uint64_t* entries_above_threshold = nullptr, float* avg_doc_sz = nullptr,
                                     int device = 0) {
    FPSparseMatrixHip* B = new FPSparseMatrixHip(vocab_size, 0, device);
    try {
      B->check(isle_hip_ingest_tdf(B->ctx_, text, nbytes, vocab_size, num_docs, (uint64_t)max_entries, nullptr, entries_in_A), "ingest_tdf");
      B->threshold_on_device(num_topics, sample_rate, original_cols, nullptr, entries_above_threshold, avg_doc_sz);
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
    if (rc == ISLE_E_NOCONV) {
      // the reference's log line says nconv = num_topics here and its assert passes; keep its line, but say what happened
      std::fprintf(stderr, "WARNING: block Krylov-Schur used all %d restarts; only %d of %d Ritz pairs passed the residual test "
                           "(tolerance %g). The unconverged Ritz vectors are used as they are, as the reference does.\n",
                   ISLE_BLOCK_KS_MAX_ITERS, nconv, (int)num_topics, (double)ISLE_BLOCK_KS_TOLERANCE);
      last_nconv_ = nconv;
    } else {
      last_nconv_ = (int)num_topics;
    }
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
  // out == NULL: the product stays on the device as the start point of run_lloyds(k, NULL, ...)
  void left_multiply_by_U_Spectra(FPTYPE* const out, const FPTYPE* in, const doc_id_t ld_in, const doc_id_t ncols) {  // :1438-1450
    assert(ld_in >= U_cols_);
    check(isle_hip_lift_centers(ctx_, in, (int)ld_in, (int)ncols, out), "left_multiply_by_U_Spectra");
  }
  // centers == NULL: start from the centres left_multiply_by_U_Spectra(NULL, ...) left on the device and do not copy the result back
  // (src/trainer.cpp reads only closest_docs after this call; V x k floats are 400 MB at vocab 100k, k = 1000).
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

  // ---- the stage after the hot path, on the count matrix this object was built from (from_counts) --------------------
  // src/trainer.cpp:577-627: A_sp->rth_highest_element(r, closest_docs[t], ...) for every topic, then
  // A_sp->find_catchwords(num_topics, catchword_thresholds, catchwords).  The partition is the one run_lloyds left on the
  // device (its closest_docs, mapped through original_cols exactly as :572-575 does).
  void find_catchwords(const doc_id_t num_topics, const uint64_t r, FPTYPE* catchword_thresholds /*vocab x topics, col-major*/,
                       std::vector<word_id_t>* catchwords /*[num_topics]*/) {
    std::vector<int32_t> catch_topic(vocab_size_);
    uint64_t n = 0;
    check(isle_hip_catchwords(ctx_, (int)num_topics, nullptr, r, ISLE_RHO_C, catchword_thresholds, catch_topic.data(), &n), "find_catchwords");
    for (doc_id_t t = 0; t < num_topics; ++t) catchwords[t].clear();
    for (word_id_t w = 0; w < vocab_size_; ++w)
      if (catch_topic[w] >= 0) catchwords[catch_topic[w]].push_back(w);
  }
  // A_sp->construct_topic_model(Model, num_topics, closest_docs, catchwords, ..., &top_topic_pairs, ...)
  // src/sparseMatrix.cpp:597-838; rank_threshold as at :720.
  void construct_topic_model(FPTYPE* Model /*vocab x topics, col-major*/, const doc_id_t num_topics, const doc_id_t num_docs_A,
                             std::vector<std::tuple<int, int, doc_id_t>>* top_topic_pairs) {
    const uint64_t rank_threshold = (doc_id_t)(ISLE_EPS3_C * ISLE_W0_C * (FPTYPE)num_docs_A / ((FPTYPE)num_topics * 2.0));
    std::vector<int32_t> t1, t2;
    if (top_topic_pairs) {
      t1.resize(num_docs_A);
      t2.resize(num_docs_A);
    }
    check(isle_hip_topic_model(ctx_, (int)num_topics, rank_threshold, Model, nullptr, top_topic_pairs ? t1.data() : nullptr,
                               top_topic_pairs ? t2.data() : nullptr, nullptr),
          "construct_topic_model");
    if (top_topic_pairs) {
      top_topic_pairs->clear();
      for (doc_id_t d = 0; d < num_docs_A; ++d)
        if (t1[d] >= 0 && t2[d] >= 0) top_topic_pairs->push_back(std::make_tuple((int)t1[d], (int)t2[d], d));
    }
  }
  // ISLETrainer::construct_edge_topics_v2 (src/trainer.cpp:1116-1167): pair selection on the host, the FPaxpy pair on the
  // device.  Ties in the count ordering are broken by (primary, secondary) ascending (the reference's sort is unstable).
  void construct_edge_topics(std::vector<std::tuple<int, int, doc_id_t>>& top_topic_pairs, const int max_edge_topics,
                             std::vector<std::tuple<int, int, uint64_t>>& selected_pairs, std::vector<FPTYPE>& EdgeModel /*vocab x #edge*/) {
    auto lt = [](const std::tuple<int, int, doc_id_t>& l, const std::tuple<int, int, doc_id_t>& r) {
      return std::get<0>(l) < std::get<0>(r) || (std::get<0>(l) == std::get<0>(r) && std::get<1>(l) < std::get<1>(r));
    };
    std::sort(top_topic_pairs.begin(), top_topic_pairs.end(), lt);
    selected_pairs.clear();
    for (size_t i = 0; i < top_topic_pairs.size();) {
      size_t j = i;
      while (j < top_topic_pairs.size() && !lt(top_topic_pairs[i], top_topic_pairs[j])) ++j;
      if ((int64_t)(j - i) >= ISLE_EDGE_TOPIC_MIN_DOCS)
        selected_pairs.push_back(std::make_tuple(std::get<0>(top_topic_pairs[i]), std::get<1>(top_topic_pairs[i]), (uint64_t)(j - i)));
      i = j;
    }
    std::cout << "#Candidates for edge topics: " << selected_pairs.size() << std::endl;
    std::stable_sort(selected_pairs.begin(), selected_pairs.end(),
                     [](const std::tuple<int, int, uint64_t>& l, const std::tuple<int, int, uint64_t>& r) { return std::get<2>(l) > std::get<2>(r); });
    if ((int64_t)selected_pairs.size() > (int64_t)max_edge_topics) {
      std::cout << "Edge topic threshold: " << std::get<2>(selected_pairs[max_edge_topics]) << std::endl;
      selected_pairs.resize(max_edge_topics);
    }
    std::cout << "#Edge topics: " << selected_pairs.size() << std::endl;
    std::vector<int64_t> pq(2 * selected_pairs.size());
    for (size_t e = 0; e < selected_pairs.size(); ++e) {
      pq[2 * e] = std::get<0>(selected_pairs[e]);
      pq[2 * e + 1] = std::get<1>(selected_pairs[e]);
    }
    EdgeModel.assign((size_t)vocab_size_ * selected_pairs.size(), 0.f);
    check(isle_hip_edge_topics(ctx_, pq.data(), (int)selected_pairs.size(), (float)ISLE_EDGE_TOPIC_PRIMARY_RATIO, EdgeModel.data()),
          "construct_edge_topics");
    std::cout << "Completed edge topic construction" << std::endl;
  }
};

}  // namespace ISLE
