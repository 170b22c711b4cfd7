// oracle/isle_post_oracle.cpp
//
// TEST INFRASTRUCTURE ONLY — NOT PART OF THE PRODUCT.  PARITY UNPINNED (see isle_oracle.cpp: the reference ships no
// golden vectors for this stage either and cannot be built here).
//
// CPU restatement of the stage immediately DOWNSTREAM of the hot path and of the edge-topic
// construction (§8a a19): catchword thresholds, catchwords, topic-model construction, edge topics.  Each function cites
// the reference file:line it follows; MKL calls are plain loops, std::sort stands where the reference sorts.
//
// Layouts: A is the word-document matrix in CSC with the NORMALISED values nv[i] = avg_doc_sz * (count_i / doc_sum)
// (normalize_docs, src/sparseMatrix.cpp:136-167); cluster_of[d] = topic whose closest_docs list holds document d of A
// (-1: none; src/trainer.cpp:572-575 maps B's columns back through original_cols); thresholds / Model are column-major
// V x k (element (word, topic) at topic * V + word), as catchword_thresholds and DenseMatrix are.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <tuple>
#include <vector>

extern "C" {

// SparseMatrix::normalize_docs, FPTYPE branch, normalize_to_one = false   src/sparseMatrix.cpp:136-167
void orc_post_normalize(uint64_t D, const int64_t* offs, const float* counts, float avg_doc_sz, float* nv) {
#pragma omp parallel for schedule(dynamic, 4096)
  for (int64_t d = 0; d < (int64_t)D; ++d) {
    float sum = 0.f;
    for (int64_t i = offs[d]; i < offs[d + 1]; ++i) sum += counts[i];
    for (int64_t i = offs[d]; i < offs[d + 1]; ++i) nv[i] = avg_doc_sz * (counts[i] / sum);
  }
}

// SparseMatrix::rth_highest_element for every topic   src/sparseMatrix.cpp:491-524, called at src/trainer.cpp:586-590
// (r is computed by the caller: src/trainer.cpp:579-583).
void orc_post_catch_thresholds(uint64_t V, uint64_t D, uint32_t k, const int64_t* offs, const uint32_t* rows, const float* nv,
                               const int32_t* cluster_of, uint64_t r, float* thr /*V*k col-major*/) {
  std::vector<std::vector<uint64_t>> part(k);
  for (uint64_t d = 0; d < D; ++d)
    if (cluster_of[d] >= 0) part[cluster_of[d]].push_back(d);
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t t = 0; t < (int64_t)k; ++t) {
    float* out = thr + (size_t)t * V;
    const std::vector<uint64_t>& docs = part[t];
    if (docs.empty()) {  // :500-504
      for (uint64_t w = 0; w < V; ++w) out[w] = 0.f;
      continue;
    }
    std::vector<std::vector<float>> freqs(V);
    for (uint64_t d : docs)
      for (int64_t i = offs[d]; i < offs[d + 1]; ++i) freqs[rows[i]].push_back(nv[i]);  // :508-510
    for (uint64_t w = 0; w < V; ++w) {
      std::vector<float>& f = freqs[w];
      if (f.size() > r) {  // :513-516
        std::sort(f.begin(), f.end(), std::greater<float>());
        out[w] = f[r - 1];
      } else {             // :517-523
        out[w] = (r >= docs.size()) ? ((f.size() == docs.size()) ? *std::min_element(f.begin(), f.end()) : 0.f) : 0.f;
      }
    }
  }
}

// SparseMatrix::find_catchwords   src/sparseMatrix.cpp:573-595.  catch_topic[w] = the topic word w is a catchword of
// (the rule admits at most one), -1 if none; returns the number of catchwords or -1 if a word qualified twice.
int64_t orc_post_find_catchwords(uint64_t V, uint32_t k, const float* thr, double rho, int32_t* catch_topic) {
  for (uint64_t w = 0; w < V; ++w) catch_topic[w] = -1;
  int64_t n = 0;
  bool twice = false;
  for (uint32_t t = 0; t < k; ++t)
    for (uint64_t w = 0; w < V; ++w) {
      bool is_catchword = false;
      for (uint32_t o = 0; o < k; ++o)
        if (t != o) {
          is_catchword = ((float)thr[w + (size_t)t * V] > rho * (float)thr[w + (size_t)o * V]);
          if (!is_catchword) break;
        }
      if (is_catchword) {
        if (catch_topic[w] >= 0) twice = true;
        catch_topic[w] = (int32_t)t;
        ++n;
      }
    }
  return twice ? -1 : n;
}

struct PostModel {
  std::vector<std::tuple<uint64_t, uint64_t, float>> dts;  // (doc, topic, sum) in (doc, topic) order
  std::vector<float> model_threshold;
  std::vector<int32_t> top1, top2;                          // per document, -1 if absent
};

// SparseMatrix::construct_topic_model   src/sparseMatrix.cpp:597-838
// (avg_null_topics is ignored by the live code path; rank_threshold from :720).
void* orc_post_topic_model(uint64_t V, uint64_t D, uint32_t k, const int64_t* offs, const uint32_t* rows, const float* nv,
                           const int32_t* cluster_of, const int32_t* catch_topic, uint64_t rank_threshold, float* Model /*V*k*/) {
  PostModel* pm = new PostModel;
  for (size_t i = 0; i < (size_t)V * k; ++i) Model[i] = 0.f;  // :612
  // :620-633 the (word, topic) list sorted by word == catch_topic[] walked in word order
  std::vector<uint64_t> topic_ncatch(k, 0);
  for (uint64_t w = 0; w < V; ++w)
    if (catch_topic[w] >= 0) topic_ncatch[catch_topic[w]]++;
  // :656-683 document-topic catchword sums (entry order within the document; non-zero sums only, topic ascending)
  std::vector<float> acc(k);
  std::vector<size_t> doc_start(1, 0);
  for (uint64_t d = 0; d < D; ++d) {
    std::fill(acc.begin(), acc.end(), 0.f);
    for (int64_t i = offs[d]; i < offs[d + 1]; ++i) {
      const int32_t t = catch_topic[rows[i]];
      if (t >= 0) acc[t] += nv[i];
    }
    for (uint32_t t = 0; t < k; ++t)
      if (acc[t]) pm->dts.emplace_back(d, t, acc[t]);
    doc_start.push_back(pm->dts.size());
  }
  // :687-708 two heaviest topics of each document
  pm->top1.assign(D, -1);
  pm->top2.assign(D, -1);
  for (uint64_t d = 0; d < D; ++d) {
    float mx = 0.f, mx2 = 0.f;
    int t1 = -1, t2 = -1;
    for (size_t j = doc_start[d]; j < doc_start[d + 1]; ++j) {
      const float v = std::get<2>(pm->dts[j]);
      if (v > mx) {
        mx2 = mx;
        t2 = t1;
        mx = v;
        t1 = (int)std::get<1>(pm->dts[j]);
      } else if (v > mx2) {
        mx2 = v;
        t2 = (int)std::get<1>(pm->dts[j]);
      }
    }
    if (t1 >= 0 && t2 >= 0) {
      pm->top1[d] = t1;
      pm->top2[d] = t2;
    }
  }
  // :713-748 per-topic threshold = rank_threshold-th largest document sum
  std::vector<std::tuple<uint64_t, uint64_t, float>> by_topic(pm->dts);
  std::sort(by_topic.begin(), by_topic.end(), [](const auto& l, const auto& r) {
    return std::get<1>(l) < std::get<1>(r) || (std::get<1>(l) == std::get<1>(r) && std::get<2>(l) > std::get<2>(r));
  });
  pm->model_threshold.assign(k, 0.f);
  {
    size_t it = 0;
    for (uint32_t t = 0; t < k; ++t) {
      while (it < by_topic.size() && std::get<1>(by_topic[it]) < t) ++it;
      size_t end = it;
      while (end < by_topic.size() && std::get<1>(by_topic[end]) == t) ++end;
      if (topic_ncatch[t] > 0 && end - it >= rank_threshold) pm->model_threshold[t] = std::get<2>(by_topic[it + rank_threshold - 1]);
      it = end;
    }
  }
  // :783-810 accumulate documents (ascending): every (doc, topic) whose sum clears the threshold, then the document's
  // own cluster
  for (uint64_t d = 0; d < D; ++d) {
    for (size_t j = doc_start[d]; j < doc_start[d + 1]; ++j) {
      const uint64_t t = std::get<1>(pm->dts[j]);
      if (std::get<2>(pm->dts[j]) > pm->model_threshold[t])
        for (int64_t i = offs[d]; i < offs[d + 1]; ++i) Model[(size_t)t * V + rows[i]] += nv[i];
    }
    if (cluster_of[d] >= 0)
      for (int64_t i = offs[d]; i < offs[d + 1]; ++i) Model[(size_t)cluster_of[d] * V + rows[i]] += nv[i];
  }
  // :816-820 L1 normalisation: FPasum, then FPscal by 1.0 / sum (double reciprocal handed to sscal as float)
  for (uint32_t t = 0; t < k; ++t) {
    float s = 0.f;
    for (uint64_t w = 0; w < V; ++w) s += std::fabs(Model[(size_t)t * V + w]);
    const float a = (float)(1.0 / s);
    for (uint64_t w = 0; w < V; ++w) Model[(size_t)t * V + w] *= a;
  }
  return pm;
}
uint64_t orc_post_dts_size(void* h) { return ((PostModel*)h)->dts.size(); }
void orc_post_dts_get(void* h, uint64_t* doc, uint32_t* topic, float* val, float* model_threshold, int32_t* top1, int32_t* top2) {
  PostModel* pm = (PostModel*)h;
  for (size_t i = 0; i < pm->dts.size(); ++i) {
    doc[i] = std::get<0>(pm->dts[i]);
    topic[i] = (uint32_t)std::get<1>(pm->dts[i]);
    val[i] = std::get<2>(pm->dts[i]);
  }
  std::copy(pm->model_threshold.begin(), pm->model_threshold.end(), model_threshold);
  std::copy(pm->top1.begin(), pm->top1.end(), top1);
  std::copy(pm->top2.begin(), pm->top2.end(), top2);
}
void orc_post_free(void* h) { delete (PostModel*)h; }

// ISLETrainer::construct_edge_topics_v2   src/trainer.cpp:1116-1167.  top1/top2 per document (-1: document has no pair).
// The reference's second sort (by count, descending) is not stable; ties are broken here by (primary, secondary) ascending.
// Returns the number of edge topics; pairs_out holds (primary, secondary, count) triples; Edge is V x n col-major.
uint64_t orc_post_edge_topics(uint64_t V, uint64_t D, const int32_t* top1, const int32_t* top2, int64_t max_edge_topics, int min_docs,
                              const float* Model, float primary_ratio, int64_t* pairs_out /*3 * max*/, float* Edge /*V * max, nullable*/) {
  std::vector<std::tuple<int, int, uint64_t>> tp;
  for (uint64_t d = 0; d < D; ++d)
    if (top1[d] >= 0 && top2[d] >= 0) tp.emplace_back(top1[d], top2[d], d);
  std::sort(tp.begin(), tp.end(), [](const auto& l, const auto& r) {
    return std::get<0>(l) < std::get<0>(r) || (std::get<0>(l) == std::get<0>(r) && std::get<1>(l) < std::get<1>(r));
  });
  std::vector<std::tuple<int, int, int64_t>> sel;
  for (size_t i = 0; i < tp.size();) {
    size_t j = i;
    while (j < tp.size() && std::get<0>(tp[j]) == std::get<0>(tp[i]) && std::get<1>(tp[j]) == std::get<1>(tp[i])) ++j;
    if ((int64_t)(j - i) >= min_docs) sel.emplace_back(std::get<0>(tp[i]), std::get<1>(tp[i]), (int64_t)(j - i));
    i = j;
  }
  std::stable_sort(sel.begin(), sel.end(), [](const auto& l, const auto& r) { return std::get<2>(l) > std::get<2>(r); });
  if ((int64_t)sel.size() > max_edge_topics) sel.resize((size_t)max_edge_topics);  // :1139-1145
  for (size_t e = 0; e < sel.size(); ++e) {
    pairs_out[3 * e] = std::get<0>(sel[e]);
    pairs_out[3 * e + 1] = std::get<1>(sel[e]);
    pairs_out[3 * e + 2] = std::get<2>(sel[e]);
    if (Edge) {
      const float a = primary_ratio, b = (float)(1.0 - (double)primary_ratio);  // FPaxpy twice, :1153-1158
      const float* p = Model + (size_t)std::get<0>(sel[e]) * V;
      const float* q = Model + (size_t)std::get<1>(sel[e]) * V;
      float* o = Edge + e * V;
      for (uint64_t w = 0; w < V; ++w) {
        float y = 0.f;
        y = a * p[w] + y;
        y = b * q[w] + y;
        o[w] = y;
      }
    }
  }
  return sel.size();
}

}  // extern "C"
