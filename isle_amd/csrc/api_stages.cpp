// isle_amd/csrc/api_stages.cpp — the stages either side of the hot path behind the C ABI: count matrix upload and tdf
// ingest, thresholding A -> B, catchwords / topic model / edge topics, inference.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "api_internal.h"

// ------------------------------------------------------------------------------------------
// upstream stage: A -> B on the device
// ------------------------------------------------------------------------------------------
extern "C" int isle_hip_upload_counts_u32(isle_ctx* c, uint64_t V, uint64_t D, uint64_t nnz, const float* counts, const uint32_t* rows,
                                          const int64_t* offs, uint64_t doc_offset, uint64_t docs_global) {
  if (!c) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  if (V == 0 || V > 0xfffffff0ull || D > 0xfffffff0ull) return isle_fail(c, ISLE_E_ARG, "vocab/doc count out of range");
  if (offs[0] != 0 || (uint64_t)offs[D] != nnz) return isle_fail(c, ISLE_E_ARG, "offsets[0] != 0 or offsets[D] != nnz");
  for (uint64_t d = 0; d < D; ++d) {
    if (offs[d + 1] < offs[d]) return isle_fail(c, ISLE_E_ARG, "offsets not monotone at column %llu", (unsigned long long)d);
    for (int64_t i = offs[d]; i < offs[d + 1]; ++i) {
      if (rows[i] >= V) return isle_fail(c, ISLE_E_ARG, "row index out of range at %lld", (long long)i);
      if (i > offs[d] && rows[i] <= rows[i - 1])
        return isle_fail(c, ISLE_E_ARG, "rows not strictly ascending in column %llu", (unsigned long long)d);
      if (!(counts[i] > 0.f)) return isle_fail(c, ISLE_E_ARG, "count not positive at %lld", (long long)i);
    }
  }
  c->a_V = V;
  c->a_D = D;
  c->a_nnz = nnz;
  c->a_doc_offset = doc_offset;
  c->a_D_global = docs_global ? docs_global : D;
  HIPCHK(c, c->a_cnt.reserve(nnz ? nnz : 1));
  HIPCHK(c, c->a_rows.reserve(nnz ? nnz : 1));
  HIPCHK(c, c->a_offs.reserve(D + 1));
  if (nnz) {
    HIPCHK(c, hipMemcpy(c->a_cnt.p, counts, nnz * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->a_rows.p, rows, nnz * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  HIPCHK(c, hipMemcpy(c->a_offs.p, offs, (D + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
  c->a_ready = true;
  c->a_avg_valid = false;
  c->p_catch_ready = false;
  c->p_model_ready = false;
  return 0;
}

extern "C" int isle_hip_ingest_tdf(isle_ctx* c, const char* text, uint64_t nbytes, uint64_t V, uint64_t D, uint64_t max_entries,
                                   uint64_t* entries_read, uint64_t* nnz) {
  if (!c) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  if (c->world > 1) return isle_fail(c, ISLE_E_ARG, "ingest_tdf: single-rank only");
  if (V == 0 || V > 0xfffffff0ull || D == 0 || D > 0xfffffff0ull) return isle_fail(c, ISLE_E_ARG, "ingest_tdf: vocab/doc count out of range");
  if (nbytes && !text) return isle_fail(c, ISLE_E_ARG, "ingest_tdf: null text");
  c->a_ready = false;
  DevBuf<unsigned char> td;
  HIPCHK(c, td.reserve(nbytes + 16));
  hipError_t he = nbytes ? hipMemcpy(td.p, text, nbytes, hipMemcpyHostToDevice) : hipSuccess;
  uint64_t nread = 0, err[2] = {0, 0};
  int rc = 0;
  if (he == hipSuccess) rc = k_ingest_tdf(c, td.p, nbytes, V, D, &nread, err);
  (void)hipStreamSynchronize(c->stream);
  td.release();
  HIPCHK(c, he);
  ISLECHK(rc);
  if (err[0]) {
    static const char* what[] = {"", "bad character", "more than three fields", "fewer than three fields", "doc/word id is 0 or exceeds <num_docs>/<vocab_size>",
                                 "count is 0"};
    return isle_fail(c, ISLE_E_ARG, "ingest_tdf: %s on line %llu", what[err[0] < 6 ? err[0] : 0], (unsigned long long)(err[1] + 1));
  }
  if (max_entries && nread != max_entries)  // include/utils.h:227
    return isle_fail(c, ISLE_E_ARG, "ingest_tdf: file has %llu entries, <max_entries> says %llu", (unsigned long long)nread, (unsigned long long)max_entries);
  c->a_doc_offset = 0;
  c->a_D_global = D;
  c->a_ready = true;
  c->a_avg_valid = false;
  c->p_catch_ready = false;
  c->p_model_ready = false;
  if (entries_read) *entries_read = nread;
  if (nnz) *nnz = c->a_nnz;
  return 0;
}

extern "C" int isle_hip_get_A(isle_ctx* c, float* counts, uint32_t* rows, int64_t* offs) {
  if (!c) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  if (!c->a_ready) return isle_fail(c, ISLE_E_ARG, "get_A: no count matrix");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (counts && c->a_nnz) HIPCHK(c, hipMemcpy(counts, c->a_cnt.p, c->a_nnz * sizeof(float), hipMemcpyDeviceToHost));
  if (rows && c->a_nnz) HIPCHK(c, hipMemcpy(rows, c->a_rows.p, c->a_nnz * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (offs) HIPCHK(c, hipMemcpy(offs, c->a_offs.p, (c->a_D + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int isle_hip_threshold(isle_ctx* c, uint64_t num_topics, double sample_rate, uint64_t sample_seed, uint64_t* docs_kept,
                                  uint64_t* nnz_kept, uint64_t* entries_above, float* avg_out) {
  if (!c) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  if (!c->a_ready) return isle_fail(c, ISLE_E_ARG, "threshold: no count matrix uploaded");
  if (num_topics == 0) return isle_fail(c, ISLE_E_ARG, "threshold: num_topics == 0");
  const bool sampling = sample_rate > 0.0 && sample_rate < 1.0;
  const uint64_t V = c->a_V, D = c->a_D;

  // corpus statistics (src/sparseMatrix.cpp:92-99), global
  HIPCHK(c, c->a_scan.reserve(isle_scan_scratch(D) + 4));
  uint64_t* st_dev = (uint64_t*)c->a_scan.p;
  ISLECHK(k_th_stats(c, st_dev));
  ISLECHK(allreduce_sum<uint64_t>(c, st_dev, 2));
  uint64_t st[2];
  HIPCHK(c, hipMemcpyAsync(st, st_dev, sizeof(st), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const uint64_t tokens = st[0], nz_docs = st[1];
  const float avg = (float)(tokens / std::max<uint64_t>(nz_docs, 1));  // :98, integer division
  if (avg_out) *avg_out = avg;
  c->a_avg = avg;
  c->a_avg_valid = true;
  const uint64_t maxv64 = (uint64_t)avg + 2;
  if (maxv64 > 65535) return isle_fail(c, ISLE_E_ARG, "threshold: average document size %g too large", (double)avg);
  const uint32_t maxv = (uint32_t)maxv64;

  // rounded normalised counts + per-word value histogram, global
  HIPCHK(c, c->a_q.reserve(c->a_nnz ? c->a_nnz : 1));
  HIPCHK(c, c->a_hist.reserve((size_t)V * (maxv + 1)));
  ISLECHK(k_th_round_hist(c, avg, maxv));
  ISLECHK(allreduce_sum<uint32_t>(c, c->a_hist.p, (size_t)V * (maxv + 1)));

  // thresholds  (src/sparseMatrix.cpp:367-368)
  uint64_t count_gr = (uint64_t)(1.0 * (float)nz_docs / (2.0 * (float)num_topics));
  uint64_t count_eq = (uint64_t)std::ceil(3.0 * (1.0 / 60.0) * 1.0 * (float)nz_docs / (float)num_topics);
  if (count_gr == 0) count_gr = 1;
  if (count_eq == 0) count_eq = 1;
  HIPCHK(c, c->zetas.reserve(V));
  ISLECHK(k_th_zetas(c, maxv, count_gr, count_eq));

  // survivors per document
  HIPCHK(c, c->a_kept.reserve(D ? D : 1));
  if (sampling) HIPCHK(c, c->a_wgt.reserve(D ? D : 1));
  ISLECHK(k_th_count(c, sampling));
  ISLECHK(k_th_scans(c));
  int64_t above_local = 0;
  HIPCHK(c, hipMemcpyAsync(&above_local, c->a_off_all.p + D, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (entries_above) {
    uint64_t g = (uint64_t)above_local;
    if (c->multi()) {
      HIPCHK(c, hipMemcpyAsync(st_dev, &g, sizeof(g), hipMemcpyHostToDevice, c->stream));
      ISLECHK(allreduce_sum<uint64_t>(c, st_dev, 1));
      HIPCHK(c, hipMemcpyAsync(&g, st_dev, sizeof(g), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    *entries_above = g;
  }

  if (sampling && (D || c->multi())) {  // sampled_threshold_and_copy, src/sparseMatrix.cpp:1383-1415 (keys on the host, like the reference)
    // Several ranks (round 5): a document's key depends on its GLOBAL number only, the pivot is the (rate x D_global)-th largest key of the
    // whole corpus — every rank gathers all keys (padded to the largest shard) and selects the same pivot; what a shard keeps is what the
    // single-rank run keeps of those documents.
    std::vector<float> wgt(D), key(D), dice(D);
    if (D) HIPCHK(c, hipMemcpy(wgt.data(), c->a_wgt.p, D * sizeof(float), hipMemcpyDeviceToHost));
    for (uint64_t dl = 0; dl < D; ++dl) {
      const uint64_t d = c->a_doc_offset + dl;  // the document's number in the corpus
      uint64_t z = (sample_seed + 1) * 0x9E3779B97F4A7C15ull ^ (d * 0xD1342543DE82EF95ull);
      z += 0x9E3779B97F4A7C15ull;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      z = z ^ (z >> 31);
      const double u = (double)(z >> 11) * (1.0 / 9007199254740992.0);
      key[dl] = (wgt[dl] == 0.f) ? 0.f : (float)std::pow(u, 1.0 / (double)wgt[dl]);
      dice[dl] = key[dl];
    }
    uint64_t Dg = D;
    if (c->multi()) {  // all keys of the corpus on every rank: shards padded with -1 to the largest one
      uint64_t dmax = D;
      HIPCHK(c, hipMemcpyAsync(st_dev, &dmax, sizeof(dmax), hipMemcpyHostToDevice, c->stream));
      ISLECHK(isle_allreduce(c, st_dev, 1, ISLE_DT_U64, true));
      HIPCHK(c, hipMemcpyAsync(&dmax, st_dev, sizeof(dmax), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      if (dmax == 0) dmax = 1;
      DevBuf<float> keys_all;
      HIPCHK(c, keys_all.reserve((size_t)c->world * dmax));
      std::vector<float> mine(dmax, -1.f);
      std::copy(key.begin(), key.end(), mine.begin());
      HIPCHK(c, hipMemcpy(keys_all.p + (size_t)c->rank * dmax, mine.data(), dmax * sizeof(float), hipMemcpyHostToDevice));
      {
        TimeScope ts(c, ISLE_T_COMM);
        ISLECHK(isle_allgather(c, keys_all.p + (size_t)c->rank * dmax, keys_all.p, dmax, ISLE_DT_F32));
      }
      std::vector<float> all((size_t)c->world * dmax);
      HIPCHK(c, hipMemcpyAsync(all.data(), keys_all.p, all.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      keys_all.release();
      dice.clear();
      for (float v : all)
        if (v >= 0.f) dice.push_back(v);
      Dg = dice.size();
    }
    float pivot = 2.f;  // (an empty corpus keeps nothing)
    if (Dg) {
      const size_t nth = std::min<size_t>((size_t)((float)sample_rate * (float)Dg), Dg - 1);
      std::nth_element(dice.begin(), dice.begin() + nth, dice.end(), std::greater<float>());
      pivot = dice[nth];
    }
    std::vector<uint8_t> drop(D ? D : 1);
    for (uint64_t d = 0; d < D; ++d) drop[d] = !(key[d] >= pivot);
    DevBuf<uint8_t> drop_dev;
    HIPCHK(c, drop_dev.reserve(D ? D : 1));
    if (D) HIPCHK(c, hipMemcpy(drop_dev.p, drop.data(), D, hipMemcpyHostToDevice));
    int rc = D ? k_th_drop(c, drop_dev.p) : 0;
    if (rc == 0) rc = k_th_scans(c);
    (void)hipStreamSynchronize(c->stream);
    drop_dev.release();
    ISLECHK(rc);
  }

  int64_t tail[2];  // nnz(B), columns of B (local)
  HIPCHK(c, hipMemcpyAsync(&tail[0], c->a_off_all.p + D, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&tail[1], c->a_col_of.p + D, sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const uint64_t bnnz = (uint64_t)tail[0], Db = (uint64_t)tail[1];

  // placement of this shard in B's global column numbering
  uint64_t b_off = 0, b_glob = Db;
  if (c->multi()) {
    DevBuf<uint64_t> all;
    HIPCHK(c, all.reserve((size_t)c->world + 1));
    HIPCHK(c, hipMemcpyAsync(all.p + c->world, &Db, sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    {
      TimeScope ts(c, ISLE_T_COMM);
      ISLECHK(isle_allgather(c, all.p + c->world, all.p, 1, ISLE_DT_U64));
    }
    std::vector<uint64_t> h(c->world);
    HIPCHK(c, hipMemcpyAsync(h.data(), all.p, c->world * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    all.release();
    b_glob = 0;
    for (int r = 0; r < c->world; ++r) {
      if (r == c->rank) b_off = b_glob;
      b_glob += h[r];
    }
  }

  isle_trim_derived(c, Db, bnnz);
  c->V = V;
  c->D = Db;
  c->nnz = bnnz;
  c->doc_offset = b_off;
  c->D_global = b_glob;
  HIPCHK(c, c->vals.reserve(bnnz ? bnnz : 1));
  HIPCHK(c, c->rows.reserve(bnnz ? bnnz : 1));
  HIPCHK(c, c->offs.reserve(Db + 1));
  HIPCHK(c, c->original_cols.reserve(Db ? Db : 1));
  if (D == 0) HIPCHK(c, hipMemsetAsync(c->offs.p, 0, sizeof(int64_t), c->stream));
  ISLECHK(k_th_emit(c, c->a_doc_offset));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->band_ready = false;
  c->gl_mode = -1;
  c->P_ready = false;
  c->Pt_ready = false;
  c->Pt2_ready = false;
  c->lift_valid = false;
  c->members_valid = false;
  c->U_k = 0;
  c->centers_ready = false;
  c->assign_valid = false;
  c->p_catch_ready = false;
  c->p_model_ready = false;
  c->b_from_threshold = true;
  if (docs_kept) *docs_kept = Db;
  if (nnz_kept) *nnz_kept = bnnz;
  return 0;
}

extern "C" int isle_hip_shape(isle_ctx* c, uint64_t* V, uint64_t* D, uint64_t* nnz, uint64_t* doc_offset, uint64_t* docs_global) {
  if (!c) return ISLE_E_ARG;
  if (V) *V = c->V;
  if (D) *D = c->D;
  if (nnz) *nnz = c->nnz;
  if (doc_offset) *doc_offset = c->doc_offset;
  if (docs_global) *docs_global = c->D_global;
  return 0;
}

extern "C" int isle_hip_get_B(isle_ctx* c, float* vals, uint32_t* rows, int64_t* offs, uint64_t* original_cols, float* zetas) {
  if (!c) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  if (c->V == 0) return isle_fail(c, ISLE_E_ARG, "get_B: no matrix");
  if ((original_cols || zetas) && !c->b_from_threshold)
    return isle_fail(c, ISLE_E_ARG, "get_B: original_cols / zetas exist only after isle_hip_threshold");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (vals && c->nnz) HIPCHK(c, hipMemcpy(vals, c->vals.p, c->nnz * sizeof(float), hipMemcpyDeviceToHost));
  if (rows && c->nnz) HIPCHK(c, hipMemcpy(rows, c->rows.p, c->nnz * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (offs) HIPCHK(c, hipMemcpy(offs, c->offs.p, (c->D + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
  if (original_cols && c->D) HIPCHK(c, hipMemcpy(original_cols, c->original_cols.p, c->D * sizeof(uint64_t), hipMemcpyDeviceToHost));
  if (zetas) HIPCHK(c, hipMemcpy(zetas, c->zetas.p, c->V * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

// ------------------------------------------------------------------------------------------
// downstream stage: catchwords, topic model, edge topics
// ------------------------------------------------------------------------------------------
static int post_prepare(isle_ctx* c, const char* who) {
  if (!c->a_ready) return isle_fail(c, ISLE_E_ARG, "%s: no count matrix uploaded (isle_hip_upload_counts_u32)", who);
  if (c->world > 1) return isle_fail(c, ISLE_E_ARG, "%s: single-rank only", who);
  return 0;
}

extern "C" int isle_hip_catchwords(isle_ctx* c, int num_topics, const uint32_t* assign, uint64_t r, double rho, float* thresholds,
                                   int32_t* catch_topic, uint64_t* num_catchwords) {
  if (!c) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  ISLECHK(post_prepare(c, "catchwords"));
  if (num_topics < 1) return isle_fail(c, ISLE_E_ARG, "catchwords: num_topics < 1");
  if (r < 1 || r > 0xfffffff0ull) return isle_fail(c, ISLE_E_ARG, "catchwords: rank r = %llu out of range (too few documents per topic?)",
                                                    (unsigned long long)r);
  const bool identity = !c->b_from_threshold;
  if (identity && c->D != c->a_D) return isle_fail(c, ISLE_E_ARG, "catchwords: B was uploaded separately and its columns do not match A's");
  if (assign) {
    for (uint64_t j = 0; j < c->D; ++j)
      if (assign[j] >= (uint32_t)num_topics) return isle_fail(c, ISLE_E_ARG, "catchwords: assign[%llu] out of range", (unsigned long long)j);
    HIPCHK(c, c->assign.reserve(c->D ? c->D : 1));
    if (c->D) HIPCHK(c, hipMemcpy(c->assign.p, assign, c->D * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->assign_valid = true;
    c->members_valid = false;
  } else if (!c->assign_valid) {
    return isle_fail(c, ISLE_E_ARG, "catchwords: no partition resident (run isle_hip_lloyds_sparse or pass assign)");
  }
  if (!c->a_avg_valid) {  // B came from the host: the corpus statistics were never computed here
    HIPCHK(c, c->a_scan.reserve(isle_scan_scratch(c->a_D) + 4));
    ISLECHK(k_th_stats(c, (uint64_t*)c->a_scan.p));
    uint64_t st[2];
    HIPCHK(c, hipMemcpyAsync(st, c->a_scan.p, sizeof(st), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->a_avg = (float)(st[0] / std::max<uint64_t>(st[1], 1));
    c->a_avg_valid = true;
  }
  ISLECHK(k_post_normalize(c, c->a_avg));
  ISLECHK(k_post_cluster_of(c, c->assign.p, identity));
  HIPCHK(c, c->counts.reserve(num_topics));
  ISLECHK(k_count_sizes(c, c->assign.p, c->D, num_topics, c->counts.p));
  ISLECHK(k_post_catch_thresholds(c, (uint32_t)num_topics, (uint32_t)r, c->counts.p));
  uint64_t nc = 0;
  ISLECHK(k_post_find_catchwords(c, (uint32_t)num_topics, rho, &nc));
  if (num_catchwords) *num_catchwords = nc;
  c->p_k = num_topics;
  c->p_catch_ready = true;
  c->p_model_ready = false;
  if (thresholds) {
    HIPCHK(c, c->p_segvals.reserve((size_t)c->a_V * num_topics));
    ISLECHK(k_post_thr_colmajor(c, (uint32_t)num_topics, c->p_segvals.p));
    HIPCHK(c, hipMemcpyAsync(thresholds, c->p_segvals.p, (size_t)c->a_V * num_topics * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  }
  if (catch_topic) HIPCHK(c, hipMemcpyAsync(catch_topic, c->p_catch.p, c->a_V * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int isle_hip_topic_model(isle_ctx* c, int num_topics, uint64_t rank_threshold, float* model, float* model_threshold, int32_t* top1,
                                    int32_t* top2, uint64_t* doc_topic_sums) {
  if (!c) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  ISLECHK(post_prepare(c, "topic_model"));
  if (!c->p_catch_ready || c->p_k != num_topics) return isle_fail(c, ISLE_E_ARG, "topic_model: run isle_hip_catchwords(num_topics = %d) first", num_topics);
  if (rank_threshold < 1 || rank_threshold > 0xfffffff0ull) return isle_fail(c, ISLE_E_ARG, "topic_model: rank_threshold out of range");  // :721
  uint64_t n = 0;
  ISLECHK(k_post_doc_topic_sums(c, (uint32_t)num_topics, &n));
  ISLECHK(k_post_model_thresholds(c, (uint32_t)num_topics, (uint32_t)rank_threshold));
  ISLECHK(k_post_model(c, (uint32_t)num_topics));
  c->p_model_ready = true;
  if (doc_topic_sums) *doc_topic_sums = n;
  if (model) HIPCHK(c, hipMemcpyAsync(model, c->p_model.p, (size_t)c->a_V * num_topics * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  if (model_threshold) HIPCHK(c, hipMemcpyAsync(model_threshold, c->p_mthr.p, num_topics * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  if (top1 && c->a_D) HIPCHK(c, hipMemcpyAsync(top1, c->p_top1.p, c->a_D * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  if (top2 && c->a_D) HIPCHK(c, hipMemcpyAsync(top2, c->p_top2.p, c->a_D * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int isle_hip_get_doc_topic_sums(isle_ctx* c, int64_t* doc_offsets, uint32_t* topic, float* val) {
  if (!c) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  if (!c->p_model_ready) return isle_fail(c, ISLE_E_ARG, "get_doc_topic_sums: run isle_hip_topic_model first");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (doc_offsets) HIPCHK(c, hipMemcpy(doc_offsets, c->p_dts_off.p, (c->a_D + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
  if (topic && c->p_dts_n) HIPCHK(c, hipMemcpy(topic, c->p_dts_topic.p, c->p_dts_n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (val && c->p_dts_n) HIPCHK(c, hipMemcpy(val, c->p_dts_val.p, c->p_dts_n * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int isle_hip_edge_topics(isle_ctx* c, const int64_t* pairs, int n, float primary_ratio, float* edge) {
  if (!c) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  if (!c->p_model_ready) return isle_fail(c, ISLE_E_ARG, "edge_topics: run isle_hip_topic_model first");
  if (n < 0 || (n && (!pairs || !edge))) return isle_fail(c, ISLE_E_ARG, "edge_topics: bad arguments");
  if (n == 0) return 0;
  for (int e = 0; e < 2 * n; ++e)
    if (pairs[e] < 0 || pairs[e] >= c->p_k) return isle_fail(c, ISLE_E_ARG, "edge_topics: topic id %lld out of range", (long long)pairs[e]);
  DevBuf<int64_t> pd;
  DevBuf<float> ed;
  HIPCHK(c, pd.reserve(2 * (size_t)n));
  hipError_t e1 = ed.reserve((size_t)c->a_V * n);
  if (e1 != hipSuccess) {
    pd.release();
    HIPCHK(c, e1);
  }
  int rc = 0;
  hipError_t he = hipMemcpy(pd.p, pairs, 2 * (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice);
  if (he == hipSuccess) rc = k_post_edge(c, pd.p, n, primary_ratio, (float)(1.0 - (double)primary_ratio), ed.p);
  if (he == hipSuccess && rc == 0) he = hipStreamSynchronize(c->stream);
  if (he == hipSuccess && rc == 0) he = hipMemcpy(edge, ed.p, (size_t)c->a_V * n * sizeof(float), hipMemcpyDeviceToHost);
  pd.release();
  ed.release();
  ISLECHK(rc);
  HIPCHK(c, he);
  return 0;
}

extern "C" int isle_hip_infer(isle_ctx* c, uint64_t V, int k, const float* model_by_word, uint64_t D, uint64_t nnz, const float* counts,
                              const uint32_t* rows, const int64_t* offs, int iters, float Lf, float avg_doc_sz, float* weights,
                              int32_t* top_topic, float* top_weight, float* llh, uint64_t* nconverged) {
  if (!c || !model_by_word || !offs || (nnz && (!counts || !rows))) return ISLE_E_ARG;
  if (iters < 1 || !(Lf > 0.f)) return isle_fail(c, ISLE_E_ARG, "infer: iters = %d, Lf = %g", iters, (double)Lf);
  if (offs[0] != 0 || (uint64_t)offs[D] != nnz) return isle_fail(c, ISLE_E_ARG, "infer: offsets do not span the %llu entries", (unsigned long long)nnz);
  ISLECHK(isle_enter(c));
  return k_infer(c, V, k, model_by_word, D, nnz, counts, rows, offs, iters, Lf, avg_doc_sz, weights, top_topic, top_weight, llh, nconverged);
}

