// isle_amd/host/prestage.h — host stages immediately upstream of the hot path, restated from the reference so that the
// ISLETrain CLI can hand a genuine B to the GPU (host C++ for now, no GPU kernels yet):
//   tdf reader                      include/utils.h:158-228   (DocWordEntriesReader::fill_doc_word_entries)
//   sort + de-duplicate             src/trainer.cpp:237-247
//   populate_CSC, avg_doc_sz        src/sparseMatrix.cpp:58-107
//   normalize_docs                  src/sparseMatrix.cpp:136-167
//   compute_thresholds              src/sparseMatrix.cpp:357-485   (FPTYPE branch; the per-word descending frequency
//                                   list of :289-333 is replaced by a per-word histogram of the rounded values)
//   threshold_and_copy(_doc_block)  src/sparseMatrix.cpp:1285-1361
//   sampled_threshold_and_copy      src/sparseMatrix.cpp:1365-1435 (seeded RNG instead of unseeded rand())
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

namespace ISLE {
namespace prestage {

struct DocWordEntry {
  uint64_t doc, word;
  uint32_t count;
};

// tdf text: one "<doc> <word> <count>" per line, 1-based ids, any mix of blanks/tabs, optional '\r', last newline optional.
inline void read_tdf(const std::string& path, uint64_t max_entries, std::vector<DocWordEntry>& entries) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("cannot open tdf file " + path);
  std::fseek(f, 0, SEEK_END);
  const long sz = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<char> buf((size_t)sz);
  if (sz && std::fread(buf.data(), 1, (size_t)sz, f) != (size_t)sz) {
    std::fclose(f);
    throw std::runtime_error("short read on " + path);
  }
  std::fclose(f);
  entries.clear();
  entries.reserve(max_entries);
  uint64_t doc = 0, word = 0, count = 0;
  int state = 1;
  bool was_ws = false, any = false;
  for (long i = 0; i < sz; ++i) {
    const char ch = buf[(size_t)i];
    switch (ch) {
      case '\r': break;
      case '\n':
        if (any && state == 3 && count == 0) throw std::runtime_error("tdf file: count is 0 (entry " + std::to_string(entries.size() + 1) + ")");
        if (any) entries.push_back({doc - 1, word - 1, (uint32_t)count});
        doc = word = count = 0;
        state = 1;
        was_ws = false;
        any = false;
        break;
      case ' ':
      case '\t': was_ws = true; break;
      default:
        if (ch < '0' || ch > '9') throw std::runtime_error("Bad format in tdf file");
        if (was_ws && any) {
          state++;
          was_ws = false;
        }
        was_ws = false;
        any = true;
        if (state == 1) doc = doc * 10 + (uint64_t)(ch - '0');
        else if (state == 2) word = word * 10 + (uint64_t)(ch - '0');
        else if (state == 3) count = count * 10 + (uint64_t)(ch - '0');
        else throw std::runtime_error("Bad line in tdf file");
    }
  }
  if (any && state == 3) entries.push_back({doc - 1, word - 1, (uint32_t)count});  // no trailing newline
  if (entries.size() != max_entries)  // include/utils.h:227 assert(nRead == max_entries)
    throw std::runtime_error("tdf file has " + std::to_string(entries.size()) + " entries, <max_entries> says " + std::to_string(max_entries));
}

struct Csc {
  uint64_t V = 0, D = 0;
  std::vector<float> vals;
  std::vector<uint64_t> rows;  // the reference's 8-byte word_id_t
  std::vector<int64_t> offs;
};

// sort by (doc, word), drop duplicates, build CSC of raw counts; returns avg_doc_sz and nz_docs like populate_CSC
inline void build_A(std::vector<DocWordEntry>& entries, uint64_t V, uint64_t D, Csc& A, float* avg_doc_sz, uint64_t* nz_docs) {
  std::sort(entries.begin(), entries.end(),
            [](const DocWordEntry& l, const DocWordEntry& r) { return (l.doc < r.doc) || (l.doc == r.doc && l.word < r.word); });
  entries.erase(std::unique(entries.begin(), entries.end(),
                            [](const DocWordEntry& l, const DocWordEntry& r) { return l.doc == r.doc && l.word == r.word; }),
                entries.end());
  if (!entries.empty() && (entries.back().doc >= D)) throw std::runtime_error("doc id exceeds <num_docs>");
  A.V = V;
  A.D = D;
  A.offs.assign(D + 1, 0);
  A.vals.resize(entries.size());
  A.rows.resize(entries.size());
  uint64_t tokens = 0;
  for (size_t i = 0; i < entries.size(); ++i) {
    if (entries[i].word >= V) throw std::runtime_error("word id exceeds <vocab_size>");
    A.vals[i] = (float)entries[i].count;
    A.rows[i] = entries[i].word;
    A.offs[entries[i].doc + 1]++;
    tokens += entries[i].count;
  }
  uint64_t nz = 0;
  for (uint64_t d = 0; d < D; ++d) {
    nz += A.offs[d + 1] > 0;
    A.offs[d + 1] += A.offs[d];
  }
  *nz_docs = nz;
  *avg_doc_sz = (float)(tokens / std::max<uint64_t>(nz, 1));  // src/sparseMatrix.cpp:98 (integer division)
}

struct Thresholded {
  Csc B;
  std::vector<uint64_t> original_cols;
  std::vector<float> zetas;
  uint64_t entries_above_threshold = 0;
};

// normalize_docs + compute_thresholds + (sampled_)threshold_and_copy.  sample_rate <= 0: no sampling.
inline void threshold(const Csc& A, float avg_doc_sz, uint64_t nz_docs, uint64_t num_topics, double sample_rate, uint64_t sample_seed,
                      Thresholded& out) {
  const uint64_t V = A.V, D = A.D, nnz = (uint64_t)A.offs[D];
  std::vector<float> rnd(nnz);
#pragma omp parallel for schedule(dynamic, 4096)
  for (int64_t d = 0; d < (int64_t)D; ++d) {
    float sum = 0.f;
    for (int64_t i = A.offs[d]; i < A.offs[d + 1]; ++i) sum += A.vals[i];
    for (int64_t i = A.offs[d]; i < A.offs[d + 1]; ++i) rnd[i] = std::round(avg_doc_sz * (A.vals[i] / sum));  // :158, :371
  }
  const uint32_t maxv = (uint32_t)avg_doc_sz + 2;
  std::vector<uint32_t> hist((size_t)V * (maxv + 1), 0);
  for (uint64_t i = 0; i < nnz; ++i) {
    const uint32_t v = (uint32_t)std::min<float>(rnd[i], (float)maxv);
    if (v > 0) hist[(size_t)A.rows[i] * (maxv + 1) + v]++;
  }
  uint64_t count_gr = (uint64_t)(1.0 * (float)nz_docs / (2.0 * (float)num_topics));                        // :367
  uint64_t count_eq = (uint64_t)std::ceil(3.0 * (1.0 / 60.0) * 1.0 * (float)nz_docs / (float)num_topics);  // :368
  if (count_gr == 0) count_gr = 1;
  if (count_eq == 0) count_eq = 1;
  out.zetas.assign(V, 1.0f);
  uint64_t new_nnzs = 0;
  for (uint64_t w = 0; w < V; ++w) {
    const uint32_t* hw = &hist[(size_t)w * (maxv + 1)];
    uint64_t size = 0;
    for (uint32_t v = 1; v <= maxv; ++v) size += hw[v];
    if (size == 0) continue;  // :477-480
    if (count_gr > size) {    // :399-411
      new_nnzs += size;
      continue;
    }
    uint32_t zeta = maxv;
    uint64_t cum = 0;
    for (uint32_t v = maxv; v >= 1; --v) {
      cum += hw[v];
      if (cum >= count_gr) {
        zeta = v;
        break;
      }
    }
    while (true) {  // :445-470
      if (hw[zeta] < count_eq) {
        out.zetas[w] = (float)zeta;
        uint64_t ge = 0;
        for (uint32_t v = zeta; v <= maxv; ++v) ge += hw[v];
        new_nnzs += ge;
        break;
      }
      uint32_t nxt = 0;
      for (uint32_t v = zeta; v-- > 1;)
        if (hw[v] > 0) {
          nxt = v;
          break;
        }
      if (nxt == 0 || zeta == 1) {
        out.zetas[w] = 1.0f;
        new_nnzs += size;
        break;
      }
      zeta = nxt;
    }
  }
  out.entries_above_threshold = new_nnzs;
  std::vector<int64_t> cnt(D, 0);
#pragma omp parallel for schedule(dynamic, 4096)
  for (int64_t d = 0; d < (int64_t)D; ++d) {
    int64_t n = 0;
    for (int64_t i = A.offs[d]; i < A.offs[d + 1]; ++i) n += (rnd[i] >= out.zetas[A.rows[i]]);
    cnt[d] = n;
  }
  if (sample_rate > 0.0 && sample_rate < 1.0) {
    std::vector<float> key(D), dice(D);
    for (uint64_t d = 0; d < D; ++d) {
      float wgt = 0.f;
      for (int64_t i = A.offs[d]; i < A.offs[d + 1]; ++i)
        if (rnd[i] >= out.zetas[A.rows[i]]) wgt += out.zetas[A.rows[i]];  // :1393-1394
      uint64_t z = (sample_seed + 1) * 0x9E3779B97F4A7C15ull ^ (d * 0xD1342543DE82EF95ull);
      z += 0x9E3779B97F4A7C15ull;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      z = z ^ (z >> 31);
      const double u = (double)(z >> 11) * (1.0 / 9007199254740992.0);
      key[d] = (wgt == 0.f) ? 0.f : (float)std::pow(u, 1.0 / (double)wgt);  // :1401-1403
      dice[d] = key[d];
    }
    const size_t nth = std::min<size_t>((size_t)((float)sample_rate * (float)D), D - 1);  // :1406-1409
    std::nth_element(dice.begin(), dice.begin() + nth, dice.end(), std::greater<float>());
    const float pivot = dice[nth];
    std::printf("sampling docs: pivot: %g\n", pivot);
    for (uint64_t d = 0; d < D; ++d)
      if (!(key[d] >= pivot)) cnt[d] = 0;
  }
  out.original_cols.clear();
  out.B.V = V;
  out.B.offs.assign(1, 0);
  for (uint64_t d = 0; d < D; ++d)
    if (cnt[d] > 0) {
      out.original_cols.push_back(d);
      out.B.offs.push_back(out.B.offs.back() + cnt[d]);
    }
  out.B.D = out.original_cols.size();
  out.B.vals.resize((size_t)out.B.offs.back());
  out.B.rows.resize((size_t)out.B.offs.back());
#pragma omp parallel for schedule(dynamic, 4096)
  for (int64_t j = 0; j < (int64_t)out.B.D; ++j) {
    const uint64_t d = out.original_cols[j];
    int64_t p = out.B.offs[j];
    for (int64_t i = A.offs[d]; i < A.offs[d + 1]; ++i) {
      const float z = out.zetas[A.rows[i]];
      if (rnd[i] >= z) {
        out.B.vals[p] = std::sqrt(z);  // :1347
        out.B.rows[p] = A.rows[i];
        ++p;
      }
    }
  }
}

}  // namespace prestage
}  // namespace ISLE
