// tools/synth_corpus.cpp — BENCH/TEST INPUT GENERATOR (not product, not oracle).
//
// Planted-topic Zipf corpus of SURVEY.md App. D, generated directly as CSC, followed by
// the pre-stages that sit upstream of the hot path in the reference so that the matrix
// handed to the hot path is a genuine ISLE "B":
//   populate_CSC avg_doc_sz           src/sparseMatrix.cpp:58-107   (avg = tokens / nz_docs, integer division)
//   normalize_docs                    src/sparseMatrix.cpp:136-167  (a = avg * (count / doc_sum))
//   compute_thresholds                src/sparseMatrix.cpp:357-485  (zeta_w rule, FPTYPE branch)
//   threshold_and_copy(_doc_block)    src/sparseMatrix.cpp:1285-1361 (B = sqrt(zeta_w) where round(a) >= zeta_w; empty cols dropped)
// The per-word descending frequency list of the reference (list_word_freqs_by_sorting,
// :289-333) is replaced by a per-word histogram of the rounded values — same multiset.
#include <algorithm>
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
struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed) {}
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double u01() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  double normal() {
    double u1 = u01(), u2 = u01();
    if (u1 < 1e-300) u1 = 1e-300;
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
  }
};

struct Corpus {
  uint64_t V = 0, D = 0;
  std::vector<float> counts;  // raw counts (A)
  std::vector<uint32_t> rows;
  std::vector<int64_t> offs;
  std::vector<uint32_t> dom;  // planted dominant topic per doc
  // thresholded B
  uint64_t Db = 0;
  std::vector<float> bvals;
  std::vector<uint32_t> brows;
  std::vector<int64_t> boffs;
  std::vector<uint64_t> original_cols;
  std::vector<float> zetas;
};
}  // namespace

extern "C" {

void* synth_generate(uint64_t V, uint64_t D, uint32_t K, double zipf_s, double L0, double dom_w, uint64_t seed, uint64_t doc_base) {
  Corpus* c = new Corpus;
  c->V = V;
  c->D = D;
  std::vector<double> cdf(V);
  {
    double s = 0;
    for (uint64_t i = 0; i < V; ++i) {
      s += 1.0 / std::pow((double)(i + 1), zipf_s);
      cdf[i] = s;
    }
    for (uint64_t i = 0; i < V; ++i) cdf[i] /= s;
  }
  // topic t = the same Zipf law over its own word order
  std::vector<uint32_t> perm((size_t)K * V);
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t t = 0; t < (int64_t)K; ++t) {
    Rng r((seed * 0x100000001B3ull + 0xABCDEF) ^ ((uint64_t)(t + 1) * 0x9E3779B97F4A7C15ull));
    uint32_t* p = &perm[(size_t)t * V];
    for (uint64_t i = 0; i < V; ++i) p[i] = (uint32_t)i;
    for (uint64_t i = V - 1; i > 0; --i) {
      uint64_t j = r.next() % (i + 1);
      std::swap(p[i], p[j]);
    }
  }
  c->offs.assign(D + 1, 0);
  c->dom.resize(D);
  const uint64_t CH = 4096;
  const uint64_t nch = (D + CH - 1) / CH;
  std::vector<std::vector<uint32_t>> crow(nch);
  std::vector<std::vector<float>> ccnt(nch);
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t ch = 0; ch < (int64_t)nch; ++ch) {
    std::vector<uint32_t> words;
    auto& rr = crow[ch];
    auto& cc = ccnt[ch];
    for (uint64_t d = ch * CH; d < std::min(D, (uint64_t)(ch + 1) * CH); ++d) {
      Rng r((seed + 77) * 0xD1342543DE82EF95ull + (d + doc_base) * 0x9E3779B97F4A7C15ull);
      double L = std::exp(std::log(L0) + 0.4 * r.normal());
      L = std::min(2000.0, std::max(30.0, L));
      const uint32_t len = (uint32_t)L;
      const uint32_t dom = (uint32_t)(r.next() % K);
      c->dom[d] = dom;
      words.resize(len);
      for (uint32_t i = 0; i < len; ++i) {
        const uint32_t t = (r.u01() < dom_w) ? dom : (uint32_t)(r.next() % K);
        const double u = r.u01();
        uint64_t rank = std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin();
        if (rank >= V) rank = V - 1;
        words[i] = perm[(size_t)t * V + rank];
      }
      std::sort(words.begin(), words.end());
      uint32_t n = 0;
      for (uint32_t i = 0; i < len;) {
        uint32_t j = i;
        while (j < len && words[j] == words[i]) ++j;
        rr.push_back(words[i]);
        cc.push_back((float)(j - i));
        ++n;
        i = j;
      }
      c->offs[d + 1] = n;
    }
  }
  for (uint64_t d = 0; d < D; ++d) c->offs[d + 1] += c->offs[d];
  const uint64_t nnz = c->offs[D];
  c->rows.resize(nnz);
  c->counts.resize(nnz);
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t ch = 0; ch < (int64_t)nch; ++ch) {
    const int64_t base = c->offs[ch * CH];
    std::memcpy(&c->rows[base], crow[ch].data(), crow[ch].size() * 4);
    std::memcpy(&c->counts[base], ccnt[ch].data(), ccnt[ch].size() * 4);
  }
  return c;
}

// Build a corpus from caller-supplied CSC counts (tests; a tdf reader can feed this too).
void* synth_from_csc(uint64_t V, uint64_t D, const float* counts, const uint32_t* rows, const int64_t* offs) {
  Corpus* c = new Corpus;
  c->V = V;
  c->D = D;
  c->offs.assign(offs, offs + D + 1);
  c->rows.assign(rows, rows + offs[D]);
  c->counts.assign(counts, counts + offs[D]);
  c->dom.assign(D, 0);
  return c;
}

// ---- thresholding, in phases so that a column-sharded corpus can all-reduce the global statistics ----
struct ThreshState {
  std::vector<float> rnd;
  std::vector<uint32_t> hist;
  uint32_t maxv = 0;
};
static ThreshState g_ts;  // one corpus at a time per process (bench/test tool)

// phase A: local token count and non-empty docs
void synth_stats(void* h, uint64_t* tokens, uint64_t* nz_docs) {
  Corpus* c = (Corpus*)h;
  const uint64_t D = c->D, nnz = c->offs[D];
  uint64_t t = 0, nz = 0;
  for (uint64_t i = 0; i < nnz; ++i) t += (uint64_t)c->counts[i];
  for (uint64_t d = 0; d < D; ++d) nz += (c->offs[d + 1] > c->offs[d]);
  *tokens = t;
  *nz_docs = nz;
}

// phase B: rounded normalised counts with the GLOBAL avg_doc_sz; returns the local per-word histogram
// (V x (maxv+1) uint32) for the caller to all-reduce in place.
uint32_t* synth_hist(void* h, uint64_t tokens_global, uint64_t nzdocs_global, uint32_t* maxv_out) {
  Corpus* c = (Corpus*)h;
  const uint64_t V = c->V, D = c->D, nnz = c->offs[D];
  const float avg = (float)(tokens_global / std::max<uint64_t>(nzdocs_global, 1));  // src/sparseMatrix.cpp:98
  g_ts.rnd.resize(nnz);
  std::vector<float>& rnd = g_ts.rnd;
#pragma omp parallel for schedule(dynamic, 4096)
  for (int64_t d = 0; d < (int64_t)D; ++d) {
    float sum = 0.f;
    for (int64_t i = c->offs[d]; i < c->offs[d + 1]; ++i) sum += c->counts[i];
    for (int64_t i = c->offs[d]; i < c->offs[d + 1]; ++i)
      rnd[i] = std::round(avg * (c->counts[i] / sum));  // :158 then :1345 / :371
  }
  const uint32_t maxv = (uint32_t)avg + 2;
  g_ts.maxv = maxv;
  g_ts.hist.assign((size_t)V * (maxv + 1), 0);
  for (uint64_t i = 0; i < nnz; ++i) {
    uint32_t v = (uint32_t)std::min<float>(rnd[i], (float)maxv);
    if (v > 0) g_ts.hist[(size_t)c->rows[i] * (maxv + 1) + v]++;
  }
  *maxv_out = maxv;
  return g_ts.hist.data();
}

// phase C: zetas from the (global) histogram, then B.  Returns nnz(B).
// sample_rate in (0,1): importance sampling of documents (sampled_threshold_and_copy, src/sparseMatrix.cpp:1365-1435):
// weight w_d = sum of zeta over the surviving entries, key = u^(1/w_d), keep keys >= the floor(rate*D)-th largest.
// (single-process only: the pivot is not all-reduced across shards)
uint64_t synth_apply(void* h, uint32_t k, uint64_t nzdocs_global, double sample_rate, uint64_t sample_seed) {
  Corpus* c = (Corpus*)h;
  const uint64_t V = c->V, D = c->D;
  const uint64_t nz_docs = nzdocs_global;
  const uint32_t maxv = g_ts.maxv;
  const std::vector<uint32_t>& hist = g_ts.hist;
  const std::vector<float>& rnd = g_ts.rnd;
  uint64_t count_gr = (uint64_t)(1.0 * (float)nz_docs / (2.0 * (float)k));               // :367
  uint64_t count_eq = (uint64_t)std::ceil(3.0 * (1.0 / 60.0) * 1.0 * (float)nz_docs / (float)k);  // :368
  if (count_gr == 0) count_gr = 1;
  if (count_eq == 0) count_eq = 1;
  c->zetas.assign(V, 1.0f);
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t w = 0; w < (int64_t)V; ++w) {
    const uint32_t* hw = &hist[(size_t)w * (maxv + 1)];
    uint64_t size = 0;
    for (uint32_t v = 1; v <= maxv; ++v) size += hw[v];
    if (size == 0 || count_gr > size) {
      c->zetas[w] = 1.0f;  // :399-411, :479
      continue;
    }
    // zeta = freqs[count_gr - 1] in the descending list
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
        c->zetas[w] = (float)zeta;
        break;
      }
      // next lower distinct value present
      uint32_t nxt = 0;
      for (uint32_t v = zeta; v-- > 1;)
        if (hw[v] > 0) {
          nxt = v;
          break;
        }
      if (nxt == 0 || zeta == 1) {
        c->zetas[w] = 1.0f;
        break;
      }
      zeta = nxt;
    }
  }
  // threshold_and_copy
  std::vector<int64_t> cnt(D + 1, 0);
#pragma omp parallel for schedule(dynamic, 4096)
  for (int64_t d = 0; d < (int64_t)D; ++d) {
    int64_t n = 0;
    for (int64_t i = c->offs[d]; i < c->offs[d + 1]; ++i) n += (rnd[i] >= c->zetas[c->rows[i]]);
    cnt[d + 1] = n;
  }
  if (sample_rate > 0.0 && sample_rate < 1.0) {
    std::vector<float> key(D), dice(D);
#pragma omp parallel for schedule(dynamic, 4096)
    for (int64_t d = 0; d < (int64_t)D; ++d) {
      float w = 0.f;
      for (int64_t i = c->offs[d]; i < c->offs[d + 1]; ++i)
        if (rnd[i] >= c->zetas[c->rows[i]]) w += c->zetas[c->rows[i]];                  // :1393-1394
      Rng r((sample_seed + 1) * 0x9E3779B97F4A7C15ull ^ ((uint64_t)d * 0xD1342543DE82EF95ull));
      key[d] = (w == 0.f) ? 0.f : (float)std::pow(r.u01(), 1.0 / (double)w);             // :1401-1403
      dice[d] = key[d];
    }
    const size_t nth = std::min<size_t>((size_t)((float)sample_rate * (float)D), D - 1);  // :1406-1409 (FPTYPE product)
    std::nth_element(dice.begin(), dice.begin() + nth, dice.end(), std::greater<float>());
    const float pivot = dice[nth];
    for (uint64_t d = 0; d < D; ++d)
      if (!(key[d] >= pivot)) cnt[d + 1] = 0;                                              // select_docs :1413-1415
  }
  c->original_cols.clear();
  c->boffs.assign(1, 0);
  std::vector<int64_t> src_first;
  for (uint64_t d = 0; d < D; ++d)
    if (cnt[d + 1] > 0) {
      c->original_cols.push_back(d);
      c->boffs.push_back(c->boffs.back() + cnt[d + 1]);
    }
  c->Db = c->original_cols.size();
  const uint64_t bnnz = c->boffs.back();
  c->bvals.resize(bnnz);
  c->brows.resize(bnnz);
#pragma omp parallel for schedule(dynamic, 4096)
  for (int64_t j = 0; j < (int64_t)c->Db; ++j) {
    const uint64_t d = c->original_cols[j];
    int64_t p = c->boffs[j];
    for (int64_t i = c->offs[d]; i < c->offs[d + 1]; ++i) {
      const float z = c->zetas[c->rows[i]];
      if (rnd[i] >= z) {
        c->bvals[p] = std::sqrt(z);
        c->brows[p] = c->rows[i];
        ++p;
      }
    }
  }
  std::vector<float>().swap(g_ts.rnd);
  std::vector<uint32_t>().swap(g_ts.hist);
  return bnnz;
}

// single-process convenience: all three phases
uint64_t synth_threshold(void* h, uint32_t k) {
  uint64_t t, nz;
  uint32_t mv;
  synth_stats(h, &t, &nz);
  (void)synth_hist(h, t, nz, &mv);
  return synth_apply(h, k, nz, 0.0, 0);
}

uint64_t synth_nnz_A(void* h) { return ((Corpus*)h)->offs.back(); }

// tdf text of A ("<doc> <word> <count>\n", 1-based) into out[cap]; returns the number of bytes (0 if cap is too small).
uint64_t synth_tdf_bytes(void* h, char* out, uint64_t cap) {
  Corpus* c = (Corpus*)h;
  const uint64_t D = c->D;
  std::vector<uint64_t> start(D + 1, 0);
  auto digits = [](uint64_t x) { int n = 1; while (x >= 10) { x /= 10; ++n; } return (uint64_t)n; };
#pragma omp parallel for schedule(dynamic, 4096)
  for (int64_t d = 0; d < (int64_t)D; ++d) {
    uint64_t b = 0;
    for (int64_t i = c->offs[d]; i < c->offs[d + 1]; ++i) b += digits(d + 1) + digits((uint64_t)c->rows[i] + 1) + digits((uint64_t)c->counts[i]) + 3;
    start[d + 1] = b;
  }
  for (uint64_t d = 0; d < D; ++d) start[d + 1] += start[d];
  if (start[D] > cap) return 0;
  auto put = [](char* p, uint64_t x) {
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + x % 10); x /= 10; } while (x);
    for (int j = 0; j < n; ++j) p[j] = tmp[n - 1 - j];
    return p + n;
  };
#pragma omp parallel for schedule(dynamic, 4096)
  for (int64_t d = 0; d < (int64_t)D; ++d) {
    char* p = out + start[d];
    for (int64_t i = c->offs[d]; i < c->offs[d + 1]; ++i) {
      p = put(p, (uint64_t)d + 1);
      *p++ = ' ';
      p = put(p, (uint64_t)c->rows[i] + 1);
      *p++ = ' ';
      p = put(p, (uint64_t)c->counts[i]);
      *p++ = '\n';
    }
  }
  return start[D];
}
uint64_t synth_docs_B(void* h) { return ((Corpus*)h)->Db; }
const float* synth_A_counts(void* h) { return ((Corpus*)h)->counts.data(); }
const uint32_t* synth_A_rows(void* h) { return ((Corpus*)h)->rows.data(); }
const int64_t* synth_A_offs(void* h) { return ((Corpus*)h)->offs.data(); }
const uint32_t* synth_dom(void* h) { return ((Corpus*)h)->dom.data(); }
const float* synth_B_vals(void* h) { return ((Corpus*)h)->bvals.data(); }
const uint32_t* synth_B_rows(void* h) { return ((Corpus*)h)->brows.data(); }
const int64_t* synth_B_offs(void* h) { return ((Corpus*)h)->boffs.data(); }
const uint64_t* synth_B_original_cols(void* h) { return ((Corpus*)h)->original_cols.data(); }
const float* synth_zetas(void* h) { return ((Corpus*)h)->zetas.data(); }
void synth_free_A(void* h) {
  Corpus* c = (Corpus*)h;
  std::vector<float>().swap(c->counts);
  std::vector<uint32_t>().swap(c->rows);
}
void synth_destroy(void* h) { delete (Corpus*)h; }
}
