// isle_amd/host/trainer_hip.h — ISLE::ISLETrainer over the MI355X path: the reference's trainer class (include/trainer.h:97-265,
// src/trainer.cpp) with its constructor arguments, its data-ingest modes and the methods its drivers call —
//   ISLETrainer(...)  load_data_from_file()  feed_data()  finalize_data()  train()  output_cluster_summary()  write_model_to_file()
//   train_edge_topics()  write_edgemodel_to_file()  get_basic_model()  get_num_edge_topics()  get_edge_model()
// — so that drivers/ISLETrain.cpp (:34-46) and the stale export layer drivers/trainer_export.cpp (:31-98) read the same against it.
// What runs where: ingest, thresholding, the hot path of train() (src/trainer.cpp:490-571), catchwords, the topic model and the edge
// topics all run on the device through FPSparseMatrixHip (fpsparse_hip.h -> include/isle_hip.h); this class is the host-side order of
// calls, the log lines (diagnosticLog.txt / timerLog.txt with the reference's formats) and the output files.
// Not mirrored (dead under the shipped hyper-parameters or outside the path): load_preprocessed_data_from_file,
// print_log_combinatorial, print_distinct_top_five_sets, compute_input_svd, the coherence / diversity outputs, construct_edge_topics_v1.
#pragma once
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <ctime>
#include <fstream>
#include <iomanip>
#include <memory>
#include <numeric>
#include <sstream>

#include "fpsparse_hip.h"
#include "prestage.h"

namespace ISLE {
namespace trainer_detail {
struct Logs {
  std::ofstream diag, timer;
  clock_t u0;
  std::chrono::high_resolution_clock::time_point s0, sbegin;
  clock_t ubegin;
  explicit Logs(const std::string& dir) : diag(dir + "/diagnosticLog.txt"), timer(dir + "/timerLog.txt") {
    u0 = ubegin = std::clock();
    s0 = sbegin = std::chrono::high_resolution_clock::now();
  }
  void print(const std::string& s) {  // LogUtils::print_string: file + stdout
    diag << s << std::flush;
    std::cout << s << std::flush;
  }
  void next_time_secs(const std::string& text, int fill_len = 40) {  // include/timer.h:72-85
    const clock_t u1 = std::clock();
    const auto s1 = std::chrono::high_resolution_clock::now();
    std::ostringstream ostr;
    ostr << "Time for " << std::setfill('.') << std::setw(fill_len) << std::left << text << ((double)(u1 - u0)) / CLOCKS_PER_SEC
         << "s(user)  " << std::chrono::duration<double>(s1 - s0).count() << "s(sys)";
    std::cout << ostr.str() << std::endl;
    timer << ostr.str() << std::endl;
    u0 = u1;
    s0 = s1;
  }
  void total(const std::string& text) {  // include/timer.h:108-120
    std::ostringstream ostr;
    ostr << "Total time for " << std::setfill('.') << std::setw(50) << std::left << text
         << ((double)(std::clock() - ubegin)) / CLOCKS_PER_SEC << "s(user)  "
         << std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - sbegin).count() << "s(secs)";
    std::cout << ostr.str() << std::endl;
    timer << ostr.str() << std::endl;
  }
};

std::string log_dir_name(uint64_t num_topics, const std::string& base, bool sample_docs, float sample_rate, bool tf_idf) {  // src/utils.cpp:28-48
  std::string s = "log_t_" + std::to_string(num_topics) + "_eps1_" + std::to_string(1.0 / 60.0) + "_eps2_" + std::to_string(1.0 / 3.0) +
                  "_eps3_" + std::to_string(5.0) + "_kMppReps_" + std::to_string(1) + "_kMLowDReps_" + std::to_string(10) + "_kMReps_" +
                  std::to_string(10) + "_sample_" + std::to_string(sample_docs) + "_tfidf_" + std::to_string((int)tf_idf);
  if (sample_docs) s += "_Rate_" + std::to_string(sample_rate);
  return base + "/" + s;
}
// DenseMatrix::write_to_file_as_sparse (src/denseMatrix.cpp:155-186, mmap branch) with MMappedOutput::concat_int /
// concat_float (include/utils.h:405-478): "<topic>\t<word>\t<weight>\n", 1-based, entries <= 1e-8 skipped, the weight
// written as integer part, '.', then SIX digits produced by repeated multiplication in FPTYPE — truncated, not rounded
// (the before_dec / after_dec arguments of concat_float never reach ftoa_mv).
// The weight's text, exactly as the reference's writer emits it: the integer part in decimal (at most its six low digits), a point, then six
// fraction digits peeled off one at a time by multiplying the float remainder by ten (single precision, truncating).  Returns the length.
inline size_t weight_text(float w, char* out) {
  size_t len = 0;
  unsigned int whole = (unsigned int)w;
  char digits[8];
  int nd = 0;
  do {
    digits[nd++] = (char)('0' + whole % 10);
    whole /= 10;
  } while (whole > 0 && nd < 6);
  while (nd > 0) out[len++] = digits[--nd];
  out[len++] = '.';
  float rest = w - (float)((int)w);
  for (int place = 0; place < 6; ++place) {
    rest *= 10;
    const int digit = (int)rest;
    out[len++] = (char)('0' + digit);
    rest -= digit;
  }
  return len;
}
void write_dense_as_sparse(const std::string& filename, const float* M, uint64_t vocab_size, uint64_t ncols) {
  constexpr size_t kFlushAt = (size_t(1) << 24) - 256;
  FILE* fp = std::fopen(filename.c_str(), "wb");
  if (!fp) throw std::runtime_error("cannot open " + filename);
  std::string pending;
  pending.reserve(size_t(1) << 24);
  char text[32];
  for (uint64_t col = 0; col < ncols; ++col) {
    const float* column = M + col * vocab_size;
    for (uint64_t row = 0; row < vocab_size; ++row) {
      const float w = column[row];
      if (!(w > 0.00000001f)) continue;  // the reference skips entries at or below 1e-8
      pending += std::to_string(col + 1);
      pending += '\t';
      pending += std::to_string(row + 1);
      pending += '\t';
      pending.append(text, weight_text(w, text));
      pending += '\n';
      if (pending.size() > kFlushAt) {
        std::fwrite(pending.data(), 1, pending.size(), fp);
        pending.clear();
      }
    }
  }
  std::fwrite(pending.data(), 1, pending.size(), fp);
  std::fclose(fp);
}
}  // namespace trainer_detail

class ISLETrainer {
 public:
  enum data_ingest { FILE_DATA_LOAD, PREPROCESSED_DATA_LOAD, ITERATIVE_DATA_LOAD };  // include/trainer.h:91-94

 private:
  const word_id_t vocab_size;
  const doc_id_t num_docs;
  const offset_t max_entries;
  const doc_id_t num_topics;
  const bool flag_tf_idf;  // a no-op in the reference too; only the directory name differs
  const bool flag_sample_docs;
  const FPTYPE sample_rate;
  const data_ingest how_data_loaded;
  const std::string input_file, vocab_file, output_path_base;
  const bool flag_construct_edge_topics;
  const int max_edge_topics;
  std::string log_dir;
  std::unique_ptr<trainer_detail::Logs> log;

  bool is_data_loaded = false, is_training_complete = false;
  // ITERATIVE_DATA_LOAD: the (doc, word, count) triples fed so far (src/trainer.cpp:200-230 keeps DocWordEntry records)
  std::vector<uint64_t> fed_doc;
  std::vector<uint32_t> fed_word;
  std::vector<float> fed_count;

  FPSparseMatrixHip* B_fl_CSC = nullptr;  // owns the device context: A (counts), B and everything derived from them live there
  std::vector<doc_id_t> original_cols;
  uint64_t entries_in_A = 0, entries_above_threshold = 0;
  float avg_doc_sz = 0.f;
  std::vector<FPTYPE> evalues;
  std::vector<doc_id_t>* closest_docs = nullptr;
  std::vector<word_id_t>* catchwords = nullptr;
  FPTYPE* catchword_thresholds = nullptr;
  FPTYPE* Model = nullptr;  // vocab_size x num_topics, column-major (DenseMatrix<FPTYPE>, include/denseMatrix.h:50-58)
  std::vector<std::tuple<int, int, doc_id_t>> top_topic_pairs;
  std::vector<std::tuple<int, int, uint64_t>> selected_pairs;
  std::vector<FPTYPE> EdgeModel;
  std::vector<std::string> vocab_words;
  std::vector<std::vector<std::pair<word_id_t, FPTYPE>>> topwords;

  void print_header() {  // src/trainer.cpp:130-143
    std::ostringstream s;
    s << "\n<<<<<<<<<<<<\t" << input_file << "\t>>>>>>>>>>>>\n\n"
      << std::setfill('.') << std::setw(10) << std::left << std::setw(15) << std::left << "#Entries" << max_entries << "\n"
      << std::setw(15) << std::left << "#Words" << vocab_size << "\n"
      << std::setw(15) << std::left << "#Docs" << num_docs << "\n"
      << std::setw(15) << std::left << "#Topics" << num_topics << "\n"
      << std::setw(15) << std::left << "TF-IDF" << flag_tf_idf << "\n"
      << std::setw(15) << std::left << "Sampling?" << flag_sample_docs << "\n"
      << std::setw(15) << std::left << "Sample rate" << sample_rate << "\n"
      << std::setw(15) << std::left << "Edge topics?" << flag_construct_edge_topics << "\n"
      << std::setw(15) << std::left << "#Edge topics" << max_edge_topics << std::endl;
    log->print(s.str());
  }
  void after_matrices_built() {  // the log lines of finalize_data / the thresholding block of train() (src/trainer.cpp:236-371, :430-485)
    log->next_time_secs("Sorting entries");
    log->next_time_secs("De-duplicating entries");
    std::cout << "Entries in sparse matrix: " << entries_in_A << std::endl << "Average document size: " << avg_doc_sz << std::endl;
    log->next_time_secs("Populating CSC");
    log->next_time_secs("Computing thresholds");
    log->print("Number of entries above threshold: " + std::to_string(entries_above_threshold) + "\n");
    std::cout << (flag_sample_docs ? "After sampling docs: cols remaining: " : "Columns remaining after thresholding: ") << B_fl_CSC->num_docs() << "\n";
    log->next_time_secs("Creating thresholded and scaled matrix");
    is_data_loaded = true;
  }

 public:
  ISLETrainer(const word_id_t vocab_size_, const doc_id_t num_docs_, const offset_t max_entries_, const doc_id_t num_topics_, const bool tf_idf_,
              const bool sample_docs_, const FPTYPE sample_rate_, const data_ingest how_data_loaded_, const std::string& input_file_ = std::string(""),
              const std::string& vocab_file_ = std::string(""), const std::string& output_path_base_ = std::string(""),
              const bool construct_edge_topics_ = false, const int max_edge_topics_ = 100000)
      : vocab_size(vocab_size_), num_docs(num_docs_), max_entries(max_entries_), num_topics(num_topics_), flag_tf_idf(tf_idf_),
        flag_sample_docs(sample_docs_), sample_rate(sample_rate_), how_data_loaded(how_data_loaded_), input_file(input_file_),
        vocab_file(vocab_file_), output_path_base(output_path_base_), flag_construct_edge_topics(construct_edge_topics_),
        max_edge_topics(max_edge_topics_) {
    // src/trainer.cpp:8-81: log directory, the two log files, then the data according to the ingest mode
    log_dir = trainer_detail::log_dir_name(num_topics, output_path_base, flag_sample_docs, sample_rate, flag_tf_idf);
    struct stat st;
    if (stat(log_dir.c_str(), &st) == -1) mkdir(log_dir.c_str(), S_IRWXU);
    else std::cerr << "Subdir exists already" << std::endl;
    log.reset(new trainer_detail::Logs(log_dir));
    if (how_data_loaded == FILE_DATA_LOAD) load_data_from_file();
    else if (how_data_loaded == PREPROCESSED_DATA_LOAD) throw std::runtime_error("PREPROCESSED_DATA_LOAD is not mirrored (binary A_sp dumps of the reference)");
  }
  ~ISLETrainer() {
    delete[] closest_docs;
    delete[] catchwords;
    delete[] catchword_thresholds;
    delete[] Model;
    delete B_fl_CSC;
  }
  ISLETrainer(const ISLETrainer&) = delete;

  // src/trainer.cpp:124-150 + finalize_data :232-371 + the thresholding block of train() :430-485: tdf text -> A -> B, all on the device
  // (include/utils.h:96-229 for the format)
  void load_data_from_file() {
    std::vector<char> text;
    FILE* f = std::fopen(input_file.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open tdf file " + input_file);
    std::fseek(f, 0, SEEK_END);
    const long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    text.resize((size_t)sz);
    if (sz && std::fread(text.data(), 1, (size_t)sz, f) != (size_t)sz) {
      std::fclose(f);
      throw std::runtime_error("short read on " + input_file);
    }
    std::fclose(f);
    print_header();
    log->next_time_secs("Reading file Entries");
    B_fl_CSC = FPSparseMatrixHip::from_tdf(vocab_size, num_docs, text.data(), text.size(), max_entries, num_topics,
                                           flag_sample_docs ? (double)sample_rate : 0.0, original_cols, &entries_in_A, &entries_above_threshold,
                                           &avg_doc_sz);
    after_matrices_built();
  }

  // include/trainer.h:139-143, src/trainer.cpp:214-230: one document's words and counts.  The reference's convention, kept: `doc` is the
  // 0-based column, `words[i]` are 1-BASED word ids as in a tdf file (it stores `words[w] - 1`, :224).  Where the reference would write
  // outside its matrix, an id out of range is an error here.
  inline void feed_data(const doc_id_t doc, const word_id_t* const words, const count_t* const counts, const offset_t num_words) {
    if (how_data_loaded != ITERATIVE_DATA_LOAD) throw std::runtime_error("feed_data needs ITERATIVE_DATA_LOAD");
    if (is_data_loaded) throw std::runtime_error("feed_data after finalize_data");
    for (offset_t i = 0; i < num_words; ++i) {
      if (doc >= num_docs || words[i] < 1 || words[i] > vocab_size) throw std::runtime_error("feed_data: id out of range (documents 0-based, words 1-based)");
      if (counts[i] == 0) continue;
      fed_doc.push_back(doc);
      fed_word.push_back((uint32_t)(words[i] - 1));
      fed_count.push_back((float)counts[i]);
    }
  }
  // src/trainer.cpp:232-371: sort by (doc, word), keep the first of equal pairs, CSC of A — then A goes to the device, where B is built
  void finalize_data() {
    if (how_data_loaded != ITERATIVE_DATA_LOAD) throw std::runtime_error("finalize_data needs ITERATIVE_DATA_LOAD");
    print_header();
    log->next_time_secs("Reading file Entries");
    const size_t n = fed_doc.size();
    std::vector<size_t> order(n);
    std::iota(order.begin(), order.end(), (size_t)0);
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
      return fed_doc[a] < fed_doc[b] || (fed_doc[a] == fed_doc[b] && fed_word[a] < fed_word[b]);
    });
    std::vector<float> counts;
    std::vector<uint32_t> rows;
    std::vector<offset_t> offsets((size_t)num_docs + 1, 0);
    counts.reserve(n);
    rows.reserve(n);
    for (size_t i = 0; i < n; ++i) {
      const size_t e = order[i];
      if (i && fed_doc[e] == fed_doc[order[i - 1]] && fed_word[e] == fed_word[order[i - 1]]) continue;  // duplicate (doc, word): the first stays
      counts.push_back(fed_count[e]);
      rows.push_back(fed_word[e]);
      offsets[fed_doc[e] + 1]++;
    }
    for (doc_id_t d = 0; d < num_docs; ++d) offsets[d + 1] += offsets[d];
    entries_in_A = counts.size();
    std::vector<uint64_t>().swap(fed_doc);
    std::vector<uint32_t>().swap(fed_word);
    std::vector<float>().swap(fed_count);
    B_fl_CSC = FPSparseMatrixHip::from_counts(vocab_size, num_docs, counts.data(), rows.data(), offsets.data(), num_topics,
                                              flag_sample_docs ? (double)sample_rate : 0.0, original_cols, nullptr, &entries_above_threshold, &avg_doc_sz);
    after_matrices_built();
  }

  // src/trainer.cpp:425-654 (thresholding already done where the data came in): the hot path :490-571, then catchwords and topic vectors
  void train() {
    if (!is_data_loaded) throw std::runtime_error("train() before the data is loaded");
    log->print("Frob(B_fl_CSC): " + std::to_string(B_fl_CSC->frobenius()) + "\n");
    B_fl_CSC->initialize_for_eigensolver(num_topics);
    log->next_time_secs("eigen solver init");
    B_fl_CSC->compute_block_ks(num_topics, evalues);
    {
      std::ostringstream ostr;  // include/logUtils.h:101-122
      ostr << "Eigvals:  ";
      for (doc_id_t t = 0; t < num_topics; ++t) ostr << "(" << t << "): " << std::sqrt(evalues[t]) << "\t";
      ostr << std::endl;
      std::vector<FPTYPE> slabs(num_topics / 100 + 1, 0.0);
      for (doc_id_t t = 0; t < num_topics; ++t) slabs[t / 100] += evalues[t];
      for (doc_id_t slab = 0; slab < num_topics / 100; ++slab)
        ostr << "Sum of Top-" << (slab + 1) * 100 << " eig vals: " << std::accumulate(slabs.begin(), slabs.begin() + 1 + slab, (FPTYPE)0.0) << "\n";
      log->print(ostr.str());
    }
    log->next_time_secs("Spectra eigen solve");  // the reference uses this label for block-KS too (App. C #13)

    std::vector<doc_id_t> best_kmeans_seeds;
    FPTYPE* centers_lowd = new FPTYPE[(size_t)num_topics * num_topics];
    log->print("k-means init method: KMEANSPP\n");
    const FPTYPE best_residual = B_fl_CSC->kmeans_init_on_projected_space((int)num_topics, 1, best_kmeans_seeds, centers_lowd);
    log->print("Best k-means init residual: " + std::to_string(best_residual) + "\n");
    log->next_time_secs("K-means seeds initialization");

    B_fl_CSC->run_lloyds_on_projected_space(num_topics, centers_lowd, NULL, 10);
    // The reference allocates centers[vocab_size * num_topics] here (src/trainer.cpp:284) and hands it through both calls below, but
    // reads nothing of it afterwards (only closest_docs, :566-575): the lifted centres and Lloyd's result stay in device memory.
    FPTYPE* centers = nullptr;
    B_fl_CSC->left_multiply_by_U_Spectra(centers, centers_lowd, num_topics, num_topics);
    delete[] centers_lowd;
    log->next_time_secs("Converging LLoyds k-means on B_k");
    B_fl_CSC->cleanup_after_eigensolver();

    closest_docs = new std::vector<doc_id_t>[num_topics];
    B_fl_CSC->run_lloyds(num_topics, centers, closest_docs, 10);
    uint64_t closest_docs_sizes_sum = 0;
    for (doc_id_t t = 0; t < num_topics; ++t) closest_docs_sizes_sum += closest_docs[t].size();
    if (closest_docs_sizes_sum != B_fl_CSC->num_docs()) throw std::runtime_error("partition incomplete");  // :567-570
    log->next_time_secs("k-means on B");
    for (doc_id_t topic = 0; topic != num_topics; ++topic)  // :573-575
      for (auto d = closest_docs[topic].begin(); d < closest_docs[topic].end(); ++d) *d = original_cols[*d];
    {
      std::ofstream o(log_dir + "/HotPathClusters.tsv");  // topic \t doc, 1-based like the reference's sparse writers
      for (doc_id_t t = 0; t < num_topics; ++t)
        for (doc_id_t d : closest_docs[t]) o << (t + 1) << "\t" << (d + 1) << "\n";
      std::ofstream sv(log_dir + "/HotPathSingularValues.txt");
      sv << std::setprecision(9);
      for (doc_id_t t = 0; t < num_topics; ++t) sv << std::sqrt(evalues[t]) << "\n";
    }

    // ---- src/trainer.cpp:577-654: catchwords and the topic model, on the device ------------------
    uint64_t r;  // :579-583
    if (flag_sample_docs)
      r = (uint64_t)std::floor(ISLE_EPS2_C * ISLE_W0_C * (FPTYPE)num_docs * sample_rate / (FPTYPE)(2.0 * num_topics));
    else
      r = (uint64_t)std::floor(ISLE_EPS2_C * ISLE_W0_C * (FPTYPE)num_docs / (FPTYPE)(2.0 * num_topics));
    catchword_thresholds = new FPTYPE[(size_t)vocab_size * num_topics];
    catchwords = new std::vector<word_id_t>[num_topics];
    B_fl_CSC->find_catchwords(num_topics, r, catchword_thresholds, catchwords);
    log->next_time_secs("Collecting word freqs in clusters");
    log->next_time_secs("Finding catchwords for clusters");
    Model = new FPTYPE[(size_t)vocab_size * num_topics];
    B_fl_CSC->construct_topic_model(Model, num_topics, num_docs, flag_construct_edge_topics ? &top_topic_pairs : NULL);
    log->next_time_secs("Constructing topic vectors");
    is_training_complete = true;
  }

  // src/trainer.cpp:776-826
  void output_cluster_summary() {
    if (!is_training_complete) throw std::runtime_error("output_cluster_summary() before train()");
    {  // create_vocab_list, src/utils.cpp:6-25
      std::ifstream in(vocab_file);
      std::string word;
      while (in.good() && !in.eof() && vocab_words.size() < vocab_size) {
        in >> word;
        vocab_words.push_back(word);
      }
      vocab_words.resize(vocab_size);
    }
    const word_id_t ntop = std::min<word_id_t>(10, vocab_size);  // max(DEFAULT_COHERENCE_NUM_WORDS, 10), :781-783
    topwords.assign(num_topics, {});
    for (doc_id_t t = 0; t < num_topics; ++t) {  // DenseMatrix::find_n_top_words, src/denseMatrix.cpp:92-107 (ties: lower word id first)
      std::vector<std::pair<word_id_t, FPTYPE>>& tw = topwords[t];
      tw.reserve(vocab_size);
      for (word_id_t w = 0; w < vocab_size; ++w) tw.push_back(std::make_pair(w, Model[(size_t)t * vocab_size + w]));
      // heaviest first, lower word id first among equal weights (what a stable sort of the word-ordered list gives)
      std::partial_sort(tw.begin(), tw.begin() + ntop, tw.end(), [](const std::pair<word_id_t, FPTYPE>& l, const std::pair<word_id_t, FPTYPE>& r2) {
        return l.second > r2.second || (l.second == r2.second && l.first < r2.first);
      });
      if (tw[ntop - 1].second == (FPTYPE)0.0) std::cout << "\n ==== WARNING: top words in topic " << t << " have zero weight\n\n";
      tw.resize(ntop);
    }
    for (doc_id_t t = 0; t < num_topics; ++t) {
      std::ostringstream o;
      o << "\n---------- Topic: " << t << ", Cluster_size: " << closest_docs[t].size() << " -----------\n";
      o << "Catchwords:\n";  // include/logUtils.h:49-64
      for (word_id_t w : catchwords[t]) o << vocab_words[w] << ":" << w << "(" << catchword_thresholds[(size_t)t * vocab_size + w] << ") ";
      o << "\n";
      o << "\n#Top words: " << topwords[t].size() << "\n";  // src/denseMatrix.cpp:110-121
      for (auto& tw : topwords[t]) o << vocab_words[tw.first] << ":" << tw.first << "(" << tw.second << ") ";
      o << "\n\n";
      log->diag << o.str();
    }
    log->diag << "\n---------------------------\n";
    log->print("\n Avg coherence: " + std::to_string(0.0f) + "\n\n");
    {  // LogUtils::print_cluster_details, include/logUtils.h:66-99
      std::vector<std::pair<int, doc_id_t>> cluster_sizes;
      for (doc_id_t t = 0; t < num_topics; ++t) cluster_sizes.push_back(std::make_pair((int)closest_docs[t].size(), t));
      std::stable_sort(cluster_sizes.begin(), cluster_sizes.end(),
                       [](const std::pair<int, doc_id_t>& l, const std::pair<int, doc_id_t>& r2) { return l.first < r2.first; });
      std::ostringstream o;
      int catchless = 0;
      for (doc_id_t i = 0; i < num_topics; ++i) {
        const doc_id_t t = cluster_sizes[i].second;
        o << std::setw(12) << std::left << "Cluster" << t << std::setw(12) << std::left << "  size:" << cluster_sizes[i].first << std::setw(15)
          << std::left << "  distsq_sum:" << 0 << std::setw(15) << std::left << "  raw_coh:" << 0 << std::setw(15) << std::left << "  flt_coh:" << 0
          << "  #catchwords: " << catchwords[t].size() << std::endl;
        if (catchwords[t].size() == 0) catchless++;
      }
      o << "\n#Topics with no catchwords: " << catchless << "(" << num_topics << ")" << std::endl;
      log->print(o.str());
    }
    log->next_time_secs("Output summary");
  }

  void output_top_words() {  // src/trainer.cpp:855-868
    std::ofstream out_top_words(log_dir + "/TopWordsPerTopic_catch.txt");
    for (doc_id_t t = 0; t < num_topics; ++t) {
      for (auto& tw : topwords[t]) out_top_words << vocab_words[tw.first] << "\t";
      out_top_words << std::endl;
    }
    log->next_time_secs("Writing top words to file");
  }
  void output_model(bool output_sparse = false) {  // src/trainer.cpp:831-838 (the CLI asks for the sparse form)
    (void)output_sparse;
    trainer_detail::write_dense_as_sparse(log_dir + "/M_hat_catch_sparse", Model, vocab_size, num_topics);
  }
  // src/trainer.cpp:656-662
  void write_model_to_file() {
    if (topwords.empty()) throw std::runtime_error("write_model_to_file() before output_cluster_summary()");
    output_top_words();
    output_model(true);
    log->next_time_secs("Output model");
    output_top_words();
    log->next_time_secs("Output topwords");
  }
  // src/trainer.cpp:673-685 -> construct_edge_topics_v2 :1116-1167
  void train_edge_topics() {
    if (!flag_construct_edge_topics) throw std::runtime_error("train_edge_topics() without construct_edge_topics");
    B_fl_CSC->construct_edge_topics(top_topic_pairs, max_edge_topics, selected_pairs, EdgeModel);
    log->next_time_secs("Constructing edge topic model");
  }
  // src/trainer.cpp:687-693
  void write_edgemodel_to_file() {
    trainer_detail::write_dense_as_sparse(log_dir + "/EdgeModel_sparse", EdgeModel.data(), vocab_size, selected_pairs.size());
    log->next_time_secs("Output edge model");
  }
  void finish_log() { log->total("TVSD"); }  // the "Total time for TVSD" line the reference's train() ends with (:652)

  // src/trainer.cpp:993-996: vocab_size x num_topics floats, column-major (element (word, topic) at word + topic * vocab_size)
  void get_basic_model(FPTYPE* const basicModel) { std::memcpy(basicModel, Model, (size_t)vocab_size * num_topics * sizeof(FPTYPE)); }
  int get_num_edge_topics() { return (int)selected_pairs.size(); }                                                        // :998-1001
  void get_edge_model(FPTYPE* const edgeModel) { std::memcpy(edgeModel, EdgeModel.data(), EdgeModel.size() * sizeof(FPTYPE)); }  // :1003-1007
  const std::vector<FPTYPE>& eigenvalues() const { return evalues; }
  const std::vector<doc_id_t>* partition() const { return closest_docs; }
};

}  // namespace ISLE
