// isle_amd/host/ISLETrain.cpp — the reference's 12-argument CLI (drivers/ISLETrain.cpp:8-51) over the MI355X hot path.
//
//   ISLETrain <tdf_file> <vocab_file> <output_dir> <vocab_size> <num_docs> <max_entries> <num_topics>
//             <apply tf-idf(0/1)> <sample(0/1)> <sample_rate> <edge topics(0/1)> <max_edge_topics>
//
// Runs ingest (device) -> thresholding (device) -> the hot path src/trainer.cpp:490-571 on the GPU, and writes into
// the reference's log directory (src/utils.cpp:28-48) diagnosticLog.txt / timerLog.txt with the reference's line
// formats for these phases, then catchwords, the topic model and (optionally) edge topics on the device
// (src/trainer.cpp:577-654, :673-693) and the reference's output files M_hat_catch_sparse, TopWordsPerTopic_catch.txt,
// EdgeModel_sparse.  Extra files: HotPathClusters.tsv / HotPathSingularValues.txt (partition and singular values).
#include <sys/stat.h>

#include <chrono>
#include <ctime>
#include <fstream>
#include <iomanip>
#include <numeric>
#include <sstream>

#include "fpsparse_hip.h"
#include "prestage.h"

using namespace ISLE;

namespace {
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
void write_dense_as_sparse(const std::string& filename, const float* M, uint64_t vocab_size, uint64_t ncols) {
  std::string buf;
  buf.reserve(1 << 24);
  FILE* f = std::fopen(filename.c_str(), "wb");
  if (!f) throw std::runtime_error("cannot open " + filename);
  char tmp[64];
  for (uint64_t topic = 0; topic < ncols; ++topic)
    for (uint64_t word = 0; word < vocab_size; ++word) {
      float num = M[topic * vocab_size + word];
      if (!(num > 0.00000001f)) continue;
      buf += std::to_string(topic + 1);
      buf += '\t';
      buf += std::to_string(word + 1);
      buf += '\t';
      int i = 0;
      unsigned int num_int = (unsigned int)num;
      if (num_int == 0) {
        tmp[i++] = '0';
      } else {
        char rev[16];
        int n = 0;
        for (int d = 0; d < 6 && num_int > 0; ++d) {
          rev[n++] = (char)('0' + num_int % 10);
          num_int /= 10;
        }
        while (n) tmp[i++] = rev[--n];
      }
      tmp[i++] = '.';
      float frac = num - (float)((int)num);
      for (int d = 0; d < 6; ++d) {
        frac *= 10;
        tmp[i++] = (char)('0' + (int)frac);
        frac -= (int)frac;
      }
      tmp[i++] = '\n';
      buf.append(tmp, (size_t)i);
      if (buf.size() > (1u << 24) - 256) {
        std::fwrite(buf.data(), 1, buf.size(), f);
        buf.clear();
      }
    }
  std::fwrite(buf.data(), 1, buf.size(), f);
  std::fclose(f);
}
}  // namespace

int main(int argv, char** argc) {
  if (argv != 13) {
    std::cout << "Incorrect usage of ISLETrain. Use: \n"
              << "trainFromFile <tdf_file> <vocab_file> <output_dir> "
              << "<vocab_size> <num_docs> <max_entries> <num_topics> "
              << "<apply tf-idf(0/1)> <sample(0/1)> <sample_rate> "
              << "<edge topics(0/1) <max_edge_topics>" << std::endl;
    exit(-1);
  }
  const std::string tdf_file = argc[1];
  const std::string vocab_file = argc[2];
  const std::string output_dir = argc[3];
  const word_id_t vocab_size = atol(argc[4]);
  const doc_id_t num_docs = atol(argc[5]);
  const offset_t max_entries = atol(argc[6]);
  const doc_id_t num_topics = atol(argc[7]);
  const bool tf_idf = atoi(argc[8]);  // a no-op in the reference too; only the directory name differs
  const bool sample = atoi(argc[9]);
  const FPTYPE sample_rate = (FPTYPE)atof(argc[10]);
  const bool compute_edge_topics = atoi(argc[11]);

  try {
    const std::string log_dir = log_dir_name(num_topics, output_dir, sample, sample_rate, tf_idf);
    struct stat st;
    if (stat(log_dir.c_str(), &st) == -1) mkdir(log_dir.c_str(), S_IRWXU);
    else std::cerr << "Subdir exists already" << std::endl;
    Logs log(log_dir);

    // ---- ingest (include/utils.h:96-229; src/trainer.cpp:232-371) and thresholding (:430-485), both on the device --------
    std::vector<char> text;
    {
      FILE* f = std::fopen(tdf_file.c_str(), "rb");
      if (!f) throw std::runtime_error("cannot open tdf file " + tdf_file);
      std::fseek(f, 0, SEEK_END);
      const long sz = std::ftell(f);
      std::fseek(f, 0, SEEK_SET);
      text.resize((size_t)sz);
      if (sz && std::fread(text.data(), 1, (size_t)sz, f) != (size_t)sz) {
        std::fclose(f);
        throw std::runtime_error("short read on " + tdf_file);
      }
      std::fclose(f);
    }
    {
      std::ostringstream s;  // src/trainer.cpp:130-143
      s << "\n<<<<<<<<<<<<\t" << tdf_file << "\t>>>>>>>>>>>>\n\n"
        << std::setfill('.') << std::setw(10) << std::left << std::setw(15) << std::left << "#Entries" << max_entries << "\n"
        << std::setw(15) << std::left << "#Words" << vocab_size << "\n"
        << std::setw(15) << std::left << "#Docs" << num_docs << "\n"
        << std::setw(15) << std::left << "#Topics" << num_topics << "\n"
        << std::setw(15) << std::left << "TF-IDF" << tf_idf << "\n"
        << std::setw(15) << std::left << "Sampling?" << sample << "\n"
        << std::setw(15) << std::left << "Sample rate" << sample_rate << "\n"
        << std::setw(15) << std::left << "Edge topics?" << compute_edge_topics << "\n"
        << std::setw(15) << std::left << "#Edge topics" << atoi(argc[12]) << std::endl;
      log.print(s.str());
    }
    log.next_time_secs("Reading file Entries");
    std::vector<doc_id_t> original_cols;
    uint64_t entries_in_A = 0, entries_above_threshold = 0;
    float avg_doc_sz = 0.f;
    FPSparseMatrixHip* B_fl_CSC = FPSparseMatrixHip::from_tdf(vocab_size, num_docs, text.data(), text.size(), max_entries, num_topics,
                                                              sample ? (double)sample_rate : 0.0, original_cols, &entries_in_A,
                                                              &entries_above_threshold, &avg_doc_sz);
    std::vector<char>().swap(text);
    log.next_time_secs("Sorting entries");
    log.next_time_secs("De-duplicating entries");
    std::cout << "Entries in sparse matrix: " << entries_in_A << std::endl << "Average document size: " << avg_doc_sz << std::endl;
    log.next_time_secs("Populating CSC");
    log.next_time_secs("Computing thresholds");
    log.print("Number of entries above threshold: " + std::to_string(entries_above_threshold) + "\n");
    std::cout << (sample ? "After sampling docs: cols remaining: " : "Columns remaining after thresholding: ") << B_fl_CSC->num_docs() << "\n";
    log.next_time_secs("Creating thresholded and scaled matrix");

    // ---- src/trainer.cpp:490-571 -----------------------------------------------------------------
    log.print("Frob(B_fl_CSC): " + std::to_string(B_fl_CSC->frobenius()) + "\n");
    std::vector<FPTYPE> evalues;
    B_fl_CSC->initialize_for_eigensolver(num_topics);
    log.next_time_secs("eigen solver init");
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
      log.print(ostr.str());
    }
    log.next_time_secs("Spectra eigen solve");  // the reference uses this label for block-KS too (App. C #13)

    std::vector<doc_id_t> best_kmeans_seeds;
    FPTYPE* centers_lowd = new FPTYPE[(size_t)num_topics * num_topics];
    log.print("k-means init method: KMEANSPP\n");
    const FPTYPE best_residual = B_fl_CSC->kmeans_init_on_projected_space((int)num_topics, 1, best_kmeans_seeds, centers_lowd);
    log.print("Best k-means init residual: " + std::to_string(best_residual) + "\n");
    log.next_time_secs("K-means seeds initialization");

    B_fl_CSC->run_lloyds_on_projected_space(num_topics, centers_lowd, NULL, 10);
    // The reference allocates centers[vocab_size * num_topics] here (src/trainer.cpp:284) and hands it through both calls below, but
    // reads nothing of it afterwards (only closest_docs, :566-575): the lifted centres and Lloyd's result stay in device memory.
    FPTYPE* centers = nullptr;
    B_fl_CSC->left_multiply_by_U_Spectra(centers, centers_lowd, num_topics, num_topics);
    delete[] centers_lowd;
    log.next_time_secs("Converging LLoyds k-means on B_k");
    B_fl_CSC->cleanup_after_eigensolver();

    std::vector<doc_id_t>* closest_docs = new std::vector<doc_id_t>[num_topics];
    B_fl_CSC->run_lloyds(num_topics, centers, closest_docs, 10);
    uint64_t closest_docs_sizes_sum = 0;
    for (doc_id_t t = 0; t < num_topics; ++t) closest_docs_sizes_sum += closest_docs[t].size();
    if (closest_docs_sizes_sum != B_fl_CSC->num_docs()) throw std::runtime_error("partition incomplete");  // :567-570
    log.next_time_secs("k-means on B");
    for (doc_id_t topic = 0; topic != num_topics; ++topic)  // :573-575
      for (auto d = closest_docs[topic].begin(); d < closest_docs[topic].end(); ++d) *d = original_cols[*d];
    // ---------------------------------------------------------------------------------------------

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
    if (sample)
      r = (uint64_t)std::floor(ISLE_EPS2_C * ISLE_W0_C * (FPTYPE)num_docs * sample_rate / (FPTYPE)(2.0 * num_topics));
    else
      r = (uint64_t)std::floor(ISLE_EPS2_C * ISLE_W0_C * (FPTYPE)num_docs / (FPTYPE)(2.0 * num_topics));
    FPTYPE* catchword_thresholds = new FPTYPE[(size_t)vocab_size * num_topics];
    std::vector<word_id_t>* catchwords = new std::vector<word_id_t>[num_topics];
    B_fl_CSC->find_catchwords(num_topics, r, catchword_thresholds, catchwords);
    log.next_time_secs("Collecting word freqs in clusters");
    log.next_time_secs("Finding catchwords for clusters");
    FPTYPE* Model = new FPTYPE[(size_t)vocab_size * num_topics];
    std::vector<std::tuple<int, int, doc_id_t>> top_topic_pairs;
    B_fl_CSC->construct_topic_model(Model, num_topics, num_docs, compute_edge_topics ? &top_topic_pairs : NULL);
    log.next_time_secs("Constructing topic vectors");

    // ---- output_cluster_summary (src/trainer.cpp:776-826) -------------------------------------------
    std::vector<std::string> vocab_words;
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
    std::vector<std::vector<std::pair<word_id_t, FPTYPE>>> topwords(num_topics);
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
      log.diag << o.str();
    }
    log.diag << "\n---------------------------\n";
    log.print("\n Avg coherence: " + std::to_string(0.0f) + "\n\n");
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
      log.print(o.str());
    }
    log.next_time_secs("Output summary");

    // ---- write_model_to_file (src/trainer.cpp:656-662) ----------------------------------------------
    auto write_top_words = [&]() {  // output_top_words, :855-868
      std::ofstream out_top_words(log_dir + "/TopWordsPerTopic_catch.txt");
      for (doc_id_t t = 0; t < num_topics; ++t) {
        for (auto& tw : topwords[t]) out_top_words << vocab_words[tw.first] << "\t";
        out_top_words << std::endl;
      }
      log.next_time_secs("Writing top words to file");
    };
    write_top_words();
    write_dense_as_sparse(log_dir + "/M_hat_catch_sparse", Model, vocab_size, num_topics);  // output_model(true), :831-838
    log.next_time_secs("Output model");
    write_top_words();
    log.next_time_secs("Output topwords");

    if (compute_edge_topics) {  // train_edge_topics + write_edgemodel_to_file, :673-693
      std::vector<std::tuple<int, int, uint64_t>> selected_pairs;
      std::vector<FPTYPE> EdgeModel;
      B_fl_CSC->construct_edge_topics(top_topic_pairs, atoi(argc[12]), selected_pairs, EdgeModel);
      log.next_time_secs("Constructing edge topic model");
      write_dense_as_sparse(log_dir + "/EdgeModel_sparse", EdgeModel.data(), vocab_size, selected_pairs.size());
      log.next_time_secs("Output edge model");
    }
    delete[] catchword_thresholds;
    delete[] catchwords;
    delete[] Model;
    log.total("TVSD");
    delete[] centers;
    delete[] closest_docs;
    delete B_fl_CSC;
  } catch (const std::exception& e) {
    // the reference prints the message and still exits with 0 (drivers/ISLETrain.cpp:48-50): a failed training
    // run then looks like a good one to a calling script.  Deliberate deviation: status 1.
    std::cerr << "ISLE Trainer failed: " << e.what() << std::endl;
    return 1;
  } catch (...) {
    std::cerr << "ISLE Trainer failed" << std::endl;
    return 1;
  }
  return 0;
}
