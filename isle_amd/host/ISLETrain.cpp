// isle_amd/host/ISLETrain.cpp — the reference's 12-argument CLI (drivers/ISLETrain.cpp:8-51) over the MI355X hot path.
//
//   ISLETrain <tdf_file> <vocab_file> <output_dir> <vocab_size> <num_docs> <max_entries> <num_topics>
//             <apply tf-idf(0/1)> <sample(0/1)> <sample_rate> <edge topics(0/1)> <max_edge_topics>
//
// Runs ingest (host, prestage.h) -> thresholding (device) -> the hot path src/trainer.cpp:490-571 on the GPU, and writes into
// the reference's log directory (src/utils.cpp:28-48) diagnosticLog.txt / timerLog.txt with the reference's line
// formats for these phases.  What comes AFTER the hot path in the reference (catchwords, topic model,
// M_hat_catch_sparse, edge topics: SURVEY §8f next-3) is not built yet: the partition and centres are written to
// HotPathClusters.tsv / HotPathSingularValues.txt instead and the program says so.
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
  (void)vocab_file;

  try {
    const std::string log_dir = log_dir_name(num_topics, output_dir, sample, sample_rate, tf_idf);
    struct stat st;
    if (stat(log_dir.c_str(), &st) == -1) mkdir(log_dir.c_str(), S_IRWXU);
    else std::cerr << "Subdir exists already" << std::endl;
    Logs log(log_dir);

    std::vector<prestage::DocWordEntry> entries;
    prestage::read_tdf(tdf_file, (uint64_t)max_entries, entries);
    {
      std::ostringstream s;  // src/trainer.cpp:130-143
      s << "\n<<<<<<<<<<<<\t" << tdf_file << "\t>>>>>>>>>>>>\n\n"
        << std::setfill('.') << std::setw(10) << std::left << std::setw(15) << std::left << "#Entries" << entries.size() << "\n"
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
    prestage::Csc A;
    float avg_doc_sz = 0.f;
    uint64_t nz_docs = 0;
    prestage::build_A(entries, vocab_size, num_docs, A, &avg_doc_sz, &nz_docs);
    std::cout << "Entries in sparse matrix: " << A.offs[num_docs] << std::endl << "Average document size: " << avg_doc_sz << std::endl;
    log.next_time_secs("Populating CSC");

    // src/trainer.cpp:430-485 on the device: thresholds from the whole corpus, B built in HBM
    std::vector<uint32_t> rows32(A.rows.begin(), A.rows.end());
    std::vector<doc_id_t> original_cols;
    uint64_t entries_above_threshold = 0;
    FPSparseMatrixHip* B_fl_CSC = FPSparseMatrixHip::from_counts(vocab_size, num_docs, A.vals.data(), rows32.data(), A.offs.data(), num_topics,
                                                                 sample ? (double)sample_rate : 0.0, original_cols, nullptr,
                                                                 &entries_above_threshold);
    std::vector<uint32_t>().swap(rows32);
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
    FPTYPE* centers = new FPTYPE[(size_t)vocab_size * num_topics];
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
    log.print("NOTE: catchwords / topic model / M_hat_catch_sparse / edge topics are not built in this implementation yet;\n"
              "      wrote HotPathClusters.tsv and HotPathSingularValues.txt (hot path src/trainer.cpp:490-571 complete).\n");
    log.total("TVSD");
    delete[] centers;
    delete[] closest_docs;
    delete B_fl_CSC;
  } catch (const std::exception& e) {
    std::cerr << "ISLE Trainer failed: " << e.what() << std::endl;  // reference: message only, exit code 0 (drivers/ISLETrain.cpp:48-50)
  } catch (...) {
    std::cerr << "ISLE Trainer failed" << std::endl;
  }
}
