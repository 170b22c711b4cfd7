// isle_amd/host/hot_path_main.cpp — the hot slice of ISLETrainer::train() (src/trainer.cpp:490-571) as a
// standalone C++ program over ISLE::FPSparseMatrixHip, with the reference's own log lines
// (the "Eigvals:" block in the format of include/logUtils.h:101-122; trainer.cpp:490 "Frob(B_fl_CSC):").
//
//   hot_path_main <B.bin> <num_topics> <out.bin>
// B.bin  : u64 V, u64 D, u64 nnz, f32 vals[nnz], u64 rows[nnz], i64 offs[D+1]   (the reference's 8-byte types)
// out.bin: u64 k, f32 evalues[k], u64 seeds[k], f32 centers[V*k] (col-major), u64 sizes[k], then the partition
//          as k lists (u64 ids, ascending), in topic order
// Exit code != 0 on failure (unlike drivers/ISLETrain.cpp:48-50, which swallows every exception).
#include <cmath>
#include <cstdio>
#include <string>

#include "fpsparse_hip.h"

using namespace ISLE;

// The "Eigvals:" log block of the reference's trainer (its text is a parity surface: include/logUtils.h:101-122 defines the
// format — singular values as "(i): sigma" separated by tabs, then the running sum of the eigenvalues after every full hundred).
// Floats are printed as the reference's ostream does (six significant digits, "%g"); hundreds are summed first and then
// chained, which is the rounding order the reference's figures have.
static void log_spectrum(const std::vector<FPTYPE>& lambda, doc_id_t k) {
  std::string line = "Eigvals:  ";
  char buf[64];
  for (doc_id_t i = 0; i < k; ++i) {
    std::snprintf(buf, sizeof buf, "(%llu): %g\t", (unsigned long long)i, (double)std::sqrt(lambda[i]));
    line += buf;
  }
  line += "\n";
  FPTYPE running = 0;
  for (doc_id_t h = 0; (h + 1) * 100 <= k; ++h) {
    FPTYPE hundred = 0;
    for (doc_id_t i = h * 100; i < (h + 1) * 100; ++i) hundred += lambda[i];
    running += hundred;
    std::snprintf(buf, sizeof buf, "Sum of Top-%llu eig vals: %g\n", (unsigned long long)((h + 1) * 100), (double)running);
    line += buf;
  }
  std::fputs(line.c_str(), stdout);
  std::fflush(stdout);
}

int main(int argc, char** argv) {
  if (argc != 4) {
    std::cerr << "usage: hot_path_main <B.bin> <num_topics> <out.bin>\n";
    return 2;
  }
  try {
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) throw std::runtime_error("cannot open input");
    uint64_t hdr[3];
    if (std::fread(hdr, 8, 3, f) != 3) throw std::runtime_error("short header");
    const word_id_t vocab_size = hdr[0];
    const doc_id_t num_docs = hdr[1];
    const offset_t nnz = (offset_t)hdr[2];
    const doc_id_t num_topics = std::atol(argv[2]);
    FPSparseMatrixHip* B_fl_CSC = new FPSparseMatrixHip(vocab_size, num_docs);
    B_fl_CSC->allocate(nnz);
    if (std::fread(B_fl_CSC->vals_CSC, 4, nnz, f) != (size_t)nnz || std::fread(B_fl_CSC->rows_CSC, 8, nnz, f) != (size_t)nnz ||
        std::fread(B_fl_CSC->offsets_CSC, 8, num_docs + 1, f) != num_docs + 1)
      throw std::runtime_error("short matrix file");
    std::fclose(f);

    // ---- src/trainer.cpp:490-571 ----------------------------------------------------------------
    std::cout << "Frob(B_fl_CSC): " << std::to_string(B_fl_CSC->frobenius()) << "\n";
    std::vector<FPTYPE> evalues;
    B_fl_CSC->initialize_for_eigensolver(num_topics);
    B_fl_CSC->compute_block_ks(num_topics, evalues);
    std::cout.flush();
    log_spectrum(evalues, num_topics);
    auto& B_fl = B_fl_CSC;

    std::vector<doc_id_t> best_kmeans_seeds;
    int num_centers_lowd = (int)num_topics;
    FPTYPE* centers_lowd = new FPTYPE[(size_t)num_topics * (size_t)num_centers_lowd];
    std::cout << "k-means init method: KMEANSPP\n";
    FPTYPE best_residual = B_fl->kmeans_init_on_projected_space(num_centers_lowd, 1 /*KMEANS_INIT_REPS*/, best_kmeans_seeds, centers_lowd);
    std::cout << "Best k-means init residual: " << std::to_string(best_residual) << "\n";

    B_fl->run_lloyds_on_projected_space(num_centers_lowd, centers_lowd, NULL, 10 /*MAX_KMEANS_LOWD_REPS*/);
    FPTYPE* centers = new FPTYPE[(size_t)vocab_size * num_topics];
    B_fl->left_multiply_by_U_Spectra(centers, centers_lowd, num_topics, num_topics);
    delete[] centers_lowd;
    B_fl->cleanup_after_eigensolver();

    std::vector<doc_id_t>* closest_docs = new std::vector<doc_id_t>[num_topics];
    B_fl->run_lloyds(num_topics, centers, closest_docs, 10 /*MAX_KMEANS_REPS*/);
    uint64_t closest_docs_sizes_sum = 0;
    for (doc_id_t t = 0; t < num_topics; ++t) closest_docs_sizes_sum += closest_docs[t].size();
    assert(closest_docs_sizes_sum == B_fl->num_docs());  // :567-570
    if (closest_docs_sizes_sum != B_fl->num_docs()) throw std::runtime_error("partition incomplete");
    // ---------------------------------------------------------------------------------------------

    FILE* o = std::fopen(argv[3], "wb");
    if (!o) throw std::runtime_error("cannot open output");
    uint64_t k64 = num_topics;
    std::fwrite(&k64, 8, 1, o);
    std::fwrite(evalues.data(), 4, num_topics, o);
    std::fwrite(best_kmeans_seeds.data(), 8, num_topics, o);
    std::fwrite(centers, 4, (size_t)vocab_size * num_topics, o);
    for (doc_id_t t = 0; t < num_topics; ++t) {
      uint64_t s = closest_docs[t].size();
      std::fwrite(&s, 8, 1, o);
    }
    for (doc_id_t t = 0; t < num_topics; ++t) std::fwrite(closest_docs[t].data(), 8, closest_docs[t].size(), o);
    std::fclose(o);
    delete[] centers;
    delete[] closest_docs;
    delete B_fl_CSC;
  } catch (const std::exception& e) {
    std::cerr << "ISLE hot path failed: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
