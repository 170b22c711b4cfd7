// isle_amd/host/ISLEInfer.cpp — the reference's inference driver (drivers/ISLEInfer.cpp) over the C ABI of include/isle_hip.h.
//
//   ISLEInfer <sparse_model_file> <infer_file> <output_dir> <num_topics> <vocab_size> <min_doc_id_in_infer_file>
//             <max_doc_id_in_infer_file> <nnzs_in_infer_file> <nnzs_in_sparse_model_file> <iters>[0 for default]
//             <Lifschitz_constant_guess>[0 for default]
//
// Same argument list (11 arguments, usage + exit(-1) otherwise, drivers/ISLEInfer.cpp:11-20), same inputs
// (M_hat_catch_sparse as written by ISLETrain: "<topic>\t<word>\t<weight>", 1-based, src/infer.cpp:125-190; tdf documents),
// same outputs: per block of 1,000,000 documents a file top_topics_iters_<iters>_Lf_<Lf>_doc_<first>_to_<last> with
// "<doc>\t<topic>\t<weight>" for the (at most five) topics heavier than 1 / num_topics (:100-112), and the summary lines on
// stdout (:159-176).  The arithmetic runs in isle_hip_infer (isle_amd/csrc/infer.hip); there is no CPU fallback.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/isle_hip.h"
#include "prestage.h"

namespace {

// read_sparse_model, src/infer.cpp:125-190 (mmap branch): three blank-separated fields per line, the weight as
// <digits>[.<digits>] assembled in FPTYPE as before + after * 0.1^n
void read_sparse_model(const std::string& path, uint64_t num_topics, uint64_t vocab_size, unsigned base, std::vector<float>& model_by_word,
                       uint64_t* entries) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("cannot open model file " + path);
  std::fseek(f, 0, SEEK_END);
  const long sz = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<char> buf((size_t)sz);
  if (sz && std::fread(buf.data(), 1, (size_t)sz, f) != (size_t)sz) {
    std::fclose(f);
    throw std::runtime_error("short read on " + path);
  }
  std::fclose(f);
  model_by_word.assign(vocab_size * num_topics, 0.f);
  uint64_t topic = 0, word = 0, n = 0;
  bool was_ws = false, before_dec = true, any = false;
  float vb = 0.f, va = 0.f;
  int pos = 0, state = 1;
  auto flush = [&]() {
    if (!any) return;
    if (state != 3) throw std::runtime_error("Bad line in sparse model file");
    if (topic < base || word < base || topic - base >= num_topics || word - base >= vocab_size)
      throw std::runtime_error("sparse model entry out of range");
    model_by_word[num_topics * (word - base) + (topic - base)] = (float)((double)vb + (double)va * std::pow(0.1, pos));
    ++n;
  };
  for (long i = 0; i < sz; ++i) {
    const char ch = buf[(size_t)i];
    switch (ch) {
      case '\r': break;
      case '\n':
        flush();
        state = 1;
        word = topic = 0;
        pos = 0;
        va = vb = 0.f;
        before_dec = true;
        was_ws = false;
        any = false;
        break;
      case ' ':
      case '\t': was_ws = true; break;
      case '.':
        if (state != 3 || !before_dec) throw std::runtime_error("Bad format in sparse model file");
        before_dec = false;
        break;
      default:
        if (ch < '0' || ch > '9') throw std::runtime_error("Bad format in sparse model file");
        if (was_ws && any) ++state;
        was_ws = false;
        any = true;
        if (state == 1) topic = topic * 10 + (uint64_t)(ch - '0');
        else if (state == 2) word = word * 10 + (uint64_t)(ch - '0');
        else if (state == 3) {
          if (before_dec) vb = vb * 10 + (float)(ch - '0');
          else {
            va = va * 10 + (float)(ch - '0');
            ++pos;
          }
        } else throw std::runtime_error("Bad line in sparse model file");
    }
  }
  flush();  // no trailing newline
  *entries = n;
}

// The weight as the reference's writer prints it (MMappedOutput::concat_float, include/utils.h:421-478): "0.0" for zero, a sign, the integer
// part, a point, and six fraction digits taken one at a time by multiplying the single-precision remainder by ten (truncating).
void append_weight(std::string& out, float w) {
  if (w == 0.0f) {
    out += "0.0";
    return;
  }
  if (w < 0.f) {
    out += '-';
    w = -w;
  }
  out += std::to_string((unsigned int)w);
  out += '.';
  float rest = w - (float)((int)w);
  for (int place = 0; place < 6; ++place) {
    rest *= 10;
    const int digit = (int)rest;
    out += (char)('0' + digit);
    rest -= digit;
  }
}

// the eleven positional arguments (drivers/ISLEInfer.cpp:11-33: this order is the interface)
struct InferArgs {
  std::string model_path, docs_path, out_dir;
  uint64_t topics = 0, vocab = 0, first_doc = 0, last_doc = 0, doc_entries = 0;
  int iterations = 0;
  float lipschitz = 0.f;
};

[[noreturn]] void usage() {
  // the reference's wording, as scripts may grep for it
  std::cout << "Incorrect usage of ISLEInfer. Use: \n"
            << "inferFromFile <sparse_model_file> <infer_file> <output_dir> "
            << "<num_topics> <vocab_size> <min_doc_id_in_infer_file> <max_doc_id_in_infer_file>"
            << "<nnzs_in_infer_file> <nnzs_in_sparse_model_file> "
            << "<iters>[0 for default]  "
            << "Lifschitz_constant_guess>[0 for default]" << std::endl;
  std::exit(-1);
}

InferArgs read_args(int argc, char** argv) {
  if (argc != 12) usage();
  InferArgs a;
  a.model_path = argv[1];
  a.docs_path = argv[2];
  a.out_dir = argv[3];
  a.topics = std::strtoull(argv[4], nullptr, 10);
  a.vocab = std::strtoull(argv[5], nullptr, 10);
  a.first_doc = std::strtoull(argv[6], nullptr, 10);
  a.last_doc = std::strtoull(argv[7], nullptr, 10);
  a.doc_entries = std::strtoull(argv[8], nullptr, 10);
  // argv[9], the model's entry count, is not needed: the model file is read to its end
  a.iterations = (int)std::strtol(argv[10], nullptr, 10);
  if (a.iterations == 0) a.iterations = 15;  // INFER_ITERS_DEFAULT, include/hyperparams.h:81
  a.lipschitz = std::strtof(argv[11], nullptr);
  if (a.lipschitz == 0.0f) a.lipschitz = 10.0f;  // INFER_LF_DEAFULT :82
  if (a.topics < 1 || a.vocab < 1 || a.last_doc < a.first_doc) throw std::runtime_error("bad <num_topics> / <vocab_size> / document range");
  return a;
}

}  // namespace

int main(int argc, char** argv) {
  try {
    const InferArgs args = read_args(argc, argv);
    const std::string &sparse_model_file = args.model_path, &infer_file = args.docs_path, &output_dir = args.out_dir;
    const uint64_t num_topics = args.topics, vocab_size = args.vocab, doc_begin = args.first_doc, doc_end = args.last_doc;
    const uint64_t max_entries = args.doc_entries;
    const int iters = args.iterations;
    const float Lfguess = args.lipschitz;

    std::cout << "Loading sparse model file: " << sparse_model_file << std::endl;
    std::vector<float> model_by_word;
    uint64_t model_entries = 0;
    read_sparse_model(sparse_model_file, num_topics, vocab_size, 1, model_by_word, &model_entries);

    std::cout << "Loading data from inference file: " << infer_file << std::endl;
    std::vector<ISLE::prestage::DocWordEntry> entries;
    ISLE::prestage::read_tdf(infer_file, max_entries, entries);
    const uint64_t num_docs = doc_end - doc_begin;  // drivers/ISLEInfer.cpp:49 (the last id of the range is not a document of its own)
    for (auto& e : entries) {  // :58: entries[i].doc -= (doc_begin - 1), ids already 0-based here
      if (e.doc + 1 < doc_begin) throw std::runtime_error("document id below <min_doc_id_in_infer_file>");
      e.doc -= (doc_begin - 1);
    }
    ISLE::prestage::Csc A;
    float avg_doc_sz = 0.f;
    uint64_t nz_docs = 0;
    ISLE::prestage::build_A(entries, vocab_size, num_docs, A, &avg_doc_sz, &nz_docs);  // sort, de-duplicate, populate_CSC (:50-59)
    std::vector<uint32_t> rows32(A.rows.begin(), A.rows.end());

    isle_ctx* ctx = isle_hip_create(0);
    if (!ctx) throw std::runtime_error("no HIP device (there is no CPU fallback)");
    std::vector<int32_t> top_topic(num_docs * 5);
    std::vector<float> top_weight(num_docs * 5), llh(num_docs * 2);
    uint64_t nconverged = 0;
    std::cout << "Creating inference engine" << std::endl;
    if (isle_hip_infer(ctx, vocab_size, (int)num_topics, model_by_word.data(), num_docs, A.vals.size(), A.vals.data(), rows32.data(),
                       A.offs.data(), iters, Lfguess, avg_doc_sz, nullptr, top_topic.data(), top_weight.data(), llh.data(), &nconverged)) {
      const std::string msg = isle_hip_last_error(ctx);
      isle_hip_destroy(ctx);
      throw std::runtime_error(msg);
    }
    isle_hip_destroy(ctx);

    const uint64_t block = 1000000;  // :66
    for (uint64_t b0 = 0; b0 < num_docs; b0 += block) {
      const uint64_t b1 = std::min(num_docs, b0 + block);
      const std::string name = output_dir + "/top_topics_iters_" + std::to_string(iters) + "_Lf_" + std::to_string(Lfguess) + "_doc_" +
                               std::to_string(doc_begin + b0) + "_to_" + std::to_string(doc_begin + b1);
      FILE* f = std::fopen(name.c_str(), "wb");
      if (!f) throw std::runtime_error("cannot open " + name);
      std::string buf;
      for (uint64_t d = b0; d < b1; ++d)
        for (int i = 0; i < 5 && top_topic[d * 5 + i] >= 0; ++i) {
          buf += std::to_string(d + doc_begin);
          buf += '\t';
          buf += std::to_string(1 + top_topic[d * 5 + i]);
          buf += '\t';
          append_weight(buf, top_weight[d * 5 + i]);
          buf += '\n';
          if (buf.size() > (1u << 24)) {
            std::fwrite(buf.data(), 1, buf.size(), f);
            buf.clear();
          }
        }
      std::fwrite(buf.data(), 1, buf.size(), f);
      std::fclose(f);
    }
    std::cout << "Number of docs for which inference converged: " << nconverged << " (of " << num_docs << ")" << std::endl;
    float sum_first = 0.f, sum_second = 0.f;  // :165-171 (fp32 sums in document order)
    for (uint64_t d = 0; d < num_docs; ++d) {
      sum_first += llh[2 * d];
      sum_second += llh[2 * d + 1];
    }
    std::cout << "Avg LLH per document for converged docs: " << ((float)num_docs / nconverged) * sum_first / nconverged << std::endl;
    std::cout << "Avg LLH per word: " << sum_second / max_entries << std::endl;
  } catch (const std::exception& e) {
    std::cerr << "ISLEInfer: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
