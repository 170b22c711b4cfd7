// isle_amd/host/ISLETrain.cpp — command line of the training path on MI355X.
//
// Contract taken from the reference's driver (drivers/ISLETrain.cpp:9-16 and :35-46): twelve positional arguments in this order,
//
//   ISLETrain <tdf_file> <vocab_file> <output_dir> <vocab_size> <num_docs> <max_entries> <num_topics>
//             <apply tf-idf(0/1)> <sample(0/1)> <sample_rate> <edge topics(0/1)> <max_edge_topics>
//
// the usage text and exit status 255 on any other count, and the order of the trainer calls.  Everything else here is this
// repository's: the arguments are described by one table and parsed by it, every number is checked against the range of the type the
// trainer takes it as (the reference passes atol() / atoi() results through casts), a flag is true for any non-zero value as with the
// reference's (bool)atoi(), and a failed run exits with status 1 (the reference prints a message and exits 0, SURVEY App. C #3 —
// deliberate deviation).
//
// The trainer (trainer_hip.h) loads the file in its constructor (ingest and thresholding on the device), train() runs the hot path
// src/trainer.cpp:490-571 plus catchwords and the topic model on the GPU; the writers leave diagnosticLog.txt, timerLog.txt,
// M_hat_catch_sparse, TopWordsPerTopic_catch.txt, EdgeModel_sparse (and HotPathClusters.tsv / HotPathSingularValues.txt).
#include <cerrno>
#include <cstdlib>
#include <limits>

#include "trainer_hip.h"

namespace {

enum class Kind { Text, Count, Flag, Rate };

struct Positional {
  const char* label;  // as it appears in the usage text
  Kind kind;
  unsigned long long max;  // largest value the receiving type holds (counts and flags)
};
constexpr unsigned long long kMaxId = 0xfffffff0ull;  // what the C ABI accepts for a vocabulary / document count (include/isle_hip.h)
constexpr unsigned long long kMaxOffset = (unsigned long long)std::numeric_limits<ISLE::offset_t>::max();
constexpr unsigned long long kMaxInt = (unsigned long long)std::numeric_limits<int>::max();

// the order IS the interface
const Positional kArgs[] = {
    {"<tdf_file>", Kind::Text, 0},          {"<vocab_file>", Kind::Text, 0},        {"<output_dir>", Kind::Text, 0},
    {"<vocab_size>", Kind::Count, kMaxId},  {"<num_docs>", Kind::Count, kMaxId},    {"<max_entries>", Kind::Count, kMaxOffset},
    {"<num_topics>", Kind::Count, kMaxInt}, {"<apply tf-idf(0/1)>", Kind::Flag, kMaxInt}, {"<sample(0/1)>", Kind::Flag, kMaxInt},
    {"<sample_rate>", Kind::Rate, 0},       {"<edge topics(0/1)", Kind::Flag, kMaxInt},   {"<max_edge_topics>", Kind::Count, kMaxInt}};
constexpr int kNumArgs = sizeof(kArgs) / sizeof(kArgs[0]);

struct Parsed {
  std::string text[kNumArgs];
  unsigned long long count[kNumArgs] = {};
  double rate[kNumArgs] = {};
};

[[noreturn]] void usage() {
  // the reference's wording (its unbalanced "<edge topics(0/1)" included): scripts grep for it
  std::cout << "Incorrect usage of ISLETrain. Use: \n"
            << "trainFromFile";
  for (const Positional& a : kArgs) std::cout << ' ' << a.label;
  std::cout << std::endl;
  std::exit(-1);
}

void parse(int n, char** v, Parsed& out) {
  for (int i = 0; i < n; ++i) {
    const char* s = v[i];
    out.text[i] = s;
    const Kind kind = kArgs[i].kind;
    if (kind == Kind::Text) continue;
    char* end = nullptr;
    errno = 0;
    if (kind == Kind::Rate) {
      out.rate[i] = std::strtod(s, &end);
    } else {
      if (*s == '-') throw std::runtime_error(std::string("argument ") + std::to_string(i + 1) + " must not be negative: " + s);
      out.count[i] = std::strtoull(s, &end, 10);
    }
    if (end == s || *end != '\0' || errno == ERANGE) throw std::runtime_error(std::string("argument ") + std::to_string(i + 1) + " is not a number: " + s);
    if (kind != Kind::Rate && out.count[i] > kArgs[i].max)
      throw std::runtime_error(std::string("argument ") + std::to_string(i + 1) + " " + kArgs[i].label + " is out of range (largest accepted " +
                               std::to_string(kArgs[i].max) + "): " + s);
  }
}

}  // namespace

int main(int argc, char** argv) {
  if (argc != kNumArgs + 1) usage();
  try {
    Parsed a;
    parse(kNumArgs, argv + 1, a);
    const bool want_edge_topics = a.count[10] != 0;
    ISLE::ISLETrainer trainer((ISLE::word_id_t)a.count[3], (ISLE::doc_id_t)a.count[4], (ISLE::offset_t)a.count[5], (ISLE::doc_id_t)a.count[6],
                              a.count[7] != 0, a.count[8] != 0, (ISLE::FPTYPE)a.rate[9], ISLE::ISLETrainer::FILE_DATA_LOAD, a.text[0], a.text[1],
                              a.text[2], want_edge_topics, (int)a.count[11]);
    // drivers/ISLETrain.cpp:38-46: the sequence a caller of the class is expected to make
    trainer.train();
    trainer.output_cluster_summary();
    trainer.write_model_to_file();
    if (want_edge_topics) {
      trainer.train_edge_topics();
      trainer.write_edgemodel_to_file();
    }
    trainer.finish_log();
    return 0;
  } catch (const std::exception& e) {
    std::cerr << "ISLE Trainer failed: " << e.what() << std::endl;
  } catch (...) {
    std::cerr << "ISLE Trainer failed" << std::endl;
  }
  return 1;
}
