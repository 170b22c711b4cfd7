// isle_amd/host/trainer_feed_main.cpp — ISLE::ISLETrainer driven the way drivers/trainer_export.cpp (:31-98) drives it:
// CreateTrainer(ITERATIVE_DATA_LOAD) -> feedData per document -> finalizeData -> Train -> GetBasicModel.  Test driver
// (tests/test_gpu_cli.py): reads a tdf file on the host, feeds it document by document in SHUFFLED order with the words of a document
// in reverse order, and writes the basic model as "<word> <topic> <weight %.9g>" lines — which must equal what the file-loading CLI
// leaves for the same corpus.
//   trainer_feed_main <tdf_file> <output_dir> <vocab_size> <num_docs> <num_topics> <model_out>
#include <map>
#include <random>

#include "trainer_hip.h"

using namespace ISLE;

int main(int argc, char** argv) {
  if (argc != 7) {
    std::cerr << "usage: trainer_feed_main <tdf_file> <output_dir> <vocab_size> <num_docs> <num_topics> <model_out>\n";
    return 2;
  }
  const word_id_t vocab_size = atol(argv[3]);
  const doc_id_t num_docs = atol(argv[4]);
  const doc_id_t num_topics = atol(argv[5]);
  try {
    std::vector<std::vector<std::pair<word_id_t, count_t>>> docs(num_docs);
    {
      std::ifstream in(argv[1]);
      uint64_t d, w, cnt;
      while (in >> d >> w >> cnt) docs.at(d - 1).push_back(std::make_pair((word_id_t)w, (count_t)cnt));  // tdf ids are 1-based; feed_data takes the word id as it is (src/trainer.cpp:224)
    }
    ISLETrainer trainer(vocab_size, num_docs, 0, num_topics, false, false, 0.0f, ISLETrainer::data_ingest::ITERATIVE_DATA_LOAD, argv[1], "", argv[2]);
    std::vector<doc_id_t> order(num_docs);
    std::iota(order.begin(), order.end(), (doc_id_t)0);
    std::mt19937_64 rng(5);
    std::shuffle(order.begin(), order.end(), rng);
    for (doc_id_t d : order) {
      std::vector<word_id_t> words;
      std::vector<count_t> counts;
      for (auto it = docs[d].rbegin(); it != docs[d].rend(); ++it) {
        words.push_back(it->first);
        counts.push_back(it->second);
      }
      trainer.feed_data(d, words.data(), counts.data(), (offset_t)words.size());
    }
    trainer.finalize_data();
    trainer.train();
    std::vector<FPTYPE> model((size_t)vocab_size * num_topics);
    trainer.get_basic_model(model.data());
    FILE* f = std::fopen(argv[6], "w");
    if (!f) throw std::runtime_error("cannot open the model output");
    for (doc_id_t t = 0; t < num_topics; ++t)
      for (word_id_t w = 0; w < vocab_size; ++w)
        if (model[(size_t)t * vocab_size + w] > 0.f) std::fprintf(f, "%llu %llu %.9g\n", (unsigned long long)w, (unsigned long long)t, model[(size_t)t * vocab_size + w]);
    std::fclose(f);
  } catch (const std::exception& e) {
    std::cerr << "trainer_feed_main failed: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
