// isle_amd/host/ISLETrain.cpp — the reference's 12-argument CLI (drivers/ISLETrain.cpp:8-51) over the MI355X path.
//
//   ISLETrain <tdf_file> <vocab_file> <output_dir> <vocab_size> <num_docs> <max_entries> <num_topics>
//             <apply tf-idf(0/1)> <sample(0/1)> <sample_rate> <edge topics(0/1)> <max_edge_topics>
//
// The same call sequence as the reference's main (:34-46) on ISLE::ISLETrainer (trainer_hip.h): the constructor loads the file (ingest
// and thresholding on the device), train() runs the hot path src/trainer.cpp:490-571, catchwords and the topic model on the GPU, the
// writers leave the reference's files in its log directory (src/utils.cpp:28-48): diagnosticLog.txt, timerLog.txt, M_hat_catch_sparse,
// TopWordsPerTopic_catch.txt, EdgeModel_sparse; extra files HotPathClusters.tsv / HotPathSingularValues.txt.
#include "trainer_hip.h"

using namespace ISLE;

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
  const bool tf_idf = atoi(argc[8]);
  const bool sample = atoi(argc[9]);
  const FPTYPE sample_rate = (FPTYPE)atof(argc[10]);
  const bool compute_edge_topics = atoi(argc[11]);
  const int max_edge_topics = atoi(argc[12]);

  try {
    ISLETrainer trainer(vocab_size, num_docs, max_entries, num_topics, tf_idf, sample, sample_rate, ISLETrainer::data_ingest::FILE_DATA_LOAD, tdf_file,
                        vocab_file, output_dir, compute_edge_topics, max_edge_topics);
    trainer.train();
    trainer.output_cluster_summary();
    trainer.write_model_to_file();
    if (compute_edge_topics) {
      trainer.train_edge_topics();
      trainer.write_edgemodel_to_file();
    }
    trainer.finish_log();
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
