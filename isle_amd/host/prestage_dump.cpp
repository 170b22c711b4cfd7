// isle_amd/host/prestage_dump.cpp — CPU-only helper for tests: runs the ISLETrain pre-stages (tdf -> A -> B) and dumps B.
//   prestage_dump <tdf> <vocab_size> <num_docs> <max_entries> <num_topics> <sample_rate or 0> <out.bin>
// out.bin: u64 V, u64 D_B, u64 nnz, u64 entries_above_threshold, f32 vals[nnz], u64 rows[nnz], i64 offs[D_B+1], u64 original_cols[D_B], f32 zetas[V]
#include <cstdlib>
#include <iostream>

#include "prestage.h"

int main(int argc, char** argv) {
  if (argc != 8) return 2;
  try {
    using namespace ISLE::prestage;
    std::vector<DocWordEntry> e;
    read_tdf(argv[1], std::atol(argv[4]), e);
    Csc A;
    float avg;
    uint64_t nz;
    build_A(e, std::atol(argv[2]), std::atol(argv[3]), A, &avg, &nz);
    Thresholded T;
    threshold(A, avg, nz, std::atol(argv[5]), std::atof(argv[6]), 0, T);
    FILE* o = std::fopen(argv[7], "wb");
    uint64_t hdr[4] = {T.B.V, T.B.D, (uint64_t)T.B.offs.back(), T.entries_above_threshold};
    std::fwrite(hdr, 8, 4, o);
    std::fwrite(T.B.vals.data(), 4, T.B.vals.size(), o);
    std::fwrite(T.B.rows.data(), 8, T.B.rows.size(), o);
    std::fwrite(T.B.offs.data(), 8, T.B.offs.size(), o);
    std::fwrite(T.original_cols.data(), 8, T.original_cols.size(), o);
    std::fwrite(T.zetas.data(), 4, T.zetas.size(), o);
    std::fclose(o);
  } catch (const std::exception& ex) {
    std::cerr << ex.what() << std::endl;
    return 1;
  }
  return 0;
}
