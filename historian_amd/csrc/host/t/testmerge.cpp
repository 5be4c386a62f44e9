// Merges alignments that share sequences (same command line and output as the reference's testmerge):
//   testmerge <alignment.fa> <alignment.fa> ...
// Rows are identified by sequence name across the files; the merged alignment goes to stdout as FASTA.
// Host only.
#include <cstdlib>
#include <iostream>
#include "../hx_host.h"
using namespace historian;

namespace {
struct RowRegistry {
  map<string, AlignRowIndex> rowOf;
  vguard<FastSeq> sequences;
  AlignRowIndex lookup(const FastSeq& ungapped) {
    const auto known = rowOf.find(ungapped.name);
    if (known != rowOf.end()) return known->second;
    const AlignRowIndex row = sequences.size();
    rowOf[ungapped.name] = row;
    sequences.push_back(ungapped);
    return row;
  }
};
}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) {
    std::cout << "Usage: " << argv[0] << " <align1> <align2> ...\n";
    return EXIT_FAILURE;
  }
  RowRegistry rows;
  vguard<AlignPath> inputs;
  for (int f = 1; f < argc; ++f) {
    const Alignment file(readFastSeqs(argv[f]));
    AlignPath renumbered;
    for (size_t k = 0; k < file.ungapped.size(); ++k) renumbered[rows.lookup(file.ungapped[k])] = file.path.at(k);
    inputs.push_back(renumbered);
  }
  writeFastaSeqs(std::cout, Alignment(rows.sequences, alignPathMerge(inputs)).gapped());
  return EXIT_SUCCESS;
}
