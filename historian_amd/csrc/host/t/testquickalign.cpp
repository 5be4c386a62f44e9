// Pairwise guide alignment of two sequences with the GPU-backed QuickAlignMatrix (same command line and
// output as the reference's testquickalign):
//   testquickalign <two-sequence fasta> <rate model json> <time>
// Full DiagonalEnvelope; the Viterbi alignment is written as gapped FASTA.
#include <cstdlib>
#include <iostream>
#include "../hx_host.h"
using namespace historian;

int main(int argc, char** argv) {
  if (argc != 4) {
    std::cout << "Usage: " << argv[0] << " <seqfile> <modelfile> <time>\n";
    return EXIT_FAILURE;
  }
  const vguard<FastSeq> pair = readFastSeqs(argv[1]);
  Require(pair.size() == 2, "Sequence file must have exactly two sequences");
  RateModel model;
  model.readFile(argv[2]);
  DiagonalEnvelope everyDiagonal(pair[0], pair[1]);
  everyDiagonal.initFull();
  const QuickAlignMatrix viterbi(everyDiagonal, model, atof(argv[3]));
  writeFastaSeqs(std::cout, viterbi.gappedSeq());
  return EXIT_SUCCESS;
}
