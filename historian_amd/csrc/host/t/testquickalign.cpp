// Mirror of the reference's t/testquickalign.cpp over the GPU-backed QuickAlignMatrix:
//   testquickalign <seqfile> <modelfile> <time>
// prints the pairwise Viterbi alignment as gapped FASTA (reference data/testquickalign.out.fa).
#include <cstdlib>
#include <iostream>
#include "../hx_host.h"
using namespace historian;

int main(int argc, char** argv) {
  if (argc != 4) {
    std::cout << "Usage: " << argv[0] << " <seqfile> <modelfile> <time>\n";
    exit(EXIT_FAILURE);
  }
  const vguard<FastSeq> seqs = readFastSeqs(argv[1]);
  Require(seqs.size() == 2, "Sequence file must have exactly two sequences");
  RateModel rates;
  rates.readFile(argv[2]);
  const double time = atof(argv[3]);
  DiagonalEnvelope env(seqs[0], seqs[1]);
  env.initFull();
  QuickAlignMatrix mx(env, rates, time);
  writeFastaSeqs(std::cout, mx.gappedSeq());
  exit(EXIT_SUCCESS);
}
