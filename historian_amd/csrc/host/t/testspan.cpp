// Mirror of the reference's t/testspan.cpp over the GPU-backed pair farm:
//   testspan [-dense] [-kmatchoff] <seqfile> <modelfile> <time>
// prints the merged maximum-spanning-tree alignment.  Without -dense the graph is the reference's sparse
// random graph (std::uniform_int_distribution: standard-library specific, which is why the reference's own
// Makefile skips this test); -dense aligns all pairs (AlignGraph's other constructor), -kmatchoff uses full
// envelopes (DiagEnvParams::sparse = false).
#include <cstdlib>
#include <cstring>
#include <iostream>
#include "../hx_host.h"
using namespace historian;

int main(int argc, char** argv) {
  bool dense = false;
  DiagEnvParams dep;
  int a = 1;
  while (a < argc && argv[a][0] == '-') {
    if (!strcmp(argv[a], "-dense")) dense = true;
    else if (!strcmp(argv[a], "-kmatchoff")) dep.sparse = false;
    else break;
    ++a;
  }
  if (argc - a != 3) {
    std::cout << "Usage: " << argv[0] << " [-dense] [-kmatchoff] <seqfile> <modelfile> <time>\n";
    exit(EXIT_FAILURE);
  }
  const vguard<FastSeq> seqs = readFastSeqs(argv[a]);
  Require(seqs.size() >= 2, "Sequence file must have at least two sequences");
  RateModel rates;
  rates.readFile(argv[a + 1]);
  const double time = atof(argv[a + 2]);
  ForwardMatrix::random_engine generator = ForwardMatrix::newRNG();
  vguard<FastSeq> gapped;
  if (dense) {
    AlignGraph ag(seqs, rates, time, dep);
    gapped = ag.mstGapped();
  } else {
    AlignGraph ag(seqs, rates, time, dep, generator);
    gapped = ag.mstGapped();
  }
  writeFastaSeqs(std::cout, gapped);
  exit(EXIT_SUCCESS);
}
