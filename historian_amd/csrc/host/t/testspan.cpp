// Guide alignment of a family through the alignment graph (AlignGraph): pairwise Viterbi alignments as one
// device batch, maximum spanning tree, merge.
//   testspan [-dense] [-kmatchoff] [-kmatch <k>] [-kmatchn <threshold>] [-kmatchband <size>] <fasta> <rate model json> <time>
// Without options this is the reference's testspan: a sparse random pair graph with k-mer seeded envelopes.
// That graph is drawn with std::uniform_int_distribution, whose output depends on the standard library (the
// reference's own Makefile skips the test for that reason).  -dense aligns every pair instead (deterministic),
// -kmatchoff uses full envelopes; -kmatch / -kmatchn / -kmatchband set the k-mer length, the match threshold
// and the band size of the sparse envelopes (the reference's command-line options of the same names).
#include <cstdlib>
#include <cstring>
#include <iostream>
#include "../hx_host.h"
using namespace historian;

int main(int argc, char** argv) {
  bool allPairs = false;
  DiagEnvParams envelopeParams;
  int first = 1;
  for (; first < argc && argv[first][0] == '-'; ++first) {
    const bool hasValue = first + 1 < argc;
    if (!strcmp(argv[first], "-dense")) allPairs = true;
    else if (!strcmp(argv[first], "-kmatchoff")) envelopeParams.sparse = false;
    else if (!strcmp(argv[first], "-kmatch") && hasValue) envelopeParams.kmerLen = atoi(argv[++first]);
    else if (!strcmp(argv[first], "-kmatchn") && hasValue) envelopeParams.kmerThreshold = atoi(argv[++first]);
    else if (!strcmp(argv[first], "-kmatchband") && hasValue) envelopeParams.bandSize = atoi(argv[++first]);
    else break;
  }
  if (argc - first != 3) {
    std::cout << "Usage: " << argv[0] << " [-dense] [-kmatchoff] [-kmatch k] [-kmatchn n] [-kmatchband b] <seqfile> <modelfile> <time>\n";
    return EXIT_FAILURE;
  }
  const vguard<FastSeq> family = readFastSeqs(argv[first]);
  Require(family.size() >= 2, "Sequence file must have at least two sequences");
  RateModel model;
  model.readFile(argv[first + 1]);
  const double branch = atof(argv[first + 2]);
  ForwardMatrix::random_engine rng = ForwardMatrix::newRNG();
  AlignGraph* graph = allPairs ? new AlignGraph(family, model, branch, envelopeParams)
                               : new AlignGraph(family, model, branch, envelopeParams, rng);
  writeFastaSeqs(std::cout, graph->mstGapped());
  delete graph;
  return EXIT_SUCCESS;
}
