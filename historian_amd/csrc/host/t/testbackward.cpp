// Mirror of the reference's t/testbackward.cpp on the GPU-backed Forward/Backward matrices.
#include <cstdlib>
#include <iostream>
#include "../hx_host.h"
using namespace historian;

int main(int argc, char** argv) {
  if (argc != 4 && argc != 5) {
    std::cout << "Usage: " << argv[0] << " <sequences> <modelfile> <xtime> [<ytime>]\n";
    exit(EXIT_FAILURE);
  }
  vguard<FastSeq> seqs = readFastSeqs(argv[1]);
  Assert(seqs.size() == 2, "Expected two sequences in file %s", argv[1]);
  RateModel rates;
  rates.readFile(argv[2]);
  ProbModel xprobs(rates, atof(argv[3]));
  ProbModel yprobs(rates, atof(argv[argc > 4 ? 4 : 3]));
  vguard<Vec> eqm = rates.insProb;
  PairHMM hmm(xprobs, yprobs, eqm);
  Profile xprof(1, rates.alphabet, seqs[0], 1);
  Profile yprof(1, rates.alphabet, seqs[1], 2);
  ForwardMatrix forward(xprof, yprof, hmm, 0, GuideAlignmentEnvelope());
  BackwardMatrix backward(forward);
  std::cout << "Forward score: " << forward.lpEnd << std::endl;
  std::cout << "Backward score: " << backward.lpStart() << std::endl;
  auto bestCells = backward.cellsAbovePostProbThreshold(.5);
  while (!bestCells.empty()) {
    std::cout << "P" << backward.cellName(bestCells.top()) << " = " << exp(bestCells.top().logPostProb) << std::endl;
    bestCells.pop();
  }
  exit(EXIT_SUCCESS);
}
