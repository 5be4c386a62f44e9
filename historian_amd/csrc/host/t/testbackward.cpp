// testbackward <sequences> <modelfile> <xtime> [<ytime>]
// Forward and Backward fills of two leaf profiles: the two scores (equal up to rounding) and every cell whose posterior
// probability exceeds one half, most probable first - the output of the reference's t/testbackward.cpp.
#include "pair_setup.h"
using namespace historian;

int main(int argc, char** argv) {
  if (argc < 4 || argc > 5) return usage(argv[0], "<sequences> <modelfile> <xtime> [<ytime>]");
  const vguard<FastSeq> seqs = twoSequences(argv[1]);
  const PairSetup setup(argv[2], argv[3], argc == 5 ? argv[4] : NULL);
  const Profile x = setup.leaf(seqs[0], 1), y = setup.leaf(seqs[1], 2);
  ForwardMatrix forward(x, y, setup.hmm(), 0, GuideAlignmentEnvelope());
  BackwardMatrix backward(forward);
  std::cout << "Forward score: " << forward.lpEnd << std::endl << "Backward score: " << backward.lpStart() << std::endl;
  for (auto likely = backward.cellsAbovePostProbThreshold(.5); !likely.empty(); likely.pop())
    std::cout << "P" << backward.cellName(likely.top()) << " = " << exp(likely.top().logPostProb) << std::endl;
  return EXIT_SUCCESS;
}
