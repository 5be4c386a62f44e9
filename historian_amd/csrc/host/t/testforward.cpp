// testforward [-all|-hubs] [-best|-matrix|<numberofpaths>] <sequences> <modelfile> <xtime> [<ytime>]
// The Forward fill of two leaf profiles, then the parent profile of: every cell (-matrix), the best path (-best), or n
// sampled paths from a fresh mt19937 - as JSON, like the reference's t/testforward.cpp (golden files: data/testforward.*).
#include "pair_setup.h"
using namespace historian;

int main(int argc, char** argv) {
  if (argc < 6 || argc > 7) return usage(argv[0], "[-all|-hubs] [-best|-matrix|<numberofpaths>] <sequences> <modelfile> <xtime> [<ytime>]");
  const string keep(argv[1]), what(argv[2]);
  if (keep != "-all" && keep != "-hubs") Abort("Unknown strategy: %s", argv[1]);
  const ForwardMatrix::ProfilingStrategy strategy = keep == "-hubs" ? ForwardMatrix::CollapseChains : ForwardMatrix::KeepAll;

  const vguard<FastSeq> seqs = twoSequences(argv[3]);
  const PairSetup setup(argv[4], argv[5], argc == 7 ? argv[6] : NULL);
  const Profile x = setup.leaf(seqs[0], 1), y = setup.leaf(seqs[1], 2);
  ForwardMatrix forward(x, y, setup.hmm(), 0, GuideAlignmentEnvelope());

  Profile parent;
  if (what == "-matrix") parent = forward.makeProfile(everyCell(forward), strategy);
  else if (what == "-best") parent = forward.bestProfile(strategy);
  else {
    auto generator = forward.newRNG();
    parent = forward.sampleProfile(generator, atoi(argv[2]), 0, strategy);
  }
  setup.print(parent);
  return EXIT_SUCCESS;
}
