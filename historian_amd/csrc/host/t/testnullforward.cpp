// testnullforward <modelfile> <xtime> [<ytime>]
// Forward over two small profiles in which one emitting state each has been turned into a null state by hand (the
// reference's t/testnullforward.cpp: its only test of the null / wait / ready branches of the fill); every cell kept.
#include "pair_setup.h"
using namespace historian;

int main(int argc, char** argv) {
  if (argc < 3 || argc > 4) return usage(argv[0], "<modelfile> <xtime> [<ytime>]");
  const PairSetup setup(argv[1], argv[2], argc == 4 ? argv[3] : NULL);
  FastSeq sx, sy;
  sx.name = "x"; sx.seq = "acg";
  sy.name = "y"; sy.seq = "cag";
  Profile x = setup.leaf(sx, 1), y = setup.leaf(sy, 2);
  x.state[2].lpAbsorb.clear();          // x's second residue and y's first no longer absorb: null states
  y.state[1].lpAbsorb.clear();
  ForwardMatrix forward(x, y, setup.hmm(), 0, GuideAlignmentEnvelope());
  Profile parent = forward.makeProfile(everyCell(forward), ForwardMatrix::KeepAll);
  setup.print(parent);
  return EXIT_SUCCESS;
}
