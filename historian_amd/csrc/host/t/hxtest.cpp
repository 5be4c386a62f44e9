// One driver for the small checks of the host mirror; every sub-command prints what the reference's golden file for it holds
// (tests/test_host_mirror.py diffs the output byte for byte):
//   hxtest logsumexp [-slow|-fast]                 the 20 x 20 grid of data/logsumexp.txt (reference Makefile:206-208)
//   hxtest seqprofile <alphabet> <sequence>        leaf Profile as JSON (Makefile:239-240)
//   hxtest quickalign <pair.fa> <model.json> <t>   guide-alignment Viterbi of two sequences as gapped FASTA (Makefile:278-279)
//   hxtest expm <model.json> <t>                   exp(R t) of every mixture component as hex floats, row by row
//   hxtest branch <pair.fa> <model.json> <t>       the two sequences as parent and child of one branch (next row N4): Viterbi
//                                                  and Forward log-likelihoods as hex floats, then the best alignment
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include "../hx_host.h"
using namespace historian;

namespace {

struct Command {
  const char* name;
  int minArgs, maxArgs;
  const char* usage;
  int (*run)(int nArgs, char** args);
};

int lseGrid(int nArgs, char** args) {
  const bool exactSum = nArgs == 1 && strcmp(args[0], "-slow") == 0;
  if (nArgs == 1 && !exactSum && strcmp(args[0], "-fast") != 0) return -1;
  std::cerr << "(running in " << (exactSum ? "slow" : "fast") << " mode)" << std::endl;
  // twenty values per axis, accumulated exactly as the golden file's producer did: repeated addition of 0.1
  vguard<double> axis;
  for (double v = 0; v < 2; v += .1) axis.push_back(v);
  for (double a : axis)
    for (double b : axis)
      std::cout << a << ' ' << b << ' ' << (exactSum ? log_sum_exp_slow(a, b) : log_sum_exp(a, b)) << std::endl;
  return 0;
}

int leafProfileJson(int, char** args) {
  FastSeq leaf;
  leaf.seq = args[1];
  Profile(1, string(args[0]), leaf, 0).writeJson(std::cout);
  return 0;
}

int guidePair(int, char** args) {
  const vguard<FastSeq> two = readFastSeqs(args[0]);
  Require(two.size() == 2, "Sequence file must have exactly two sequences");
  RateModel rates;
  rates.readFile(args[1]);
  DiagonalEnvelope all(two[0], two[1]);
  all.initFull();
  writeFastaSeqs(std::cout, QuickAlignMatrix(all, rates, atof(args[2])).gappedSeq());
  return 0;
}

int substitutionMatrix(int, char** args) {
  RateModel rates;
  rates.readFile(args[0]);
  for (const Mat& m : rates.getSubProbMatrix(atof(args[1])))
    for (const Vec& row : m) {
      for (double v : row) printf("%a ", v);
      printf("\n");
    }
  return 0;
}

int branchPair(int, char** args) {
  const vguard<FastSeq> two = readFastSeqs(args[0]);
  Require(two.size() == 2, "Sequence file must have exactly two sequences");
  RateModel rates;
  rates.readFile(args[1]);
  const int C = (int)rates.subRate.size();
  const auto parent = TreeAlignFuncs::leafPWM(two[0], rates.alphabet, C), child = TreeAlignFuncs::leafPWM(two[1], rates.alphabet, C);
  vguard<SeqIdx> xPos(parent.size() + 1), yPos(child.size() + 1);
  for (size_t k = 0; k < xPos.size(); ++k) xPos[k] = (SeqIdx)k;
  for (size_t k = 0; k < yPos.size(); ++k) yPos[k] = (SeqIdx)k;
  const GuideAlignmentEnvelope everywhere;
  const Refiner::BranchMatrix viterbi(rates, parent, child, atof(args[2]), everywhere, xPos, yPos, 0, 1);
  const Sampler::BranchMatrix forward(rates, parent, child, atof(args[2]), everywhere, xPos, yPos, 0, 1);
  printf("viterbi %a\nforward %a\n", viterbi.lpEnd, forward.lpEnd);
  const AlignPath best = viterbi.best();
  for (AlignRowIndex row = 0; row < 2; ++row) {
    size_t next = 0;
    for (bool here : best.at(row)) putchar(here ? two[row].seq[next++] : '-');
    putchar('\n');
  }
  return 0;
}

const Command commands[] = {
    {"logsumexp", 0, 1, "[-slow|-fast]", lseGrid},
    {"seqprofile", 2, 2, "<alphabet> <sequence>", leafProfileJson},
    {"quickalign", 3, 3, "<seqfile> <modelfile> <time>", guidePair},
    {"expm", 2, 2, "<modelfile> <time>", substitutionMatrix},
    {"branch", 3, 3, "<seqfile> <modelfile> <time>", branchPair},
};

}  // namespace

int main(int argc, char** argv) {
  const Command* chosen = nullptr;
  for (const Command& c : commands)
    if (argc > 1 && strcmp(argv[1], c.name) == 0) chosen = &c;
  const int nArgs = argc - 2;
  if (!chosen || nArgs < chosen->minArgs || nArgs > chosen->maxArgs || chosen->run(nArgs, argv + 2) != 0) {
    std::cout << "Usage:\n";
    for (const Command& c : commands)
      if (!chosen || chosen == &c) std::cout << "  " << argv[0] << ' ' << c.name << ' ' << c.usage << "\n";
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
