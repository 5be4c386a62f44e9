// Shared by the three pair-DP drivers (testforward, testbackward, testnullforward): what each of the reference's test
// mains (t/testforward.cpp, t/testbackward.cpp, t/testnullforward.cpp) sets up before it touches the matrix - the rate
// model, one ProbModel per branch, the pair HMM with the model's insertion distribution at the root - and the "all
// cells" selection two of them print.  The drivers keep the reference's command lines and standard output (the golden
// files are diffed byte for byte); the fills run on the GPU.
#pragma once
#include <cstdlib>
#include <iostream>
#include <memory>
#include "../hx_host.h"

namespace historian {

class PairSetup {
public:
  RateModel rates;
  // branch lengths as the drivers take them: the y time defaults to the x time
  PairSetup(const char* modelFile, const char* xTime, const char* yTime) {
    rates.readFile(modelFile);
    left.reset(new ProbModel(rates, atof(xTime)));
    right.reset(new ProbModel(rates, atof(yTime ? yTime : xTime)));
    pair.reset(new PairHMM(*left, *right, rates.insProb));
  }
  const PairHMM& hmm() const { return *pair; }
  Profile leaf(const FastSeq& seq, AlignRowIndex row) const { return Profile(1, rates.alphabet, seq, row); }
  // the profile's own log-likelihood annotations, then its JSON on stdout
  void print(Profile& prof) const {
    prof.calcSumPathAbsorbProbs(vguard<LogProb>(1, 0), pair->logRoot);
    prof.writeJson(std::cout);
  }

private:
  std::unique_ptr<ProbModel> left, right;
  std::unique_ptr<PairHMM> pair;
};

// start, end and every (cell, state) of the lattice
inline set<ForwardMatrix::CellCoords> everyCell(const ForwardMatrix& fwd) {
  set<ForwardMatrix::CellCoords> cells{fwd.startCell, fwd.endCell};
  for (ProfileStateIndex i = 0; i + 1 < fwd.xSize; ++i)
    for (ProfileStateIndex j = 0; j + 1 < fwd.ySize; ++j)
      if (i || j)
        for (PairHMM::State s : PairHMM::states()) cells.emplace(i, j, s);
  return cells;
}

inline vguard<FastSeq> twoSequences(const char* file) {
  vguard<FastSeq> seqs = readFastSeqs(file);
  Assert(seqs.size() == 2, "Expected two sequences in file %s", file);
  return seqs;
}

inline int usage(const char* prog, const char* args) {
  std::cout << "Usage: " << prog << " " << args << "\n";
  return EXIT_FAILURE;
}

}  // namespace historian
