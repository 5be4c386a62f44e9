// Mirror of the reference's t/testmerge.cpp: testmerge <align1> <align2> ...  (host only)
#include <cstdlib>
#include <iostream>
#include "../hx_host.h"
using namespace historian;

int main(int argc, char** argv) {
  if (argc < 3) {
    std::cout << "Usage: " << argv[0] << " <align1> <align2> ...\n";
    exit(EXIT_FAILURE);
  }
  map<string, AlignRowIndex> nameToRowIndex;
  vguard<FastSeq> ungapped;
  vguard<AlignPath> paths;
  for (int n = 1; n < argc; ++n) {
    vguard<FastSeq> gapped = readFastSeqs(argv[n]);
    Alignment align(gapped);
    AlignPath path;
    for (size_t k = 0; k < gapped.size(); ++k) {
      if (nameToRowIndex.find(gapped[k].name) == nameToRowIndex.end()) {
        nameToRowIndex[gapped[k].name] = ungapped.size();
        ungapped.push_back(align.ungapped[k]);
      }
      path[nameToRowIndex[gapped[k].name]] = align.path[k];
    }
    paths.push_back(path);
  }
  const AlignPath path = alignPathMerge(paths);
  const Alignment align(ungapped, path);
  writeFastaSeqs(std::cout, align.gapped());
  exit(EXIT_SUCCESS);
}
