// Progressive reconstruction driver over the GPU-backed mirror.  Input is a small job file (the
// reference's own readers for Newick / Stockholm / guide files are outside this build's scope):
//   model <rate model json>
//   seqs <fasta of ungapped leaf sequences>
//   guide <fasta of the gapped guide alignment>      (optional)
//   band <n> | samples <n> | maxstates <n> | seed <n> | posterior <minPostProb> | batch <0|1>   (optional)
//   tree <N>   followed by N lines:  <parent index or -1> <branch length> <name>   (post-order, root last)
// Output: final Forward / trace log-likelihoods as hex floats, the band used per node, and the
// gapped reconstruction (one row per tree node on the root path).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include "../hx_host.h"
using namespace historian;

int main(int argc, char** argv) {
  if (argc != 2) {
    std::cout << "Usage: " << argv[0] << " <jobfile>\n";
    exit(EXIT_FAILURE);
  }
  std::ifstream in(argv[1]);
  Require(in.good(), "Couldn't open %s", argv[1]);
  Reconstructor recon;
  Reconstructor::Dataset ds;
  string key, seqFile, guideFile;
  while (in >> key) {
    if (key == "model") { string f; in >> f; recon.model.readFile(f.c_str()); }
    else if (key == "seqs") in >> seqFile;
    else if (key == "guide") in >> guideFile;
    else if (key == "band") in >> recon.maxDistanceFromGuide;
    else if (key == "samples") in >> recon.profileSamples;
    else if (key == "maxstates") in >> recon.profileMaxStates;
    else if (key == "seed") in >> recon.rndSeed;
    else if (key == "batch") { int b; in >> b; recon.batchReadyNodes = b != 0; }
    else if (key == "posterior") { in >> recon.minPostProb; recon.usePosteriorsForProfile = true; }
    else if (key == "tree") {
      int n; in >> n;
      for (int k = 0; k < n; ++k) {
        int parent; double len; string name;
        in >> parent >> len >> name;
        ds.tree.addNode(parent, len, name);
      }
      ds.tree.finish();
    } else Fail("Unknown key %s in %s", key.c_str(), argv[1]);
  }
  map<string, string> ungapped, gapped;
  for (const auto& fs : readFastSeqs(seqFile.c_str())) ungapped[fs.name] = fs.seq;
  if (!guideFile.empty())
    for (const auto& fs : readFastSeqs(guideFile.c_str())) gapped[fs.name] = fs.seq;
  for (TreeNodeIndex n = 0; n < ds.tree.nodes(); ++n)
    if (ds.tree.isLeaf(n)) {
      const string& name = ds.tree.nodeName[n];
      Require(ungapped.count(name), "Can't find sequence for leaf node %s", name.c_str());
      FastSeq fs;
      fs.name = name;
      fs.seq = ungapped[name];
      ds.seqs[n] = fs;
      if (!guideFile.empty()) {
        Require(gapped.count(name), "Can't find guide row for leaf node %s", name.c_str());
        AlignRowPath row;
        for (char c : gapped[name]) row.push_back(!Alignment::isGap(c));
        ds.guide[n] = row;
      }
    }
  ds.prepareRecon();
  const double t0 = wallSeconds();
  recon.reconstruct(ds);
  if (getenv("HX_TIMING")) {
    const double total = wallSeconds() - t0;
    const FillTiming& f = fillTiming;
    fprintf(stderr, "timing: reconstruct %.3f s = one-off HIP/device initialisation %.3f s + %.3f s; %ld fills over %lld lattice cells: "
                    "flatten+upload %.3f s, forward (launch..lpEnd) %.3f s of which fill kernels %.3f s, matrix D2H %.3f s in %ld reads, "
                    "device best-path tracebacks %.3f s in %ld calls, cell gathers %.3f s in %ld calls; host traceback/profile building/other %.3f s\n",
            total, f.deviceInit, total - f.deviceInit, f.fills, f.cells, f.flattenAndUpload, f.forwardWait, f.forwardKernel, f.readMatrix,
            f.matrixReads, f.deviceTrace, f.deviceTraces, f.cellGather, f.cellGathers,
            total - f.deviceInit - f.flattenAndUpload - f.forwardWait - f.readMatrix - f.backwardWait - f.deviceTrace - f.cellGather);
  }
  printf("lpFinalFwd %a %.6f\n", ds.lpFinalFwd, ds.lpFinalFwd);
  printf("lpFinalTrace %a %.6f\n", ds.lpFinalTrace, ds.lpFinalTrace);
  for (const auto& nb : ds.bandUsed) printf("band %d %d\n", nb.first, nb.second);
  for (const auto& row_path : ds.path) {
    const TreeNodeIndex node = (TreeNodeIndex)row_path.first;
    std::cout << "row " << node << " " << ds.tree.nodeName[node] << " ";
    const bool leaf = ds.seqs.count(node) > 0;
    size_t k = 0;
    for (bool b : row_path.second) std::cout << (b ? (leaf ? ds.seqs.at(node).seq[k++] : Alignment::wildcardChar) : Alignment::gapChar);
    std::cout << "\n";
  }
  exit(EXIT_SUCCESS);
}
