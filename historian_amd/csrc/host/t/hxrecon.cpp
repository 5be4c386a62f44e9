// Progressive reconstruction driver over the GPU-backed mirror.  Input is one small job file per family (the
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
#include <list>
#include <sstream>
#include "../hx_host.h"
using namespace historian;

// one job file -> the reconstruction parameters (the first file's stand for all) and one family
static void readJob(const char* file, Reconstructor& recon, bool setParams, Reconstructor::Dataset& ds) {
  std::ifstream in(file);
  Require(in.good(), "Couldn't open %s", file);
  Reconstructor scratch;
  Reconstructor& r = setParams ? recon : scratch;
  string key, seqFile, guideFile;
  while (in >> key) {
    if (key == "model") { string f; in >> f; if (setParams) r.model.readFile(f.c_str()); }
    else if (key == "seqs") in >> seqFile;
    else if (key == "guide") in >> guideFile;
    else if (key == "band") in >> r.maxDistanceFromGuide;
    else if (key == "samples") in >> r.profileSamples;
    else if (key == "maxstates") in >> r.profileMaxStates;
    else if (key == "seed") in >> r.rndSeed;
    else if (key == "batch") { int b; in >> b; r.batchReadyNodes = b != 0; }
    else if (key == "posterior") { in >> r.minPostProb; r.usePosteriorsForProfile = true; }
    else if (key == "tree") {
      int n; in >> n;
      for (int k = 0; k < n; ++k) {
        int parent; double len; string name;
        in >> parent >> len >> name;
        ds.tree.addNode(parent, len, name);
      }
      ds.tree.finish();
    } else Fail("Unknown key %s in %s", key.c_str(), file);
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
}

static void printFamily(const Reconstructor::Dataset& ds) {
  printf("lpFinalFwd %a %.6f\n", ds.lpFinalFwd, ds.lpFinalFwd);
  printf("lpFinalTrace %a %.6f\n", ds.lpFinalTrace, ds.lpFinalTrace);
  for (const auto& nb : ds.bandUsed) printf("band %d %d\n", nb.first, nb.second);
  fflush(stdout);
  for (const auto& row_path : ds.path) {
    const TreeNodeIndex node = (TreeNodeIndex)row_path.first;
    std::cout << "row " << node << " " << ds.tree.nodeName[node] << " ";
    const bool leaf = ds.seqs.count(node) > 0;
    size_t k = 0;
    for (bool b : row_path.second) std::cout << (b ? (leaf ? ds.seqs.at(node).seq[k++] : Alignment::wildcardChar) : Alignment::gapChar);
    std::cout << "\n";
  }
  std::cout.flush();
}

// hxrecon [-devices d0,d1,...] <jobfile> [<jobfile> ...]
// Several job files = several families (the first file's parameters apply to all); with -devices they are farmed over
// those devices, one host thread each (an ordinal may be repeated: two threads sharing a device).  Output per family as
// for a single one, preceded by a line "family <k>" when there are several.
int main(int argc, char** argv) {
  vguard<int> devices;
  int first = 1;
  if (argc > 2 && string(argv[1]) == "-devices") {
    std::stringstream list(argv[2]);
    string tok;
    while (std::getline(list, tok, ',')) devices.push_back(atoi(tok.c_str()));
    first = 3;
  }
  if (argc <= first) {
    std::cout << "Usage: " << argv[0] << " [-devices d0,d1,...] <jobfile> [<jobfile> ...]\n";
    exit(EXIT_FAILURE);
  }
  Reconstructor recon;
  recon.devices = devices;
  std::list<Reconstructor::Dataset> families;
  vguard<Reconstructor::Dataset*> all;
  for (int k = first; k < argc; ++k) {
    families.emplace_back();
    readJob(argv[k], recon, k == first, families.back());
    all.push_back(&families.back());
  }
  const double t0 = wallSeconds();
  if (all.size() == 1) recon.reconstruct(*all[0]);
  else recon.reconstructAll(all);
  if (getenv("HX_TIMING")) {
    const double total = wallSeconds() - t0;
    const FillTiming& f = fillTiming;
    fprintf(stderr, "timing: reconstruct %.3f s = one-off HIP/device initialisation %.3f s + %.3f s; %ld fills over %lld lattice cells: "
                    "flatten+upload %.3f s, forward (launch..lpEnd) %.3f s of which fill kernels %.3f s, backward (launch..lpStart) %.3f s, matrix D2H %.3f s in %ld reads (page-locking beyond the warmed buffers: %.3f s for %ld buffers, while a fill runs where the size is known by then), "
                    "device best-path tracebacks %.3f s in %ld calls, cell gathers %.3f s in %ld calls; sorting sampled cells %.3f s, keeping their values %.3f s; envelopes %.3f s, prepared vectors (read + lpAbsorb) %.3f s; host traceback/profile building/other %.3f s\n",
            total, f.deviceInit, total - f.deviceInit, f.fills, f.cells, f.flattenAndUpload, f.forwardWait, f.forwardKernel, f.backwardWait, f.readMatrix,
            f.matrixReads, f.pinnedAlloc, f.pinnedAllocs, f.deviceTrace, f.deviceTraces, f.cellGather, f.cellGathers, f.cellSets, f.retain, f.construct, f.readPrepared,
            total - f.deviceInit - f.flattenAndUpload - f.forwardWait - f.readMatrix - f.backwardWait - f.deviceTrace - f.cellGather);    fprintf(stderr, "timing: best traces taken again from an exact-policy fill because the walk met a near tie (HX_TIE_REFILL=1): %ld, %.3f s\n",
            f.tieRefills, f.tieRefill);
  }
  for (size_t k = 0; k < all.size(); ++k) {
    if (all.size() > 1) { printf("family %zu\n", k); fflush(stdout); }
    printFamily(*all[k]);
  }
  exit(EXIT_SUCCESS);
}
