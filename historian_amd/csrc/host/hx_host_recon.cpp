#include <cstdlib>
#include <cstdio>
// Progressive reconstruction loop of the host mirror (see hx_host.h): the part of
// Reconstructor that drives the DP (reference src/recon.cpp:864-915, 917-1052).
#include "hx_host.h"

namespace historian {

static const double NEG_INF = -std::numeric_limits<double>::infinity();

void ReconTree::addNode(TreeNodeIndex parentNode, double len, const string& name) {
  parent.push_back(parentNode);
  branchLen.push_back(len);
  nodeName.push_back(name);
}

void ReconTree::finish() {
  child.assign(parent.size(), vguard<TreeNodeIndex>());
  for (TreeNodeIndex n = 0; n < nodes(); ++n)
    if (parent[n] >= 0) {
      Assert(parent[n] > n, "Tree nodes are not sorted in postorder");
      child[parent[n]].push_back(n);
    }
  Assert(nodes() > 0 && parent[root()] < 0, "Tree nodes are not sorted in postorder");
  for (TreeNodeIndex n = 0; n < nodes(); ++n)
    if (!isLeaf(n)) Assert(child[n].size() == 2, "Tree is not binary: node %d has %d children", n, (int)child[n].size());
}

Reconstructor::Reconstructor()
    : maxDistanceFromGuide(20), profileSamples(10), profileMaxStates(0), includeBestTraceInProfile(true), keepGapsOpen(false),
      usePosteriorsForProfile(false), reconstructRoot(true), minPostProb(.01), rndSeed(std::mt19937::default_seed) {}

void Reconstructor::seedGenerator() { generator = DPMatrix::random_engine(rndSeed); }

// reference src/recon.cpp:885-906
void Reconstructor::Dataset::prepareRecon() {
  vguard<double> closestLeafDistance;
  closestLeaf.clear();
  for (TreeNodeIndex node = 0; node < tree.nodes(); ++node)
    if (tree.isLeaf(node)) {
      Assert(seqs.count(node), "Can't find sequence for leaf node %s", tree.nodeName[node].c_str());
      closestLeaf.push_back(node);
      closestLeafDistance.push_back(0);
    } else {
      int cl = -1;
      double dcl = 0;
      for (size_t nc = 0; nc < tree.child[node].size(); ++nc) {
        const TreeNodeIndex c = tree.getChild(node, nc);
        const double dc = closestLeafDistance[c] + tree.branchLength(c);
        if (nc == 0 || dc < dcl) {
          cl = closestLeaf[c];
          dcl = dc;
        }
      }
      closestLeaf.push_back(cl);
      closestLeafDistance.push_back(dcl);
    }
}

// reference src/recon.cpp:917-1052
void Reconstructor::reconstruct(Dataset& dataset) {
  if (!usePosteriorsForProfile) seedGenerator();
  const vguard<Vec>& rootProb = model.insProb;
  dataset.lpFinalFwd = dataset.lpFinalTrace = NEG_INF;
  const ForwardMatrix::ProfilingStrategy strategy = (ForwardMatrix::ProfilingStrategy)(
      ForwardMatrix::CollapseChains | (keepGapsOpen ? ForwardMatrix::KeepGapsOpen : ForwardMatrix::DontKeepGapsOpen) |
      (includeBestTraceInProfile ? ForwardMatrix::IncludeBestTrace : ForwardMatrix::DontIncludeBestTrace));
  vguard<vguard<LogProb> > logRootProb;
  for (const auto& rv : rootProb) logRootProb.push_back(log_vector(rv));

  AlignPath path;
  map<int, Profile> prof;
  const bool timing = getenv("HX_TIMING") != NULL;
  double tLeaf = 0, tHmm = 0, tFwd = 0, tProf = 0, tCheck = 0;
  for (TreeNodeIndex node = 0; node < dataset.tree.nodes(); ++node) {
    const double ta = wallSeconds();
    if (dataset.tree.isLeaf(node)) {
      prof[node] = Profile(model.components(), model.alphabet, dataset.seqs.at(node), node);
      tLeaf += wallSeconds() - ta;
    } else {
      const int lChildNode = dataset.tree.getChild(node, 0);
      const int rChildNode = dataset.tree.getChild(node, 1);
      const Profile& lProf = prof[lChildNode];
      const Profile& rProf = prof[rChildNode];
      ProbModel lProbs(model, dataset.tree.branchLength(lChildNode));
      ProbModel rProbs(model, dataset.tree.branchLength(rChildNode));
      PairHMM hmm(lProbs, rProbs, rootProb);
      const double tb = wallSeconds();
      tHmm += tb - ta;

      ForwardMatrix* forward = NULL;
      int maxDist = maxDistanceFromGuide;
      while (true) {
        forward = new ForwardMatrix(lProf, rProf, hmm, node,
                                    dataset.guide.empty() ? GuideAlignmentEnvelope()
                                                          : GuideAlignmentEnvelope(dataset.guide, dataset.closestLeaf[lChildNode],
                                                                                   dataset.closestLeaf[rChildNode], maxDist));
        if (forward->lpEnd > NEG_INF) break;
        if (maxDist < 0) Abort("Zero forward likelihood even in the absence of guide alignment constraints - this is not good");
        if (maxDist * 2 > (int)alignPathColumns(dataset.guide))
          maxDist = -1;
        else if (maxDist == 0)
          maxDist = 1;
        else
          maxDist *= 2;
        delete forward;
        forward = NULL;
      }
      dataset.bandUsed[node] = maxDist;
      const double tc = wallSeconds();
      tFwd += tc - tb;

      BackwardMatrix* backward = NULL;
      if (usePosteriorsForProfile && node != dataset.tree.root()) backward = new BackwardMatrix(*forward);

      Profile& nodeProf = prof[node];
      if (node == dataset.tree.root()) {
        if (reconstructRoot) {
          path = forward->bestAlignPath();
          nodeProf = forward->bestProfile();
        }
      } else if (usePosteriorsForProfile)
        nodeProf = backward->postProbProfile(minPostProb, profileMaxStates, strategy);
      else
        nodeProf = forward->sampleProfile(generator, profileSamples, profileMaxStates, strategy);

      if (backward) delete backward;
      const double td = wallSeconds();
      tProf += td - tc;
      if (node == dataset.tree.root()) dataset.lpFinalFwd = forward->lpEnd;
      if (nodeProf.size()) {
        const LogProb lpTrace = nodeProf.calcSumPathAbsorbProbs(log_vector(model.cptWeight), logRootProb, NULL);
        if (node == dataset.tree.root()) dataset.lpFinalTrace = lpTrace;
      }
      delete forward;
      tCheck += wallSeconds() - td;
    }
  }
  if (timing)
    fprintf(stderr, "timing: leaf profiles %.3f s, ProbModel+PairHMM %.3f s, ForwardMatrix ctor %.3f s, traceback+profile %.3f s, "
                    "calcSumPathAbsorbProbs+delete %.3f s\n", tLeaf, tHmm, tFwd, tProf, tCheck);
  dataset.path = path;
}

// Alignment(ungapped, path).gapped(): leaves show residues, internal nodes the wildcard character
vguard<FastSeq> Reconstructor::Dataset::gappedRecon() const {
  vguard<FastSeq> g;
  for (const auto& row_path : path) {
    FastSeq fs;
    const TreeNodeIndex node = (TreeNodeIndex)row_path.first;
    fs.name = tree.nodeName[node];
    const bool leaf = seqs.count(node) > 0;
    size_t k = 0;
    for (bool b : row_path.second)
      if (b)
        fs.seq.push_back(leaf ? seqs.at(node).seq[k++] : Alignment::wildcardChar);
      else
        fs.seq.push_back(Alignment::gapChar);
    g.push_back(fs);
  }
  return g;
}

}  // namespace historian
