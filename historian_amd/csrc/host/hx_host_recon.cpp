#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstdio>
#include <exception>
#include <mutex>
#include <thread>
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
      usePosteriorsForProfile(false), reconstructRoot(true), minPostProb(.01), rndSeed(std::mt19937::default_seed),
      batchReadyNodes(true), maxBatchLatticeCells(4e8) {}

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

// reference src/recon.cpp:917-1052.  The per-node work is the reference's; what differs is when a node's
// Forward fill is launched: as soon as both children's profiles exist, together with every other such
// node (ForwardMatrix::fillBatch), instead of one node at a time.  Tracebacks, sampling and profile
// building consume the shared random generator and therefore still run strictly in node order.
void Reconstructor::reconstruct(Dataset& dataset) {
  if (!usePosteriorsForProfile) seedGenerator();
  const vguard<Vec>& rootProb = model.insProb;
  dataset.lpFinalFwd = dataset.lpFinalTrace = NEG_INF;
  const ForwardMatrix::ProfilingStrategy strategy = (ForwardMatrix::ProfilingStrategy)(
      ForwardMatrix::CollapseChains | (keepGapsOpen ? ForwardMatrix::KeepGapsOpen : ForwardMatrix::DontKeepGapsOpen) |
      (includeBestTraceInProfile ? ForwardMatrix::IncludeBestTrace : ForwardMatrix::DontIncludeBestTrace));
  vguard<vguard<LogProb> > logRootProb;
  for (const auto& rv : rootProb) logRootProb.push_back(log_vector(rv));

  const TreeNodeIndex N = dataset.tree.nodes();
  AlignPath path;
  map<int, Profile> prof;
  const bool timing = getenv("HX_TIMING") != NULL;
  double tLeaf = 0, tHmm = 0, tFwd = 0, tProf = 0, tCheck = 0, tSample = 0, tBuild = 0;

  // per internal node: the branch models (the matrices keep references to them), the filled matrix, the band
  struct NodeWork {
    ProbModel* lProbs;
    ProbModel* rProbs;
    PairHMM* hmm;
    ForwardMatrix* forward;
    int maxDist;
    NodeWork() : lProbs(NULL), rProbs(NULL), hmm(NULL), forward(NULL), maxDist(0) {}
  };
  vguard<NodeWork> work(N);
  vguard<char> done(N, 0);
  {
    const double ta = wallSeconds();
    for (TreeNodeIndex node = 0; node < N; ++node)
      if (dataset.tree.isLeaf(node)) {
        prof[node] = Profile(model.components(), model.alphabet, dataset.seqs.at(node), node);
        done[node] = 1;
      }
    tLeaf += wallSeconds() - ta;
  }
  // sampled tracebacks read whole matrices on the host, through page-locked buffers (the node being walked, the copies
  // in flight for the next two, one spare): have them page-locked by the time the first fill is done.
  // An internal node's profile is a little longer than the longest sequence below it; buffers that turn out too small are
  // replaced on demand.
  if (!usePosteriorsForProfile && profileSamples > 0) {
    size_t longest = 0;
    for (TreeNodeIndex node = 0; node < N; ++node)
      if (dataset.tree.isLeaf(node)) longest = std::max(longest, (size_t)dataset.seqs.at(node).length());
    const size_t side = longest + longest / 4 + 130;
    detail::warmHostBuffers(5 * side * side, 4);
  }
  auto envelopeFor = [&](TreeNodeIndex node, int maxDist) {
    struct Timed { double t0; ~Timed() { fillTiming.construct += wallSeconds() - t0; } } timed{wallSeconds()};
    return dataset.guide.empty() ? GuideAlignmentEnvelope()
                                 : GuideAlignmentEnvelope(dataset.guide, dataset.closestLeaf[dataset.tree.getChild(node, 0)],
                                                          dataset.closestLeaf[dataset.tree.getChild(node, 1)], maxDist);
  };

  TreeNodeIndex next = 0;
  while (next < N) {
    if (dataset.tree.isLeaf(next)) { ++next; continue; }
    // ---- launch the fills of the ready nodes (always including `next`, whose children are done) ----
    const double tb0 = wallSeconds();
    vguard<TreeNodeIndex> ready;
    double cells = 0;
    for (TreeNodeIndex node = next; node < N; ++node) {
      if (dataset.tree.isLeaf(node) || work[node].forward) continue;
      const int lc = dataset.tree.getChild(node, 0), rc = dataset.tree.getChild(node, 1);
      if (!done[lc] || !done[rc]) continue;
      const double c = (double)prof[lc].size() * (double)prof[rc].size();
      if (!ready.empty() && (!batchReadyNodes || cells + c > maxBatchLatticeCells)) continue;
      ready.push_back(node);
      cells += c;
    }
    for (TreeNodeIndex node : ready) {
      NodeWork& w = work[node];
      const int lc = dataset.tree.getChild(node, 0), rc = dataset.tree.getChild(node, 1);
      w.lProbs = new ProbModel(model, dataset.tree.branchLength(lc));
      w.rProbs = new ProbModel(model, dataset.tree.branchLength(rc));
      w.hmm = new PairHMM(*w.lProbs, *w.rProbs, rootProb);
      w.maxDist = maxDistanceFromGuide;
    }
    const double tb1 = wallSeconds();
    tHmm += tb1 - tb0;
    if (ready.size() == 1) {
      const TreeNodeIndex node = ready[0];
      NodeWork& w = work[node];
      w.forward = new ForwardMatrix(prof[dataset.tree.getChild(node, 0)], prof[dataset.tree.getChild(node, 1)], *w.hmm, node,
                                    envelopeFor(node, w.maxDist));
    } else {
      vguard<ForwardMatrix::JobSpec> specs;
      for (TreeNodeIndex node : ready) {
        ForwardMatrix::JobSpec js;
        js.x = &prof[dataset.tree.getChild(node, 0)];
        js.y = &prof[dataset.tree.getChild(node, 1)];
        js.hmm = work[node].hmm;
        js.parentRowIndex = node;
        js.env = envelopeFor(node, work[node].maxDist);
        specs.push_back(js);
      }
      const vguard<ForwardMatrix*> filled = devices.empty() ? ForwardMatrix::fillBatch(specs) : ForwardMatrix::fillBatch(specs, devices);
      for (size_t k = 0; k < ready.size(); ++k) work[ready[k]].forward = filled[k];
    }
    tFwd += wallSeconds() - tb1;

    // ---- finish nodes in node order while their matrices are there ----
    // Three passes over the nodes whose matrices are filled.  (1) In node order, everything that draws from the shared
    // generator or talks to the device: refills of zero-likelihood bands, tracebacks, sampled traces.  (2) The profiles of
    // the sampled cell sets - deterministic, host-only, independent between nodes - on as many host threads as there are
    // nodes (up to the machine's cores).  (3) In node order again: the path-sum check and the clean-up.
    struct Finishing { TreeNodeIndex node; ForwardMatrix* forward; set<ForwardMatrix::CellCoords> cells; bool deferred, checked; LogProb lpTrace; };
    vguard<Finishing> finishing;
    const double tp0 = wallSeconds();
    // sampled tracebacks walk the whole matrix on the host: the copy of the next node's matrix is started before this
    // node's traces are drawn, so that it runs underneath them (two page-locked buffers in use at a time)
    const auto sampled = [&](TreeNodeIndex node) {
      return !usePosteriorsForProfile && profileSamples > 0 && !dataset.tree.isLeaf(node) && node != dataset.tree.root() && work[node].forward &&
             work[node].forward->lpEnd > NEG_INF;
    };
    const auto copyAhead = [&](TreeNodeIndex from) {
      int started = 0;
      for (TreeNodeIndex node = from; node < N && started < 2; ++node)
        if (sampled(node)) { work[node].forward->startHostCopy(); ++started; }
    };
    copyAhead(next);
    while (next < N && (dataset.tree.isLeaf(next) || work[next].forward)) {
      const TreeNodeIndex node = next++;
      if (dataset.tree.isLeaf(node)) continue;
      NodeWork& w = work[node];
      const double tc0 = wallSeconds();
      // zero likelihood inside the band: widen and refill (reference src/recon.cpp:956-975)
      while (!(w.forward->lpEnd > NEG_INF)) {
        if (w.maxDist < 0) Abort("Zero forward likelihood even in the absence of guide alignment constraints - this is not good");
        if (w.maxDist * 2 > (int)alignPathColumns(dataset.guide))
          w.maxDist = -1;
        else if (w.maxDist == 0)
          w.maxDist = 1;
        else
          w.maxDist *= 2;
        delete w.forward;
        w.forward = new ForwardMatrix(prof[dataset.tree.getChild(node, 0)], prof[dataset.tree.getChild(node, 1)], *w.hmm, node,
                                      envelopeFor(node, w.maxDist));
      }
      ForwardMatrix* forward = w.forward;
      dataset.bandUsed[node] = w.maxDist;
      tFwd += wallSeconds() - tc0;

      Finishing f;
      f.node = node;
      f.forward = forward;
      f.deferred = f.checked = false;
      f.lpTrace = NEG_INF;
      Profile& nodeProf = prof[node];
      if (node == dataset.tree.root()) {
        if (reconstructRoot) {
          path = forward->bestAlignPath();
          nodeProf = forward->bestProfile();
        }
        dataset.lpFinalFwd = forward->lpEnd;
      } else if (usePosteriorsForProfile) {
        BackwardMatrix backward(*forward);
        nodeProf = backward.postProbProfile(minPostProb, profileMaxStates, strategy);
      } else {
        copyAhead(next);
        f.cells = forward->sampleCells(generator, profileSamples, profileMaxStates, strategy);
        forward->retainCells(f.cells);
        f.deferred = !forward->onHost();
        if (!f.deferred) nodeProf = forward->makeProfile(f.cells, strategy);
      }
      finishing.push_back(std::move(f));
    }
    const double tp1 = wallSeconds();
    tSample += tp1 - tp0;
    {
      vguard<size_t> todo;
      for (size_t k = 0; k < finishing.size(); ++k)
        if (finishing[k].deferred) todo.push_back(k);
      const size_t cores = std::max(1u, std::thread::hardware_concurrency());
      const size_t nThreads = std::min(todo.size(), cores);
      const vguard<LogProb> logCptWeight = log_vector(model.cptWeight);
      const auto build = [&](Finishing& f) {
        Profile& p = prof[f.node];
        p = f.forward->makeProfile(f.cells, strategy);
        if (p.size()) f.lpTrace = p.calcSumPathAbsorbProbs(logCptWeight, logRootProb, NULL);
        f.checked = true;
      };
      if (nThreads <= 1) {
        for (size_t k : todo) build(finishing[k]);
      } else {
        std::atomic<size_t> cursor(0);
        vguard<FillTiming> spent(nThreads);
        vguard<std::exception_ptr> failed(nThreads);
        vguard<std::thread> pool;
        for (size_t t = 0; t < nThreads; ++t)
          pool.emplace_back([&, t]() {
            try {
              for (size_t at = cursor++; at < todo.size(); at = cursor++) build(finishing[todo[at]]);
            } catch (...) {
              failed[t] = std::current_exception();
            }
            spent[t] = fillTiming;
          });
        for (auto& th : pool) th.join();
        for (size_t t = 0; t < nThreads; ++t) {
          if (failed[t]) std::rethrow_exception(failed[t]);
          detail::mergeTiming(spent[t], fillTiming);
        }
      }
    }
    const double td = wallSeconds();
    tProf += td - tp0;
    tBuild += td - tp1;
    if (timing && getenv("HX_TIMING_LEVELS"))
      fprintf(stderr, "timing: level of %zu nodes: sampling %.4f s, profiles %.4f s\n", finishing.size(), tp1 - tp0, td - tp1);
    for (Finishing& f : finishing) {
      NodeWork& w = work[f.node];
      Profile& nodeProf = prof[f.node];
      if (!f.checked && nodeProf.size()) f.lpTrace = nodeProf.calcSumPathAbsorbProbs(log_vector(model.cptWeight), logRootProb, NULL);
      if (f.node == dataset.tree.root() && nodeProf.size()) dataset.lpFinalTrace = f.lpTrace;
      delete f.forward;
      w.forward = NULL;
      delete w.hmm;
      delete w.lProbs;
      delete w.rProbs;
      w.hmm = NULL;
      w.lProbs = w.rProbs = NULL;
      done[f.node] = 1;
    }
    tCheck += wallSeconds() - td;
  }
  if (timing)
    fprintf(stderr, "timing: leaf profiles %.3f s, ProbModel+PairHMM %.3f s, ForwardMatrix fills %.3f s, traceback+profile %.3f s "
                    "(%.3f s tracebacks and sampling in node order, of which host tracebacks %.3f s; %.3f s profile building on host threads, "
                    "makeProfile %.3f s summed over threads), path-sum check+delete %.3f s\n", tLeaf, tHmm, tFwd, tProf, tSample,
            fillTiming.hostTraces, tBuild, fillTiming.hostMakeProfile, tCheck);
  dataset.path = path;
}

// A family's pair DPs in lattice cells, estimated before any profile exists: every internal node pairs two subtrees, and
// a subtree's profile is about as long as its longest leaf sequence.
double Reconstructor::familyCost(const Dataset& dataset) {
  const ReconTree& tree = dataset.tree;
  vguard<double> len((size_t)tree.nodes(), 0.);
  double cells = 0;
  for (TreeNodeIndex n = 0; n < tree.nodes(); ++n) {
    if (tree.isLeaf(n)) { len[n] = (double)dataset.seqs.at(n).length() + 2; continue; }
    const double l = len[tree.getChild(n, 0)], r = len[tree.getChild(n, 1)];
    cells += l * r;
    len[n] = std::max(l, r);
  }
  return cells;
}

// Every family (reference src/recon.cpp:1368-1372 runs them one after the other).  With several devices the families
// are the farm's jobs: sorted by estimated cost, longest first, and taken from that list by one host thread per device
// (list scheduling = longest-processing-time-first); each thread drives its own device.  A family is reconstructed by a
// copy of this Reconstructor, so its generator is seeded exactly as in a sequential run: same results, any device count.
void Reconstructor::reconstructAll(vguard<Dataset*>& datasets) {
  if (devices.size() < 2 || datasets.size() < 2) {
    for (Dataset* d : datasets) reconstruct(*d);
    return;
  }
  vguard<double> cost;
  for (const Dataset* d : datasets) cost.push_back(familyCost(*d));
  vguard<size_t> order(datasets.size());
  for (size_t k = 0; k < order.size(); ++k) order[k] = k;
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return cost[a] > cost[b]; });
  std::atomic<size_t> nextJob(0);
  std::mutex merge;
  FillTiming& mine = fillTiming;
  vguard<std::thread> workers;
  for (int device : devices)
    workers.emplace_back([&, device]() {
      detail::setThreadDevice(device);
      Reconstructor own(*this);
      own.devices.clear();                         // (this worker's fills all go to its device)
      for (size_t k = nextJob++; k < order.size(); k = nextJob++) own.reconstruct(*datasets[order[k]]);
      std::lock_guard<std::mutex> lock(merge);
      detail::mergeTiming(fillTiming, mine);
    });
  for (std::thread& w : workers) w.join();
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
