// Host mirror of the alignment graph around the guide-alignment pair DP (reference src/span.cpp) and of
// the alignment merge it ends with (src/alignpath.cpp:9-20,93-217,232-280).  The pairwise fills of
// buildGraph run as ONE device batch (QuickAlignMatrix::fillBatch); everything else is the reference's
// host logic.
#include <algorithm>
#include <cmath>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include "hx_host.h"

namespace historian {

// ---- src/alignpath.cpp: Alignment, AlignSeqMap, alignPathMerge ---------------------------------
static AlignColIndex gappedSeqColumns(const vguard<FastSeq>& gapped) {
  AlignColIndex cols = 0;
  for (size_t row = 0; row < gapped.size(); ++row)
    if (row == 0)
      cols = gapped[row].length();
    else
      Assert(cols == gapped[row].length(), "Alignment is not flush: sequence %s has %u chars, but sequence %s has %u chars",
             gapped[0].name.c_str(), (unsigned)cols, gapped[row].name.c_str(), gapped[row].length());
  return cols;
}

Alignment::Alignment(const vguard<FastSeq>& gapped) : ungapped(gapped.size()) {
  (void)gappedSeqColumns(gapped);   // has the effect of checking that the alignment is flush
  for (AlignRowIndex row = 0; row < gapped.size(); ++row) {
    ungapped[row].name = gapped[row].name;
    ungapped[row].comment = gapped[row].comment;
    AlignRowPath rowPath(gapped[row].length(), false);
    for (AlignColIndex col = 0; col < rowPath.size(); ++col)
      if (!isGap(gapped[row].seq[col])) {
        rowPath[col] = true;
        ungapped[row].seq.push_back(gapped[row].seq[col]);
      }
    path[row] = rowPath;
  }
}

Alignment::Alignment(const vguard<FastSeq>& ungapped, const AlignPath& path) : ungapped(ungapped), path(path) {}

vguard<FastSeq> Alignment::gapped() const {
  vguard<FastSeq> gs(ungapped.size());
  for (auto& row_path : path) {
    FastSeq& g = gs[row_path.first];
    const FastSeq& ug = ungapped[row_path.first];
    const AlignColIndex cols = row_path.second.size();
    g.name = ug.name;
    g.comment = ug.comment;
    g.seq.reserve(cols);
    SeqIdx pos = 0;
    for (AlignColIndex col = 0; col < cols; ++col)
      if (row_path.second[col]) {
        Assert(ug.seq.size() > pos, "Sequence position %u out of bounds for sequence %s", (unsigned)col, ug.name.c_str());
        g.seq.push_back(ug.seq[pos]);
        ++pos;
      } else
        g.seq.push_back(gapChar);
  }
  return gs;
}

namespace {
// map used by alignPathMerge
struct AlignSeqMap {
  typedef size_t AlignNum;
  const vguard<AlignPath>& alignments;
  map<AlignRowIndex, SeqIdx> seqLen;
  vguard<AlignColIndex> alignCols;
  map<AlignNum, map<AlignColIndex, map<AlignRowIndex, SeqIdx> > > alignColRowToPos;
  map<AlignRowIndex, map<SeqIdx, map<AlignNum, AlignColIndex> > > rowPosAlignToCol;
  AlignSeqMap(const vguard<AlignPath>& alignments);
  map<AlignNum, AlignColIndex> linkedColumns(AlignNum nAlign, AlignColIndex col) const;
};

AlignSeqMap::AlignSeqMap(const vguard<AlignPath>& alignments) : alignments(alignments) {
  // get row indices and sequence lengths; confirm row & sequence lengths match
  for (auto& align : alignments) {
    if (align.size() == 0)
      alignCols.push_back(0);
    else {
      alignCols.push_back(alignPathColumns(align));
      for (auto& row_path : align) {
        const AlignRowIndex row = row_path.first;
        const SeqIdx len = alignPathResiduesInRow(row_path.second);
        if (seqLen.find(row) == seqLen.end())
          seqLen[row] = len;
        else
          Assert(seqLen[row] == len, "Incompatible number of residues for row #%d of alignment (%d != %d)", (int)row,
                 (int)seqLen[row], (int)len);
      }
    }
  }
  // build bidirectional map from (align#,column#) <==> (row#,residue#)
  for (size_t nAlign = 0; nAlign < alignments.size(); ++nAlign) {
    auto& align = alignments[nAlign];
    map<AlignRowIndex, SeqIdx> rowPos;
    for (auto& row_path : align) rowPos[row_path.first] = 0;
    for (AlignColIndex col = 0; col < alignCols[nAlign]; ++col) {
      bool allGaps = true;
      for (auto& row_path : align)
        if (row_path.second[col]) {
          const SeqIdx pos = rowPos[row_path.first]++;
          alignColRowToPos[nAlign][col][row_path.first] = pos;
          rowPosAlignToCol[row_path.first][pos][nAlign] = col;
          allGaps = false;
        }
      Assert(!allGaps, "Column %d of alignment %d in AlignSeqMap is empty", (int)col, (int)nAlign);
    }
  }
}

map<AlignSeqMap::AlignNum, AlignColIndex> AlignSeqMap::linkedColumns(AlignNum nAlign, AlignColIndex col) const {
  map<AlignNum, AlignColIndex> ac, acQueue;
  acQueue[nAlign] = col;
  while (acQueue.size() > ac.size()) {
    for (auto& nAlign_col : acQueue)
      if (ac.find(nAlign_col.first) == ac.end()) {
        ac.insert(nAlign_col);
        for (auto& row_pos : alignColRowToPos.at(nAlign_col.first).at(nAlign_col.second))
          for (auto& linked_nAlign_col : rowPosAlignToCol.at(row_pos.first).at(row_pos.second)) {
            if (ac.find(linked_nAlign_col.first) != ac.end())
              Assert(ac[linked_nAlign_col.first] == linked_nAlign_col.second,
                     "Inconsistent alignments\nColumn %u of alignment %u points to position %u of sequence %u, which points "
                     "back to column %u of alignment %u",
                     (unsigned)col, (unsigned)nAlign, (unsigned)row_pos.second, (unsigned)row_pos.first,
                     (unsigned)linked_nAlign_col.second, (unsigned)linked_nAlign_col.first);
            acQueue.insert(linked_nAlign_col);
          }
      }
  }
  return ac;
}
}  // namespace

AlignPath alignPathMerge(const vguard<AlignPath>& alignments) {
  const AlignSeqMap alignSeqMap(alignments);
  AlignPath a;
  for (auto& row_seqlen : alignSeqMap.seqLen) a[row_seqlen.first].clear();
  vguard<AlignColIndex> nextCol(alignments.size(), 0);
  bool allDone, noneReady;
  do {
    allDone = noneReady = true;
    map<AlignSeqMap::AlignNum, AlignColIndex> linkedCols;
    for (AlignSeqMap::AlignNum n = 0; n < alignments.size(); ++n)
      if (nextCol[n] < alignSeqMap.alignCols[n]) {
        allDone = false;
        bool ready = true;
        linkedCols = alignSeqMap.linkedColumns(n, nextCol[n]);
        for (const auto& nAlign_col : linkedCols)
          if (nextCol[nAlign_col.first] != nAlign_col.second) {
            ready = false;
            break;
          }
        if (ready) {
          noneReady = false;
          if (linkedCols.size()) {
            for (auto& idx_path : a) idx_path.second.push_back(false);
            for (const auto& nAlign_col : linkedCols) {
              for (const auto& row_path : alignments.at(nAlign_col.first))
                if (alignments.at(nAlign_col.first).at(row_path.first).at(nAlign_col.second)) a[row_path.first].back() = true;
              ++nextCol[nAlign_col.first];
            }
          } else
            ++nextCol[n];   // empty column
          break;
        }
      }
    if (noneReady && !allDone) {
      for (AlignSeqMap::AlignNum n = 0; n < alignments.size(); ++n)
        std::cerr << "Alignment #" << n << ": next column " << nextCol[n] << std::endl;
      Abort("%s fail, no alignments ready", __func__);
    }
  } while (!allDone);
  (void)alignPathColumns(a);   // this will also test if alignment is flush
  return a;
}

// ---- src/diagenv.cpp:12-20,92-102 --------------------------------------------------------------
DiagEnvParams::DiagEnvParams()
    : sparse(true), autoMemSize(true), kmerLen(DEFAULT_KMER_LENGTH), kmerThreshold(DEFAULT_KMER_THRESHOLD),
      bandSize(DEFAULT_BAND_SIZE), maxSize(0) {}

size_t DiagEnvParams::effectiveMaxSize() const {
  size_t ms = 0;
  if (autoMemSize) {   // getMemorySize() (src/memsize.c): physical memory of the host
    const long pages = sysconf(_SC_PHYS_PAGES), pageSize = sysconf(_SC_PAGE_SIZE);
    ms = (pages > 0 && pageSize > 0) ? (size_t)pages * (size_t)pageSize : 0;
    Require(ms > 0, "Can't figure out available system memory; you will need to specify a size");
  } else
    ms = maxSize;
  return ms;
}

// ---- src/span.cpp --------------------------------------------------------------------------------
AlignGraph::Partition::Partition(size_t n) : seqSetIdx(n), seqSet(n), nSets(n) {
  for (size_t i = 0; i < n; ++i) {
    seqSetIdx[i] = i;
    seqSet[i].insert(i);
  }
}

bool AlignGraph::Partition::inSameSet(const AlignGraph::TrialEdge& e) const { return seqSetIdx[e.row1] == seqSetIdx[e.row2]; }

void AlignGraph::Partition::merge(const AlignGraph::TrialEdge& e) {
  if (!inSameSet(e)) {
    size_t idx1 = seqSetIdx[e.row1];
    size_t idx2 = seqSetIdx[e.row2];
    if (idx1 > idx2) std::swap(idx1, idx2);
    set<size_t>& set1 = seqSet[idx1];
    set<size_t>& set2 = seqSet[idx2];
    for (auto n2 : set2) seqSetIdx[n2] = idx1;
    set1.insert(set2.begin(), set2.end());
    set2.clear();
    --nSets;
  }
}

AlignGraph::AlignGraph(const vguard<FastSeq>& seqs, const RateModel& model, const double time, const DiagEnvParams& diagEnvParams,
                       ForwardMatrix::random_engine& generator)
    : seqs(seqs), model(model), time(time), diagEnvParams(diagEnvParams), edges(seqs.size()), edgePath(seqs.size()) {
  buildSparseRandomGraph(generator);
}

AlignGraph::AlignGraph(const vguard<FastSeq>& seqs, const RateModel& model, const double time, const DiagEnvParams& diagEnvParams)
    : seqs(seqs), model(model), time(time), diagEnvParams(diagEnvParams), edges(seqs.size()), edgePath(seqs.size()) {
  buildDenseGraph();
}

void AlignGraph::buildDenseGraph() {
  list<TrialEdge> e;
  for (AlignRowIndex src = 0; src + 1 < seqs.size(); ++src)
    for (AlignRowIndex dest = src + 1; dest < seqs.size(); ++dest) e.push_back(TrialEdge(src, dest));
  buildGraph(e, "all-vs-all");
}

void AlignGraph::buildSparseRandomGraph(ForwardMatrix::random_engine& generator) {
  list<TrialEdge> trialEdges;
  map<AlignRowIndex, set<AlignRowIndex> > targets;
  Partition part(seqs.size());
  const size_t nEdges = std::min((size_t)(seqs.size() * (seqs.size() - 1) / 2),
                                 (size_t)ceil(log(seqs.size()) * (double)seqs.size() / log(2)));
  std::uniform_int_distribution<size_t> dist(0, seqs.size() - 1);
  for (size_t n = 0; n < nEdges || part.nSets > 1; ++n) {
    size_t src, dest;
    do {
      src = dist(generator);
      dest = dist(generator);
      if (dest < src) std::swap(src, dest);
    } while (src == dest || targets[src].count(dest));
    targets[src].insert(dest);
    trialEdges.push_back(TrialEdge(src, dest));
    part.merge(trialEdges.back());
  }
  buildGraph(trialEdges, std::to_string(trialEdges.size()) + " random pairs");
}

// The reference aligns the pairs one after the other (src/span.cpp:92-120); here the envelopes are
// built first and all fills go to the device as one batch.  Edges are pushed in the same order.
void AlignGraph::buildGraph(const list<TrialEdge>& trialEdges, const string&) {
  vguard<DiagonalEnvelope*> envs;
  vguard<const DiagonalEnvelope*> cenvs;
  for (auto& trialEdge : trialEdges) {
    const size_t src = trialEdge.row1, dest = trialEdge.row2;
    DiagonalEnvelope* env = new DiagonalEnvelope(seqs[src], seqs[dest]);
    if (diagEnvParams.sparse) {
      KmerIndex yKmerIndex(seqs[dest], model.alphabet, diagEnvParams.kmerLen);
      env->initSparse(yKmerIndex, diagEnvParams.bandSize, diagEnvParams.kmerThreshold, ForwardMatrix::cellSize(),
                      diagEnvParams.effectiveMaxSize());
    } else
      env->initFull();
    envs.push_back(env);
    cenvs.push_back(env);
  }
  const vguard<QuickAlignMatrix*> mxs = QuickAlignMatrix::fillBatch(cenvs, model, time);
  size_t k = 0;
  for (auto& trialEdge : trialEdges) {
    const size_t src = trialEdge.row1, dest = trialEdge.row2;
    QuickAlignMatrix& mx = *mxs[k];
    edgePath[src][dest] = mx.alignPath(src, dest);
    Edge e;
    e.row1 = src;
    e.row2 = dest;
    e.lp = mx.end;
    edges[src].push(e);
    edges[dest].push(e);
    delete mxs[k];
    delete envs[k];
    ++k;
  }
}

list<AlignPath> AlignGraph::minSpanTree() {
  list<AlignPath> paths;
  Partition part(seqs.size());
  while (part.nSets > 1) {
    Edge best;
    bool foundBest = false;
    for (auto src : part.seqSet.front()) {
      while (!edges[src].empty() && part.inSameSet(edges[src].top())) edges[src].pop();
      if (!edges[src].empty() && (!foundBest || best < edges[src].top())) {
        best = edges[src].top();
        foundBest = true;
      }
    }
    Assert(foundBest, "Found no valid edge");
    paths.push_back(edgePath[best.row1][best.row2]);
    part.merge(best);
    if (getenv("HX_DEBUG_SPAN")) fprintf(stderr, "mst %d %d %a\n", (int)best.row1, (int)best.row2, best.lp);
  }
  return paths;
}

AlignPath AlignGraph::mstPath() {
  const list<AlignPath> pathList = minSpanTree();
  const vguard<AlignPath> pathVec(pathList.begin(), pathList.end());
  return alignPathMerge(pathVec);
}

Alignment AlignGraph::mstAlign() { return Alignment(seqs, mstPath()); }

vguard<FastSeq> AlignGraph::mstGapped() { return mstAlign().gapped(); }

}  // namespace historian
