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

// ---- Alignment: gapped rows <-> ungapped sequences + path (behaviour of reference src/alignpath.cpp:22-31,
// 232-280) -----------------------------------------------------------------------------------------
Alignment::Alignment(const vguard<FastSeq>& gapped) : ungapped(gapped.size()) {
  for (size_t r = 1; r < gapped.size(); ++r)
    Assert(gapped[0].length() == gapped[r].length(), "Alignment is not flush: sequence %s has %u chars, but sequence %s has %u chars",
           gapped[0].name.c_str(), gapped[0].length(), gapped[r].name.c_str(), gapped[r].length());
  for (AlignRowIndex r = 0; r < gapped.size(); ++r) {
    FastSeq& plain = ungapped[r];
    plain.name = gapped[r].name;
    plain.comment = gapped[r].comment;
    AlignRowPath& occupied = path[r];
    occupied.reserve(gapped[r].seq.size());
    for (char c : gapped[r].seq) {
      occupied.push_back(!isGap(c));
      if (!isGap(c)) plain.seq.push_back(c);
    }
  }
}

Alignment::Alignment(const vguard<FastSeq>& ungapped, const AlignPath& path) : path(path), ungapped(ungapped) {}

vguard<FastSeq> Alignment::gapped() const {
  vguard<FastSeq> rows(ungapped.size());
  for (const auto& rp : path) {
    const FastSeq& plain = ungapped[rp.first];
    FastSeq& out = rows[rp.first];
    out.name = plain.name;
    out.comment = plain.comment;
    out.seq.reserve(rp.second.size());
    size_t used = 0;
    for (bool occupied : rp.second) {
      if (occupied) Assert(used < plain.seq.size(), "Sequence position %u out of bounds for sequence %s", (unsigned)used, plain.name.c_str());
      out.seq.push_back(occupied ? plain.seq[used++] : gapChar);
    }
  }
  return rows;
}

// ---- merging alignments that share rows (behaviour of reference src/alignpath.cpp:93-203) --------
// Every input alignment is a list of columns; two columns of different alignments are "linked" when
// they hold the same residue of the same row, and linked columns must come out as one merged column.
// The merged alignment is produced by repeatedly taking, for the lowest-numbered input alignment whose
// next column can go out (every column linked to it is also the next one of its alignment), that whole
// linked group.  Inconsistent inputs (a cycle, or a residue aligned to two columns of one alignment)
// abort, as in the reference.
namespace {
struct ColumnRef { size_t align; AlignColIndex col; };

class MergeIndex {
public:
  explicit MergeIndex(const vguard<AlignPath>& in) : in(in), nCols(in.size(), 0), residues(in.size()) {
    for (size_t a = 0; a < in.size(); ++a) {
      if (in[a].empty()) continue;
      nCols[a] = alignPathColumns(in[a]);
      for (const auto& rp : in[a]) {
        const SeqIdx len = alignPathResiduesInRow(rp.second);
        auto known = rowLength.find(rp.first);
        if (known == rowLength.end())
          rowLength[rp.first] = len;
        else
          Assert(known->second == len, "Incompatible number of residues for row #%d of alignment (%d != %d)", (int)rp.first,
                 (int)known->second, (int)len);
      }
    }
    for (const auto& rl : rowLength) where[rl.first].assign(rl.second, vguard<ColumnRef>());
    for (size_t a = 0; a < in.size(); ++a) {
      residues[a].assign(nCols[a], vguard<std::pair<AlignRowIndex, SeqIdx> >());
      map<AlignRowIndex, SeqIdx> consumed;
      for (AlignColIndex c = 0; c < nCols[a]; ++c) {
        for (const auto& rp : in[a])
          if (rp.second[c]) {
            const SeqIdx pos = consumed[rp.first]++;
            residues[a][c].push_back(std::make_pair(rp.first, pos));
            where[rp.first][pos].push_back(ColumnRef{a, c});
          }
        Assert(!residues[a][c].empty(), "Column %d of alignment %d in AlignSeqMap is empty", (int)c, (int)a);
      }
    }
  }

  // the columns (one per alignment at most) transitively linked to column `col` of alignment `a`
  map<size_t, AlignColIndex> group(size_t a, AlignColIndex col) const {
    map<size_t, AlignColIndex> g;
    vguard<ColumnRef> todo(1, ColumnRef{a, col});
    g[a] = col;
    while (!todo.empty()) {
      const ColumnRef cur = todo.back();
      todo.pop_back();
      for (const auto& row_pos : residues[cur.align][cur.col])
        for (const ColumnRef& other : where.at(row_pos.first)[row_pos.second]) {
          auto seen = g.find(other.align);
          if (seen == g.end()) {
            g[other.align] = other.col;
            todo.push_back(other);
          } else
            Assert(seen->second == other.col,
                   "Inconsistent alignments\nColumn %u of alignment %u points to position %u of sequence %u, which points "
                   "back to column %u of alignment %u",
                   (unsigned)col, (unsigned)a, (unsigned)row_pos.second, (unsigned)row_pos.first, (unsigned)other.col,
                   (unsigned)other.align);
        }
    }
    return g;
  }

  const vguard<AlignPath>& in;
  vguard<AlignColIndex> nCols;
  map<AlignRowIndex, SeqIdx> rowLength;

private:
  vguard<vguard<vguard<std::pair<AlignRowIndex, SeqIdx> > > > residues;   // [alignment][column] -> (row, residue index)
  map<AlignRowIndex, vguard<vguard<ColumnRef> > > where;                  // row -> residue index -> columns holding it
};
}  // namespace

AlignPath alignPathMerge(const vguard<AlignPath>& alignments) {
  const MergeIndex index(alignments);
  AlignPath merged;
  for (const auto& rl : index.rowLength) merged[rl.first];
  vguard<AlignColIndex> next(alignments.size(), 0);
  for (;;) {
    bool pending = false, progressed = false;
    for (size_t a = 0; a < alignments.size() && !progressed; ++a) {
      if (next[a] >= index.nCols[a]) continue;
      pending = true;
      const map<size_t, AlignColIndex> g = index.group(a, next[a]);
      bool ready = true;
      for (const auto& ac : g) ready = ready && next[ac.first] == ac.second;
      if (!ready) continue;
      for (auto& rp : merged) rp.second.push_back(false);
      for (const auto& ac : g) {
        for (const auto& rp : alignments[ac.first])
          if (rp.second[ac.second]) merged[rp.first].back() = true;
        ++next[ac.first];
      }
      progressed = true;
    }
    if (!pending) break;
    if (!progressed) {
      for (size_t a = 0; a < alignments.size(); ++a) std::cerr << "Alignment #" << a << ": next column " << next[a] << std::endl;
      Abort("%s fail, no alignments ready", __func__);
    }
  }
  (void)alignPathColumns(merged);   // also checks that the result is flush
  return merged;
}

// ---- src/diagenv.cpp:12-20,92-102 --------------------------------------------------------------
DiagEnvParams::DiagEnvParams()
    : kmerLen(DEFAULT_KMER_LENGTH), kmerThreshold(DEFAULT_KMER_THRESHOLD), bandSize(DEFAULT_BAND_SIZE), sparse(true),
      autoMemSize(true), maxSize(0) {}

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
// Disjoint sets of sequence indices as a union-find forest: parent links with path halving, and at every root the set's
// members in ascending order (the spanning tree below scans the members of sequence 0's set in that order, which is what
// decides between equally good edges).
AlignGraph::Components::Components(size_t n) : up(n), members(n), count(n) {
  for (size_t k = 0; k < n; ++k) {
    up[k] = k;
    members[k].assign(1, k);
  }
}

size_t AlignGraph::Components::root(size_t k) {
  while (up[k] != k) {
    up[k] = up[up[k]];
    k = up[k];
  }
  return k;
}

void AlignGraph::Components::join(size_t a, size_t b) {
  size_t big = root(a), small = root(b);
  if (big == small) return;
  if (members[big].size() < members[small].size()) std::swap(big, small);
  vguard<size_t> both(members[big].size() + members[small].size());
  std::merge(members[big].begin(), members[big].end(), members[small].begin(), members[small].end(), both.begin());
  members[big].swap(both);
  vguard<size_t>().swap(members[small]);
  up[small] = big;
  --count;
}

AlignGraph::AlignGraph(const vguard<FastSeq>& sequences, const RateModel& rates, const double branchLength,
                       const DiagEnvParams& envelopeParams, ForwardMatrix::random_engine& rng)
    : AlignGraph(sequences, rates, branchLength, envelopeParams, &rng) {}

AlignGraph::AlignGraph(const vguard<FastSeq>& sequences, const RateModel& rates, const double branchLength, const DiagEnvParams& envelopeParams)
    : AlignGraph(sequences, rates, branchLength, envelopeParams, nullptr) {}

// (with a generator: the sparse random graph; without: every pair)
AlignGraph::AlignGraph(const vguard<FastSeq>& sequences, const RateModel& rates, const double branchLength, const DiagEnvParams& envelopeParams,
                       ForwardMatrix::random_engine* rng)
    : model(rates), time(branchLength), seqs(sequences), diagEnvParams(envelopeParams) {
  edges.resize(seqs.size());
  edgePath.resize(seqs.size());
  if (rng) buildSparseRandomGraph(*rng);
  else buildDenseGraph();
}

// every unordered pair once, lower index first, in row-major order
void AlignGraph::buildDenseGraph() {
  const size_t n = seqs.size();
  list<TrialEdge> everyPair;
  for (size_t k = 0; k < n * n; ++k)
    if (k / n < k % n) everyPair.emplace_back(k / n, k % n);
  buildGraph(everyPair, "all-vs-all");
}

// Random pairs until there are n log2(n) of them (or all) AND the graph is connected.  The draws are the reference's
// (src/span.cpp:57-85): two indices per attempt from one uniform_int_distribution, ordered, attempts that repeat a pair or
// hit the diagonal are drawn again - so the generator is left where the reference leaves it.
void AlignGraph::buildSparseRandomGraph(ForwardMatrix::random_engine& generator) {
  const size_t n = seqs.size();
  const size_t wanted = std::min(n * (n - 1) / 2, (size_t)ceil(log((double)n) * (double)n / log(2)));
  std::uniform_int_distribution<size_t> anySequence(0, n - 1);
  std::set<std::pair<size_t, size_t>> taken;
  Components linked(n);
  list<TrialEdge> picked;
  while (picked.size() < wanted || linked.count > 1) {
    std::pair<size_t, size_t> p;
    do {
      p.first = anySequence(generator);
      p.second = anySequence(generator);
      if (p.second < p.first) std::swap(p.first, p.second);
    } while (p.first == p.second || !taken.insert(p).second);
    picked.emplace_back(p.first, p.second);
    linked.join(p.first, p.second);
  }
  buildGraph(picked, std::to_string(picked.size()) + " random pairs");
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
    Edge scored;                       // the pair's Viterbi score, queued at both of its sequences
    static_cast<TrialEdge&>(scored) = trialEdge;
    scored.lp = mx.end;
    for (const size_t at : {src, dest}) edges[at].push(scored);
    delete mxs[k];
    delete envs[k];
    ++k;
  }
}

// Maximum spanning tree over the pairwise alignments, grown from the component of sequence 0: every round
// takes the best-scoring edge leaving that component (each sequence keeps its edges in a max-heap; edges that
// have become internal are discarded lazily).  Returns the alignments of the chosen edges in selection order.
list<AlignPath> AlignGraph::minSpanTree() {
  list<AlignPath> chosen;
  Components components(seqs.size());
  while (components.count > 1) {
    const Edge* best = NULL;
    for (size_t member : components.members[components.root(0)]) {
      std::priority_queue<Edge>& heap = edges[member];
      while (!heap.empty() && components.root(heap.top().row1) == components.root(heap.top().row2)) heap.pop();
      if (!heap.empty() && (best == NULL || *best < heap.top())) best = &heap.top();
    }
    Assert(best != NULL, "Found no valid edge");
    const Edge taken = *best;
    chosen.push_back(edgePath[taken.row1][taken.row2]);
    components.join(taken.row1, taken.row2);
    if (getenv("HX_DEBUG_SPAN")) fprintf(stderr, "mst %d %d %a\n", (int)taken.row1, (int)taken.row2, taken.lp);
  }
  return chosen;
}

AlignPath AlignGraph::mstPath() {
  const list<AlignPath> tree = minSpanTree();
  return alignPathMerge(vguard<AlignPath>(tree.begin(), tree.end()));
}

Alignment AlignGraph::mstAlign() {
  const AlignPath merged = mstPath();
  return Alignment(seqs, merged);
}

vguard<FastSeq> AlignGraph::mstGapped() { return mstAlign().gapped(); }

}  // namespace historian
