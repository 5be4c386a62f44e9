// Walking a filled pair-DP matrix on the host: the moves between cells, best and sampled tracebacks, posterior
// decoding, and the construction of the parent profile from a set of chosen cells.
//
// Behaviour follows the reference (src/forward.cpp:225-577, 654-895, 1099-1379): which moves exist, the order in
// which candidates are compared (cell order x, y, state; the first maximum wins a tie; sampling subtracts in that
// order), the order in which path weights are accumulated into a profile's transitions.  The structure does not: a
// cell's neighbours are produced as a flat list (DPMatrix::Moves) that is sorted once, tracebacks never build a map,
// and profile construction works on a sorted vector of cells with index arrays instead of maps keyed by cell.
// The map-returning members of the reference's interface (sourceTransitions, destTransitions ...) are thin adapters.
#include "hx_host.h"

#include <algorithm>
#include <iomanip>
#include <sstream>

#include "../../../include/historian_hip.h"

namespace historian {

namespace {
const double kNegInf = -std::numeric_limits<double>::infinity();
#define FWD_BACK_ERROR_TOLERANCE .01

typedef DPMatrix::CellCoords Cell;
typedef PairHMM::State State;

// sources of a move into each pair-HMM state, in the order the fill sums them (PairHMM::sources without the vector)
struct StateList { int n; State s[5]; };
const StateList kSources[6] = {
    {5, {PairHMM::IMM, PairHMM::IMD, PairHMM::IDM, PairHMM::IMI, PairHMM::IIW}},   // into IMM
    {4, {PairHMM::IMM, PairHMM::IMD, PairHMM::IDM, PairHMM::IMI, PairHMM::IMM}},   // into IMD
    {4, {PairHMM::IMM, PairHMM::IMD, PairHMM::IDM, PairHMM::IIW, PairHMM::IMM}},   // into IDM
    {2, {PairHMM::IMM, PairHMM::IMI, PairHMM::IMM, PairHMM::IMM, PairHMM::IMM}},   // into IMI
    {3, {PairHMM::IMM, PairHMM::IIW, PairHMM::IMI, PairHMM::IMM, PairHMM::IMM}},   // into IIW
    {5, {PairHMM::IMM, PairHMM::IMD, PairHMM::IDM, PairHMM::IMI, PairHMM::IIW}}};  // into EEE

// |a - b| <= eps * 2^exponent(larger magnitude): the comparison gsl_fcmp makes (reference src/forward.cpp:1091)
bool nearlyEqual(double a, double b, double eps) {
  int exponent;
  frexp(fabs(a) > fabs(b) ? a : b, &exponent);
  return fabs(a - b) <= ldexp(eps, exponent);
}

std::map<Cell, LogProb> asMap(const DPMatrix::Moves& m) { return std::map<Cell, LogProb>(m.begin(), m.end()); }
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// moves as flat lists
// ---------------------------------------------------------------------------------------------------------------------
// sort by cell; where a cell was produced twice (two profile transitions between the same states) the later one stands,
// as a map assignment would leave it
void DPMatrix::settle(Moves& m) {
  if (m.size() < 2) return;
  if (m.size() <= 32) {
    // (the usual case - a handful of candidates: a stable insertion sort, no scratch allocation)
    for (size_t k = 1; k < m.size(); ++k) {
      const Move v = m[k];
      size_t at = k;
      while (at > 0 && v.first < m[at - 1].first) { m[at] = m[at - 1]; --at; }
      m[at] = v;
    }
  } else
    std::stable_sort(m.begin(), m.end(), [](const Move& a, const Move& b) { return a.first < b.first; });
  size_t w = 0;
  for (size_t r = 0; r < m.size(); ++r) {
    if (w > 0 && m[w - 1].first == m[r].first) m[w - 1] = m[r];
    else m[w++] = m[r];
  }
  m.resize(w);
}

// arg-max; a later candidate has to be strictly better, so the first cell in cell order wins a tie
DPMatrix::CellCoords DPMatrix::pickBest(const Moves& m) {
  Assert(!m.empty(), "%s traceback failure", __func__);
  CellCoords best;
  double top = kNegInf;
  for (const Move& c : m)
    if (c.second > top) {
      top = c.second;
      best = c.first;
    }
  return best;
}

// proportional to exp(weight): weights relative to the largest, one uniform draw on [0, total), the first cell at which
// the running remainder is used up (reference src/forward.cpp:225-243: same draws from the same generator state)
DPMatrix::CellCoords DPMatrix::pickSampled(const Moves& m, random_engine& generator) const {
  double top = kNegInf;
  for (const Move& c : m) top = std::max(top, c.second);
  double fewShares[64];
  vguard<double> manyShares(m.size() > 64 ? m.size() : 0);
  double* const share = m.size() > 64 ? manyShares.data() : fewShares;
  double total = 0;
  for (size_t k = 0; k < m.size(); ++k) total += share[k] = exp(m[k].second - top);
  std::uniform_real_distribution<double> pick(0, total);
  const double drawn = pick(generator);
  double left = drawn;
  for (size_t k = 0; k < m.size(); ++k)
    if ((left -= share[k]) <= 0) return m[k].first;
  for (const Move& c : m) std::cerr << "Log P" << cellName(c.first) << " = " << c.second << std::endl;
  Abort("%s fail (ptot=%g, p=%g)", "sampleCell", total, drawn);
  return CellCoords();
}

DPMatrix::CellCoords DPMatrix::sampleCell(const map<CellCoords, LogProb>& cellLogProb, random_engine& generator) const {
  return pickSampled(Moves(cellLogProb.begin(), cellLogProb.end()), generator);
}

DPMatrix::CellCoords DPMatrix::bestCell(const map<CellCoords, LogProb>& cellLogProb) {
  return pickBest(Moves(cellLogProb.begin(), cellLogProb.end()));
}

// what a cell emits or absorbs on entry: the branch-summed likelihood of the absorbed residue(s), or the insertion's
LogProb DPMatrix::lpCellEmitOrAbsorb(const CellCoords& c) {
  const bool xEmits = x.state[c.xpos].isEmit(), yEmits = y.state[c.ypos].isEmit();
  switch (c.state) {
    case PairHMM::IMM: return xEmits && yEmits ? computeLogProbAbsorb(c.xpos, c.ypos) : 0;
    case PairHMM::IMD: return xEmits ? rootsubx[c.xpos] : 0;
    case PairHMM::IIW: return xEmits ? insx[c.xpos] : 0;
    case PairHMM::IDM: return yEmits ? rootsuby[c.ypos] : 0;
    case PairHMM::IMI: return yEmits ? insy[c.ypos] : 0;
    default: return 0;
  }
}

// ---- kinds of cell (reference src/forward.cpp:528-577) ---------------------------------------------------------------
bool DPMatrix::isAbsorbing(const CellCoords& c) const {
  const bool xEmits = x.state[c.xpos].isEmit(), yEmits = y.state[c.ypos].isEmit();
  return (c.state == PairHMM::IMM && xEmits && yEmits) || (c.state == PairHMM::IMD && xEmits) || (c.state == PairHMM::IDM && yEmits);
}

// does entering the cell advance the x (y) profile?  Null moves of the two profiles are ordered: y's go first while x
// sits in an emitting (or the start) state
bool DPMatrix::changesX(const CellCoords& c) const {
  if (c.state == PairHMM::IMM) return x.state[c.xpos].isNull() || y.state[c.ypos].isEmit();
  return c.state == PairHMM::IMD || c.state == PairHMM::IIW || c.state == PairHMM::EEE;
}

bool DPMatrix::changesY(const CellCoords& c) const {
  if (c.state == PairHMM::IMM) return x.state[c.xpos].isEmitOrStart();
  return c.state == PairHMM::IDM || c.state == PairHMM::IMI || c.state == PairHMM::EEE;
}

// the absorbing cell an insertion or null cell stands in for when gaps are kept open
list<DPMatrix::CellCoords> DPMatrix::equivAbsorbCells(const CellCoords& c) const {
  list<CellCoords> out;
  const ProfileState& xs = x.state[c.xpos];
  const ProfileState& ys = y.state[c.ypos];
  if (c.state == PairHMM::IIW && xs.isEmit()) out.emplace_back(c.xpos, c.ypos, PairHMM::IMD);
  else if (c.state == PairHMM::IMI && ys.isEmit()) out.emplace_back(c.xpos, c.ypos, PairHMM::IDM);
  else if (changesX(c) && xs.isNull() && x.equivAbsorbState.count(c.xpos)) out.emplace_back(x.equivAbsorbState.at(c.xpos), c.ypos, PairHMM::IMD);
  else if (changesY(c) && ys.isNull() && y.equivAbsorbState.count(c.ypos)) out.emplace_back(c.xpos, y.equivAbsorbState.at(c.ypos), PairHMM::IDM);
  return out;
}

string DPMatrix::cellName(const CellCoords& c) const {
  string name("(");
  name += hmm.stateName(c.state, c.xpos == 0, c.ypos == 0);
  name += ',';
  name += x.state[c.xpos].name;
  name += ',';
  name += y.state[c.ypos].name;
  name += ')';
  return name;
}

void DPMatrix::write(std::ostream& out, bool edgeOnly) const {
  for (ProfileStateIndex i = 0; i + 1 < xSize; ++i)
    for (ProfileStateIndex j = 0; j + 1 < ySize; ++j) {
      if (!(edgeOnly ? atEdge(i, j) : inEnvelope(i, j))) continue;
      for (int s = 0; s < PairHMM::TotalStates; ++s)
        out << std::setw(16) << cell(i, j, (State)s) << std::setw(6) << i << std::setw(6) << j << std::setw(6)
            << PairHMM::stateName((State)s, i == 0, j == 0) << std::endl;
    }
}

string DPMatrix::toString(bool edgeOnly) const {
  std::ostringstream text;
  write(text, edgeOnly);
  return text.str();
}

DPMatrix::random_engine DPMatrix::newRNG() { return random_engine(); }

// ---------------------------------------------------------------------------------------------------------------------
// Forward: the moves into a cell
// ---------------------------------------------------------------------------------------------------------------------
// Every cell a move into `dest` can come from, with the log-weight of the move (pair-HMM transition + the profiles'
// transitions), emission of `dest` excluded.  This is the recursion of the fill read backwards (reference
// src/forward.cpp:98-199 and :326-398 must agree, and slowFillTest checks that they do):
//   an x move  (IMD, IIW, or IMM while x is in a null state) comes along x's in-transitions, y staying put - allowed
//              while y is ready; into an emitting x state from any pair-HMM source state, into a null x state within
//              the same pair-HMM state;
//   a y move   (IDM, IMI, or IMM while y is null and x emits or is the start) likewise along y's in-transitions;
//   an xy move (IMM with both emitting; the move into EEE from the two END states) along pairs of in-transitions.
void ForwardMatrix::movesInto(const CellCoords& dest, Moves& out) const {
  out.clear();
  const ProfileState& xs = x.state[dest.xpos];
  const ProfileState& ys = y.state[dest.ypos];
  const bool yHolds = ys.isReady() || yEmpty, xHolds = xs.isReady() || xEmpty;
  const StateList& from = kSources[dest.state];
  const auto alongX = [&](bool sameState) {
    for (ProfileTransitionIndex ti : xs.in) {
      const ProfileTransition& t = x.trans[ti];
      if (sameState) out.emplace_back(Cell(t.src, dest.ypos, dest.state), t.lpTrans);
      else
        for (int k = 0; k < from.n; ++k) out.emplace_back(Cell(t.src, dest.ypos, from.s[k]), hmm.lpTrans(from.s[k], dest.state) + t.lpTrans);
    }
  };
  const auto alongY = [&](bool sameState) {
    for (ProfileTransitionIndex ti : ys.in) {
      const ProfileTransition& t = y.trans[ti];
      if (sameState) out.emplace_back(Cell(dest.xpos, t.src, dest.state), t.lpTrans);
      else
        for (int k = 0; k < from.n; ++k) out.emplace_back(Cell(dest.xpos, t.src, from.s[k]), hmm.lpTrans(from.s[k], dest.state) + t.lpTrans);
    }
  };
  const auto alongBoth = [&](const ProfileState& xTo, const ProfileState& yTo) {
    for (ProfileTransitionIndex xi : xTo.in)
      for (ProfileTransitionIndex yi : yTo.in) {
        const ProfileTransition& xt = x.trans[xi];
        const ProfileTransition& yt = y.trans[yi];
        for (int k = 0; k < from.n; ++k)
          out.emplace_back(Cell(xt.src, yt.src, from.s[k]), hmm.lpTrans(from.s[k], dest.state) + xt.lpTrans + yt.lpTrans);
      }
  };
  const bool xInside = dest.xpos + 1 < xSize, yInside = dest.ypos + 1 < ySize;
  switch (dest.state) {
    case PairHMM::IMD:
    case PairHMM::IIW:
      if (xs.isNull()) { if (yHolds && xInside) alongX(true); }
      else if (yHolds) alongX(false);
      break;
    case PairHMM::IDM:
    case PairHMM::IMI:
      if (ys.isNull()) { if (yInside) alongY(true); }
      else if (xHolds) alongY(false);
      break;
    case PairHMM::IMM:
      if (ys.isNull() && xs.isEmitOrStart()) { if (yInside) alongY(true); }
      else if (xs.isNull()) { if (yHolds && xInside) alongX(true); }
      else if (ys.isEmit()) alongBoth(xs, ys);
      break;
    case PairHMM::EEE:
      if (!xInside && !yInside) alongBoth(x.end(), y.end());
      break;
    default: Abort("%s fail", __func__);
  }
  settle(out);
}

map<DPMatrix::CellCoords, LogProb> ForwardMatrix::sourceTransitionsWithoutEmitOrAbsorb(const CellCoords& destCell) {
  Moves m;
  movesInto(destCell, m);
  return asMap(m);
}

map<DPMatrix::CellCoords, LogProb> ForwardMatrix::sourceTransitions(const CellCoords& destCell) {
  Moves m;
  movesInto(destCell, m);
  const LogProb absorbed = lpCellEmitOrAbsorb(destCell);
  for (Move& mv : m) mv.second += absorbed;
  return asMap(m);
}

map<DPMatrix::CellCoords, LogProb> ForwardMatrix::sourceCells(const CellCoords& destCell) {
  Moves m;
  scoredSources(destCell, m);
  return asMap(m);
}

// the moves into a cell, each weighted with the destination's emission and the source cell's Forward value
void ForwardMatrix::scoredSources(const CellCoords& dest, Moves& m) {
  movesInto(dest, m);
  const LogProb absorbed = lpCellEmitOrAbsorb(dest);
  for (Move& mv : m) mv.second = (mv.second + absorbed) + cell(mv.first);
}

// ---- tracebacks -------------------------------------------------------------------------------------------------------
ForwardMatrix::Path ForwardMatrix::sampleTrace(random_engine& generator) {
  Assert(lpEnd > kNegInf, "Forward likelihood is zero; traceback fail");
  const double t0 = wallSeconds();
  Path path(1, endCell);
  Moves m;
  CellCoords at = endCell;
  do {
    scoredSources(at, m);
    at = pickSampled(m, generator);
    path.push_front(at);
  } while (at.xpos != 0 || at.ypos != 0);
  fillTiming.hostTraces += wallSeconds() - t0;
  return path;
}

ForwardMatrix::Path ForwardMatrix::bestTrace(const CellCoords& end) { return bestTrace(end, NULL); }

ForwardMatrix::Path ForwardMatrix::bestTrace(const CellCoords& end, bool* nearTie) {
  const double t0 = wallSeconds();
  Path path(1, end);
  Moves m;
  CellCoords at = end;
  bool tie = false;
  while (at.xpos != 0 || at.ypos != 0) {
    scoredSources(at, m);
    at = pickBest(m);
    if (nearTie && !tie) {
      // as k_best_trace does (hx_trace.hip, HX_TRACE_TIE_TOL): another source cell within 1e-9 of the best, relative
      double top = kNegInf, second = kNegInf;
      for (const Move& c : m) {
        if (c.second > top) { second = top; top = c.second; }
        else if (c.second > second) second = c.second;
      }
      tie = second > kNegInf && top - second <= 1e-9 * std::max(1.0, std::fabs(top));
    }
    path.push_front(at);
  }
  if (nearTie) *nearTie = tie;
  fillTiming.hostTraces += wallSeconds() - t0;
  return path;
}

ForwardMatrix::Path ForwardMatrix::bestTrace() {
  Assert(lpEnd > kNegInf, "Forward likelihood is zero; traceback fail");
  // Where the walk meets a near tie between two source cells - two equally probable routes through a general profile - the
  // choice hangs on the last bits of the arithmetic, and only the exact policy has the reference's bits (DESIGN.md section 6:
  // 3 of 640 random internal-node pairs part from the reference's path in the fast policy, every one of them at such a step).
  // With HX_TIE_REFILL=1 such a pair is filled once more under the exact policy and THAT trace returned (off by default: near
  // ties are common and the second fill is not cheap, see refillAtNearTies).
  if (haveHostCells || !batch || !handle || !deviceTraceback()) {
    bool tie = false;
    Path walked = bestTrace(endCell, &tie);
    return tie && refillAtNearTies() ? exactBestTrace() : walked;
  }
  // the matrix is still device-resident: walk it there (one wavefront per job, all jobs of the batch at once)
  BatchHandle& h = *handle;
  if (!h.bestTracesDone) {
    const double t0 = wallSeconds();
    long long cap = 0;
    for (int k = 0; k < h.nJobs; ++k) {
      hx_layout lay;
      detail::check(hx_batch_layout(h.b, k, 0, &lay), "hx_batch_layout");
      cap = std::max(cap, (long long)lay.n_rows + lay.n_cols + 4);
    }
    h.bestTraceCap = cap;
    h.bestTraceCells.resize(3 * (size_t)cap * h.nJobs);
    h.bestTraceLen.assign(h.nJobs, 0);
    detail::check(hx_batch_best_trace(h.b, reinterpret_cast<hx_trace_cell*>(h.bestTraceCells.data()), cap, h.bestTraceLen.data()),
                  "hx_batch_best_trace");
    h.bestTraceTies.assign(h.nJobs, 0);
    detail::check(hx_batch_best_trace_ties(h.b, h.bestTraceTies.data()), "hx_batch_best_trace_ties");
    h.bestTracesDone = true;
    fillTiming.deviceTrace += wallSeconds() - t0;
    fillTiming.deviceTraces += 1;
  }
  const int len = h.bestTraceLen[jobIndex];
  Assert(len > 0, "traceback failure");
  if (h.bestTraceTies[jobIndex] && refillAtNearTies()) return exactBestTrace();
  const hx_trace_cell* tc = reinterpret_cast<const hx_trace_cell*>(h.bestTraceCells.data()) + (size_t)h.bestTraceCap * jobIndex;
  Path path;
  for (int k = 0; k < len; ++k) path.emplace_back(tc[k].xpos, tc[k].ypos, (State)tc[k].state);
  return path;
}

AlignPath ForwardMatrix::bestAlignPath() { return traceAlignPath(bestTrace()); }

// ---------------------------------------------------------------------------------------------------------------------
// alignment paths of cells, moves and whole traces (reference src/forward.cpp:470-526, 654-684)
// ---------------------------------------------------------------------------------------------------------------------
LogProb ForwardMatrix::eliminatedLogProbInsert(const CellCoords& c) const {
  if (c.state == PairHMM::IIW) return x.state[c.xpos].isNull() ? 0 : insx[c.xpos];
  if (c.state == PairHMM::IMI) return y.state[c.ypos].isNull() ? 0 : insy[c.ypos];
  if ((int)c.state < 0 || c.state > PairHMM::EEE) Abort("%s fail", __func__);
  return 0;
}

ProfileState::SeqCoords ForwardMatrix::cellSeqCoords(const CellCoords& c) const {
  ProfileState::SeqCoords both(y.state[c.ypos].seqCoords);
  for (const auto& rc : x.state[c.xpos].seqCoords)
    if (!both.count(rc.first)) both.insert(rc);       // (a row in both: y's coordinate stands)
  return both;
}

// the alignment columns a cell contributes: those of the profile state(s) it enters, plus the parent's residue when it absorbs
AlignPath ForwardMatrix::cellAlignPath(const CellCoords& c) const {
  const ProfileState& xs = x.state[c.xpos];
  const ProfileState& ys = y.state[c.ypos];
  AlignPath cols;
  switch (c.state) {
    case PairHMM::IMM:
      if (xs.isEmit() && ys.isEmit()) cols = alignPathUnion(xs.alignPath, ys.alignPath);
      else cols = xs.isEmitOrStart() ? ys.alignPath : xs.alignPath;
      break;
    case PairHMM::IMD:
    case PairHMM::IIW: cols = xs.alignPath; break;
    case PairHMM::IDM:
    case PairHMM::IMI: cols = ys.alignPath; break;
    case PairHMM::EEE: break;
    default: Abort("%s fail", __func__);
  }
  if (isAbsorbing(c)) cols[parentRowIndex].push_back(true);
  return cols;
}

AlignPath ForwardMatrix::transitionAlignPath(const CellCoords& src, const CellCoords& dest) const {
  AlignPath cols;
  if (src.xpos != dest.xpos) cols = x.getTrans(src.xpos, dest.xpos)->alignPath;
  if (src.ypos != dest.ypos) cols = alignPathConcat(cols, y.getTrans(src.ypos, dest.ypos)->alignPath);
  return cols;
}

AlignPath ForwardMatrix::traceAlignPath(const Path& path) const {
  AlignPath whole;
  map<AlignRowIndex, SeqIdx> residues;               // residues of every row laid down so far
  const auto countInto = [&](const AlignPath& part) {
    for (const auto& row : part) residues[row.first] += alignPathResiduesInRow(row.second);
  };
  const auto agree = [&](const ProfileState& st, const char* axis) {
    for (const auto& rc : st.seqCoords)
      Assert(residues[rc.first] == rc.second, "Sequence %d: cell %s-coord is %d, path %s-coord is %d", (int)rc.first, axis, (int)rc.second, axis,
             (int)residues[rc.first]);
  };
  for (auto at = path.begin(); at != path.end(); ++at) {
    const AlignPath here = cellAlignPath(*at);
    whole = alignPathConcat(whole, here);
    const auto next = std::next(at);
    if (next == path.end()) break;
    countInto(here);
    agree(x.state[at->xpos], "x");
    agree(y.state[at->ypos], "y");
    const AlignPath step = transitionAlignPath(*at, *next);
    whole = alignPathConcat(whole, step);
    countInto(step);
  }
  for (AlignRowIndex row : {parentRowIndex, x.rootRowIndex, y.rootRowIndex}) ensureAlignPathHasRow(whole, row);
  (void)alignPathColumns(whole);
  return whole;
}

// ---------------------------------------------------------------------------------------------------------------------
// The parent profile of a set of cells (reference src/forward.cpp:686-843).
//
// Absorbing cells, the start and end cells and cells where paths fork become profile states; every other chosen cell is
// summed out: an "effective transition" between two retained cells carries the total weight of all paths through
// eliminated cells between them, and the alignment columns of the best such path.  Cells are handled in reverse cell
// order, so that when a cell is processed the effective transitions leaving it are complete.
// ---------------------------------------------------------------------------------------------------------------------
ForwardMatrix::EffectiveTransition::EffectiveTransition() : lpBestAlignPath(kNegInf), lpPath(kNegInf) {}

Profile ForwardMatrix::makeProfile(const set<CellCoords>& cells, ProfilingStrategy strategy) {
  Assert(cells.count(startCell), "Missing SSS");
  Assert(cells.count(endCell), "Missing EEE");
  const double tStart = wallSeconds();
  if (!haveHostCells && batch) prefetchCells(cells);   // the fwdLogProb annotations below read these cells

  const vguard<Cell> chosen(cells.begin(), cells.end());            // cell order
  const size_t n = chosen.size();
  const auto indexOf = [&](const Cell& c) -> size_t {              // n: not a chosen cell
    const auto at = std::lower_bound(chosen.begin(), chosen.end(), c);
    return (at != chosen.end() && *at == c) ? (size_t)(at - chosen.begin()) : n;
  };

  // moves into every chosen cell (emission excluded), restricted to chosen sources; and how many moves leave each cell
  vguard<Moves> into(n);
  vguard<vguard<size_t> > fromIdx(n);
  vguard<int> leaving(n, 0);
  {
    Moves m;
    for (size_t k = 0; k < n; ++k) {
      movesInto(chosen[k], m);
      for (const Move& mv : m) {
        const size_t s = indexOf(mv.first);
        if (s == n) continue;
        into[k].push_back(mv);
        fromIdx[k].push_back(s);
        ++leaving[s];
      }
    }
  }

  Profile prof(hmm.components(), alphSize, parentRowIndex);
  prof.name = pairParentName(x.name, hmm.l.t, y.name, hmm.r.t);
  prof.meta["node"] = std::to_string(parentRowIndex);

  // which cells become states
  const bool keepEverything = (strategy & KeepGapsOpen) != 0 || (strategy & CollapseChains) == 0;
  const ProfileStateIndex none = (ProfileStateIndex)-1;
  vguard<ProfileStateIndex> stateOf(n, none);
  for (size_t k = 0; k < n; ++k) {
    const Cell& c = chosen[k];
    const bool absorbs = isAbsorbing(c);
    if (!(absorbs || c == startCell || c == endCell || leaving[k] > 1 || keepEverything)) continue;
    stateOf[k] = prof.state.size();
    prof.state.push_back(ProfileState());
    ProfileState& st = prof.state.back();
    if (absorbs) {
      // the parent's likelihood vector: the product of the two branch-multiplied child vectors (no root prior)
      if (c.state == PairHMM::IMM) {
        initAbsorbScratch(c.xpos, c.ypos);
        st.lpAbsorb = absorbScratch;
      } else
        st.lpAbsorb = c.state == PairHMM::IMD ? subx.state[c.xpos].lpAbsorb : suby.state[c.ypos].lpAbsorb;
    }
    st.alignPath = cellAlignPath(c);
    st.seqCoords = cellSeqCoords(c);
    st.name = cellName(c);
    st.meta["fwdLogProb"] = std::to_string(c.state == PairHMM::EEE ? lpEnd : cell(c.xpos, c.ypos, c.state));
  }
  if (strategy & KeepGapsOpen)
    for (size_t k = 0; k < n; ++k) {
      if (stateOf[k] == none || isAbsorbing(chosen[k])) continue;
      const list<CellCoords> twin = equivAbsorbCells(chosen[k]);
      if (twin.empty()) continue;
      const size_t t = indexOf(twin.front());
      if (t != n && stateOf[t] != none) prof.equivAbsorbState[stateOf[k]] = stateOf[t];
    }

  // effective transitions leaving each chosen cell, by destination state index (ascending)
  typedef std::pair<ProfileStateIndex, EffectiveTransition> Reach;
  vguard<vguard<Reach> > reach(n);
  const auto slot = [](vguard<Reach>& list, ProfileStateIndex to) -> EffectiveTransition& {
    auto at = std::lower_bound(list.begin(), list.end(), to, [](const Reach& r, ProfileStateIndex v) { return r.first < v; });
    if (at == list.end() || at->first != to) at = list.insert(at, Reach(to, EffectiveTransition()));
    return at->second;
  };
  for (size_t k = n; k-- > 0;) {
    const Cell& c = chosen[k];
    const LogProb inserted = eliminatedLogProbInsert(c);
    if (stateOf[k] != none) {
      // a retained cell: every move into it is a transition into its state
      for (size_t q = 0; q < into[k].size(); ++q) {
        EffectiveTransition& e = slot(reach[fromIdx[k][q]], stateOf[k]);
        e.lpPath = e.lpBestAlignPath = into[k][q].second + inserted;
        e.bestAlignPath = transitionAlignPath(into[k][q].first, c);
      }
      continue;
    }
    // an eliminated cell: whatever it reaches, its sources reach through it
    const vguard<Reach> onward(reach[k]);
    if (onward.empty()) continue;
    const AlignPath here = cellAlignPath(c);
    for (size_t q = 0; q < into[k].size(); ++q) {
      const LogProb through = into[k][q].second;
      const AlignPath step = transitionAlignPath(into[k][q].first, c);
      vguard<Reach>& ofSource = reach[fromIdx[k][q]];
      for (const Reach& r : onward) {
        EffectiveTransition& e = slot(ofSource, r.first);
        log_accum_exp(e.lpPath, through + inserted + r.second.lpPath);
        const LogProb best = through + inserted + r.second.lpBestAlignPath;
        if (best > e.lpBestAlignPath) {
          e.lpBestAlignPath = best;
          e.bestAlignPath = alignPathConcat(step, here, r.second.bestAlignPath);
        }
      }
    }
  }

  // the profile's transitions: retained cells in cell order, destinations ascending
  for (size_t k = 0; k < n; ++k) {
    if (stateOf[k] == none) continue;
    for (const Reach& r : reach[k]) {
      const ProfileTransitionIndex ti = prof.trans.size();
      ProfileTransition t;
      t.src = stateOf[k];
      t.dest = r.first;
      t.lpTrans = r.second.lpPath;
      t.alignPath = r.second.bestAlignPath;
      prof.trans.push_back(t);
      ProfileState& from = prof.state[t.src];
      (prof.state[t.dest].isNull() ? from.nullOut : from.absorbOut).push_back(ti);
      prof.state[t.dest].in.push_back(ti);
    }
  }

  prof.seq = x.seq;
  prof.seq.insert(y.seq.begin(), y.seq.end());
  prof.assertTransitionsConsistent();
  prof.assertPathToEndExists();
  prof = Profile::withReadyStates(std::move(prof));
  prof.assertSeqCoordsConsistent();
  fillTiming.hostMakeProfile += wallSeconds() - tStart;
  return prof;
}

// The cells of the best trace (counted twice) and of sampled traces (reference src/forward.cpp:845-889).  Each accepted
// sample takes draws from the shared generator; a sample whose ancestral length is out of bounds ends the sampling.
Profile ForwardMatrix::sampleProfile(random_engine& generator, size_t profileSamples, size_t maxCells, ProfilingStrategy strategy,
                                     size_t minLen, size_t maxLen) {
  return makeProfile(sampleCells(generator, profileSamples, maxCells, strategy, minLen, maxLen), strategy);
}

// The walks sampleTrace would make from the generator's present state, made on the device in the matrix where it is
// (hx_batch_sample_traces; HX_DEVICE_SAMPLING=1): the generator's canonical uniforms are drawn from a COPY of it, and the
// caller advances the generator itself by what the walks it keeps have used (two 32-bit draws per step).  Empty when a walk
// could not be made there: the caller then samples on the host as before, from the untouched generator.
bool ForwardMatrix::deviceSampling() {
  static const bool on = getenv("HX_DEVICE_SAMPLING") != NULL && atoi(getenv("HX_DEVICE_SAMPLING")) != 0;
  return on;
}
bool ForwardMatrix::sampleTracesOnDevice(const random_engine& generator, size_t walks, vguard<Path>& paths, vguard<long long>& draws) {
  paths.clear();
  draws.clear();
  if (!batch || haveHostCells || walks == 0) return false;
  const double t0 = wallSeconds();
  const long long cap = (long long)xSize + ySize + 4;
  random_engine ahead(generator);
  vguard<double> uniforms((size_t)cap * walks);
  for (double& u : uniforms) u = std::generate_canonical<double, 53>(ahead);
  vguard<hx_trace_cell> cells((size_t)cap * walks);
  vguard<int32_t> len(walks, 0);
  vguard<int64_t> used(walks, 0);
  detail::check(hx_batch_sample_traces(batch, jobIndex, (int32_t)walks, uniforms.data(), (int64_t)uniforms.size(), cells.data(), cap,
                                 len.data(), used.data()), "hx_batch_sample_traces");
  for (size_t w = 0; w < walks; ++w)
    if (len[w] <= 0) return false;
  paths.resize(walks);
  draws.resize(walks);
  for (size_t w = 0; w < walks; ++w) {
    const hx_trace_cell* tc = cells.data() + (size_t)cap * w;
    for (int k = 0; k < len[w]; ++k) paths[w].emplace_back(tc[k].xpos, tc[k].ypos, (State)tc[k].state);
    draws[w] = used[w];
  }
  fillTiming.deviceTrace += wallSeconds() - t0;
  fillTiming.deviceTraces += 1;
  return true;
}

set<ForwardMatrix::CellCoords> ForwardMatrix::sampleCells(random_engine& generator, size_t profileSamples, size_t maxCells,
                                                          ProfilingStrategy strategy, size_t minLen, size_t maxLen) {
  Require((strategy & IncludeBestTrace) || profileSamples > 0, "Must allow at least one sample path in the profile");
  const auto ancestralLength = [](const Path& p) {
    return (size_t)std::count_if(p.begin(), p.end(), [](const CellCoords& c) {
      return c.state == PairHMM::IMM || c.state == PairHMM::IDM || c.state == PairHMM::IMD;
    });
  };
  set<CellCoords> keep;
  // sampled traces walk the matrix on the host, so it is coming over anyway: the best trace then walks it there too
  // (a thousand-odd steps) instead of waiting for a device traceback kernel
  vguard<Path> devicePaths;
  vguard<long long> deviceDraws;
  const bool onDevice = profileSamples > 0 && deviceSampling() && sampleTracesOnDevice(generator, profileSamples, devicePaths, deviceDraws);
  if (profileSamples > 0 && batch && !onDevice) ensureHostCells();
  // the next walk: the device's, in order, or one more walk on the host; the generator moves past the walks taken
  size_t taken = 0;
  const auto nextWalk = [&]() -> Path {
    if (!onDevice) return sampleTrace(generator);
    generator.discard(2 * (unsigned long long)(deviceDraws[taken] - (taken ? deviceDraws[taken - 1] : 0)));
    return devicePaths[taken++];
  };
  if (maxCells == 0) {
    // no cell budget: every visited cell stays, so the visits need not be counted - collect, sort, drop repeats
    vguard<CellCoords> seen;
    if (strategy & IncludeBestTrace) {
      const Path best = bestTrace();
      seen.assign(best.begin(), best.end());
    }
    for (size_t accepted = 0; accepted < profileSamples; ++accepted) {
      const Path sampled = nextWalk();
      const size_t ancestral = ancestralLength(sampled);
      if (ancestral < minLen || ancestral > maxLen) break;
      seen.insert(seen.end(), sampled.begin(), sampled.end());
    }
    const double t0 = wallSeconds();
    std::sort(seen.begin(), seen.end());
    seen.erase(std::unique(seen.begin(), seen.end()), seen.end());
    keep.insert(seen.begin(), seen.end());        // (sorted input: linear-time construction)
    fillTiming.cellSets += wallSeconds() - t0;
    return keep;
  }
  map<CellCoords, size_t> visits;
  size_t traces = 0;
  if (strategy & IncludeBestTrace) {
    for (const CellCoords& c : bestTrace()) visits[c] = 2;
    ++traces;
  }
  for (size_t accepted = 0; accepted < profileSamples && visits.size() < maxCells; ++accepted) {
    const Path sampled = nextWalk();
    const size_t ancestral = ancestralLength(sampled);
    if (ancestral < minLen || ancestral > maxLen) break;
    for (const CellCoords& c : sampled) ++visits[c];
    ++traces;
  }
  // with a cell budget that the traces have used up, only cells seen twice stay
  const size_t needed = (traces > 1 && visits.size() >= maxCells) ? 2 : 1;
  for (const auto& v : visits)
    if (v.second >= needed) keep.insert(keep.end(), v.first);
  return keep;
}

Profile ForwardMatrix::bestProfile(ProfilingStrategy strategy) {
  const Path best = bestTrace();
  return makeProfile(set<CellCoords>(best.begin(), best.end()), strategy);
}

// Self-check (reference src/forward.cpp:1099-1126): every stored Forward cell must equal the sum, in libm arithmetic,
// over the moves into it of move weight x source cell.  Run when Forward and Backward disagree; warns per cell.
void ForwardMatrix::slowFillTest() {
  Moves m;
  for (ProfileStateIndex i = 0; i < xSize; ++i)
    for (ProfileStateIndex j = 0; j < ySize; ++j) {
      if (!inEnvelope(i, j)) continue;
      for (int s = 0; s <= PairHMM::EEE; ++s) {
        const bool isEnd = s == PairHMM::EEE && i + 1 == xSize && j + 1 == ySize;
        const bool stored = i + 1 < xSize && j + 1 < ySize && s != PairHMM::EEE;
        if (!stored && !isEnd) continue;
        const CellCoords c(i, j, (State)s);
        const LogProb have = isEnd ? lpEnd : cell(c);
        LogProb slow = (s == PairHMM::SSS && i == 0 && j == 0) ? 0 : kNegInf;
        scoredSources(c, m);
        for (const Move& mv : m)
          if (mv.second > kNegInf) log_accum_exp_slow(slow, mv.second);
        Test(nearlyEqual(slow, have, FWD_BACK_ERROR_TOLERANCE), "Forward cell %s score (%g) doesn't match slow computation (%g)",
             cellName(c).c_str(), have, slow);
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward: the moves out of a cell, posterior decoding
// ---------------------------------------------------------------------------------------------------------------------
// Every cell a move out of `src` can lead to, with the log-weight of the move including what the destination emits
// (reference src/forward.cpp:1224-1285); the mirror image of ForwardMatrix::movesInto.
void BackwardMatrix::movesOutOf(const CellCoords& src, Moves& out) {
  out.clear();
  const ProfileState& xs = x.state[src.xpos];
  const ProfileState& ys = y.state[src.ypos];
  const bool yHolds = ys.isReady() || yEmpty, xHolds = xs.isReady() || xEmpty;
  const ProfileStateIndex xEnd = xSize - 1, yEnd = ySize - 1;
  const State s = src.state;
  for (ProfileTransitionIndex xi : xs.absorbOut)
    for (ProfileTransitionIndex yi : ys.absorbOut)
      out.emplace_back(Cell(x.trans[xi].dest, y.trans[yi].dest, PairHMM::IMM), hmm.lpTrans(s, PairHMM::IMM) + x.trans[xi].lpTrans + y.trans[yi].lpTrans);
  if (yHolds)
    for (ProfileTransitionIndex xi : xs.absorbOut) {
      const ProfileTransition& t = x.trans[xi];
      out.emplace_back(Cell(t.dest, src.ypos, PairHMM::IMD), hmm.lpTrans(s, PairHMM::IMD) + t.lpTrans);
      out.emplace_back(Cell(t.dest, src.ypos, PairHMM::IIW), hmm.lpTrans(s, PairHMM::IIW) + t.lpTrans);
    }
  if (xHolds)
    for (ProfileTransitionIndex yi : ys.absorbOut) {
      const ProfileTransition& t = y.trans[yi];
      out.emplace_back(Cell(src.xpos, t.dest, PairHMM::IDM), hmm.lpTrans(s, PairHMM::IDM) + t.lpTrans);
      out.emplace_back(Cell(src.xpos, t.dest, PairHMM::IMI), hmm.lpTrans(s, PairHMM::IMI) + t.lpTrans);
    }
  // null moves keep the pair-HMM state; x's are made while y is ready, y's come first for IMM while x emits or is the start
  if (yHolds && (s == PairHMM::IMD || s == PairHMM::IIW || s == PairHMM::IMM))
    for (ProfileTransitionIndex xi : xs.nullOut)
      if (x.trans[xi].dest != xEnd) out.emplace_back(Cell(x.trans[xi].dest, src.ypos, s), x.trans[xi].lpTrans);
  if (s == PairHMM::IDM || s == PairHMM::IMI || (s == PairHMM::IMM && xs.isEmitOrStart()))
    for (ProfileTransitionIndex yi : ys.nullOut)
      if (y.trans[yi].dest != yEnd) out.emplace_back(Cell(src.xpos, y.trans[yi].dest, s), y.trans[yi].lpTrans);
  // both profiles step into their END states together
  for (ProfileTransitionIndex xi : xs.nullOut)
    if (x.trans[xi].dest == xEnd)
      for (ProfileTransitionIndex yi : ys.nullOut)
        if (y.trans[yi].dest == yEnd)
          out.emplace_back(Cell(xEnd, yEnd, PairHMM::EEE), x.trans[xi].lpTrans + y.trans[yi].lpTrans + hmm.lpTrans(s, PairHMM::EEE));
  settle(out);
  for (Move& mv : out) mv.second += lpCellEmitOrAbsorb(mv.first);
}

map<DPMatrix::CellCoords, LogProb> BackwardMatrix::destTransitions(const CellCoords& srcCell) {
  Moves m;
  movesOutOf(srcCell, m);
  return asMap(m);
}

void BackwardMatrix::scoredDestinations(const CellCoords& src, Moves& m) {
  movesOutOf(src, m);
  for (Move& mv : m)
    if (mv.first.state != PairHMM::EEE) mv.second += cell(mv.first);
}

map<DPMatrix::CellCoords, LogProb> BackwardMatrix::destCells(const CellCoords& srcCell) {
  Moves m;
  scoredDestinations(srcCell, m);
  return asMap(m);
}

BackwardMatrix::BackwardMatrix(ForwardMatrix& fwd) : DPMatrix(fwd.x, fwd.y, fwd.hmm, fwd.envelope), fwd(fwd) {
  // the Forward object's device job already holds the prepared vectors; the reference recomputes them
  // (src/forward.cpp:976) -- here they are shared
  handle = fwd.handle;
  batch = fwd.batch;
  jobIndex = fwd.jobIndex;
  which = 1;
  stripStride = fwd.stripStride;
  planeStride = fwd.planeStride;
  blockStride = fwd.blockStride;
  matrixDoubles = fwd.matrixDoubles;
  subx = fwd.subx;
  suby = fwd.suby;
  insx = fwd.insx; insy = fwd.insy; rootsubx = fwd.rootsubx; rootsuby = fwd.rootsuby;
  lpEnd = 0;
  const double tBack = wallSeconds();
  if (!handle->backwardDone) {      // one launch fills the Backward matrices of every job of the batch
    detail::check(hx_batch_backward(batch, NULL), "hx_batch_backward");
    handle->backwardDone = true;
    // posterior decoding walks both matrices on the host: their page-locked buffers, while the Backward fill runs
    detail::pinnedReserve((size_t)matrixDoubles, 2);
  }
  vguard<double> lpStarts((size_t)handle->nJobs, kNegInf);
  detail::check(hx_batch_lp_start(batch, lpStarts.data()), "hx_batch_lp_start");
  fillTiming.backwardWait += wallSeconds() - tBack;
  const double lpStartDev = lpStarts[(size_t)jobIndex];
  if (!nearlyEqual(lpStartDev, fwd.lpEnd, FWD_BACK_ERROR_TOLERANCE)) {
    // the two fills disagree by more than 1 %: find the cells and moves that do not add up (reference src/forward.cpp:1091-1096)
    fwd.slowFillTest();
    slowFillTest();
    sourceDestTransTest();
    Warn("Forward log-likelihood is %g, Backward log-likelihood is %g", fwd.lpEnd, lpStartDev);
  }
}

double BackwardMatrix::cellPostProb(const CellCoords& c) const { return exp(fwd.cell(c) + cell(c) - fwd.lpEnd); }

double BackwardMatrix::transPostProb(const CellCoords& src, const CellCoords& dest) const {
  Moves m;
  fwd.movesInto(dest, m);
  const auto at = std::lower_bound(m.begin(), m.end(), src, [](const Move& mv, const CellCoords& c) { return mv.first < c; });
  if (at == m.end() || !(at->first == src)) return 0;
  return exp(fwd.cell(src) + (at->second + fwd.lpCellEmitOrAbsorb(dest)) + cell(dest) - fwd.lpEnd);
}

BackwardMatrix::Path BackwardMatrix::bestTrace(const CellCoords& traceStart) {
  Path path;
  Moves m;
  CellCoords at = traceStart;
  while (at.xpos + 1 < xSize && at.ypos + 1 < ySize) {
    scoredDestinations(at, m);
    at = pickBest(m);
    path.push_back(at);
  }
  path.push_back(endCell);
  return path;
}

// reference src/forward.cpp:1302-1319: the O(cells) scan runs on the device (stream compaction); the
// candidates are pushed in the reference's visiting order so that the heap -- and therefore the pop
// order of exact ties -- is the one std::priority_queue builds there.
std::priority_queue<BackwardMatrix::CellPostProb> BackwardMatrix::cellsAbovePostProbThreshold(double minPostProb) const {
  std::priority_queue<CellPostProb> bc;
  int64_t n = 0;
  detail::check(hx_batch_posterior_scan(batch, jobIndex, minPostProb, NULL, 0, &n), "hx_batch_posterior_scan");
  vguard<hx_cell> found((size_t)n);
  if (n > 0) {
    int64_t n2 = 0;
    detail::check(hx_batch_posterior_scan(batch, jobIndex, minPostProb, found.data(), n, &n2), "hx_batch_posterior_scan");
    found.resize((size_t)std::min(n, n2));
  }
  std::sort(found.begin(), found.end(), [](const hx_cell& a, const hx_cell& b) {
    if (a.xpos != b.xpos) return a.xpos > b.xpos;     // i descending, j descending, state ascending
    if (a.ypos != b.ypos) return a.ypos > b.ypos;
    return a.state < b.state;
  });
  for (const auto& c : found) bc.push(CellPostProb(c.xpos, c.ypos, (State)c.state, c.log_post_prob));
  return bc;
}

// Add the best path through `via` (best Forward trace into it + best Backward trace out of it) to the cell set; false when
// the cell budget does not allow it (reference src/forward.cpp:1343-1379).  Only the parts of the two traces that are
// not yet in the set are added: each trace is followed away from `via` until it meets the set.
bool BackwardMatrix::addCells(set<CellCoords>& cells, size_t maxCells, const list<CellCoords>& fwdTrace, const list<CellCoords>& backTrace,
                              bool keepGapsOpen) {
  vguard<CellCoords> fresh;
  for (auto c = fwdTrace.rbegin(); c != fwdTrace.rend() && !cells.count(*c); ++c) fresh.push_back(*c);
  for (auto c = backTrace.begin(); c != backTrace.end() && !cells.count(*c); ++c) fresh.push_back(*c);
  if (maxCells > 0 && !cells.empty() && cells.size() + fresh.size() > maxCells) return false;
  cells.insert(fresh.begin(), fresh.end());
  if (keepGapsOpen)
    for (const CellCoords& c : fresh)
      for (const CellCoords& twin : equivAbsorbCells(c))
        if (!cells.count(twin) && cellPostProb(twin) > 0 && inEnvelope(twin.xpos, twin.ypos)) addTrace(twin, cells, maxCells, false);
  return true;
}

bool BackwardMatrix::addTrace(const CellCoords& via, set<CellCoords>& cells, size_t maxCells, bool keepGapsOpen) {
  const Path before = fwd.bestTrace(via), after = bestTrace(via);
  return addCells(cells, maxCells, before, after, keepGapsOpen);
}

Profile BackwardMatrix::bestProfile(ProfilingStrategy strategy) {
  set<CellCoords> cells;
  addTrace(endCell, cells, 0, (strategy & KeepGapsOpen) != 0);
  return fwd.makeProfile(cells, strategy);
}

// cells by decreasing posterior probability, each brought in with the best path through it, until the budget is used
Profile BackwardMatrix::postProbProfile(double minPostProb, size_t maxCells, ProfilingStrategy strategy) {
  const bool open = (strategy & KeepGapsOpen) != 0;
  std::priority_queue<CellPostProb> ranked = cellsAbovePostProbThreshold(minPostProb);
  set<CellCoords> cells;
  if (ranked.empty() || (strategy & IncludeBestTrace)) addCells(cells, 0, fwd.bestTrace(), list<CellCoords>(), open);
  while (!ranked.empty() && (maxCells == 0 || cells.size() < maxCells)) {
    const CellCoords top = ranked.top();
    if (cells.count(top)) ranked.pop();
    else if (!addTrace(top, cells, maxCells, open)) break;
  }
  return fwd.makeProfile(cells, strategy);
}

// Self-checks (reference src/forward.cpp:1128-1170), run when Forward and Backward disagree.
// Every stored Backward cell must equal the libm sum over the moves out of it of move weight x destination cell.
void BackwardMatrix::slowFillTest() {
  Moves m;
  for (ProfileStateIndex i = xSize - 1; i-- > 0;)
    for (ProfileStateIndex j = ySize - 1; j-- > 0;) {
      if (!inEnvelope(i, j)) continue;
      for (int s = 0; s < PairHMM::TotalStates; ++s) {
        const CellCoords c(i, j, (State)s);
        LogProb slow = kNegInf;
        scoredDestinations(c, m);
        for (const Move& mv : m)
          if (mv.second > kNegInf) log_accum_exp_slow(slow, mv.second);
        Test(nearlyEqual(slow, cell(c), FWD_BACK_ERROR_TOLERANCE), "Backward cell %s score (%g) doesn't match slow computation (%g)",
             cellName(c).c_str(), cell(c), slow);
      }
    }
}

// Every move the Forward enumeration finds into a cell must be found, with the same weight, by the Backward enumeration
// out of its source, and the other way round.
void BackwardMatrix::sourceDestTransTest() {
  Moves in, out;
  const auto find = [](const Moves& m, const CellCoords& c) -> const Move* {
    const auto at = std::lower_bound(m.begin(), m.end(), c, [](const Move& mv, const CellCoords& v) { return mv.first < v; });
    return (at != m.end() && at->first == c) ? &*at : NULL;
  };
  for (ProfileStateIndex i = 0; i < xSize; ++i)
    for (ProfileStateIndex j = 0; j < ySize; ++j) {
      if (!inEnvelope(i, j)) continue;
      for (int s = 0; s < PairHMM::TotalStates; ++s) {
        const CellCoords c(i, j, (State)s);
        fwd.movesInto(c, in);
        const LogProb absorbed = fwd.lpCellEmitOrAbsorb(c);
        for (const Move& mv : in) {
          const LogProb w = mv.second + absorbed;
          if (!(w > kNegInf)) continue;
          movesOutOf(mv.first, out);
          const Move* back = find(out, c);
          if (!back) Warn("Backward matrix is missing transition between %s and %s that is present in Forward matrix", cellName(mv.first).c_str(), cellName(c).c_str());
          else Test(nearlyEqual(w, back->second, FWD_BACK_ERROR_TOLERANCE), "Forward (%g) & Backward (%g) transitions between %s and %s don't match",
                    w, back->second, cellName(mv.first).c_str(), cellName(c).c_str());
        }
        movesOutOf(c, out);
        for (const Move& mv : out) {
          if (!(mv.second > kNegInf)) continue;
          fwd.movesInto(mv.first, in);
          const Move* there = find(in, c);
          if (!there) Warn("Forward matrix is missing transition between %s and %s that is present in Backward matrix", cellName(c).c_str(), cellName(mv.first).c_str());
          else {
            const LogProb w = there->second + fwd.lpCellEmitOrAbsorb(mv.first);
            Test(nearlyEqual(mv.second, w, FWD_BACK_ERROR_TOLERANCE), "Forward (%g) & Backward (%g) transitions between %s and %s don't match",
                 w, mv.second, cellName(c).c_str(), cellName(mv.first).c_str());
          }
        }
      }
    }
}

}  // namespace historian
