// Profile and PairHMM of the host mirror (interfaces: hx_host.h, after reference src/profile.h and src/pairhmm.h).
//
// A Profile is the DP's input structure - a toposorted state graph with per-state absorption vectors - and, through
// ForwardMatrix::makeProfile, also its output.  What is here: the leaf profile of a sequence, the transformation that
// gives every state a pure "wait" or "ready" role, the check value "profile log-likelihood", the consistency checks the
// reference runs after every construction, and the JSON form the reference's test mains print (their golden files are
// diffed byte for byte, tests/test_host_mirror.py).  PairHMM holds the 23 transition weights of the five-state
// transducer pair as a table; the reference's named members are views of it.
#include "hx_host.h"

#include <algorithm>
#include <cfloat>
#include <sstream>

namespace historian {

namespace {
const double kNegInf = -std::numeric_limits<double>::infinity();
const char* const kWaitSuffix = ";";      // a state that still has null moves to make (reference src/profile.cpp:8-9)
const char* const kReadySuffix = ".";     // its twin, from which only absorbing moves leave
}  // namespace

ProfileTransition::ProfileTransition() : src(0), dest(0), lpTrans(kNegInf) {}
ProfileState::ProfileState() {}
ProfileState::ProfileState(size_t components, AlphTok alphSize) : lpAbsorb(components, vguard<LogProb>(alphSize, kNegInf)) {}

// ------------------------------------------------------------------------------------------------------------------
// leaf profile (what reference src/profile.cpp:23-76 builds): START, one emitting state per residue, END, chained by
// transitions of log-weight 0; the move into END is a null move, every other one absorbs a residue
// ------------------------------------------------------------------------------------------------------------------
Profile::Profile(size_t components, const string& alphabet, const FastSeq& seq, AlignRowIndex rowIndex)
    : components(components), alphSize((AlphTok)alphabet.size()), rootRowIndex(rowIndex), name(seq.name) {
  const size_t len = seq.seq.size();
  state.assign(len + 2, ProfileState());
  ProfileState& first = state.front();
  ProfileState& last = state.back();
  first.name = "START";
  first.seqCoords[rowIndex] = 0;
  last.name = "END";
  last.seqCoords[rowIndex] = (SeqIdx)len;

  // one absorption row per residue: 0 at the residue's token, -inf elsewhere; a wildcard or a character outside the
  // alphabet absorbs anything (all 0)
  size_t outsideAlphabet = 0;
  for (size_t pos = 1; pos <= len; ++pos) {
    const char residue = seq.seq[pos - 1];
    ProfileState& st = state[pos];
    st.name = string(1, residue) + std::to_string(pos);
    st.alignPath[rowIndex] = AlignRowPath(1, true);
    st.seqCoords[rowIndex] = (SeqIdx)pos;
    UnvalidatedAlphTok tok = InvalidAlphabetToken;
    if (!Alignment::isWildcard(residue)) {
      tok = tokenize(residue, alphabet);
      if (tok < 0) outsideAlphabet += components;     // (counted once per mixture component, as the reference does)
    }
    vguard<LogProb> row(alphabet.size(), tok < 0 ? 0. : kNegInf);
    if (tok >= 0) row[tok] = 0;
    st.lpAbsorb.assign(components, row);
  }

  trans.resize(len + 1);
  for (size_t k = 0; k <= len; ++k) {
    ProfileTransition& link = trans[k];
    link.src = k;
    link.dest = k + 1;
    link.lpTrans = 0;
    (k == len ? state[k].nullOut : state[k].absorbOut).push_back(k);
    state[k + 1].in.push_back(k);
  }
  this->seq[rowIndex] = seq.seq;
  if (outsideAlphabet) Warn("%d invalid characters found in sequence %s", (int)outsideAlphabet, seq.name.c_str());
  assertTransitionsConsistent();
  assertSeqCoordsConsistent();
  assertAllStatesWaitOrReady();
  assertPathToEndExists();
}

const ProfileTransition* Profile::getTrans(ProfileStateIndex src, ProfileStateIndex dest) const {
  const vguard<ProfileTransitionIndex>& incoming = state[dest].in;
  const auto hit = std::find_if(incoming.begin(), incoming.end(), [&](ProfileTransitionIndex t) { return trans[t].src == src; });
  return hit == incoming.end() ? NULL : &trans[*hit];
}

bool Profile::isEmpty() const {
  return std::none_of(state.begin(), state.end(), [](const ProfileState& s) { return s.isEmit(); });
}

// ------------------------------------------------------------------------------------------------------------------
// The profile's own likelihood under the insertion distribution: the sum over all START..END paths of transition
// weights times absorption probabilities, by one sweep in state order (reference src/profile.cpp:112-131).  Also the
// per-state annotation the test mains print.
// ------------------------------------------------------------------------------------------------------------------
LogProb Profile::calcSumPathAbsorbProbs(const vguard<LogProb>& logCptWeight, const vguard<vguard<LogProb> >& logInsProb, const char* tag) {
  vguard<LogProb> reach(state.size(), kNegInf);      // log-probability of all paths from START up to and including a state
  reach[0] = 0;
  for (ProfileStateIndex pos = 1; pos < state.size(); ++pos) {
    ProfileState& st = state[pos];
    LogProb absorb = 0;
    if (st.isEmit()) {
      absorb = kNegInf;
      for (size_t cpt = 0; cpt < components; ++cpt)
        log_accum_exp(absorb, logCptWeight[cpt] + logInnerProduct(logInsProb[cpt], st.lpAbsorb[cpt]));
    }
    LogProb& here = reach[pos];
    for (ProfileTransitionIndex ti : st.in) {
      const ProfileTransition& t = trans[ti];
      Assert(t.src < pos, "Transition #%u from %u -> %u is not toposorted", (unsigned)ti, (unsigned)t.src, (unsigned)t.dest);
      log_accum_exp(here, reach[t.src] + t.lpTrans + absorb);
    }
    if (tag) st.meta[tag] = std::to_string(here);
  }
  return reach.back();
}

// ------------------------------------------------------------------------------------------------------------------
// JSON.  The layout (indentation, separators, six-decimal numbers from std::to_string) is the reference's
// (src/profile.cpp:147-204, src/jsonutil.cpp:151-176): its golden files are the byte-for-byte yardstick.
// ------------------------------------------------------------------------------------------------------------------
namespace {
string number(double d) {
  if (d < -DBL_MAX) return "\"-inf\"";
  if (d > DBL_MAX) return "\"inf\"";
  return std::to_string(d);
}

// a string->string object: "{ }" when empty, on one line for a single entry, one entry per line otherwise
string tagObject(const map<string, string>& tags, size_t indent) {
  if (tags.empty()) return "{ }";
  const bool single = tags.size() == 1;
  const string pad(indent, ' ');
  string out = single ? "{ " : "\n" + pad + "{";
  const char* sep = "";
  for (const auto& kv : tags) {
    out += sep;
    if (!single) out += "\n" + pad + " ";
    out += "\"" + kv.first + "\": \"" + kv.second + "\"";
    sep = ",";
  }
  return out + (single ? " }" : "\n" + pad + "}");
}

// [ [ row, "*-*" ], ... ]
string pathArray(const AlignPath& path) {
  string out = "[";
  for (const auto& row : path) {
    if (out.size() > 1) out += ",";
    out += " [ " + std::to_string(row.first) + ", \"";
    for (bool residue : row.second) out += residue ? Alignment::wildcardChar : Alignment::gapChar;
    out += "\" ]";
  }
  return out + " ]";
}
}  // namespace

void Profile::writeJson(std::ostream& out) const {
  out << "{\n";
  if (!name.empty()) out << " \"name\": \"" << name << "\",\n";
  if (!meta.empty()) out << " \"meta\": " << tagObject(meta, 2) << ",\n";
  out << " \"alphSize\": " << alphSize << ",\n \"state\": [\n";
  for (ProfileStateIndex n = 0; n < state.size(); ++n) {
    const ProfileState& st = state[n];
    out << "  {\n   \"n\": " << n << ",\n";
    if (!st.name.empty()) out << "   \"name\": \"" << st.name << "\",\n";
    if (!st.meta.empty()) out << "   \"meta\": " << tagObject(st.meta, 4) << ",\n";
    if (!st.alignPath.empty()) out << "   \"path\": " << pathArray(st.alignPath) << ",\n";
    if (!st.seqCoords.empty()) {
      out << "   \"seqPos\": [";
      const char* sep = " ";
      for (const auto& rc : st.seqCoords) {
        out << sep << "[ " << rc.first << ", " << rc.second << " ]";
        sep = ", ";
      }
      out << " ],\n";
    }
    if (st.isEmit()) {
      out << "   \"lpAbsorb\": [";
      for (size_t cpt = 0; cpt < components; ++cpt) {
        out << (cpt ? ", [" : "[");
        for (AlphTok a = 0; a < alphSize; ++a) out << (a ? ", " : " ") << number(st.lpAbsorb[cpt][a]);
        out << " ]";
      }
      out << "],\n";
    }
    // outgoing transitions in transition-index order, null and absorbing ones together
    vguard<ProfileTransitionIndex> leaving(st.nullOut);
    leaving.insert(leaving.end(), st.absorbOut.begin(), st.absorbOut.end());
    std::sort(leaving.begin(), leaving.end());
    leaving.erase(std::unique(leaving.begin(), leaving.end()), leaving.end());
    out << "   \"trans\": [";
    for (size_t k = 0; k < leaving.size(); ++k) {
      const ProfileTransition& t = trans[leaving[k]];
      if (k) out << ",\n             ";
      out << " { \"to\": " << t.dest << ", \"lpTrans\": " << number(t.lpTrans);
      if (!t.alignPath.empty()) out << ", \"path\": " << pathArray(t.alignPath);
      out << " }";
    }
    out << " ]\n  }" << (n + 1 < state.size() ? "," : "") << "\n";
  }
  out << " ]\n}" << std::endl;
}

string Profile::toJson() const {
  std::ostringstream text;
  writeJson(text);
  return text.str();
}

// ------------------------------------------------------------------------------------------------------------------
// consistency checks (what the reference asserts after every construction, src/profile.cpp:206-266, 321-361)
// ------------------------------------------------------------------------------------------------------------------
// residues per row: source coordinates + residues on the transition + residues on the destination = destination's
void ProfileState::assertSeqCoordsConsistent(const SeqCoords& srcCoords, const SeqCoords& destCoords, const AlignPath& transPath,
                                             const AlignPath& destPath) {
  SeqCoords reached(srcCoords);
  for (const AlignPath* part : {&transPath, &destPath})
    for (const auto& row : *part) reached[row.first] += alignPathResiduesInRow(row.second);
  for (const auto& want : destCoords) {
    const auto got = reached.find(want.first);
    Assert(got != reached.end(), "Missing coordinate for sequence %d", (int)want.first);
    Assert(got->second == want.second, "Sequence coord %d: source state + transition path + dest state path != dest state (%d)",
           (int)want.first, (int)want.second);
  }
}

void ProfileState::assertSeqCoordsConsistent(const SeqCoords& srcCoords, const ProfileState& dest, const AlignPath& transPath) {
  assertSeqCoordsConsistent(srcCoords, dest.seqCoords, transPath, dest.alignPath);
}

void Profile::assertSeqCoordsConsistent() const {
  for (const ProfileTransition& t : trans) ProfileState::assertSeqCoordsConsistent(state[t.src].seqCoords, state[t.dest], t.alignPath);
}

void Profile::assertAllStatesWaitOrReady() const {
  for (const ProfileState& s : state)
    Assert(s.isReady() || s.isWait(), "State %s has %d null transitions and %d absorbing transitions, so is neither Wait nor Ready",
           s.name.c_str(), (int)s.nullOut.size(), (int)s.absorbOut.size());
}

void Profile::assertTransitionsConsistent() const {
  for (ProfileStateIndex i = 0; i < state.size(); ++i) {
    const ProfileState& s = state[i];
    const auto allFrom = [&](const vguard<ProfileTransitionIndex>& list) {
      return std::all_of(list.begin(), list.end(), [&](ProfileTransitionIndex t) { return trans[t].src == i; });
    };
    Assert(std::all_of(s.in.begin(), s.in.end(), [&](ProfileTransitionIndex t) { return trans[t].dest == i; }),
           "Incoming transition destination index doesn't match state index");
    Assert(allFrom(s.nullOut), "Null transition source index doesn't match state index");
    Assert(allFrom(s.absorbOut), "Absorbing transition source index doesn't match state index");
  }
}

// one START..END path (states in order), found by marking what START reaches; also the toposort check
vguard<ProfileStateIndex> Profile::examplePathToEnd() const {
  const ProfileStateIndex n = state.size();
  vguard<ProfileStateIndex> cameFrom(n, n);          // n = not reached
  cameFrom[0] = 0;
  for (ProfileStateIndex i = 0; i < n; ++i) {
    if (cameFrom[i] == n) continue;
    for (ProfileTransitionIndex t : state[i].nullOut) {
      Assert(trans[t].dest > i, "Null transition violates toposort");
      cameFrom[trans[t].dest] = i;
    }
    for (ProfileTransitionIndex t : state[i].absorbOut) {
      Assert(trans[t].dest > i, "Absorbing transition violates toposort");
      cameFrom[trans[t].dest] = i;
    }
  }
  Assert(cameFrom[n - 1] != n, "No path from start to end");
  vguard<ProfileStateIndex> path(1, n - 1);
  while (path.back() != 0) path.push_back(cameFrom[path.back()]);
  std::reverse(path.begin(), path.end());
  return path;
}

void Profile::assertPathToEndExists() const { (void)examplePathToEnd(); }

// ------------------------------------------------------------------------------------------------------------------
// Wait / Ready split (what reference src/profile.cpp:268-319 produces).  The DP's null-move ordering needs every state
// to be either a Wait state (only null moves leave) or a Ready state (only absorbing moves leave).  A state with both is
// replaced by a Wait state that keeps its incoming and null transitions, followed immediately by a null Ready twin that
// takes over its absorbing transitions, joined by a null transition of log-weight 0.  States keep their relative order;
// the new joining transitions are appended to the transition list in state order.
// ------------------------------------------------------------------------------------------------------------------
Profile Profile::addReadyStates() const { return withReadyStates(Profile(*this)); }

// The same, consuming the profile: states and transitions (with their alignment paths and coordinate maps) are moved into
// the result, not copied - makeProfile builds a profile only to pass it through here.
Profile Profile::withReadyStates(Profile&& src) {
  const ProfileStateIndex n = src.size();
  const ProfileTransitionIndex nTrans = src.trans.size();
  vguard<char> mixed(n, 0);
  vguard<ProfileStateIndex> moved(n);                // new index of every old state
  ProfileStateIndex next = 0;
  for (ProfileStateIndex s = 0; s < n; ++s) {
    mixed[s] = !src.state[s].isReady() && !src.state[s].isWait();
    moved[s] = next;
    next += mixed[s] ? 2 : 1;
  }

  Profile out(src.components, src.alphSize, src.rootRowIndex);
  out.name = std::move(src.name);
  out.meta = std::move(src.meta);
  out.seq = std::move(src.seq);
  out.trans = std::move(src.trans);
  out.state.reserve(next);
  for (ProfileStateIndex s = 0; s < n; ++s) {
    out.state.push_back(std::move(src.state[s]));
    if (!mixed[s]) continue;
    const ProfileTransitionIndex joinIdx = out.trans.size();
    ProfileState& wait = out.state.back();
    ProfileState ready;
    ready.name = wait.name + kReadySuffix;
    wait.name += kWaitSuffix;
    ready.meta = wait.meta;
    ready.seqCoords = wait.seqCoords;
    ready.absorbOut.swap(wait.absorbOut);
    ready.in.push_back(joinIdx);
    wait.nullOut.push_back(joinIdx);
    ProfileTransition join;
    join.src = s;                                     // (old numbering; renumbered with all the others below)
    join.dest = n + (joinIdx - nTrans);               // a provisional index past the old states: the k-th twin
    join.lpTrans = 0;
    out.trans.push_back(join);
    out.state.push_back(ready);
  }
  // old index -> new index for the twins too (provisional indices n, n+1, ... in state order)
  vguard<ProfileStateIndex> twin;
  for (ProfileStateIndex s = 0; s < n; ++s)
    if (mixed[s]) twin.push_back(moved[s] + 1);
  const auto renumber = [&](ProfileStateIndex old) { return old < n ? moved[old] : twin[old - n]; };
  for (ProfileTransitionIndex t = 0; t < out.trans.size(); ++t) {
    ProfileTransition& tr = out.trans[t];
    // an absorbing transition of a split state now leaves its Ready twin (which took over the state's absorbOut list)
    bool leavesTwin = false;
    if (t < nTrans && mixed[tr.src]) {
      const vguard<ProfileTransitionIndex>& taken = out.state[moved[tr.src] + 1].absorbOut;
      leavesTwin = std::find(taken.begin(), taken.end(), t) != taken.end();
    }
    tr.src = leavesTwin ? moved[tr.src] + 1 : renumber(tr.src);
    tr.dest = renumber(tr.dest);
  }
  for (const auto& eq : src.equivAbsorbState) out.equivAbsorbState[moved[eq.first]] = moved[eq.second];
  out.assertTransitionsConsistent();
  out.assertAllStatesWaitOrReady();
  out.assertPathToEndExists();
  return out;
}

// ------------------------------------------------------------------------------------------------------------------
// PairHMM: the composite of the two branch transducers (reference src/pairhmm.cpp:5-153).
// States: IMM both children absorb; IMD x absorbs, y's residue was deleted; IDM the other way round; IMI an insertion
// on y's branch; IIW an insertion on x's branch.  The weights below are products of the per-branch probabilities of
// {no insertion, insertion, insertion extended, ...}; moves that would reorder commuting indels are absent.
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct Branch {
  double ins, del, insExt, delExt;
  double noIns() const { return 1 - ins; }
  double noDel() const { return 1 - del; }
  double noInsExt() const { return 1 - insExt; }
  double noDelExt() const { return 1 - delExt; }
};
}  // namespace

PairHMM::PairHMM(const ProbModel& l, const ProbModel& r, const vguard<Vec>& root)
    : AlphabetOwner(l), l(l), r(r), logl(l), logr(r) {
  // root prior per component, with the component's weight folded in
  logRoot.reserve(root.size());
  for (size_t cpt = 0; cpt < root.size(); ++cpt) {
    vguard<LogProb> row = log_vector(root[cpt]);
    if ((int)cpt < l.components())
      for (LogProb& lp : row) lp += logl.logCptWeight[cpt];
    logRoot.push_back(row);
  }
  const Branch L{l.ins, l.del, l.insExt, l.delExt}, R{r.ins, r.del, r.insExt, r.delExt};
  // after a match, or an insertion on y's branch that has just ended (the factor `open`): both branches may start anything
  const auto fromOpen = [&](double open, LogProb& toImm, LogProb& toImd, LogProb& toEee) {
    toImm = log(L.noIns() * open * L.noDel() * R.noDel());
    toImd = log(L.noIns() * open * L.noDel() * R.del);
    toEee = log(L.noIns() * open);
  };
  fromOpen(R.noIns(), imm_imm, imm_imd, imm_eee);
  fromOpen(R.noInsExt(), imi_imm, imi_imd, imi_eee);
  imm_idm = log(L.noIns() * R.noIns() * L.del * R.noDel());
  imm_imi = log(R.ins);
  imm_iiw = log(L.ins * R.noIns());
  imi_imi = log(R.insExt);
  imi_iiw = log(L.ins * R.noInsExt());
  // inside a deletion on y's branch (IMD): that deletion extends or ends; x's branch is free
  imd_imm = log(L.noIns() * L.noDel() * R.noDelExt());
  imd_imd = log(L.noIns() * L.noDel() * R.delExt);
  imd_idm = log(L.noIns() * L.del * R.noDelExt());
  imd_eee = log(L.noIns() * R.noDelExt());
  // inside a deletion on x's branch (IDM)
  idm_imm = log(R.noIns() * L.noDelExt() * R.noDel());
  idm_imd = log(R.noIns() * L.noDelExt() * R.del);
  idm_idm = log(R.noIns() * L.delExt * R.noDel());
  idm_eee = log(R.noIns() * L.noDelExt());
  // inside an insertion on x's branch (IIW)
  iiw_iiw = log(L.insExt);
  iiw_imm = log(L.noInsExt() * L.noDel() * R.noDel());
  iiw_idm = log(L.noInsExt() * L.del * R.noDel());
  iiw_eee = log(L.noInsExt());
  // the same as a table [src][dest], dest 5 = EEE; null = no such move
  const LogProb* const named[TotalStates][TotalStates + 1] = {
      {&imm_imm, &imm_imd, &imm_idm, &imm_imi, &imm_iiw, &imm_eee},
      {&imd_imm, &imd_imd, &imd_idm, NULL, NULL, &imd_eee},
      {&idm_imm, &idm_imd, &idm_idm, NULL, NULL, &idm_eee},
      {&imi_imm, &imi_imd, NULL, &imi_imi, &imi_iiw, &imi_eee},
      {&iiw_imm, NULL, &iiw_idm, NULL, &iiw_iiw, &iiw_eee}};
  for (int s = 0; s < TotalStates; ++s)
    for (int d = 0; d <= TotalStates; ++d) weight[s][d] = named[s][d] ? *named[s][d] : kNegInf;
}

LogProb PairHMM::lpTrans(State src, State dest) const {
  if ((int)src < 0 || src >= TotalStates || (int)dest < 0 || dest > EEE) return kNegInf;
  return weight[src][dest];
}

vguard<PairHMM::State> PairHMM::states() { return {IMM, IMD, IDM, IMI, IIW}; }

// the states a move into `dest` can come from, in the order the DP sums them (and tracebacks enumerate them)
vguard<PairHMM::State> PairHMM::sources(State dest) {
  switch (dest) {
    case IMD: return {IMM, IMD, IDM, IMI};
    case IDM: return {IMM, IMD, IDM, IIW};
    case IMI: return {IMM, IMI};
    case IIW: return {IMM, IIW, IMI};
    case IMM:
    case EEE: return {IMM, IMD, IDM, IMI, IIW};
    default: return {};
  }
}

const char* PairHMM::stateName(State s, bool xAtStart, bool yAtStart) {
  static const char* const plain[] = {"IMM", "IMD", "IDM", "IMI", "IIW", "EEE"};
  if ((int)s < 0 || s > EEE) {
    Abort("Don't know name of state %u", (unsigned)s);
    return "?";
  }
  if (s == IMM && xAtStart && yAtStart) return "SSS";
  if (s == IMI && xAtStart) return "SSI";
  if (s == IIW && yAtStart) return "SIW";
  return plain[s];
}

}  // namespace historian
