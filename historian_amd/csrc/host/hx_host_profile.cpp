// Profile and PairHMM parts of the host mirror (see hx_host.h).
#include "hx_host.h"

#include <cfloat>
#include <sstream>

namespace historian {

#define WaitStateSuffix ";"
#define ReadyStateSuffix "."

static const double NEG_INF = -std::numeric_limits<double>::infinity();

// JsonUtil::toString(double) / (map) -- reference src/jsonutil.cpp:151-176
static string jsonDouble(double d) {
  if (d < -DBL_MAX) return string("\"-inf\"");
  if (d > DBL_MAX) return string("\"inf\"");
  return std::to_string(d);
}

static string jsonTags(const map<string, string>& tags, size_t indent) {
  string s;
  if (tags.empty())
    s = "{ }";
  else {
    bool first = true;
    for (auto& tag_val : tags) {
      if (first)
        s += tags.size() == 1 ? string("{ ") : (string("\n") + string(indent, ' ') + "{");
      else
        s += ",";
      first = false;
      if (tags.size() > 1) s += "\n" + string(indent + 1, ' ');
      s += "\"" + tag_val.first + "\": \"" + tag_val.second + "\"";
    }
    s += (tags.size() == 1 ? string(" ") : (string("\n") + string(indent, ' '))) + "}";
  }
  return s;
}

static string alignPathJson(const AlignPath& a) {
  string s = "[";
  for (auto& row_path : a) {
    if (s.size() > 1) s += ",";
    s += " [ " + std::to_string(row_path.first) + ", \"";
    for (auto col : row_path.second) s += (col ? Alignment::wildcardChar : Alignment::gapChar);
    s += "\" ]";
  }
  s += " ]";
  return s;
}

ProfileTransition::ProfileTransition() : src(0), dest(0), lpTrans(NEG_INF) {}
ProfileState::ProfileState() {}
ProfileState::ProfileState(size_t components, AlphTok alphSize) : lpAbsorb(components, vguard<LogProb>(alphSize, NEG_INF)) {}

// leaf profile, reference src/profile.cpp:23-76
Profile::Profile(size_t components, const string& alphabet, const FastSeq& seq, AlignRowIndex rowIndex)
    : alphSize((AlphTok)alphabet.size()), components(components),
      state(seq.length() + 2, ProfileState(components, (AlphTok)alphabet.size())), trans(seq.length() + 1), rootRowIndex(rowIndex) {
  name = seq.name;
  state.front() = state.back() = ProfileState();
  state.front().name = "START";
  state.front().seqCoords[rowIndex] = 0;
  state.back().name = "END";
  state.back().seqCoords[rowIndex] = seq.length();
  set<char> invalidChars;
  int nInvalidToks = 0;
  for (size_t pos = 0; pos <= seq.seq.size(); ++pos) {
    ProfileTransition& t = trans[pos];
    t.src = pos;
    t.dest = pos + 1;
    t.lpTrans = 0;
    if (pos == seq.seq.size())
      state[pos].nullOut.push_back(pos);
    else
      state[pos].absorbOut.push_back(pos);
    state[pos + 1].in.push_back(pos);
    if (pos < seq.seq.size()) {
      state[pos + 1].name = string(1, seq.seq[pos]) + std::to_string(pos + 1);
      state[pos + 1].alignPath[rowIndex].push_back(true);
      state[pos + 1].seqCoords[rowIndex] = pos + 1;
      for (auto& lpa : state[pos + 1].lpAbsorb)
        if (Alignment::isWildcard(seq.seq[pos]))
          std::fill(lpa.begin(), lpa.end(), 0);
        else {
          const UnvalidatedAlphTok tok = tokenize(seq.seq[pos], alphabet);
          if (tok < 0) {
            invalidChars.insert(seq.seq[pos]);
            ++nInvalidToks;
            std::fill(lpa.begin(), lpa.end(), 0);
          } else
            lpa[tok] = 0;
        }
    }
  }
  this->seq[rowIndex] = seq.seq;
  if (nInvalidToks) Warn("%d invalid characters found in sequence %s", nInvalidToks, seq.name.c_str());
  assertTransitionsConsistent();
  assertSeqCoordsConsistent();
  assertAllStatesWaitOrReady();
  assertPathToEndExists();
}

const ProfileTransition* Profile::getTrans(ProfileStateIndex src, ProfileStateIndex dest) const {
  for (auto t : state[dest].in)
    if (trans[t].src == src) return &trans[t];
  return NULL;
}

LogProb Profile::calcSumPathAbsorbProbs(const vguard<LogProb>& logCptWeight, const vguard<vguard<LogProb> >& logInsProb, const char* tag) {
  vguard<LogProb> lpCumAbs(state.size(), NEG_INF);
  lpCumAbs[0] = 0;
  for (ProfileStateIndex pos = 1; pos < state.size(); ++pos) {
    LogProb lpAbs = 0;
    if (!state[pos].isNull()) {
      lpAbs = NEG_INF;
      for (size_t cpt = 0; cpt < components; ++cpt)
        log_accum_exp(lpAbs, logCptWeight[cpt] + logInnerProduct(logInsProb[cpt], state[pos].lpAbsorb[cpt]));
    }
    for (auto ti : state[pos].in) {
      const ProfileTransition& t = trans[ti];
      Assert(t.src < pos, "Transition #%u from %u -> %u is not toposorted", (unsigned)ti, (unsigned)t.src, (unsigned)t.dest);
      log_accum_exp(lpCumAbs[pos], lpCumAbs[t.src] + t.lpTrans + lpAbs);
    }
    if (tag != NULL) state[pos].meta[string(tag)] = std::to_string(lpCumAbs[pos]);
  }
  return lpCumAbs.back();
}

void Profile::writeJson(std::ostream& out) const {
  using std::endl;
  out << "{" << endl;
  if (name.size()) out << " \"name\": \"" << name << "\"," << endl;
  if (meta.size()) out << " \"meta\": " << jsonTags(meta, 2) << "," << endl;
  out << " \"alphSize\": " << alphSize << "," << endl;
  out << " \"state\": [" << endl;
  for (ProfileStateIndex s = 0; s < state.size(); ++s) {
    out << "  {" << endl;
    out << "   \"n\": " << s << "," << endl;
    if (state[s].name.size()) out << "   \"name\": \"" << state[s].name << "\"," << endl;
    if (state[s].meta.size()) out << "   \"meta\": " << jsonTags(state[s].meta, 4) << "," << endl;
    if (state[s].alignPath.size()) out << "   \"path\": " << alignPathJson(state[s].alignPath) << "," << endl;
    if (state[s].seqCoords.size()) {
      out << "   \"seqPos\": [";
      size_t nSeqPos = 0;
      for (const auto& s_c : state[s].seqCoords) out << (nSeqPos++ ? ", " : " ") << "[ " << s_c.first << ", " << s_c.second << " ]";
      out << " ]," << endl;
    }
    if (!state[s].isNull()) {
      out << "   \"lpAbsorb\": [";
      for (size_t cpt = 0; cpt < components; ++cpt) {
        out << (cpt > 0 ? ", " : "") << "[";
        for (AlphTok a = 0; a < alphSize; ++a) out << (a > 0 ? ", " : " ") << jsonDouble(state[s].lpAbsorb[cpt][a]);
        out << " ]";
      }
      out << "]," << endl;
    }
    out << "   \"trans\": [";
    set<ProfileTransitionIndex> s_out(state[s].nullOut.begin(), state[s].nullOut.end());
    s_out.insert(state[s].absorbOut.begin(), state[s].absorbOut.end());
    bool first_t = true;
    for (auto ti : s_out) {
      const ProfileTransition& tr = trans[ti];
      if (!first_t) out << ",\n             ";
      first_t = false;
      out << " { \"to\": " << tr.dest << ",";
      out << " \"lpTrans\": " << jsonDouble(tr.lpTrans);
      if (tr.alignPath.size()) out << ", \"path\": " << alignPathJson(tr.alignPath);
      out << " }";
    }
    out << " ]" << endl;
    out << "  }";
    if (s < state.size() - 1) out << ",";
    out << endl;
  }
  out << " ]" << endl;
  out << "}" << endl;
}

string Profile::toJson() const {
  std::ostringstream s;
  writeJson(s);
  return s.str();
}

void Profile::assertSeqCoordsConsistent() const {
  for (const auto& t : trans) ProfileState::assertSeqCoordsConsistent(state[t.src].seqCoords, state[t.dest], t.alignPath);
}

void ProfileState::assertSeqCoordsConsistent(const SeqCoords& srcCoords, const ProfileState& dest, const AlignPath& transPath) {
  assertSeqCoordsConsistent(srcCoords, dest.seqCoords, transPath, dest.alignPath);
}

void ProfileState::assertSeqCoordsConsistent(const SeqCoords& srcCoords, const SeqCoords& destCoords, const AlignPath& transPath, const AlignPath& destPath) {
  SeqCoords seqCoords = srcCoords;
  for (const auto& rp : transPath) seqCoords[rp.first] += alignPathResiduesInRow(rp.second);
  for (const auto& rp : destPath) seqCoords[rp.first] += alignPathResiduesInRow(rp.second);
  for (const auto& sc : destCoords) {
    Assert(seqCoords.count(sc.first), "Missing coordinate for sequence %d", (int)sc.first);
    Assert(seqCoords.at(sc.first) == sc.second, "Sequence coord %d: source state + transition path + dest state path != dest state (%d)",
           (int)sc.first, (int)sc.second);
  }
}

void Profile::assertAllStatesWaitOrReady() const {
  for (auto& s : state)
    Assert(s.isReady() || s.isWait(), "State %s has %d null transitions and %d absorbing transitions, so is neither Wait nor Ready",
           s.name.c_str(), (int)s.nullOut.size(), (int)s.absorbOut.size());
}

// reference src/profile.cpp:268-319
Profile Profile::addReadyStates() const {
  vguard<ProfileStateIndex> old2newStateIndex(size());
  Profile prof;
  prof.alphSize = alphSize;
  prof.components = components;
  prof.name = name;
  prof.meta = meta;
  prof.seq = seq;
  prof.trans = trans;
  prof.rootRowIndex = rootRowIndex;
  vguard<ProfileState> profState(state);
  for (ProfileStateIndex s = 0, n = 0; s < size(); ++s) {
    old2newStateIndex[s] = n++;
    if (!state[s].isReady() && !state[s].isWait()) {
      ProfileState readyState;
      ProfileTransition readyTrans;
      const ProfileStateIndex oldReadyStateIdx = profState.size();
      const ProfileStateIndex newReadyStateIdx = n++;
      const ProfileTransitionIndex readyTransIdx = prof.trans.size();
      profState[s].name += WaitStateSuffix;
      readyState.name = state[s].name + ReadyStateSuffix;
      readyState.meta = state[s].meta;
      readyState.seqCoords = state[s].seqCoords;
      std::swap(profState[s].absorbOut, readyState.absorbOut);
      for (auto t : readyState.absorbOut) prof.trans[t].src = oldReadyStateIdx;
      readyTrans.src = s;
      readyTrans.dest = oldReadyStateIdx;
      readyTrans.lpTrans = 0;
      profState[s].nullOut.push_back(readyTransIdx);
      readyState.in.push_back(readyTransIdx);
      profState.push_back(readyState);
      prof.trans.push_back(readyTrans);
      old2newStateIndex.push_back(newReadyStateIdx);
    }
  }
  prof.state = vguard<ProfileState>(profState.size());
  for (ProfileStateIndex s = 0; s < profState.size(); ++s) std::swap(profState[s], prof.state[old2newStateIndex[s]]);
  for (auto& t : prof.trans) {
    t.src = old2newStateIndex[t.src];
    t.dest = old2newStateIndex[t.dest];
  }
  for (const auto& ss : equivAbsorbState) prof.equivAbsorbState[old2newStateIndex[ss.first]] = old2newStateIndex[ss.second];
  prof.assertTransitionsConsistent();
  prof.assertAllStatesWaitOrReady();
  prof.assertPathToEndExists();
  return prof;
}

void Profile::assertTransitionsConsistent() const {
  for (ProfileStateIndex i = 0; i < state.size(); ++i) {
    const ProfileState& s = state[i];
    for (ProfileTransitionIndex t : s.in) Assert(trans[t].dest == i, "Incoming transition destination index doesn't match state index");
    for (ProfileTransitionIndex t : s.nullOut) Assert(trans[t].src == i, "Null transition source index doesn't match state index");
    for (ProfileTransitionIndex t : s.absorbOut) Assert(trans[t].src == i, "Absorbing transition source index doesn't match state index");
  }
}

void Profile::assertPathToEndExists() const { (void)examplePathToEnd(); }

vguard<ProfileStateIndex> Profile::examplePathToEnd() const {
  vguard<bool> fromStart(state.size(), false);
  vguard<ProfileStateIndex> prev(state.size(), 0);
  fromStart.front() = true;
  for (ProfileStateIndex i = 0; i < state.size(); ++i)
    if (fromStart[i]) {
      const ProfileState& s = state[i];
      for (ProfileTransitionIndex t : s.nullOut) {
        Assert(trans[t].dest > i, "Null transition violates toposort");
        fromStart[trans[t].dest] = true;
        prev[trans[t].dest] = i;
      }
      for (ProfileTransitionIndex t : s.absorbOut) {
        Assert(trans[t].dest > i, "Absorbing transition violates toposort");
        fromStart[trans[t].dest] = true;
        prev[trans[t].dest] = i;
      }
    }
  Assert(fromStart.back(), "No path from start to end");
  vguard<ProfileStateIndex> revPath;
  for (ProfileStateIndex j = state.size() - 1; j != 0; j = prev[j]) revPath.push_back(j);
  revPath.push_back(0);
  return vguard<ProfileStateIndex>(revPath.rbegin(), revPath.rend());
}

bool Profile::isEmpty() const {
  for (const auto& s : state)
    if (!s.isNull()) return false;
  return true;
}

// ---- PairHMM (reference src/pairhmm.cpp:5-153) ----------------------------------------------
PairHMM::PairHMM(const ProbModel& l, const ProbModel& r, const vguard<Vec>& root)
    : AlphabetOwner(l), l(l), r(r), logl(l), logr(r) {
  for (const auto& rv : root) logRoot.push_back(log_vector(rv));
  for (int cpt = 0; cpt < l.components(); ++cpt)
    for (auto& lr : logRoot[cpt]) lr += logl.logCptWeight[cpt];
  const double lIns = l.ins, lDel = l.del, lInsExt = l.insExt, lDelExt = l.delExt;
  const double rIns = r.ins, rDel = r.del, rInsExt = r.insExt, rDelExt = r.delExt;
  const double lNoIns = 1 - lIns, lNoDel = 1 - lDel, lNoInsExt = 1 - lInsExt, lNoDelExt = 1 - lDelExt;
  const double rNoIns = 1 - rIns, rNoDel = 1 - rDel, rNoInsExt = 1 - rInsExt, rNoDelExt = 1 - rDelExt;
  imm_imi = log(rIns);
  imm_iiw = log(lIns * rNoIns);
  imm_imm = log(lNoIns * rNoIns * lNoDel * rNoDel);
  imm_imd = log(lNoIns * rNoIns * lNoDel * rDel);
  imm_idm = log(lNoIns * rNoIns * lDel * rNoDel);
  imm_eee = log(lNoIns * rNoIns);
  imd_imm = log(lNoIns * lNoDel * rNoDelExt);
  imd_imd = log(lNoIns * lNoDel * rDelExt);
  imd_idm = log(lNoIns * lDel * rNoDelExt);
  imd_eee = log(lNoIns * rNoDelExt);
  idm_imm = log(rNoIns * lNoDelExt * rNoDel);
  idm_imd = log(rNoIns * lNoDelExt * rDel);
  idm_idm = log(rNoIns * lDelExt * rNoDel);
  idm_eee = log(rNoIns * lNoDelExt);
  imi_imi = log(rInsExt);
  imi_iiw = log(lIns * rNoInsExt);
  imi_imm = log(lNoIns * rNoInsExt * lNoDel * rNoDel);
  imi_imd = log(lNoIns * rNoInsExt * lNoDel * rDel);
  imi_eee = log(lNoIns * rNoInsExt);
  iiw_iiw = log(lInsExt);
  iiw_imm = log(lNoInsExt * lNoDel * rNoDel);
  iiw_idm = log(lNoInsExt * lDel * rNoDel);
  iiw_eee = log(lNoInsExt);
}

LogProb PairHMM::lpTrans(State src, State dest) const {
  switch (src) {
    case IMM:
      switch (dest) {
        case IMM: return imm_imm;
        case IMD: return imm_imd;
        case IDM: return imm_idm;
        case IMI: return imm_imi;
        case IIW: return imm_iiw;
        case EEE: return imm_eee;
        default: break;
      }
      break;
    case IMD:
      switch (dest) {
        case IMM: return imd_imm;
        case IMD: return imd_imd;
        case IDM: return imd_idm;
        case EEE: return imd_eee;
        default: break;
      }
      break;
    case IDM:
      switch (dest) {
        case IMM: return idm_imm;
        case IMD: return idm_imd;
        case IDM: return idm_idm;
        case EEE: return idm_eee;
        default: break;
      }
      break;
    case IMI:
      switch (dest) {
        case IMM: return imi_imm;
        case IMD: return imi_imd;
        case IMI: return imi_imi;
        case IIW: return imi_iiw;
        case EEE: return imi_eee;
        default: break;
      }
      break;
    case IIW:
      switch (dest) {
        case IMM: return iiw_imm;
        case IIW: return iiw_iiw;
        case IDM: return iiw_idm;
        case EEE: return iiw_eee;
        default: break;
      }
      break;
    default: break;
  }
  return NEG_INF;
}

vguard<PairHMM::State> PairHMM::states() { return vguard<State>{IMM, IMD, IDM, IMI, IIW}; }

vguard<PairHMM::State> PairHMM::sources(State dest) {
  switch (dest) {
    case IMM:
    case EEE: return vguard<State>{IMM, IMD, IDM, IMI, IIW};
    case IMD: return vguard<State>{IMM, IMD, IDM, IMI};
    case IDM: return vguard<State>{IMM, IMD, IDM, IIW};
    case IMI: return vguard<State>{IMM, IMI};
    case IIW: return vguard<State>{IMM, IIW, IMI};
    default: break;
  }
  return vguard<State>();
}

const char* PairHMM::stateName(State s, bool xAtStart, bool yAtStart) {
  switch (s) {
    case IMM: return xAtStart && yAtStart ? "SSS" : "IMM";
    case IMD: return "IMD";
    case IDM: return "IDM";
    case IMI: return xAtStart ? "SSI" : "IMI";
    case IIW: return yAtStart ? "SIW" : "IIW";
    case EEE: return "EEE";
    default: break;
  }
  Abort("Don't know name of state %u", (unsigned)s);
  return "?";
}

}  // namespace historian
