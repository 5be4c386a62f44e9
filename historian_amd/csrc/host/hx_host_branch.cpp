// Host mirror of the per-branch pair DPs: Refiner::BranchMatrix (reference src/refiner.cpp:10-104)
// and Sampler::BranchMatrix (src/sampler.cpp:1005-1084) over TreeAlignFuncs::SparseDPMatrix<3> (src/sampler.h:66-215).
// Everything that is per position - the child profile through the branch's substitution matrix, its insertion scores, the
// eleven transition scores - is prepared here in the reference's arithmetic; the lattice is filled on the device
// (hx_branch.hip) and read back dense; the traceback walks the copy.
#include <cmath>
#include <limits>
#include "hx_host.h"
#include "../../../include/historian_hip.h"

namespace historian {

static const double kNegInf = -std::numeric_limits<double>::infinity();

double TreeAlignFuncs::transProb(const ProbModel& p, State src, State dest) {
  // rows: leaving Match, leaving Insert, leaving Delete; columns: into Match, Insert, Delete, End
  const double table[3][4] = {
      {(1 - p.ins) * (1 - p.del), p.ins, (1 - p.ins) * p.del, 1 - p.ins},
      {(1 - p.insExt) * (1 - p.del), p.insExt, (1 - p.insExt) * p.del, 1 - p.insExt},
      {1 - p.delExt, 0., p.delExt, 1 - p.delExt}};
  return table[src][dest];
}

TreeAlignFuncs::PosWeightMatrix TreeAlignFuncs::preMultiply(const PosWeightMatrix& child, const vguard<Mat>& logSubProb) {
  PosWeightMatrix through(child.size());
  for (size_t pos = 0; pos < child.size(); ++pos) {
    through[pos].resize(logSubProb.size());
    for (size_t cpt = 0; cpt < logSubProb.size(); ++cpt) {
      const Mat& sub = logSubProb[cpt];
      Vec& out = through[pos][cpt];
      out.assign(sub.size(), kNegInf);
      for (size_t a = 0; a < sub.size(); ++a)
        for (size_t b = 0; b < child[pos][cpt].size(); ++b) log_accum_exp(out[a], sub[a][b] + child[pos][cpt][b]);
    }
  }
  return through;
}

vguard<LogProb> TreeAlignFuncs::calcInsProbs(const PosWeightMatrix& child, const vguard<vguard<LogProb>>& logInsProb,
                                             const vguard<LogProb>& logCptWeight) {
  vguard<LogProb> scores(child.size(), kNegInf);
  for (size_t pos = 0; pos < child.size(); ++pos)
    for (size_t cpt = 0; cpt < logInsProb.size(); ++cpt)
      for (size_t a = 0; a < child[pos][cpt].size(); ++a)
        log_accum_exp(scores[pos], logCptWeight[cpt] + logInsProb[cpt][a] + child[pos][cpt][a]);
  return scores;
}

TreeAlignFuncs::PosWeightMatrix TreeAlignFuncs::leafPWM(const FastSeq& seq, const string& alphabet, int components) {
  PosWeightMatrix pwm(seq.seq.size(), vguard<vguard<LogProb>>(components, vguard<LogProb>(alphabet.size(), kNegInf)));
  for (size_t pos = 0; pos < seq.seq.size(); ++pos) {
    const int tok = tokenize(seq.seq[pos], alphabet);
    for (int cpt = 0; cpt < components; ++cpt)
      for (size_t a = 0; a < alphabet.size(); ++a)
        if (tok < 0 || (size_t)tok == a) pwm[pos][cpt][a] = 0;        // (a wildcard is every residue)
  }
  return pwm;
}

static vguard<Mat> logOf(const vguard<Mat>& subMat) {
  vguard<Mat> out = subMat;
  for (Mat& m : out)
    for (Vec& row : m)
      for (double& v : row) v = log(v);
  return out;
}

TreeAlignFuncs::BranchMatrixBase::BranchMatrixBase(const RateModel& rates, const PosWeightMatrix& parent, const PosWeightMatrix& child,
                                                   double branchLength, const GuideAlignmentEnvelope& envelope,
                                                   const vguard<SeqIdx>& xEnvelopePos, const vguard<SeqIdx>& yEnvelopePos,
                                                   AlignRowIndex parentRow, AlignRowIndex childRow, bool viterbi)
    : model(rates), probModel(rates, std::max(1e-9 /* Tree::minBranchLength, src/tree.h:24 */, branchLength)), logProbModel(probModel),
      xRow(parentRow), yRow(childRow), xSeq(parent), ySub(preMultiply(child, logOf(probModel.subMat))),
      yEmit(calcInsProbs(child, logProbModel.logInsProb, logProbModel.logCptWeight)),
      xSize((SeqIdx)xEnvelopePos.size()), ySize((SeqIdx)yEnvelopePos.size()), lpEnd(kNegInf), env(envelope), xEnvPos(xEnvelopePos),
      yEnvPos(yEnvelopePos) {
  Assert(xSize == parent.size() + 1 && ySize == child.size() + 1, "Envelope positions do not match the profiles");
  mm = lpTrans(Match, Match); mi = lpTrans(Match, Insert); md = lpTrans(Match, Delete); me = lpTrans(Match, End);
  im = lpTrans(Insert, Match); ii = lpTrans(Insert, Insert); id = lpTrans(Insert, Delete); ie = lpTrans(Insert, End);
  dm = lpTrans(Delete, Match); dd = lpTrans(Delete, Delete); de = lpTrans(Delete, End);

  // flatten for the C ABI
  const int C = probModel.components(), A = (int)rates.alphabetSize();
  vguard<double> xFlat, yFlat;
  for (const auto& col : parent) for (const auto& cpt : col) xFlat.insert(xFlat.end(), cpt.begin(), cpt.end());
  for (const auto& col : ySub) for (const auto& cpt : col) yFlat.insert(yFlat.end(), cpt.begin(), cpt.end());
  vguard<int32_t> xe(xSize, 0), ye(ySize, 0);
  if (env.initialized()) {
    for (SeqIdx i = 0; i < xSize; ++i) xe[i] = env.cumulativeMatches[env.row1PosToCol[xEnvPos[i]]];
    for (SeqIdx j = 0; j < ySize; ++j) ye[j] = env.cumulativeMatches[env.row2PosToCol[yEnvPos[j]]];
  }
  hx_branch_job job;
  job.x_len = (int32_t)parent.size(); job.y_len = (int32_t)child.size();
  job.components = C; job.alphabet = A;
  job.x_pwm = xFlat.data(); job.y_sub = yFlat.data(); job.y_emit = yEmit.data();
  const LogProb t[3][4] = {{mm, mi, md, me}, {im, ii, id, ie}, {dm, lpTrans(Delete, Insert), dd, de}};
  for (int s = 0; s < 3; ++s)
    for (int d = 0; d < 4; ++d) job.trans[s][d] = t[s][d];
  job.x_env = env.initialized() ? xe.data() : nullptr;
  job.y_env = env.initialized() ? ye.data() : nullptr;
  job.max_distance = env.maxDistance;
  detail::ensureDevice();
  hx_branch_batch* b = nullptr;
  detail::check(hx_branch_batch_create(&job, 1, &b), "hx_branch_batch_create");
  detail::check(hx_branch_batch_run(b, viterbi ? 1 : 0, nullptr), "hx_branch_batch_run");
  detail::check(hx_branch_batch_results(b, &lpEnd), "hx_branch_batch_results");
  cells.resize((size_t)3 * xSize * ySize);
  detail::check(hx_branch_batch_read_matrix(b, 0, cells.data()), "hx_branch_batch_read_matrix");
  hx_branch_batch_destroy(b);
}

LogProb TreeAlignFuncs::BranchMatrixBase::cell(SeqIdx xpos, SeqIdx ypos, unsigned int state) const {
  if (state == End) return (xpos == xSize - 1 && ypos == ySize - 1) ? lpEnd : kNegInf;
  Assert(xpos < xSize && ypos < ySize && state < 3, "cell out of range");
  return cells[((size_t)xpos * ySize + ypos) * 3 + state];
}

bool TreeAlignFuncs::BranchMatrixBase::inEnvelope(SeqIdx xpos, SeqIdx ypos) const {
  return xpos == 0 || ypos == 0 || xpos == xSize - 1 || ypos == ySize - 1 || env.inRange(xEnvPos[xpos], yEnvPos[ypos]);
}

LogProb TreeAlignFuncs::BranchMatrixBase::logMatch(SeqIdx xpos, SeqIdx ypos) const {
  LogProb total = kNegInf;      // over the components, of the sum over the residues (src/logsumexp.h:146-151)
  for (size_t cpt = 0; cpt < xSeq[xpos - 1].size(); ++cpt) log_accum_exp(total, logInnerProduct(xSeq[xpos - 1][cpt], ySub[ypos - 1][cpt]));
  return total;
}

LogProb TreeAlignFuncs::BranchMatrixBase::lpTrans(State src, State dest) const { return log(transProb(probModel, src, dest)); }

LogProb TreeAlignFuncs::BranchMatrixBase::lpEmit(const CellCoords& at) const {
  if (at.state == Match) return (at.xpos > 0 && at.ypos > 0) ? logMatch(at.xpos, at.ypos) : kNegInf;
  if (at.state == Insert) return at.ypos > 0 ? yEmit[at.ypos - 1] : kNegInf;
  return 0;
}

void TreeAlignFuncs::BranchMatrixBase::getColumn(const CellCoords& at, bool& xUngapped, bool& yUngapped) {
  xUngapped = at.state == Delete || (at.state == Match && at.xpos > 0 && at.ypos > 0);
  yUngapped = at.state == Insert || (at.state == Match && at.xpos > 0 && at.ypos > 0);
}

// The best alignment: from the End state back to the start, at every cell the source state whose score plus transition plus
// this cell's emission is largest (the first such state in the order Match, Insert, Delete), src/refiner.cpp:62-104.
AlignPath Refiner::BranchMatrix::best() const {
  CellCoords at{(SeqIdx)(xSize - 1), (SeqIdx)(ySize - 1), End};
  vguard<bool> xBack, yBack;
  while (at.xpos > 0 || at.ypos > 0) {
    bool xHere, yHere;
    getColumn(at, xHere, yHere);
    if (xHere || yHere) { xBack.push_back(xHere); yBack.push_back(yHere); }
    const SeqIdx px = at.xpos - (xHere ? 1 : 0), py = at.ypos - (yHere ? 1 : 0);
    const LogProb emit = lpEmit(at);
    LogProb top = kNegInf;
    int from = -1;
    for (int s = 0; s < 3; ++s) {
      const LogProb via = cell(px, py, s) + lpTrans((State)s, (State)at.state) + emit;
      if (via > top) { top = via; from = s; }
    }
    Assert(from >= 0, "Could not find traceback state from cell (%u,%u,%u)", at.xpos, at.ypos, at.state);
    at = CellCoords{px, py, (unsigned)from};
  }
  AlignPath path;
  path[xRow] = AlignRowPath(xBack.rbegin(), xBack.rend());
  path[yRow] = AlignRowPath(yBack.rbegin(), yBack.rend());
  return path;
}

}  // namespace historian
