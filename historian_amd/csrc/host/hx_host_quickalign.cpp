// Host mirror of the guide-alignment pair DP (reference src/fastseq.cpp k-mer helpers,
// src/diagenv.cpp, src/quickalign.cpp): same names, arguments and error behaviour; the
// QuickAlignMatrix fill runs on the device through the C ABI (hx_quick_batch_*), the traceback
// reads the device-filled matrix.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <set>
extern "C" {
#include "../../../include/historian_hip.h"
}
#include "hx_host.h"

namespace historian {

static const double NEG_INF = -std::numeric_limits<double>::infinity();

// ---- src/fastseq.cpp:142-163,255-266 ----------------------------------------------------------
UnvalidatedTokSeq unvalidatedTokens(const FastSeq& seq, const string& alphabet) {
  UnvalidatedTokSeq tok;
  tok.reserve(seq.seq.size());
  for (char c : seq.seq) tok.push_back(tokenize(c, alphabet));
  return tok;
}

// a window of k tokens is a k-mer when none of them lies outside the alphabet (token -1) ...
bool kmerValid(SeqIdx k, vguard<int>::const_iterator tok) {
  return std::find_if(tok, tok + k, [](int token) { return token < 0; }) == tok + k;
}

// ... and its code is the window read as a base-|alphabet| number, first token most significant
Kmer makeKmer(SeqIdx k, vguard<int>::const_iterator tok, AlphTok alphabetSize) {
  Kmer code = 0;
  for (auto t = tok; t != tok + k; ++t) {
    Assert(*t >= 0, "Invalid token in makeKmer");
    code = code * alphabetSize + (Kmer)*t;
  }
  return code;
}

KmerIndex::KmerIndex(const FastSeq& seq, const string& alphabet, SeqIdx kmerLen) : kmerLen(kmerLen), alphabet(alphabet), seq(seq) {
  // every window of kmerLen tokens that lies inside the alphabet, filed under its code by start position
  const UnvalidatedTokSeq tok = unvalidatedTokens(seq, alphabet);
  if (tok.size() < kmerLen) return;
  for (auto window = tok.begin(); window + kmerLen <= tok.end(); ++window)
    if (kmerValid(kmerLen, window))
      kmerLocations[makeKmer(kmerLen, window, (AlphTok)alphabet.size())].push_back((SeqIdx)(window - tok.begin()));
}

void writeFastaSeqs(std::ostream& out, const vguard<FastSeq>& fastSeqs) {
  for (const auto& s : fastSeqs) {   // FastSeq::writeFasta (src/fastseq.cpp): name, optional comment, sequence on one line
    out << '>' << s.name;
    if (s.comment.size()) out << ' ' << s.comment;
    out << '\n' << s.seq << '\n';
  }
}

// ---- src/diagenv.cpp:104-199 ------------------------------------------------------------------
#define MIN_KMERS_FOR_SPARSE_ENVELOPE 2

void DiagonalEnvelope::initFull() {
  diagonals.clear();
  diagonals.reserve(xLen + yLen - 1);
  for (int d = minDiagonal(); d <= maxDiagonal(); ++d) diagonals.push_back(d);
  full = true;
}

// Sparse envelope (behaviour of reference src/diagenv.cpp:113-199).  Diagonals are ranked by how many k-mers
// of x match a k-mer of y on them; the best-supported ones seed bands of bandSize diagonals.  With a
// non-negative kmerThreshold every diagonal with at least that many matches seeds a band; with a negative
// one, whole tiers of equally supported diagonals are added, best tier first, while the stored diagonals
// (bands plus one guard diagonal each side) stay below maxSize bytes.  Diagonal 0 is always present.
// Short sequences (or a full matrix that fits maxSize) get the full envelope instead.
void DiagonalEnvelope::initSparse(const KmerIndex& yKmerIndex, unsigned int bandSize, int kmerThreshold, size_t cellSize,
                                  size_t maxSize) {
  const unsigned int k = yKmerIndex.kmerLen;
  const bool byMemory = kmerThreshold < 0;
  if (byMemory ? (size_t)xLen * yLen * cellSize < maxSize
               : (px->length() < MIN_KMERS_FOR_SPARSE_ENVELOPE * (k + kmerThreshold) ||
                  py->length() < MIN_KMERS_FOR_SPARSE_ENVELOPE * (k + kmerThreshold))) {
    initFull();
    return;
  }
  // k-mer matches per diagonal
  const UnvalidatedTokSeq xTok = unvalidatedTokens(*px, yKmerIndex.alphabet);
  const AlphTok alphabetSize = (AlphTok)yKmerIndex.alphabet.size();
  map<int, unsigned int> matchesOn;
  for (SeqIdx i = 0; i + k <= xLen; ++i) {
    if (!kmerValid(k, xTok.begin() + i)) continue;
    const auto hit = yKmerIndex.kmerLocations.find(makeKmer(k, xTok.begin() + i, alphabetSize));
    if (hit == yKmerIndex.kmerLocations.end()) continue;
    for (SeqIdx j : hit->second) ++matchesOn[get_diag(i, j)];
  }
  // tiers of equally supported diagonals, best first
  map<unsigned int, vguard<int>, std::greater<unsigned int> > tiers;
  for (const auto& dm : matchesOn) tiers[dm.second].push_back(dm.first);

  std::set<int> visited, stored;
  visited.insert(0);
  stored.insert(0);
  const int half = (int)(bandSize / 2);
  const size_t bytesPerDiagonal = std::min(xLen, yLen) * cellSize;
  for (const auto& tier : tiers) {
    if (!byMemory && tier.first < (unsigned int)kmerThreshold) break;
    std::set<int> visitedWith = visited, storedWith = stored;
    for (int seed : tier.second) {
      const int lo = std::max(minDiagonal(), seed - half), hi = std::min(maxDiagonal(), seed + half);
      for (int d = lo; d <= hi; ++d) visitedWith.insert(d);
      for (int d = lo - 1; d <= hi + 1; ++d) storedWith.insert(d);
    }
    if (byMemory && storedWith.size() * bytesPerDiagonal >= maxSize) break;
    visited.swap(visitedWith);
    stored.swap(storedWith);
  }
  diagonals.assign(visited.begin(), visited.end());
  full = false;
}

bool DiagonalEnvelope::contains(SeqIdx i, SeqIdx j) const {
  const int diag = (int)i - (int)j;
  const auto iter = std::lower_bound(diagonals.begin(), diagonals.end(), diag);
  return iter != diagonals.end() && *iter == diag;
}

// The x positions of column j's envelope cells, ascending: diagonals are kept sorted, and a diagonal d meets column j at row
// d + j, inside the matrix for d in (-j, xLen - j] - a contiguous stretch of the sorted list.
vguard<SeqIdx> DiagonalEnvelope::forward_i(SeqIdx j) const {
  const auto lo = std::upper_bound(diagonals.begin(), diagonals.end(), -(int)j);
  const auto hi = std::upper_bound(lo, diagonals.end(), (int)xLen - (int)j);
  vguard<SeqIdx> rows(hi - lo);
  std::transform(lo, hi, rows.begin(), [j](int d) { return (SeqIdx)(d + (int)j); });
  return rows;
}

vguard<SeqIdx> DiagonalEnvelope::reverse_i(SeqIdx j) const {
  vguard<SeqIdx> rows = forward_i(j);
  std::reverse(rows.begin(), rows.end());
  return rows;
}

// ---- src/quickalign.cpp -----------------------------------------------------------------------
QuickAlignMatrix::QuickHandle::~QuickHandle() {
  if (b) hx_quick_batch_destroy(b);
}

QuickAlignMatrix::QuickAlignMatrix(const DiagonalEnvelope& env, const RateModel& model, double time, Deferred)
    : penv(&env), px(env.px), py(env.py), xTok(unvalidatedTokens(*env.px, model.alphabet)),
      yTok(unvalidatedTokens(*env.py, model.alphabet)), xLen(px->length()), yLen(py->length()), xEnd(0), yEnd(0),
      start(NEG_INF), end(NEG_INF), result(NEG_INF), model(model), time(time), jobIndex(0), hostCells(NULL), hostCellsCap(0),
      stripStride(0), planeStride(0), blockStride(128), matrixDoubles(0) {
  computeScores();
}

// scores: src/quickalign.cpp:25-54
void QuickAlignMatrix::computeScores() {
  const ProbModel branch(model, time);
  const LogProbModel logBranch(branch);
  const size_t A = model.alphabetSize();
  // substitution log-odds against the insertion distribution, first mixture component
  submat.assign(A, vguard<LogProb>(A));
  for (AlphTok i = 0; i < A; ++i)
    for (AlphTok j = 0; j < A; ++j) submat[i][j] = log(branch.subMat.front()[i][j]) - logBranch.logInsProb.front()[j];
  // one gap state for insertions and deletions: probability of opening a gap, and of extending one (the harmonic mix of the
  // two extension probabilities, weighted by how often a gap is an insertion)
  const double open = branch.ins + (1 - branch.ins) * branch.del, stay = 1 - open;
  const double insShare = branch.ins / open;
  const double extend = 1 / (insShare / branch.insExt + (1 - insShare) / branch.delExt), close = 1 - extend;
  noGap = log(stay); gapOpen = log(open) + log(close); gapExtend = log(extend);
  m2m = log(stay * stay);  m2i = log(open);   m2d = log(stay * open);
  i2m = log(close * stay); i2i = log(extend); i2d = log(close * open); i2e = i2m;
  d2m = log(close);        d2d = log(extend); d2e = d2m;
  static bool dumped = false;
  if (!dumped && getenv("HX_DEBUG_SPAN")) {   // test hook: the inputs of the DP, exactly (tests/test_host_mirror.py)
    dumped = true;
    fprintf(stderr, "scores %a %a %a %a %a %a %a %a %a %a %a\n", m2m, m2i, m2d, i2i, i2m, i2d, d2d, d2m, gapOpen, gapExtend, noGap);
    for (AlphTok i = 0; i < A; ++i)
      for (AlphTok j = 0; j < A; ++j) fprintf(stderr, "submat %u %u %a\n", i, j, submat[i][j]);
  }
}

void QuickAlignMatrix::fillJob(hx_quick_job& job, vguard<double>& flatSub) const {
  const size_t A = model.alphabetSize();
  flatSub.clear();
  for (size_t i = 0; i < A; ++i)
    for (size_t j = 0; j < A; ++j) flatSub.push_back(submat[i][j]);
  job.x_tok = xTok.data();
  job.y_tok = yTok.data();
  job.x_len = (int32_t)xLen;
  job.y_len = (int32_t)yLen;
  job.alph_size = (int32_t)A;
  job.submat = flatSub.data();
  job.diagonals = penv->full ? NULL : penv->diagonals.data();
  job.n_diagonals = penv->full ? 0 : (int32_t)penv->diagonals.size();
  const double sc[11] = {m2m, m2i, m2d, i2i, i2m, i2d, d2d, d2m, gapOpen, gapExtend, noGap};
  for (int k = 0; k < 11; ++k) job.scores[k] = sc[k];
}

void QuickAlignMatrix::attach(const std::shared_ptr<QuickHandle>& h, int job, double score, int xe, int ye) {
  handle = h;
  jobIndex = job;
  start = 0;
  end = result = score;
  xEnd = (SeqIdx)xe;
  yEnd = (SeqIdx)ye;
  hx_layout lay;
  detail::check(hx_quick_batch_layout(h->b, job, &lay), "hx_quick_batch_layout");
  stripStride = lay.strip_stride;
  planeStride = lay.plane_stride;
  blockStride = lay.block_stride;
  matrixDoubles = lay.matrix_doubles;
}

QuickAlignMatrix::QuickAlignMatrix(const DiagonalEnvelope& env, const RateModel& model, double time)
    : QuickAlignMatrix(env, model, time, Deferred()) {
  Require(xLen > 0 && yLen > 0, "Can't align an empty sequence (%s vs %s)", px->name.c_str(), py->name.c_str());
  detail::ensureDevice();
  hx_quick_job job;
  vguard<double> flatSub;
  fillJob(job, flatSub);
  hx_quick_batch* b = NULL;
  detail::check(hx_quick_batch_create(&job, 1, &b), "hx_quick_batch_create");
  std::shared_ptr<QuickHandle> h(new QuickHandle(b));
  detail::check(hx_quick_batch_run(b, NULL), "hx_quick_batch_run");
  double score = NEG_INF;
  int32_t xe = 0, ye = 0;
  detail::check(hx_quick_batch_results(b, &score, &xe, &ye), "hx_quick_batch_results");
  attach(h, 0, score, xe, ye);
}

vguard<QuickAlignMatrix*> QuickAlignMatrix::fillBatch(const vguard<const DiagonalEnvelope*>& envs, const RateModel& model,
                                                      double time) {
  vguard<QuickAlignMatrix*> out;
  if (envs.empty()) return out;
  detail::ensureDevice();
  const size_t n = envs.size();
  vguard<hx_quick_job> jobs(n);
  vguard<vguard<double> > flat(n);
  for (size_t k = 0; k < n; ++k) {
    QuickAlignMatrix* m = new QuickAlignMatrix(*envs[k], model, time, Deferred());
    Require(m->xLen > 0 && m->yLen > 0, "Can't align an empty sequence (%s vs %s)", m->px->name.c_str(), m->py->name.c_str());
    out.push_back(m);
    m->fillJob(jobs[k], flat[k]);
  }
  hx_quick_batch* b = NULL;
  detail::check(hx_quick_batch_create(jobs.data(), (int32_t)n, &b), "hx_quick_batch_create");
  std::shared_ptr<QuickHandle> h(new QuickHandle(b));
  detail::check(hx_quick_batch_run(b, NULL), "hx_quick_batch_run");
  vguard<double> score(n, NEG_INF);
  vguard<int32_t> xe(n, 0), ye(n, 0);
  detail::check(hx_quick_batch_results(b, score.data(), xe.data(), ye.data()), "hx_quick_batch_results");
  for (size_t k = 0; k < n; ++k) out[k]->attach(h, (int)k, score[k], xe[k], ye[k]);
  return out;
}

QuickAlignMatrix::~QuickAlignMatrix() {
  if (hostCells) detail::pinnedGive(hostCells, hostCellsCap);
}

// reference getCell const (src/quickalign.h:30-33): `dummy` (-inf) outside the stored cells
LogProb QuickAlignMatrix::getCell(SeqIdx i, SeqIdx j, unsigned int offset) const {
  if (i < 1 || j < 1 || i > xLen || j > yLen) return NEG_INF;
  if (!hostCells) {
    hostCells = detail::pinnedTake((size_t)matrixDoubles, hostCellsCap);
    detail::check(hx_quick_batch_read_matrix(handle->b, jobIndex, hostCells), "hx_quick_batch_read_matrix");
  }
  const long long r = (long long)i - 1, c = (long long)j - 1;
  const long long l = r & 63, t = c + l;
  const long long slot = (r >> 6) * stripStride + (t >> 1) * blockStride + (l << 1) + (t & 1);
  return hostCells[(size_t)offset * planeStride + slot];
}

LogProb QuickAlignMatrix::cellScore(SeqIdx i, SeqIdx j, State state) const {
  if (state == Match) return mat(i, j);
  if (state == Insert) return ins(i, j);
  return state == Delete ? del(i, j) : std::numeric_limits<double>::quiet_NaN();     // (Start has no cell)
}

const char* QuickAlignMatrix::stateToString(State state) {
  switch (state) {
    case Start: return "Start";
    case Match: return "Match";
    case Insert: return "Insert";
    case Delete: return "Delete";
    default: break;
  }
  return "Unknown";
}

// (a later candidate replaces the best so far only when strictly better: the first source in consideration order wins ties)
void QuickAlignMatrix::updateMax(double& best, State& bestState, double candidate, State candidateState) {
  const bool better = candidate > best;
  best = better ? candidate : best;
  bestState = better ? candidateState : bestState;
}

// Viterbi traceback (behaviour of reference src/quickalign.cpp:147-207).  Walks back from (xEnd, yEnd, Match),
// at every step re-deriving the cell's score from its possible sources and moving to the first source (in the
// order Match, Insert, Delete, Start) that attains the maximum; the local alignment is then padded with the
// unaligned ends: y's prefix, x's prefix, the aligned core, x's suffix, y's suffix.
AlignPath QuickAlignMatrix::alignPath() const {
  Require(resultIsFinite(), "Can't do Viterbi traceback if final score is -infinity");
  SeqIdx i = xEnd, j = yEnd;
  Assert(i > 0 && j > 0, "Traceback error at (%u,%u,End)", i, j);
  vguard<bool> xCore, yCore;   // the aligned core, last column first
  State state = Match;
  while (state != Start) {
    LogProb bestScore = NEG_INF;
    State from = state;
    const auto consider = [&](LogProb score, State source) { updateMax(bestScore, from, score, source); };
    if (state == Match) {
      const LogProb emit = matchEmitScore(i, j);
      --i;
      --j;
      xCore.push_back(true);
      yCore.push_back(true);
      consider(mat(i, j) + m2m + emit, Match);
      consider(ins(i, j) + i2m + emit, Insert);
      consider(del(i, j) + d2m + emit, Delete);
      consider(start + startGapScore(i + 1, j + 1) + emit, Start);
      Assert(bestScore == mat(i + 1, j + 1), "Traceback error at (%u,%u,Match)", i + 1, j + 1);
    } else if (state == Insert) {
      --j;
      xCore.push_back(false);
      yCore.push_back(true);
      consider(mat(i, j) + m2i, Match);
      consider(ins(i, j) + i2i, Insert);
      Assert(bestScore == ins(i, j + 1), "Traceback error at (%u,%u,Insert)", i, j + 1);
    } else if (state == Delete) {
      --i;
      xCore.push_back(true);
      yCore.push_back(false);
      consider(mat(i, j) + m2d, Match);
      consider(ins(i, j) + i2d, Insert);
      consider(del(i, j) + d2d, Delete);
      Assert(bestScore == del(i + 1, j), "Traceback error at (%u,%u,Delete)", i + 1, j);
    } else
      Abort("Traceback error");
    state = from;
  }
  AlignPath path;
  AlignRowPath& xRow = path[0];
  AlignRowPath& yRow = path[1];
  const auto pad = [&](size_t columns, bool inX) {
    xRow.insert(xRow.end(), columns, inX);
    yRow.insert(yRow.end(), columns, !inX);
  };
  pad(j, false);
  pad(i, true);
  xRow.insert(xRow.end(), xCore.rbegin(), xCore.rend());
  yRow.insert(yRow.end(), yCore.rbegin(), yCore.rend());
  pad(xLen - xEnd, true);
  pad(yLen - yEnd, false);
  Assert(alignPathResiduesInRow(xRow) == xLen, "Traceback error: x row has %u steps, expected %u", alignPathResiduesInRow(xRow), xLen);
  Assert(alignPathResiduesInRow(yRow) == yLen, "Traceback error: y row has %u steps, expected %u", alignPathResiduesInRow(yRow), yLen);
  return path;
}

AlignPath QuickAlignMatrix::alignPath(AlignRowIndex row1, AlignRowIndex row2) const {
  const AlignPath two = alignPath();          // rows 0 (x) and 1 (y), renumbered
  return AlignPath{{row1, two.at(0)}, {row2, two.at(1)}};
}

vguard<FastSeq> QuickAlignMatrix::gappedSeq() const {
  vguard<FastSeq> pair;
  pair.push_back(*px);
  pair.push_back(*py);
  return Alignment(pair, alignPath()).gapped();
}

}  // namespace historian
