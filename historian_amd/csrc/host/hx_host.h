// Host-side C++ mirror of the reference's interface for the pair-HMM hot path, on top of
// the C ABI (include/historian_hip.h).  Class, member and function names, argument
// meaning and error behaviour follow the reference so that its callers
// (Reconstructor::reconstruct, reference src/recon.cpp:938-1033) and its test mains
// (t/testforward.cpp, t/testbackward.cpp, t/testnullforward.cpp, t/testseqprofile.cpp,
// t/testlogsumexp.cpp) compile against it with only the GSL vector/matrix types replaced
// by std::vector.  The Forward/Backward fills, leftMultiply and the insx/rootsubx vectors
// are computed on the GPU; traceback, sampling and profile construction are host code over
// the device-filled matrix, as O(path length) routines are in the reference.
//
//   reference file                         here
//   src/util.h, util.cpp                   Abort / Warn / Fail / Assert / Require / Test
//   src/logsumexp.h, logsumexp.cpp         log_sum_exp*, LogSumExpLookupTable, logInnerProduct
//   src/alignpath.h, alignpath.cpp         AlignPath algebra, Alignment, GuideAlignmentEnvelope
//   src/fastseq.h (subset)                 FastSeq, tokenize, readFastSeqs
//   src/model.h (subset)                   AlphabetOwner, RateModel, ProbModel, LogProbModel
//   src/profile.h                          ProfileTransition, ProfileState, Profile
//   src/pairhmm.h                          PairHMM
//   src/forward.h                          DPMatrix, ForwardMatrix, BackwardMatrix
// Out of scope: event/eigen counts -- the CountSubstEvents /
// CountIndelEvents strategy bits are accepted and ignored, SumProduct* must be NULL.
#pragma once
#include <cmath>
#include <limits>
#include <list>
#include <map>
#include <memory>
#include <queue>
#include <random>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>
#include <iostream>

struct hx_batch;
struct hx_quick_batch;
struct hx_quick_job;

namespace historian {

using std::list;
using std::map;
using std::set;
using std::string;
using std::vector;

template <class T> using vguard = std::vector<T>;

// ---- src/util.h:30-37, src/util.cpp:37-54 -------------------------------------------------
void Abort(const char* format, ...);   // message + terminate (reference: `throw;` with no active exception)
void Warn(const char* format, ...);
void Fail(const char* format, ...);    // message + exit(EXIT_FAILURE)

// wall-clock accounting of the device-backed constructors (seconds, whole process); printed by hxrecon
// when HX_TIMING is set.  Not part of the reference's interface.
struct FillTiming {
  double deviceInit = 0, flattenAndUpload = 0, forwardWait = 0, forwardKernel = 0, backwardWait = 0, readMatrix = 0;
  double construct = 0, readPrepared = 0;      // DPMatrix constructor (host-side vectors, envelope coordinates); hx_batch_read_prepared + lpAbsorb fill
  double deviceTrace = 0, cellGather = 0;
  double hostTraces = 0, hostMakeProfile = 0;      // host tracebacks (sampled, or best without the device kernel), makeProfile
  double tieRefill = 0; long tieRefills = 0;       // best traces taken again from an exact-policy fill because the walk met a near tie
  double pinnedAlloc = 0; long pinnedAllocs = 0;   // page-locked buffers allocated for matrix copies (inside readMatrix)
  double cellSets = 0, retain = 0;                 // sorting / merging the sampled cells; keeping their values (retainCells)
  long fills = 0, matrixReads = 0, deviceTraces = 0, cellGathers = 0;
  long long cells = 0;
};
extern thread_local FillTiming fillTiming;     // per host thread; worker threads of a device farm are merged into the caller's
double wallSeconds();
#define Test(assertion, ...) ((assertion) ? true : (::historian::Warn(__VA_ARGS__), false))
#define Assert(assertion, ...) do { if (!(assertion)) ::historian::Abort("Assertion Failed: " __VA_ARGS__); } while (0)
#define Require(assertion, ...) do { if (!(assertion)) ::historian::Fail(__VA_ARGS__); } while (0)

// ---- src/logsumexp.h ----------------------------------------------------------------------
#define LOG_SUM_EXP_LOOKUP_MAX 10
#define LOG_SUM_EXP_LOOKUP_PRECISION .0001
#define LOG_SUM_EXP_LOOKUP_ENTRIES (((int)(LOG_SUM_EXP_LOOKUP_MAX / LOG_SUM_EXP_LOOKUP_PRECISION)) + 1)

using LogProb = double;

double log_sum_exp_unary_slow(double gap);

struct LogSumExpLookupTable final {
  double* lookup;   // LOG_SUM_EXP_LOOKUP_ENTRIES + 1 entries (the guard entry the reference reads past its end)
  ~LogSumExpLookupTable();
  LogSumExpLookupTable();                           // fills the table with the host's libm (reference src/logsumexp.cpp:8-16)
};
extern LogSumExpLookupTable logSumExpLookupTable;

// The table operator (reference src/logsumexp.h:42-100).  T(gap) = log(1 + exp(-gap)) read from the table with linear
// interpolation for 0 <= gap < 10 and taken as 0 beyond (also for an infinite or NaN gap); the sum of two
// log-probabilities is the larger one plus T of their distance.  The arithmetic - quotient gap / precision truncated to
// the bin, the bin's offset divided by the precision again, one multiply, one add - is what every fixture carries, so it
// is kept operation for operation; the device's hx_lse.h is the same sequence.
inline double log_sum_exp_unary(double gap) {
  const bool tabulated = gap < LOG_SUM_EXP_LOOKUP_MAX && std::isfinite(gap);
  if (!tabulated) return 0;
  if (gap < 0) {
    std::cerr << "Called log_sum_exp_unary(x) for negative x = " << gap << std::endl;
    return -gap;
  }
  const int bin = (int)(gap / LOG_SUM_EXP_LOOKUP_PRECISION);
  const double* entry = logSumExpLookupTable.lookup + bin;
  const double within = (gap - (bin * LOG_SUM_EXP_LOOKUP_PRECISION)) / LOG_SUM_EXP_LOOKUP_PRECISION;
  return entry[0] + (entry[1] - entry[0]) * within;
}

inline double log_sum_exp(double p, double q) {
  // (equal operands first: -inf and -inf must not reach the subtraction)
  if (p == q) return p + log_sum_exp_unary(0);
  return p < q ? q + log_sum_exp_unary(q - p) : p + log_sum_exp_unary(p - q);
}
// more operands: folded from the left, as the reference's overloads are
inline double log_sum_exp(double a, double b, double c) { return log_sum_exp(log_sum_exp(a, b), c); }
inline double log_sum_exp(double a, double b, double c, double d) { return log_sum_exp(log_sum_exp(a, b, c), d); }
inline double log_sum_exp(double a, double b, double c, double d, double e) { return log_sum_exp(log_sum_exp(a, b, c, d), e); }
inline void log_accum_exp(double& a, double b) { a = log_sum_exp(a, b); }
double log_sum_exp_slow(double p, double q);
double log_sum_exp_slow(double p, double q, double r);
double log_sum_exp_slow(double p, double q, double r, double s);
void log_accum_exp_slow(double& total, double term);

// log of sum_k exp(v1[k] + v2[k]), accumulated from -inf in index order (src/logsumexp.h:132-151); the nested form
// sums the inner products of the rows the same way
inline LogProb logInnerProduct(const vguard<LogProb>& left, const vguard<LogProb>& right) {
  LogProb total = -std::numeric_limits<double>::infinity();
  auto q = right.begin();
  for (auto p = left.begin(); p != left.end(); ++p, ++q) log_accum_exp(total, *p + *q);
  return total;
}
inline LogProb logInnerProduct(const vguard<vguard<LogProb>>& left, const vguard<vguard<LogProb>>& right) {
  LogProb total = -std::numeric_limits<double>::infinity();
  auto q = right.begin();
  for (auto p = left.begin(); p != left.end(); ++p, ++q) log_accum_exp(total, logInnerProduct(*p, *q));
  return total;
}
vguard<LogProb> log_vector(const vguard<double>& probabilities);

// ---- src/fastseq.h (subset) -----------------------------------------------------------------
using SeqIdx = unsigned int;              // position in a sequence
using AlphTok = unsigned int;             // index of a symbol in the alphabet
using UnvalidatedAlphTok = int;           // ... or InvalidAlphabetToken
constexpr UnvalidatedAlphTok InvalidAlphabetToken = -1;

UnvalidatedAlphTok tokenize(char symbol, const string& alphabet);

struct FastSeq {
  string name, comment;
  string seq, qual;
  SeqIdx length() const { return static_cast<SeqIdx>(seq.size()); }
};
vguard<FastSeq> readFastSeqs(const char* fastaFile);

// ---- src/alignpath.h ------------------------------------------------------------------------
using AlignRowIndex = size_t;
using AlignColIndex = size_t;
using AlignRowPath = vguard<bool>;                       // per column: does the row have a residue there?
using AlignPath = map<AlignRowIndex, AlignRowPath>;

AlignColIndex alignPathColumns(const AlignPath& path);
SeqIdx alignPathResiduesInRow(const AlignRowPath& row);
AlignPath alignPathUnion(const AlignPath& left, const AlignPath& right);
AlignPath alignPathConcat(const AlignPath& first, const AlignPath& second);
AlignPath alignPathConcat(const AlignPath& first, const AlignPath& second, const AlignPath& third);
void ensureAlignPathHasRow(AlignPath& path, AlignRowIndex row);
string alignPathString(const AlignPath& path);

AlignPath alignPathMerge(const vguard<AlignPath>& paths);   // one alignment out of several that share rows

struct Alignment {
  AlignPath path;
  vguard<FastSeq> ungapped;
  static const char gapChar, wildcardChar;
  Alignment() = default;
  explicit Alignment(const vguard<FastSeq>& gappedRows);
  Alignment(const vguard<FastSeq>& ungappedRows, const AlignPath& rowPaths);
  vguard<FastSeq> gapped() const;
  static bool isGap(char symbol) { return symbol == '-' || symbol == '.'; }
  static bool isWildcard(char symbol) { return symbol == wildcardChar; }
};

struct GuideAlignmentEnvelope {
  AlignRowIndex row1 = 0, row2 = 0;                          // the two rows of the guide the envelope is about
  int maxDistance = -1;                                      // < 0: no envelope
  vguard<AlignColIndex> row1PosToCol, row2PosToCol;          // sequence position -> guide column
  vguard<int> cumulativeMatches;                             // per guide column: columns so far in which both rows have a residue
  GuideAlignmentEnvelope() = default;
  GuideAlignmentEnvelope(const AlignPath& guide, AlignRowIndex firstRow, AlignRowIndex secondRow, int band);
  bool initialized() const { return maxDistance >= 0; }
  bool inRange(SeqIdx pos1, SeqIdx pos2) const {
    if (maxDistance < 0) return true;
    const int apart = cumulativeMatches[row1PosToCol[pos1]] - cumulativeMatches[row2PosToCol[pos2]];
    return (apart < 0 ? -apart : apart) <= maxDistance;
  }
};

// ---- src/model.h (subset) -------------------------------------------------------------------
using Vec = vguard<double>;             // stands in for gsl_vector*
using Mat = vguard<vguard<double>>;    // stands in for gsl_matrix*

struct AlphabetOwner {
  char wildcard = '*';
  string alphabet;
  size_t alphabetSize() const { return alphabet.size(); }
};

struct RateModel : AlphabetOwner {
  double insRate = 0, delRate = 0;           // indel rates and extension probabilities
  double insExtProb = 0, delExtProb = 0;
  vguard<double> cptWeight;                  // mixture components: weight, root / insertion distribution, rate matrix
  vguard<Vec> insProb;
  vguard<Mat> subRate;
  int components() const { return static_cast<int>(subRate.size()); }
  void readFile(const char* jsonFile);       // JSON, reference src/model.cpp:172-232
  void read(const string& jsonText);
  static Vec getEqmProbVector(const Mat& rates);       // src/model.cpp:282-320
  vguard<Mat> getSubProbMatrix(double time) const;     // src/model.cpp:322-334 (gsl_linalg_exponential_ss restated)
};

struct ProbModel : AlphabetOwner {
  double t;                                  // branch length, and what the rates come to over it
  double ins, del, insExt, delExt;
  vguard<Mat> subMat;
  vguard<Vec> insVec;
  vguard<double> cptWeight;
  ProbModel(const RateModel& model, double branchLength);
  int components() const { return static_cast<int>(subMat.size()); }
};

struct LogProbModel {
  vguard<vguard<LogProb>> logInsProb;
  vguard<LogProb> logCptWeight;
  explicit LogProbModel(const ProbModel& probs);
  int components() const { return static_cast<int>(logCptWeight.size()); }
};

// ---- src/profile.h --------------------------------------------------------------------------
using ProfileStateIndex = size_t;
using ProfileTransitionIndex = size_t;

struct ProfileTransition {
  ProfileStateIndex src, dest;
  AlignPath alignPath;               // the alignment columns the transition steps over
  LogProb lpTrans;
  ProfileTransition();
};

struct ProfileState {
  using SeqCoords = map<AlignRowIndex, SeqIdx>;     // per alignment row: residues of that row absorbed so far
  vguard<vguard<LogProb>> lpAbsorb;                // [component][symbol]; empty: a null state
  vguard<ProfileTransitionIndex> in;                // transitions into the state, and out of it by kind of destination
  vguard<ProfileTransitionIndex> nullOut, absorbOut;
  SeqCoords seqCoords;
  AlignPath alignPath;
  map<string, string> meta;
  string name;
  ProfileState();
  ProfileState(size_t nComponents, AlphTok nSymbols);
  bool isNull() const { return lpAbsorb.empty(); }
  bool isEmit() const { return !isNull(); }
  bool isStart() const { return in.empty(); }
  bool isEmitOrStart() const { return !isNull() || in.empty(); }
  bool isReady() const { return nullOut.empty(); }
  bool isWait() const { return absorbOut.empty(); }
  static void assertSeqCoordsConsistent(const SeqCoords& before, const ProfileState& after, const AlignPath& step);
  static void assertSeqCoordsConsistent(const SeqCoords& before, const SeqCoords& after, const AlignPath& step, const AlignPath& atDest);
};

struct Profile {
  vguard<ProfileState> state;                       // START first, END last, topologically sorted
  vguard<ProfileTransition> trans;
  size_t components = 0;
  AlphTok alphSize = 0;
  AlignRowIndex rootRowIndex = 0;
  map<AlignRowIndex, string> seq;                   // the sequences below this node
  map<ProfileStateIndex, ProfileStateIndex> equivAbsorbState;
  map<string, string> meta;
  string name;
  Profile() = default;
  Profile(size_t nComponents, AlphTok nSymbols, AlignRowIndex row) : components(nComponents), alphSize(nSymbols), rootRowIndex(row) {}
  Profile(size_t nComponents, const string& alphabet, const FastSeq& leaf, AlignRowIndex row);
  ProfileStateIndex size() const { return state.size(); }
  const ProfileState& start() const { return state[0]; }
  const ProfileState& end() const { return state[state.size() - 1]; }
  const ProfileTransition* getTrans(ProfileStateIndex from, ProfileStateIndex to) const;
  LogProb calcSumPathAbsorbProbs(const vguard<LogProb>& logWeightOfComponent, const vguard<vguard<LogProb>>& logInsertProb,
                                 const char* tag = "cumLogProb");
  string toJson() const;
  void writeJson(std::ostream& to) const;
  // consistency checks (abort with a message): every state Wait or Ready; END reachable; coordinates; transition indices
  void assertAllStatesWaitOrReady() const;   void assertPathToEndExists() const;
  void assertSeqCoordsConsistent() const;    void assertTransitionsConsistent() const;
  Profile addReadyStates() const;
  static Profile withReadyStates(Profile&& src);   // addReadyStates of a profile that is not needed afterwards (moves its states)
  vguard<ProfileStateIndex> examplePathToEnd() const;
  bool isEmpty() const;                             // no emitting state
};

// ---- src/pairhmm.h --------------------------------------------------------------------------
struct PairHMM : AlphabetOwner {
  enum State { IMM = 0, IMD = 1, IDM = 2, IMI = 3, IIW = 4, TotalStates = 5,
               SSS = 0, SSI = 3, SIW = 4, EEE = 5 };      // (start aliases of IMM, IMI, IIW; END)
  const ProbModel& l;                                    // the two branches: probabilities and their logarithms
  const ProbModel& r;
  const LogProbModel logl, logr;
  vguard<vguard<LogProb>> logRoot;
  int components() const { return static_cast<int>(logRoot.size()); }
  static const char* stateName(State state, bool xAtStart, bool yAtStart);
  // one weight per move of the composite machine, named <source>_<destination>; listed by destination
  LogProb imm_imm, imd_imm, idm_imm, imi_imm, iiw_imm;
  LogProb imm_imd, imd_imd, idm_imd, imi_imd;
  LogProb imm_idm, imd_idm, idm_idm, iiw_idm;
  LogProb imm_imi, imi_imi;
  LogProb imm_iiw, imi_iiw, iiw_iiw;
  LogProb imm_eee, imd_eee, idm_eee, imi_eee, iiw_eee;
  PairHMM(const ProbModel& left, const ProbModel& right, const vguard<Vec>& rootDistribution);
  LogProb weight[TotalStates][TotalStates + 1];   // the same weights as a table [src][dest], -inf where there is no such move
  LogProb lpTrans(State from, State to) const;
  static vguard<State> sources(State to);
  static vguard<State> states();
};

// ---- src/forward.h --------------------------------------------------------------------------
class SumProduct;   // counts path: out of scope, only ever passed as NULL

class DPMatrix {
public:
  struct XYCell {
    LogProb lp[PairHMM::TotalStates];
    XYCell() { for (size_t s = 0; s < PairHMM::TotalStates; ++s) lp[s] = -std::numeric_limits<double>::infinity(); }
    LogProb operator()(PairHMM::State which) const { return lp[which]; }
  };
  struct CellCoords {
    PairHMM::State state;
    ProfileStateIndex xpos, ypos;
    CellCoords() : state(PairHMM::EEE), xpos(0), ypos(0) {}
    CellCoords(ProfileStateIndex x, ProfileStateIndex y, PairHMM::State s) : state(s), xpos(x), ypos(y) {}
    bool operator<(const CellCoords& c) const { return xpos == c.xpos ? ypos == c.ypos ? state < c.state : ypos < c.ypos : xpos < c.xpos; }
    bool operator==(const CellCoords& c) const { return xpos == c.xpos && ypos == c.ypos && state == c.state; }
  };
  enum ProfilingStrategy { KeepAll = 0, CollapseChains = 1, DontCountSubstEvents = 0, CountSubstEvents = 2,
                           DontCountIndelEvents = 0, CountIndelEvents = 4, DontIncludeBestTrace = 0, IncludeBestTrace = 8,
                           DontKeepGapsOpen = 0, KeepGapsOpen = 16 };
  using Path = list<CellCoords>;
  // (hx_host_walk.cpp) the cells next to a cell with the log-weight of the move between them, as a flat list: sorted by
  // cell, one entry per cell.  Tracebacks and profile construction work on these; the map-returning members of the
  // reference's interface are adapters.
  typedef std::pair<CellCoords, LogProb> Move;
  typedef vguard<Move> Moves;
  typedef std::mt19937 random_engine;
  static const char* random_engine_name() { return "mt" "19937"; }

  const Profile& x;
  const Profile& y;
  const bool xEmpty;
  const bool yEmpty;
  Profile subx, suby;           // lpAbsorb of x, y left-multiplied by the branch matrices (device result); shells: states carry lpAbsorb only
  const PairHMM& hmm;
  const AlphTok alphSize;
  const ProfileStateIndex xSize;
  const ProfileStateIndex ySize;
  const CellCoords startCell;
  const CellCoords endCell;
  LogProb lpEnd;
  const GuideAlignmentEnvelope envelope;
  vguard<SeqIdx> xClosestLeafPos;
  vguard<SeqIdx> yClosestLeafPos;
  vguard<bool> xNearStart;
  vguard<bool> yNearEnd;

  DPMatrix(const Profile& xProfile, const Profile& yProfile, const PairHMM& pairHmm, const GuideAlignmentEnvelope& band);
  virtual ~DPMatrix();

  // cell accessors: -inf outside storage (src/forward.h:68-88)
  LogProb cell(ProfileStateIndex xpos, ProfileStateIndex ypos, PairHMM::State state) const;
  LogProb cell(const CellCoords& c) const { return cell(c.xpos, c.ypos, c.state); }
  XYCell xyCell(ProfileStateIndex xpos, ProfileStateIndex ypos) const;
  LogProb lpStart() const { return cell(0, 0, PairHMM::IMM); }

  bool atEdge(ProfileStateIndex xpos, ProfileStateIndex ypos) const { return xNearStart[xpos] || yNearEnd[ypos]; }
  bool inEnvelope(ProfileStateIndex xpos, ProfileStateIndex ypos) const {
    return atEdge(xpos, ypos) || envelope.inRange(xClosestLeafPos[xpos], yClosestLeafPos[ypos]);
  }
  void write(std::ostream& out, bool edgeOnly = false) const;
  string cellName(const CellCoords& which) const;
  string toString(bool onlyEdges = false) const;
  static size_t cellSize() { return sizeof(LogProb) * PairHMM::TotalStates; }
  static random_engine newRNG();
  int components() const { return hmm.components(); }

  // log-sum-exp policy of the device fills: 0 = exact (bit-identical to the reference, default), 1 = fast
  static void setFillMode(unsigned hxFlags);
  // device-side best-path traceback (default on; HX_HOST_TRACEBACK=1 in the environment or false here keeps the
  // reference's host loop over a host copy of the matrix - same paths either way)
  static void setDeviceTraceback(bool on);
  static bool deviceTraceback();
  static unsigned fillMode();

protected:
  vguard<LogProb> insx, insy, rootsubx, rootsuby;
  vguard<vguard<LogProb>> absorbScratch;
  // device-resident batch this matrix is one job of (shared by the ForwardMatrix objects of one
  // ForwardMatrix::fillBatch call and by their BackwardMatrix objects; destroyed with the last of them)
  struct BatchHandle {
    hx_batch* b;
    int nJobs;
    vguard<int> jobOf;                       // fillBatch over several devices: caller's job index of each job of this batch
    bool backwardDone;
    // ForwardMatrix::bestTrace of every job, found on the device on first use (hx_batch_best_trace)
    bool bestTracesDone;
    long long bestTraceCap;
    vguard<int32_t> bestTraceCells;         // [nJobs][bestTraceCap] hx_trace_cell = {xpos, ypos, state}
    vguard<int32_t> bestTraceLen;           // [nJobs]
    vguard<int32_t> bestTraceTies;          // [nJobs] near-tie flags of those walks (hx_batch_best_trace_ties)
    BatchHandle(hx_batch* b, int nJobs) : b(b), nJobs(nJobs), backwardDone(false), bestTracesDone(false), bestTraceCap(0) {}
    ~BatchHandle();
  };
  std::shared_ptr<BatchHandle> handle;
  hx_batch* batch;               // == handle->b
  int jobIndex;                  // this matrix's job in the batch
  int which;                     // 0 Forward, 1 Backward matrix of the batch
  mutable double* hostCells;     // lazy copy of the device matrix (strip-skewed layout), page-locked, pooled
  mutable size_t hostCellsCap;
  mutable bool haveHostCells;
  mutable bool hostCopyInFlight;  // hx_batch_read_matrix_async issued into hostCells, not yet waited for
  // cells gathered from the device without copying the matrix (hx_batch_read_cells): used while the full
  // host copy does not exist, so that a best-path profile never moves 40 B/cell over PCIe
  mutable map<std::pair<ProfileStateIndex, ProfileStateIndex>, XYCell> sparseCells;
  void prefetchCells(const set<CellCoords>& cells) const;
public:
  // starts the device-to-host copy of the matrix without waiting for it; the first host access waits (ensureHostCells)
  void startHostCopy() const;
  // keeps the values of these cells and releases the host copy of the matrix (makeProfile reads only its chosen cells)
  void retainCells(const set<CellCoords>& cells) const;
  bool onHost() const { return haveHostCells; }
protected:
  long long stripStride, planeStride, blockStride, matrixDoubles;   // hx_layout of this matrix

  void createBatchAndPrepare();  // flatten inputs -> hx_batch_create; run the fill; fetch the prepared vectors
  void attach(const std::shared_ptr<BatchHandle>& h, int job, double lpEndOfJob);   // adopt job `job` of a filled batch
  void fetchPrepared();
  void ensureHostCells() const;
  void initAbsorbScratch(ProfileStateIndex xpos, ProfileStateIndex ypos) {
    for (int cpt = 0; cpt < components(); ++cpt)
      for (size_t n = 0; n < hmm.alphabetSize(); ++n)
        absorbScratch[cpt][n] = subx.state[xpos].lpAbsorb[cpt][n] + suby.state[ypos].lpAbsorb[cpt][n];
  }
  // (memoised per cell: sampled tracebacks revisit the same cells, and the sum is C x (A + 1) table look-ups)
  std::unordered_map<unsigned long long, LogProb> absorbMemo;
  LogProb computeLogProbAbsorb(ProfileStateIndex xpos, ProfileStateIndex ypos) {
    const unsigned long long key = ((unsigned long long)xpos << 32) | (unsigned long long)ypos;
    const auto hit = absorbMemo.find(key);
    if (hit != absorbMemo.end()) return hit->second;
    initAbsorbScratch(xpos, ypos);
    return absorbMemo[key] = logInnerProduct(hmm.logRoot, absorbScratch);
  }
  static void settle(Moves& m);
  static CellCoords pickBest(const Moves& m);
  CellCoords pickSampled(const Moves& m, random_engine& generator) const;
  bool isAbsorbing(const CellCoords& at) const;
  bool changesX(const CellCoords& at) const;
  bool changesY(const CellCoords& at) const;
  LogProb lpCellEmitOrAbsorb(const CellCoords& at);
  list<CellCoords> equivAbsorbCells(const CellCoords& at) const;
  static CellCoords bestCell(const map<CellCoords, LogProb>& scored);
  CellCoords sampleCell(const map<CellCoords, LogProb>& scored, random_engine& rng) const;
  friend class BackwardMatrix;
};

class ForwardMatrix : public DPMatrix {
public:
  SumProduct* sumProd;                     // always NULL here
  const AlignRowIndex parentRowIndex;

  struct EffectiveTransition {
    AlignPath bestAlignPath;
    LogProb lpBestAlignPath, lpPath;
    EffectiveTransition();
  };

  ForwardMatrix(const Profile& x, const Profile& y, const PairHMM& hmm, AlignRowIndex parentRowIndex,
                const GuideAlignmentEnvelope& env, SumProduct* sumProd = NULL);

  // Not in the reference: n independent Forward fills as ONE device batch (tree nodes whose children
  // are ready, nodes of different families).  Same results as n constructor calls; the caller deletes
  // the matrices as usual.  x, y and hmm must outlive the matrices, as for the constructor.
  struct JobSpec {
    const Profile* x;
    const Profile* y;
    const PairHMM* hmm;
    AlignRowIndex parentRowIndex;
    GuideAlignmentEnvelope env;
  };
  static vguard<ForwardMatrix*> fillBatch(const vguard<JobSpec>& jobs);
  // ... farmed over several devices: jobs sorted by lattice cells, longest first, each dealt to the least loaded device
  // (lptAssign); one device batch per device, all launched before the first result is awaited.  Same results.
  static vguard<ForwardMatrix*> fillBatch(const vguard<JobSpec>& jobs, const vguard<int>& devices);

  Path bestTrace();
  Path bestTrace(const CellCoords& from);
  // the same walk; *nearTie is set when at some step another source cell came within 1e-9 (relative) of the best one
  Path bestTrace(const CellCoords& from, bool* nearTie);
  // the best trace of a fresh fill of this pair under the exact policy (the reference's choice at near ties, see bestTrace())
  Path exactBestTrace();
  static bool refillAtNearTies();
  Path sampleTrace(random_engine& rng);
  // sampled walks made on the device (HX_DEVICE_SAMPLING=1, hx_batch_sample_traces); see hx_host_walk.cpp
  static bool deviceSampling();
  bool sampleTracesOnDevice(const random_engine& generator, size_t walks, vguard<Path>& paths, vguard<long long>& draws);
  AlignPath bestAlignPath();

  Profile makeProfile(const set<CellCoords>& chosen, ProfilingStrategy how = CollapseChains);
  Profile sampleProfile(random_engine& generator, size_t profileSamples, size_t maxCells = 0,
                        ProfilingStrategy strategy = CollapseChains, size_t minLen = 0,
                        size_t maxLen = std::numeric_limits<size_t>::max());
  // the first half of sampleProfile: the cells of the best trace and of the sampled traces (all the generator draws);
  // makeProfile(cells) is the second half and touches neither the generator nor the device once the matrix is on the host
  set<CellCoords> sampleCells(random_engine& generator, size_t profileSamples, size_t maxCells = 0, ProfilingStrategy strategy = CollapseChains,
                              size_t minLen = 0, size_t maxLen = (size_t)-1);
  bool hostMatrixReady() const { return haveHostCells || !batch; }
  Profile bestProfile(ProfilingStrategy how = CollapseChains);

  map<CellCoords, LogProb> sourceTransitionsWithoutEmitOrAbsorb(const CellCoords& into);
  map<CellCoords, LogProb> sourceTransitions(const CellCoords& into);
  void slowFillTest();   // (Forward)

private:
  void movesInto(const CellCoords& dest, Moves& out) const;     // sourceTransitionsWithoutEmitOrAbsorb as a flat list
  void scoredSources(const CellCoords& dest, Moves& out);       // ... + the destination's emission + the source's Forward cell
  map<CellCoords, LogProb> sourceCells(const CellCoords& into);
  AlignPath cellAlignPath(const CellCoords& at) const;
  AlignPath transitionAlignPath(const CellCoords& from, const CellCoords& to) const;
  AlignPath traceAlignPath(const Path& cells) const;
  ProfileState::SeqCoords cellSeqCoords(const CellCoords& at) const;
  LogProb eliminatedLogProbInsert(const CellCoords& at) const;
  friend class BackwardMatrix;

private:
  struct Deferred {};
  ForwardMatrix(const Profile& x, const Profile& y, const PairHMM& hmm, AlignRowIndex parentRowIndex,
                const GuideAlignmentEnvelope& env, Deferred);   // no fill yet (fillBatch)
};

class BackwardMatrix : public DPMatrix {
public:
  struct CellPostProb : CellCoords {
    LogProb logPostProb;
    CellPostProb(ProfileStateIndex xpos, ProfileStateIndex ypos, PairHMM::State state, LogProb lpp) : CellCoords(xpos, ypos, state), logPostProb(lpp) {}
    bool operator<(const CellPostProb& cpp) const { return logPostProb < cpp.logPostProb; }
  };
  ForwardMatrix& fwd;

  explicit BackwardMatrix(ForwardMatrix& forward);

  double transPostProb(const CellCoords& from, const CellCoords& to) const;
  double cellPostProb(const CellCoords& at) const;
  Path bestTrace(const CellCoords& from);
  std::priority_queue<CellPostProb> cellsAbovePostProbThreshold(double minPostProb) const;
  Profile bestProfile(ProfilingStrategy how = CollapseChains);
  Profile postProbProfile(double minPostProb, size_t cellBudget = 0, ProfilingStrategy how = CollapseChains);
  map<CellCoords, LogProb> destTransitions(const CellCoords& outOf);
  void sourceDestTransTest();
  void slowFillTest();   // (Backward)

private:
  void movesOutOf(const CellCoords& src, Moves& out);           // destTransitions as a flat list
  void scoredDestinations(const CellCoords& src, Moves& out);   // ... + the destination's Backward cell
  map<CellCoords, LogProb> destCells(const CellCoords& outOf);
  bool addCells(set<CellCoords>& cells, size_t maxCells, const list<CellCoords>& fwdTrace, const list<CellCoords>& backTrace, bool keepGapsOpen);
  bool addTrace(const CellCoords& through, set<CellCoords>& chosen, size_t cellBudget, bool gapsOpen);
};

string pairParentName(const string& lChildName, double lTime, const string& rChildName, double rTime);  // Tree::pairParentName

// ---- src/recon.h / recon.cpp:864-915, 917-1052 (subset) ----------------------------------------
// The progressive loop that calls the DP: one ForwardMatrix per internal node in post-order,
// band-doubling retry on zero likelihood, sampled (or posterior) profiles for non-root nodes,
// best alignment path at the root.  Tree building, guide-alignment construction, file formats,
// refinement and counts are outside this build's scope: the tree arrives as post-order arrays
// and the guide as an AlignPath over leaf rows.
typedef int TreeNodeIndex;

struct ReconTree {
  vguard<TreeNodeIndex> parent;          // -1 for the root (last node)
  vguard<double> branchLen;              // length of the branch above each node
  vguard<string> nodeName;
  vguard<vguard<TreeNodeIndex>> child;
  void addNode(TreeNodeIndex parentNode, double len, const string& name);
  void finish();                         // builds child lists, asserts post-order and binary
  TreeNodeIndex nodes() const { return (TreeNodeIndex)parent.size(); }
  bool isLeaf(TreeNodeIndex n) const { return child[n].empty(); }
  TreeNodeIndex root() const { return nodes() - 1; }
  TreeNodeIndex getChild(TreeNodeIndex n, size_t k) const { return child[n][k]; }
  double branchLength(TreeNodeIndex n) const { return branchLen[n]; }
};

struct Reconstructor {
  RateModel model;
  int maxDistanceFromGuide;              // -band, default 20 (src/recon.h:16)
  size_t profileSamples;                 // -profsamples, default 10
  size_t profileMaxStates;               // -profmaxstates; 0 = unlimited (the reference default is RAM dependent)
  bool includeBestTraceInProfile, keepGapsOpen, usePosteriorsForProfile, reconstructRoot;
  double minPostProb;                    // -profminpost
  unsigned rndSeed;                      // mt19937::default_seed
  DPMatrix::random_engine generator;
  // Not in the reference: fill every tree node whose children are ready as one device batch (the fills
  // are independent; tracebacks and sampling still run in node order, so the results are unchanged).
  bool batchReadyNodes;                  // default true
  double maxBatchLatticeCells;           // upper bound on the lattice cells of one batch (device memory); default 4e8

  struct Dataset {
    ReconTree tree;
    map<TreeNodeIndex, FastSeq> seqs;    // ungapped sequence of every leaf node
    vguard<TreeNodeIndex> closestLeaf;
    AlignPath guide, path;               // guide: rows = leaf node indices, empty = no band; path: the result (root alignment)
    LogProb lpFinalFwd, lpFinalTrace;
    map<TreeNodeIndex, int> bandUsed;
    void prepareRecon();
    vguard<FastSeq> gappedRecon() const; // Alignment(ungapped, path).gapped()
  };

  // Not in the reference (which is one process, one core): the devices independent pair DPs are farmed over.
  // Empty = the thread's device (HX_DEVICE or 0).  With several: the ready nodes of a tree level are dealt to the devices
  // (reconstruct), and whole families are dealt to one host thread per device (reconstructAll) - in both cases longest
  // job first.  Tracebacks and sampling stay in node order per family, each family has its own generator seeded as the
  // reference seeds it, so the results do not depend on the number of devices.
  vguard<int> devices;

  Reconstructor();
  void reconstruct(Dataset& family);
  void seedGenerator();
  void reconstructAll(vguard<Dataset*>& datasets);   // reference src/recon.cpp:1368-1372: every family
  static double familyCost(const Dataset& dataset);   // estimated lattice cells of a family's pair DPs
};

// Longest-processing-time-first list scheduling: jobs in order of decreasing cost, each to the device with the least
// load so far (ties: the lower device, the lower job index first).  Returns the device index of every job.
vguard<int> lptAssign(const vguard<double>& cost, int nDevices);

namespace detail {          // shared by the device-backed classes (hx_host_forward.cpp)
void ensureDevice();                  // the calling thread's device
void ensureDevice(int ordinal);       // hx_init for that device, once per process
int threadDevice();                   // the device this host thread creates its matrices on (HX_DEVICE or 0 by default)
void setThreadDevice(int ordinal);
void warmHostBuffers(size_t doubles, int count);   // page-lock `count` matrix buffers on a helper thread
void mergeTiming(const FillTiming& from, FillTiming& into);
double* pinnedTake(size_t doubles, size_t& capacity);
void pinnedReserve(size_t doubles, int count);     // make sure `count` free page-locked buffers of that size exist (call while the device is busy)
void pinnedGive(double* p, size_t capacity);
void check(int rc, const char* what);
}  // namespace detail

// ---- src/fastseq.h (k-mers), src/diagenv.h, src/quickalign.h ---------------------------------
// The guide-alignment pair DP.  DiagonalEnvelope keeps the reference's
// members and methods that callers use (the storage-index bookkeeping is the reference's CPU
// layout and has no counterpart here: the matrix lives in the device layout); QuickAlignMatrix
// keeps the reference's interface with the fill running on the device.
using Kmer = unsigned long long;
using UnvalidatedTokSeq = vguard<UnvalidatedAlphTok>;
UnvalidatedTokSeq unvalidatedTokens(const FastSeq& seq, const string& alphabet);
bool kmerValid(SeqIdx k, vguard<int>::const_iterator tok);
Kmer makeKmer(SeqIdx k, vguard<int>::const_iterator tok, AlphTok alphabetSize);
void writeFastaSeqs(std::ostream& out, const vguard<FastSeq>& fastSeqs);

struct KmerIndex {
  const SeqIdx kmerLen;
  const string& alphabet;
  const FastSeq& seq;
  map<Kmer, vguard<SeqIdx>> kmerLocations;
  KmerIndex(const FastSeq& sequence, const string& symbols, SeqIdx k);
};

enum { DEFAULT_KMER_LENGTH = 6, DEFAULT_BAND_SIZE = 64, DEFAULT_KMER_THRESHOLD = -1 };

struct DiagonalEnvelope {
  const FastSeq* px;
  const FastSeq* py;
  const SeqIdx xLen;
  const SeqIdx yLen;
  vguard<int> diagonals;   // sorted ascending; (i,j) is on diagonal d if i-j=d
  bool full;               // set by initFull (lets the device skip the per-cell membership test)
  DiagonalEnvelope(const FastSeq& x, const FastSeq& y) : px(&x), py(&y), xLen(x.length()), yLen(y.length()), full(false) {}
  void initFull();
  void initSparse(const KmerIndex& yKmerIndex, unsigned int bandSize = DEFAULT_BAND_SIZE,
                  int kmerThreshold = DEFAULT_KMER_THRESHOLD, size_t cellSize = sizeof(double), size_t maxSize = 0);
  int minDiagonal() const { return 1 - static_cast<int>(yLen); }
  int maxDiagonal() const { return (int)xLen - 1; }
  bool contains(SeqIdx i, SeqIdx j) const;
  static SeqIdx get_i(SeqIdx j, int diagonal) { return static_cast<SeqIdx>(diagonal + (int)j); }
  static int get_diag(SeqIdx i, SeqIdx j) { return static_cast<int>(i) - static_cast<int>(j); }
  bool intersects(SeqIdx j, int diag) const { const int i = diag + (int)j; return i > 0 && i <= (int)xLen; }
  vguard<SeqIdx> forward_i(SeqIdx column) const;
  vguard<SeqIdx> reverse_i(SeqIdx column) const;
};

class QuickAlignMatrix {
public:
  enum State { Start, Match, Insert, Delete };
  const DiagonalEnvelope* penv;
  const FastSeq *px, *py;
  UnvalidatedTokSeq xTok, yTok;
  SeqIdx xLen, yLen;
  SeqIdx xEnd, yEnd;                // where the best local alignment ends
  LogProb start, end;
  LogProb result;
  const RateModel& model;
  const double time;
  vguard<vguard<LogProb>> submat;  // log odds-ratio
  LogProb m2m, m2i, m2d;           // affine-gap transition scores by source state
  LogProb i2i, i2m, i2d, i2e;
  LogProb d2d, d2m, d2e;
  LogProb noGap, gapOpen, gapExtend;

  QuickAlignMatrix(const DiagonalEnvelope& envelope, const RateModel& rates, double branchLength);
  ~QuickAlignMatrix();
  // Not in the reference: n independent fills as one device batch (the pairs of an alignment graph)
  static vguard<QuickAlignMatrix*> fillBatch(const vguard<const DiagonalEnvelope*>& envs, const RateModel& model, double time);

  LogProb mat(SeqIdx i, SeqIdx j) const { return getCell(i, j, 0); }
  LogProb ins(SeqIdx i, SeqIdx j) const { return getCell(i, j, 1); }
  LogProb del(SeqIdx i, SeqIdx j) const { return getCell(i, j, 2); }
  double matchEmitScore(SeqIdx i, SeqIdx j) const {
    Assert(i > 0 && j > 0 && i <= xLen && j <= yLen, "Out of range: (i,j)=(%u,%u) (xLen,yLen)=(%u,%u)", i, j, xLen, yLen);
    const UnvalidatedAlphTok xt = xTok[i - 1], yt = yTok[j - 1];
    return (xt < 0 || yt < 0) ? 0 : submat[xt][yt];
  }
  static const char* stateToString(State which);
  static size_t cellSize() { return sizeof(double) * 3; }
  LogProb cellScore(SeqIdx row, SeqIdx column, State which) const;
  bool resultIsFinite() const { return result > -std::numeric_limits<double>::infinity(); }
  AlignPath alignPath(AlignRowIndex xRow, AlignRowIndex yRow) const;
  AlignPath alignPath() const;
  vguard<FastSeq> gappedSeq() const;   // Alignment(seqs, alignPath()).gapped()

protected:
  struct Deferred {};
  QuickAlignMatrix(const DiagonalEnvelope& env, const RateModel& model, double time, Deferred);
  void computeScores();
  void fillJob(hx_quick_job& job, vguard<double>& flatSub) const;
  LogProb getCell(SeqIdx i, SeqIdx j, unsigned int offset) const;   // -inf outside the envelope / the lattice
  static void updateMax(LogProb& best, State& bestState, double candidate, State candidateState);
  LogProb startGapScore(SeqIdx i, SeqIdx j) const {
    return (i == 1 ? noGap : (gapOpen + (i - 2) * gapExtend)) + (j == 1 ? noGap : (gapOpen + (j - 2) * gapExtend));
  }
  LogProb endGapScore(SeqIdx i, SeqIdx j) const {
    return (i == xLen ? noGap : (gapOpen + (xLen - i - 2) * gapExtend)) + (j == yLen ? noGap : (gapOpen + (yLen - j - 2) * gapExtend));
  }
  struct QuickHandle {
    hx_quick_batch* b;
    explicit QuickHandle(hx_quick_batch* b) : b(b) {}
    ~QuickHandle();
  };
  std::shared_ptr<QuickHandle> handle;
  int jobIndex;
  mutable double* hostCells;
  mutable size_t hostCellsCap;
  long long stripStride, planeStride, blockStride, matrixDoubles;   // hx_layout of the matrix
  void attach(const std::shared_ptr<QuickHandle>& h, int job, double score, int xe, int ye);
};

// ---- src/diagenv.h (parameters), src/span.h ---------------------------------------------------
struct DiagEnvParams {
  int kmerLen, kmerThreshold, bandSize;
  bool sparse, autoMemSize;
  size_t maxSize;
  size_t effectiveMaxSize() const;
  DiagEnvParams();
};

// The alignment graph of the guide alignment: which pairs are aligned, the pair farm (one device batch
// of QuickAlignMatrix fills instead of the reference's one-at-a-time loop, same results), the maximum
// spanning tree of the pairwise alignments and their merge.
struct AlignGraph {
  struct TrialEdge {
    AlignRowIndex row1 = 0, row2 = 0;
    TrialEdge() = default;
    TrialEdge(AlignRowIndex src, AlignRowIndex dest) : row1(src), row2(dest) {}
  };
  struct Edge : TrialEdge {
    LogProb lp;
    Edge() : lp(-std::numeric_limits<double>::infinity()) {}
    bool operator<(const Edge& other) const { return lp < other.lp; }
  };
  struct Components {              // disjoint sets of sequences: union-find forest + each set's members in ascending order
    vguard<size_t> up;
    vguard<vguard<size_t>> members;
    size_t count;
    explicit Components(size_t nSequences);
    size_t root(size_t sequence);
    void join(size_t a, size_t b);
  };

  const RateModel& model;
  const double time;
  const vguard<FastSeq>& seqs;
  const DiagEnvParams& diagEnvParams;
  vguard<std::priority_queue<Edge>> edges;
  vguard<map<AlignRowIndex, AlignPath>> edgePath;

  AlignGraph(const vguard<FastSeq>& seqs, const RateModel& model, const double time, const DiagEnvParams& diagEnvParams,
             ForwardMatrix::random_engine& generator);
  AlignGraph(const vguard<FastSeq>& sequences, const RateModel& rates, double branchLength, const DiagEnvParams& envelopeParams);
  AlignGraph(const vguard<FastSeq>& sequences, const RateModel& rates, double branchLength, const DiagEnvParams& envelopeParams,
             ForwardMatrix::random_engine* rngOrNull);
  void buildDenseGraph();
  void buildSparseRandomGraph(ForwardMatrix::random_engine& rng);
  void buildGraph(const list<TrialEdge>& candidates, const string& description);
  list<AlignPath> minSpanTree();            // the spanning tree's pairwise alignments; then merged, as rows, as gapped sequences
  AlignPath mstPath();
  Alignment mstAlign();
  vguard<FastSeq> mstGapped();
};

// ---- src/sampler.h:19-215, src/refiner.h:8-18: the per-branch pair DPs ----------------
// Parent profile x against child profile y across one branch: three states per cell (Match, Insert, Delete), the lattice of
// TreeAlignFuncs::SparseDPMatrix<3> inside a GuideAlignmentEnvelope.  The fill runs on the device (hx_branch.hip through the
// C ABI hx_branch_batch_*): Viterbi for Refiner::BranchMatrix, the reference's log_sum_exp for Sampler::BranchMatrix; the
// matrix comes back dense and cell(), best() walk it on the host as the reference does.
struct TreeAlignFuncs {
  typedef vguard<vguard<vguard<LogProb>>> PosWeightMatrix;      // pwm[pos][cpt][tok]
  enum State { Start = 0, Match = 0, Insert = 1, Delete = 2, End = 3 };   // ProbModel::State (src/model.h:135-137)
  static double transProb(const ProbModel& probs, State src, State dest);                       // src/model.cpp:400-447
  static PosWeightMatrix preMultiply(const PosWeightMatrix& child, const vguard<Mat>& logSubProb);    // src/sampler.cpp:452-463
  static vguard<LogProb> calcInsProbs(const PosWeightMatrix& child, const vguard<vguard<LogProb>>& logInsProb,
                                      const vguard<LogProb>& logCptWeight);                     // src/sampler.cpp:465-476
  static PosWeightMatrix leafPWM(const FastSeq& seq, const string& alphabet, int components);   // a sequence as certain columns (0 / -inf)

  class BranchMatrixBase {
  public:
    struct CellCoords { SeqIdx xpos, ypos; unsigned int state; };
    const RateModel& model;
    const ProbModel probModel;
    const LogProbModel logProbModel;
    LogProb mm, mi, md, me, im, ii, id, ie, dm, dd, de;
    AlignRowIndex xRow, yRow;
    const PosWeightMatrix& xSeq;
    const PosWeightMatrix ySub;
    const vguard<LogProb> yEmit;
    const SeqIdx xSize, ySize;
    LogProb lpEnd;

    BranchMatrixBase(const RateModel& rates, const PosWeightMatrix& parent, const PosWeightMatrix& child, double branchLength,
                     const GuideAlignmentEnvelope& envelope, const vguard<SeqIdx>& xEnvelopePos, const vguard<SeqIdx>& yEnvelopePos,
                     AlignRowIndex parentRow, AlignRowIndex childRow, bool viterbi);
    LogProb cell(SeqIdx xpos, SeqIdx ypos, unsigned int state) const;   // -inf outside the envelope; state End: lpEnd at the last cell
    bool inEnvelope(SeqIdx xpos, SeqIdx ypos) const;
    LogProb logMatch(SeqIdx xpos, SeqIdx ypos) const;
    LogProb lpTrans(State src, State dest) const;
    LogProb lpEmit(const CellCoords& at) const;
    static void getColumn(const CellCoords& at, bool& xUngapped, bool& yUngapped);
  private:
    const GuideAlignmentEnvelope& env;
    const vguard<SeqIdx>& xEnvPos;
    const vguard<SeqIdx>& yEnvPos;
    vguard<double> cells;      // dense [xSize][ySize][3] copy of the device matrix
  };
};

struct Refiner : TreeAlignFuncs {
  class BranchMatrix : public BranchMatrixBase {          // src/refiner.cpp:10-104
  public:
    BranchMatrix(const RateModel& rates, const PosWeightMatrix& parent, const PosWeightMatrix& child, double branchLength,
                 const GuideAlignmentEnvelope& envelope, const vguard<SeqIdx>& xEnvelopePos, const vguard<SeqIdx>& yEnvelopePos,
                 AlignRowIndex parentRow, AlignRowIndex childRow)
        : BranchMatrixBase(rates, parent, child, branchLength, envelope, xEnvelopePos, yEnvelopePos, parentRow, childRow, true) {}
    AlignPath best() const;
  };
};

struct Sampler : TreeAlignFuncs {
  class BranchMatrix : public BranchMatrixBase {          // src/sampler.cpp:1034-1084 (the fill; the sampling moves are not built)
  public:
    BranchMatrix(const RateModel& rates, const PosWeightMatrix& parent, const PosWeightMatrix& child, double branchLength,
                 const GuideAlignmentEnvelope& envelope, const vguard<SeqIdx>& xEnvelopePos, const vguard<SeqIdx>& yEnvelopePos,
                 AlignRowIndex parentRow, AlignRowIndex childRow)
        : BranchMatrixBase(rates, parent, child, branchLength, envelope, xEnvelopePos, yEnvelopePos, parentRow, childRow, false) {}
  };
};

}  // namespace historian
