// DPMatrix / ForwardMatrix / BackwardMatrix of the host mirror (see hx_host.h): the fills run
// on the GPU through the C ABI; everything here is the O(path) host logic of the reference
// (src/forward.cpp:225-577, 654-895, 1172-1379) over the device-filled matrix.
#include "hx_host.h"

#include <algorithm>
#include <cstdlib>
#include <iomanip>
#include <sstream>

#include "../../../include/historian_hip.h"

namespace historian {

static const double NEG_INF = -std::numeric_limits<double>::infinity();
#define FWD_BACK_ERROR_TOLERANCE .01

static unsigned g_fillMode = HX_LSE_EXACT;
static bool g_deviceReady = false;

void DPMatrix::setFillMode(unsigned hxFlags) { g_fillMode = hxFlags & HX_LSE_LINEAR; }
static int g_deviceTraceback = -1;
void DPMatrix::setDeviceTraceback(bool on) { g_deviceTraceback = on ? 1 : 0; }
bool DPMatrix::deviceTraceback() {
  if (g_deviceTraceback < 0) g_deviceTraceback = getenv("HX_HOST_TRACEBACK") ? 0 : 1;
  return g_deviceTraceback != 0;
}
unsigned DPMatrix::fillMode() { return g_fillMode; }

static void hxCheck(int rc, const char* what) {
  if (rc != HX_OK) Abort("%s failed (%d): %s", what, rc, hx_last_error());
}

static void ensureDevice() {
  if (g_deviceReady) return;
  const double t0 = wallSeconds();
  const char* dev = getenv("HX_DEVICE");
  // the device gets the table the host built with its own libm (reference src/logsumexp.cpp:6-16)
  hxCheck(hx_init(dev ? atoi(dev) : 0, logSumExpLookupTable.lookup, HX_LSE_TABLE_ENTRIES), "hx_init");
  const char* mode = getenv("HX_FILL_MODE");   // "fast" selects the fast log-sum-exp policy for this process
  if (mode && string(mode) == "fast") g_fillMode = HX_LSE_FAST;
  if (mode && string(mode) == "linear") g_fillMode = HX_LSE_LINEAR;
  g_deviceReady = true;
  fillTiming.deviceInit += wallSeconds() - t0;
}

// POD image of a Profile for the C ABI
namespace {
struct FlatProfile {
  vguard<int32_t> tsrc, tdst, inOff, inIdx, aoOff, aoIdx, noOff, noIdx, envPos;
  vguard<double> tlp, lpAbsorb;
  vguard<uint8_t> isNull;
  hx_profile pod;
  void build(const Profile& p, size_t C, size_t A, const vguard<int>* env) {
    const size_t N = p.size();
    for (const auto& t : p.trans) {
      tsrc.push_back((int32_t)t.src);
      tdst.push_back((int32_t)t.dest);
      tlp.push_back(t.lpTrans);
    }
    inOff.push_back(0); aoOff.push_back(0); noOff.push_back(0);
    lpAbsorb.assign(N * C * A, NEG_INF);
    for (size_t i = 0; i < N; ++i) {
      const ProfileState& s = p.state[i];
      for (auto t : s.in) inIdx.push_back((int32_t)t);
      for (auto t : s.absorbOut) aoIdx.push_back((int32_t)t);
      for (auto t : s.nullOut) noIdx.push_back((int32_t)t);
      inOff.push_back((int32_t)inIdx.size());
      aoOff.push_back((int32_t)aoIdx.size());
      noOff.push_back((int32_t)noIdx.size());
      isNull.push_back(s.isNull() ? 1 : 0);
      if (!s.isNull())
        for (size_t c = 0; c < C; ++c)
          for (size_t a = 0; a < A; ++a) lpAbsorb[(i * C + c) * A + a] = s.lpAbsorb[c][a];
    }
    if (env) envPos.assign(env->begin(), env->end());
    pod.n_states = (int32_t)N;
    pod.n_trans = (int32_t)p.trans.size();
    pod.trans_src = tsrc.data(); pod.trans_dst = tdst.data(); pod.trans_lp = tlp.data();
    pod.in_off = inOff.data(); pod.in_idx = inIdx.data();
    pod.aout_off = aoOff.data(); pod.aout_idx = aoIdx.data();
    pod.nout_off = noOff.data(); pod.nout_idx = noIdx.data();
    pod.is_null = isNull.data(); pod.lp_absorb = lpAbsorb.data();
    pod.env_pos = env ? envPos.data() : NULL;
  }
};
}  // namespace

DPMatrix::DPMatrix(const Profile& x, const Profile& y, const PairHMM& hmm, const GuideAlignmentEnvelope& env)
    : x(x), y(y), xEmpty(x.isEmpty()), yEmpty(y.isEmpty()), subx(x), suby(y), hmm(hmm),
      alphSize((AlphTok)hmm.alphabetSize()), xSize(x.size()), ySize(y.size()),
      startCell(0, 0, PairHMM::SSS), endCell(xSize - 1, ySize - 1, PairHMM::EEE),
      lpEnd(NEG_INF), envelope(env), xClosestLeafPos(xSize, 0), yClosestLeafPos(ySize, 0),
      xNearStart(xSize, false), yNearEnd(ySize, false),
      insx(x.size(), NEG_INF), insy(y.size(), NEG_INF), rootsubx(x.size(), NEG_INF), rootsuby(y.size(), NEG_INF),
      absorbScratch(hmm.components(), vguard<LogProb>(hmm.alphabetSize())),
      batch(NULL), jobIndex(0), which(0), hostCells(NULL), hostCellsCap(0), haveHostCells(false), stripStride(0), planeStride(0), blockStride(128), matrixDoubles(0) {
  if (env.initialized()) {
    for (ProfileStateIndex i = 1; i < xSize; ++i) xClosestLeafPos[i] = x.state[i].seqCoords.at(env.row1);
    for (ProfileStateIndex j = 1; j < ySize; ++j) yClosestLeafPos[j] = y.state[j].seqCoords.at(env.row2);
  }
  xNearStart[0] = true;
  for (ProfileStateIndex i = 0; i < xSize; ++i)
    if (xNearStart[i])
      for (const auto& t : x.state[i].nullOut) xNearStart[x.trans[t].dest] = true;
  for (auto yt : y.end().in) yNearEnd[y.trans[yt].src] = true;
}

// Page-locked buffers for the host copies of device matrices, recycled across matrices: pinning is not
// free (it is what makes the D2H copy run at link rate), and consecutive tree nodes need similar sizes.
namespace {
struct PinnedPool {
  std::vector<std::pair<size_t, double*> > freeList;   // (capacity in doubles, buffer)
  double* take(size_t n, size_t& cap) {
    size_t best = freeList.size();
    for (size_t k = 0; k < freeList.size(); ++k)
      if (freeList[k].first >= n && (best == freeList.size() || freeList[k].first < freeList[best].first)) best = k;
    if (best < freeList.size()) {
      cap = freeList[best].first;
      double* p = freeList[best].second;
      freeList.erase(freeList.begin() + (long)best);
      return p;
    }
    if (freeList.size() >= 4) {                         // keep the pool small: drop the smallest buffer
      size_t small = 0;
      for (size_t k = 1; k < freeList.size(); ++k)
        if (freeList[k].first < freeList[small].first) small = k;
      hx_host_free(freeList[small].second);
      freeList.erase(freeList.begin() + (long)small);
    }
    void* p = NULL;
    cap = n + n / 8;
    hxCheck(hx_host_alloc(cap * sizeof(double), &p), "hx_host_alloc");
    return static_cast<double*>(p);
  }
  void give(double* p, size_t cap) { freeList.push_back(std::make_pair(cap, p)); }
};
PinnedPool g_pinned;
}  // namespace

DPMatrix::~DPMatrix() {
  if (hostCells) g_pinned.give(hostCells, hostCellsCap);
}

namespace detail {
void ensureDevice() { historian::ensureDevice(); }
double* pinnedTake(size_t doubles, size_t& capacity) { return g_pinned.take(doubles, capacity); }
void pinnedGive(double* p, size_t capacity) { g_pinned.give(p, capacity); }
void check(int rc, const char* what) { hxCheck(rc, what); }
}  // namespace detail

DPMatrix::BatchHandle::~BatchHandle() {
  if (b) hx_batch_destroy(b);
}

// Replaces the member initialisers subx/suby/insx/... of reference src/forward.cpp:20-56 and
// the fill loops of src/forward.cpp:78-223: flatten, create the device job, run the Forward fill.
namespace {
// everything hx_batch_create reads for one job (must stay put until the call returns)
struct JobImage {
  FlatProfile fx, fy;
  vguard<int> envx, envy;
  vguard<double> logRoot, logSubL, logSubR, logInsL, logInsR;
  hx_hmm hh;
  hx_pair_job job;
};
}  // namespace

static void buildJobImage(const DPMatrix& m, const vguard<SeqIdx>& xClosestLeafPos, const vguard<SeqIdx>& yClosestLeafPos,
                          JobImage& im) {
  const PairHMM& hmm = m.hmm;
  const size_t C = hmm.components(), A = hmm.alphabetSize();
  const bool banded = m.envelope.initialized();
  if (banded) {
    im.envx.resize(m.xSize);
    im.envy.resize(m.ySize);
    for (ProfileStateIndex i = 0; i < m.xSize; ++i) im.envx[i] = m.envelope.cumulativeMatches[m.envelope.row1PosToCol[xClosestLeafPos[i]]];
    for (ProfileStateIndex j = 0; j < m.ySize; ++j) im.envy[j] = m.envelope.cumulativeMatches[m.envelope.row2PosToCol[yClosestLeafPos[j]]];
  }
  im.fx.build(m.x, C, A, banded ? &im.envx : NULL);
  im.fy.build(m.y, C, A, banded ? &im.envy : NULL);
  hx_hmm& hh = im.hh;
  hh.alph_size = (int32_t)A;
  hh.components = (int32_t)C;
  for (int s = 0; s < 5; ++s)
    for (int d = 0; d < 6; ++d) hh.lp_trans[s][d] = hmm.lpTrans((PairHMM::State)s, (PairHMM::State)d);
  for (size_t c = 0; c < C; ++c)
    for (size_t a = 0; a < A; ++a) {
      im.logRoot.push_back(hmm.logRoot[c][a]);
      im.logInsL.push_back(hmm.logl.logInsProb[c][a]);
      im.logInsR.push_back(hmm.logr.logInsProb[c][a]);
      for (size_t d = 0; d < A; ++d) {
        im.logSubL.push_back(log(hmm.l.subMat[c][a][d]));   // host libm log, as Profile::leftMultiply takes it
        im.logSubR.push_back(log(hmm.r.subMat[c][a][d]));
      }
    }
  hh.log_root = im.logRoot.data();
  hh.log_sub_l = im.logSubL.data();
  hh.log_sub_r = im.logSubR.data();
  hh.log_ins_l = im.logInsL.data();
  hh.log_ins_r = im.logInsR.data();
  hh.log_cptw_l = hmm.logl.logCptWeight.data();
  hh.log_cptw_r = hmm.logr.logCptWeight.data();
  im.job.x = &im.fx.pod;
  im.job.y = &im.fy.pod;
  im.job.hmm = &im.hh;
  im.job.max_distance = m.envelope.maxDistance;
}

void DPMatrix::createBatchAndPrepare() {
  ensureDevice();
  const double t0 = wallSeconds();
  JobImage im;
  buildJobImage(*this, xClosestLeafPos, yClosestLeafPos, im);
  hx_batch* b = NULL;
  hxCheck(hx_batch_create(&im.job, 1, g_fillMode | HX_SPARSE_ENVELOPE, &b), "hx_batch_create");
  std::shared_ptr<BatchHandle> h(new BatchHandle(b, 1));
  const double t1 = wallSeconds();
  hxCheck(hx_batch_forward(b, NULL), "hx_batch_forward");
  double lp = NEG_INF;
  hxCheck(hx_batch_lp_end(b, &lp), "hx_batch_lp_end");
  const double t2 = wallSeconds();
  float kms = 0;
  if (hx_batch_last_kernel_ms(b, 0, &kms) == HX_OK) fillTiming.forwardKernel += 1e-3 * kms;
  fillTiming.flattenAndUpload += t1 - t0;
  fillTiming.forwardWait += t2 - t1;
  fillTiming.fills += 1;
  fillTiming.cells += (long long)(xSize - 1) * (long long)(ySize - 1);
  attach(h, 0, lp);
}

void DPMatrix::attach(const std::shared_ptr<BatchHandle>& h, int job, double lpEndOfJob) {
  handle = h;
  batch = h->b;
  jobIndex = job;
  lpEnd = lpEndOfJob;
  hx_layout lay;
  hxCheck(hx_batch_layout(batch, jobIndex, 0, &lay), "hx_batch_layout");
  stripStride = lay.strip_stride;
  planeStride = lay.plane_stride;
  blockStride = lay.block_stride;
  matrixDoubles = lay.matrix_doubles;
  fetchPrepared();
}

// n independent fills, one device batch (not in the reference; see hx_host.h)
vguard<ForwardMatrix*> ForwardMatrix::fillBatch(const vguard<JobSpec>& jobs) {
  vguard<ForwardMatrix*> out;
  if (jobs.empty()) return out;
  ensureDevice();
  const double t0 = wallSeconds();
  const size_t n = jobs.size();
  vguard<JobImage> images(n);          // sized once: hx_pair_job holds pointers into its elements
  vguard<hx_pair_job> pods(n);
  for (size_t k = 0; k < n; ++k) {
    const JobSpec& js = jobs[k];
    ForwardMatrix* f = new ForwardMatrix(*js.x, *js.y, *js.hmm, js.parentRowIndex, js.env, Deferred());
    out.push_back(f);
    buildJobImage(*f, f->xClosestLeafPos, f->yClosestLeafPos, images[k]);
    pods[k] = images[k].job;
  }
  hx_batch* b = NULL;
  hxCheck(hx_batch_create(pods.data(), (int32_t)n, g_fillMode | HX_SPARSE_ENVELOPE, &b), "hx_batch_create");
  std::shared_ptr<BatchHandle> h(new BatchHandle(b, (int)n));
  const double t1 = wallSeconds();
  hxCheck(hx_batch_forward(b, NULL), "hx_batch_forward");
  vguard<double> lp(n, NEG_INF);
  hxCheck(hx_batch_lp_end(b, lp.data()), "hx_batch_lp_end");
  const double t2 = wallSeconds();
  float kms = 0;
  if (hx_batch_last_kernel_ms(b, 0, &kms) == HX_OK) fillTiming.forwardKernel += 1e-3 * kms;
  fillTiming.flattenAndUpload += t1 - t0;
  fillTiming.forwardWait += t2 - t1;
  for (size_t k = 0; k < n; ++k) {
    out[k]->attach(h, (int)k, lp[k]);
    fillTiming.fills += 1;
    fillTiming.cells += (long long)(out[k]->xSize - 1) * (long long)(out[k]->ySize - 1);
  }
  return out;
}

void DPMatrix::fetchPrepared() {
  const size_t C = hmm.components(), A = hmm.alphabetSize();
  vguard<double> sx(xSize * C * A), sy(ySize * C * A);
  hxCheck(hx_batch_read_prepared(batch, jobIndex, sx.data(), sy.data(), insx.data(), rootsubx.data(), insy.data(), rootsuby.data()),
          "hx_batch_read_prepared");
  for (ProfileStateIndex i = 0; i < xSize; ++i)
    if (!subx.state[i].isNull())
      for (size_t c = 0; c < C; ++c)
        for (size_t a = 0; a < A; ++a) subx.state[i].lpAbsorb[c][a] = sx[(i * C + c) * A + a];
  for (ProfileStateIndex j = 0; j < ySize; ++j)
    if (!suby.state[j].isNull())
      for (size_t c = 0; c < C; ++c)
        for (size_t a = 0; a < A; ++a) suby.state[j].lpAbsorb[c][a] = sy[(j * C + c) * A + a];
}

void DPMatrix::ensureHostCells() const {
  if (haveHostCells) return;
  const double t0 = wallSeconds();
  hostCells = g_pinned.take((size_t)matrixDoubles, hostCellsCap);
  hxCheck(hx_batch_read_matrix(batch, jobIndex, which, hostCells), "hx_batch_read_matrix");
  haveHostCells = true;
  fillTiming.readMatrix += wallSeconds() - t0;
  fillTiming.matrixReads += 1;
}

void DPMatrix::prefetchCells(const set<CellCoords>& cells) const {
  vguard<int32_t> ij;
  vguard<std::pair<ProfileStateIndex, ProfileStateIndex> > keys;
  for (const auto& c : cells) {
    if (c.xpos + 1 >= xSize || c.ypos + 1 >= ySize) continue;
    const std::pair<ProfileStateIndex, ProfileStateIndex> key(c.xpos, c.ypos);
    if (sparseCells.count(key) || (!keys.empty() && keys.back() == key)) continue;   // the set is ordered by (x, y, state)
    keys.push_back(key);
    ij.push_back((int32_t)c.xpos);
    ij.push_back((int32_t)c.ypos);
  }
  if (keys.empty()) return;
  const double t0 = wallSeconds();
  vguard<double> vals(5 * keys.size());
  hxCheck(hx_batch_read_cells(batch, jobIndex, which, ij.data(), (int64_t)keys.size(), vals.data()), "hx_batch_read_cells");
  for (size_t k = 0; k < keys.size(); ++k) {
    XYCell xy;
    for (int s = 0; s < PairHMM::TotalStates; ++s) xy.lp[s] = vals[5 * k + s];
    sparseCells[keys[k]] = xy;
  }
  fillTiming.cellGather += wallSeconds() - t0;
  fillTiming.cellGathers += 1;
}

LogProb DPMatrix::cell(ProfileStateIndex xpos, ProfileStateIndex ypos, PairHMM::State state) const {
  if (xpos + 1 >= xSize || ypos + 1 >= ySize || state >= PairHMM::TotalStates) return NEG_INF;
  if (!inEnvelope(xpos, ypos)) return NEG_INF;   // not stored (the batch is created with HX_SPARSE_ENVELOPE)
  if (!haveHostCells && !sparseCells.empty()) {
    const auto it = sparseCells.find(std::make_pair(xpos, ypos));
    if (it != sparseCells.end()) return it->second.lp[state];
  }
  ensureHostCells();
  if (which == 1) {   // the Backward matrix is stored in mirrored coordinates (hx_layout::mirrored)
    xpos = xSize - 2 - xpos;
    ypos = ySize - 2 - ypos;
  }
  const long long l = xpos & 63, t = ypos + l;
  const long long slot = (long long)(xpos >> 6) * stripStride + (t >> 1) * blockStride + (l << 1) + (t & 1);
  return hostCells[(size_t)state * planeStride + slot];
}

DPMatrix::XYCell DPMatrix::xyCell(ProfileStateIndex xpos, ProfileStateIndex ypos) const {
  XYCell c;
  for (int s = 0; s < PairHMM::TotalStates; ++s) c.lp[s] = cell(xpos, ypos, (PairHMM::State)s);
  return c;
}

DPMatrix::random_engine DPMatrix::newRNG() { return random_engine(); }

LogProb DPMatrix::lpCellEmitOrAbsorb(const CellCoords& c) {
  LogProb lp = 0;
  const ProfileState& xState = x.state[c.xpos];
  const ProfileState& yState = y.state[c.ypos];
  switch (c.state) {
    case PairHMM::IMD: if (!xState.isNull()) lp = rootsubx[c.xpos]; break;
    case PairHMM::IIW: if (!xState.isNull()) lp = insx[c.xpos]; break;
    case PairHMM::IDM: if (!yState.isNull()) lp = rootsuby[c.ypos]; break;
    case PairHMM::IMI: if (!yState.isNull()) lp = insy[c.ypos]; break;
    case PairHMM::IMM: if (!xState.isNull() && !yState.isNull()) lp = computeLogProbAbsorb(c.xpos, c.ypos);
    default: break;
  }
  return lp;
}

string DPMatrix::toString(bool edgeOnly) const {
  std::ostringstream out;
  write(out, edgeOnly);
  return out.str();
}

void DPMatrix::write(std::ostream& out, bool edgeOnly) const {
  const auto states = PairHMM::states();
  CellCoords coords;
  for (coords.xpos = 0; coords.xpos < xSize - 1; ++coords.xpos)
    for (coords.ypos = 0; coords.ypos < ySize - 1; ++coords.ypos)
      if (edgeOnly ? atEdge(coords.xpos, coords.ypos) : inEnvelope(coords.xpos, coords.ypos))
        for (auto state : states) {
          coords.state = state;
          out << std::setw(16) << cell(coords) << std::setw(6) << coords.xpos << std::setw(6) << coords.ypos << std::setw(6)
              << PairHMM::stateName(state, coords.xpos == 0, coords.ypos == 0) << std::endl;
        }
}

string DPMatrix::cellName(const CellCoords& c) const {
  return string("(") + hmm.stateName(c.state, c.xpos == 0, c.ypos == 0) + "," + x.state[c.xpos].name + "," + y.state[c.ypos].name + ")";
}

bool DPMatrix::isAbsorbing(const CellCoords& c) const {
  return (c.state == PairHMM::IMM && !x.state[c.xpos].isNull() && !y.state[c.ypos].isNull()) ||
         (c.state == PairHMM::IMD && !x.state[c.xpos].isNull()) || (c.state == PairHMM::IDM && !y.state[c.ypos].isNull());
}

bool DPMatrix::changesX(const CellCoords& c) const {
  return (c.state == PairHMM::IMM && (x.state[c.xpos].isNull() || !y.state[c.ypos].isNull())) || c.state == PairHMM::IMD ||
         c.state == PairHMM::IIW || c.state == PairHMM::EEE;
}

bool DPMatrix::changesY(const CellCoords& c) const {
  return (c.state == PairHMM::IMM && x.state[c.xpos].isEmitOrStart()) || c.state == PairHMM::IDM || c.state == PairHMM::IMI ||
         c.state == PairHMM::EEE;
}

list<DPMatrix::CellCoords> DPMatrix::equivAbsorbCells(const CellCoords& c) const {
  list<CellCoords> eq;
  if (c.state == PairHMM::IIW && !x.state[c.xpos].isNull())
    eq.push_back(CellCoords(c.xpos, c.ypos, PairHMM::IMD));
  else if (c.state == PairHMM::IMI && !y.state[c.ypos].isNull())
    eq.push_back(CellCoords(c.xpos, c.ypos, PairHMM::IDM));
  else if (changesX(c) && x.state[c.xpos].isNull() && x.equivAbsorbState.count(c.xpos))
    eq.push_back(CellCoords(x.equivAbsorbState.at(c.xpos), c.ypos, PairHMM::IMD));
  else if (changesY(c) && y.state[c.ypos].isNull() && y.equivAbsorbState.count(c.ypos))
    eq.push_back(CellCoords(c.xpos, y.equivAbsorbState.at(c.ypos), PairHMM::IDM));
  return eq;
}

// reference src/forward.cpp:225-243
DPMatrix::CellCoords DPMatrix::sampleCell(const map<CellCoords, LogProb>& cellLogProb, random_engine& generator) const {
  double ptot = 0, lpmax = NEG_INF;
  for (auto& iter : cellLogProb) lpmax = std::max(lpmax, iter.second);
  for (auto& iter : cellLogProb) ptot += exp(iter.second - lpmax);
  std::uniform_real_distribution<double> dist(0, ptot);
  const double p0 = dist(generator);
  double p = p0;
  for (auto& iter : cellLogProb)
    if ((p -= exp(iter.second - lpmax)) <= 0) return iter.first;
  for (auto& iter : cellLogProb) std::cerr << "Log P" << cellName(iter.first) << " = " << iter.second << std::endl;
  Abort("%s fail (ptot=%g, p=%g)", __func__, ptot, p0);
  return CellCoords();
}

// reference src/forward.cpp:245-255: strict >, so the first cell in map order wins ties
DPMatrix::CellCoords DPMatrix::bestCell(const map<CellCoords, LogProb>& cellLogProb) {
  CellCoords best;
  double pBest = NEG_INF;
  Assert(!cellLogProb.empty(), "%s traceback failure", __func__);
  for (auto& iter : cellLogProb)
    if (iter.second > pBest) {
      pBest = iter.second;
      best = iter.first;
    }
  return best;
}

// ---- ForwardMatrix ----------------------------------------------------------------------------
ForwardMatrix::EffectiveTransition::EffectiveTransition() : lpPath(NEG_INF), lpBestAlignPath(NEG_INF) {}

ForwardMatrix::ForwardMatrix(const Profile& x, const Profile& y, const PairHMM& hmm, AlignRowIndex parentRowIndex,
                             const GuideAlignmentEnvelope& env, SumProduct* sumProd)
    : DPMatrix(x, y, hmm, env), parentRowIndex(parentRowIndex), sumProd(sumProd) {
  Require(sumProd == NULL, "substitution counts (SumProduct) are outside this build's scope");
  createBatchAndPrepare();
}

ForwardMatrix::ForwardMatrix(const Profile& x, const Profile& y, const PairHMM& hmm, AlignRowIndex parentRowIndex,
                             const GuideAlignmentEnvelope& env, Deferred)
    : DPMatrix(x, y, hmm, env), parentRowIndex(parentRowIndex), sumProd(NULL) {}

ForwardMatrix::Path ForwardMatrix::sampleTrace(random_engine& generator) {
  Assert(lpEnd > NEG_INF, "Forward likelihood is zero; traceback fail");
  Path path;
  path.push_back(endCell);
  map<CellCoords, LogProb> clp = sourceCells(endCell);
  CellCoords current;
  while (true) {
    current = sampleCell(clp, generator);
    path.push_front(current);
    if (current.xpos == 0 && current.ypos == 0) break;
    clp = sourceCells(current);
  }
  return path;
}

ForwardMatrix::Path ForwardMatrix::bestTrace() {
  Assert(lpEnd > NEG_INF, "Forward likelihood is zero; traceback fail");
  if (haveHostCells || !batch || !handle || !deviceTraceback()) return bestTrace(endCell);
  // the matrix is still device-resident: walk it there (one wavefront per job, all jobs of the batch at once)
  BatchHandle& h = *handle;
  if (!h.bestTracesDone) {
    const double t0 = wallSeconds();
    long long cap = 0;
    for (int k = 0; k < h.nJobs; ++k) {
      hx_layout lay;
      hxCheck(hx_batch_layout(h.b, k, 0, &lay), "hx_batch_layout");
      cap = std::max(cap, (long long)lay.n_rows + lay.n_cols + 4);
    }
    h.bestTraceCap = cap;
    h.bestTraceCells.resize(3 * (size_t)cap * h.nJobs);
    h.bestTraceLen.assign(h.nJobs, 0);
    hxCheck(hx_batch_best_trace(h.b, reinterpret_cast<hx_trace_cell*>(h.bestTraceCells.data()), cap, h.bestTraceLen.data()), "hx_batch_best_trace");
    h.bestTracesDone = true;
    fillTiming.deviceTrace += wallSeconds() - t0;
    fillTiming.deviceTraces += 1;
  }
  const int len = h.bestTraceLen[jobIndex];
  Assert(len > 0, "traceback failure");
  Path path;
  const hx_trace_cell* tc = reinterpret_cast<const hx_trace_cell*>(h.bestTraceCells.data()) + (size_t)h.bestTraceCap * jobIndex;
  for (int k = 0; k < len; ++k) path.push_back(CellCoords(tc[k].xpos, tc[k].ypos, (PairHMM::State)tc[k].state));
  return path;
}

ForwardMatrix::Path ForwardMatrix::bestTrace(const CellCoords& end) {
  Path path;
  path.push_back(end);
  if (end.xpos > 0 || end.ypos > 0) {
    map<CellCoords, LogProb> clp = sourceCells(end);
    CellCoords current;
    while (true) {
      current = bestCell(clp);
      path.push_front(current);
      if (current.xpos == 0 && current.ypos == 0) break;
      clp = sourceCells(current);
    }
  }
  return path;
}

AlignPath ForwardMatrix::bestAlignPath() {
  Path trace = bestTrace();
  return traceAlignPath(trace);
}

map<DPMatrix::CellCoords, LogProb> ForwardMatrix::sourceCells(const CellCoords& destCell) {
  map<CellCoords, LogProb> sc = sourceTransitions(destCell);
  for (auto& c_lp : sc) c_lp.second += cell(c_lp.first);
  return sc;
}

map<DPMatrix::CellCoords, LogProb> ForwardMatrix::sourceTransitions(const CellCoords& destCell) {
  auto clp = sourceTransitionsWithoutEmitOrAbsorb(destCell);
  const LogProb lpAbs = lpCellEmitOrAbsorb(destCell);
  for (auto& src_lp : clp) src_lp.second += lpAbs;
  return clp;
}

// reference src/forward.cpp:326-398: the single source of truth for which predecessors exist
map<DPMatrix::CellCoords, LogProb> ForwardMatrix::sourceTransitionsWithoutEmitOrAbsorb(const CellCoords& destCell) {
  map<CellCoords, LogProb> clp;
  const ProfileState& xState = x.state[destCell.xpos];
  const ProfileState& yState = y.state[destCell.ypos];
  switch (destCell.state) {
    case PairHMM::IMD:
    case PairHMM::IIW:
      if (xState.isNull()) {
        if (yState.isReady() || yEmpty)
          if (destCell.xpos < xSize - 1)
            for (auto xt : xState.in) clp[CellCoords(x.trans[xt].src, destCell.ypos, destCell.state)] = x.trans[xt].lpTrans;
      } else if (yState.isReady() || yEmpty)
        for (auto xt : xState.in)
          for (auto s : hmm.sources(destCell.state))
            clp[CellCoords(x.trans[xt].src, destCell.ypos, s)] = hmm.lpTrans(s, destCell.state) + x.trans[xt].lpTrans;
      break;
    case PairHMM::IDM:
    case PairHMM::IMI:
      if (yState.isNull()) {
        if (destCell.ypos < ySize - 1)
          for (auto yt : yState.in) clp[CellCoords(destCell.xpos, y.trans[yt].src, destCell.state)] = y.trans[yt].lpTrans;
      } else if (xState.isReady() || xEmpty)
        for (auto yt : yState.in)
          for (auto s : hmm.sources(destCell.state))
            clp[CellCoords(destCell.xpos, y.trans[yt].src, s)] = hmm.lpTrans(s, destCell.state) + y.trans[yt].lpTrans;
      break;
    case PairHMM::IMM:
      if (yState.isNull() && xState.isEmitOrStart()) {
        if (destCell.ypos < ySize - 1)
          for (auto yt : yState.in) clp[CellCoords(destCell.xpos, y.trans[yt].src, destCell.state)] = y.trans[yt].lpTrans;
      } else if (xState.isNull()) {
        if (yState.isReady() || yEmpty)
          if (destCell.xpos < xSize - 1)
            for (auto xt : xState.in) clp[CellCoords(x.trans[xt].src, destCell.ypos, destCell.state)] = x.trans[xt].lpTrans;
      } else if (!xState.isNull() && !yState.isNull())
        for (auto xt : xState.in)
          for (auto yt : yState.in)
            for (auto s : hmm.sources(destCell.state))
              clp[CellCoords(x.trans[xt].src, y.trans[yt].src, s)] = hmm.lpTrans(s, destCell.state) + x.trans[xt].lpTrans + y.trans[yt].lpTrans;
      break;
    case PairHMM::EEE:
      if (destCell.xpos == xSize - 1 && destCell.ypos == ySize - 1)
        for (auto xt : x.end().in)
          for (auto yt : y.end().in)
            for (auto s : hmm.sources(destCell.state))
              clp[CellCoords(x.trans[xt].src, y.trans[yt].src, s)] = hmm.lpTrans(s, destCell.state) + x.trans[xt].lpTrans + y.trans[yt].lpTrans;
      break;
    default: Abort("%s fail", __func__); break;
  }
  return clp;
}

LogProb ForwardMatrix::eliminatedLogProbInsert(const CellCoords& cell) const {
  switch (cell.state) {
    case PairHMM::IIW: return x.state[cell.xpos].isNull() ? 0 : insx[cell.xpos];
    case PairHMM::IMI: return y.state[cell.ypos].isNull() ? 0 : insy[cell.ypos];
    case PairHMM::IMM:
    case PairHMM::IMD:
    case PairHMM::IDM:
    case PairHMM::EEE: return 0;
    default: Abort("%s fail", __func__); break;
  }
  return NEG_INF;
}

ProfileState::SeqCoords ForwardMatrix::cellSeqCoords(const CellCoords& c) const {
  ProfileState::SeqCoords coords = x.state[c.xpos].seqCoords;
  for (const auto& s_c : y.state[c.ypos].seqCoords) coords[s_c.first] = s_c.second;
  return coords;
}

AlignPath ForwardMatrix::cellAlignPath(const CellCoords& c) const {
  AlignPath alignPath;
  switch (c.state) {
    case PairHMM::IMM:
      if (!x.state[c.xpos].isNull() && !y.state[c.ypos].isNull())
        alignPath = alignPathUnion(x.state[c.xpos].alignPath, y.state[c.ypos].alignPath);
      else if (x.state[c.xpos].isEmitOrStart())
        alignPath = y.state[c.ypos].alignPath;
      else
        alignPath = x.state[c.xpos].alignPath;
      break;
    case PairHMM::IMD:
    case PairHMM::IIW: alignPath = x.state[c.xpos].alignPath; break;
    case PairHMM::IDM:
    case PairHMM::IMI: alignPath = y.state[c.ypos].alignPath; break;
    case PairHMM::EEE: break;
    default: Abort("%s fail", __func__); break;
  }
  if (isAbsorbing(c)) alignPath[parentRowIndex].push_back(true);
  return alignPath;
}

AlignPath ForwardMatrix::transitionAlignPath(const CellCoords& src, const CellCoords& dest) const {
  AlignPath path;
  if (src.xpos != dest.xpos) path = x.getTrans(src.xpos, dest.xpos)->alignPath;
  if (src.ypos != dest.ypos) path = alignPathConcat(path, y.getTrans(src.ypos, dest.ypos)->alignPath);
  return path;
}

AlignPath ForwardMatrix::traceAlignPath(const Path& path) const {
  AlignPath p;
  const vguard<CellCoords> pv(path.begin(), path.end());
  map<AlignRowIndex, SeqIdx> seqCoords;
  for (size_t n = 0; n + 1 < pv.size(); ++n) {
    const AlignPath cap = cellAlignPath(pv[n]), tap = transitionAlignPath(pv[n], pv[n + 1]);
    p = alignPathConcat(p, cap, tap);
    for (const auto& rp : cap) seqCoords[rp.first] += alignPathResiduesInRow(rp.second);
    for (const auto& sc : x.state[pv[n].xpos].seqCoords)
      Assert(seqCoords[sc.first] == sc.second, "Sequence %d: cell x-coord is %d, path x-coord is %d", (int)sc.first, (int)sc.second, (int)seqCoords[sc.first]);
    for (const auto& sc : y.state[pv[n].ypos].seqCoords)
      Assert(seqCoords[sc.first] == sc.second, "Sequence %d: cell y-coord is %d, path y-coord is %d", (int)sc.first, (int)sc.second, (int)seqCoords[sc.first]);
    for (const auto& rp : tap) seqCoords[rp.first] += alignPathResiduesInRow(rp.second);
  }
  p = alignPathConcat(p, cellAlignPath(pv.back()));
  ensureAlignPathHasRow(p, parentRowIndex);
  ensureAlignPathHasRow(p, x.rootRowIndex);
  ensureAlignPathHasRow(p, y.rootRowIndex);
  (void)alignPathColumns(p);
  return p;
}

// reference src/forward.cpp:686-843 (event counts not built: SURVEY.md 8f N3)
Profile ForwardMatrix::makeProfile(const set<CellCoords>& cells, ProfilingStrategy strategy) {
  Profile prof(hmm.components(), alphSize, parentRowIndex);
  prof.name = pairParentName(x.name, hmm.l.t, y.name, hmm.r.t);
  prof.meta["node"] = std::to_string(parentRowIndex);
  Assert(cells.find(startCell) != cells.end(), "Missing SSS");
  Assert(cells.find(endCell) != cells.end(), "Missing EEE");
  if (!haveHostCells && batch) prefetchCells(cells);   // the fwdLogProb annotations below read these cells

  map<CellCoords, ProfileStateIndex> profStateIndex;
  map<CellCoords, int> outgoingTransitionCount;
  for (const auto& dest : cells)
    for (const auto& src_lp : sourceTransitions(dest)) ++outgoingTransitionCount[src_lp.first];

  for (const auto& c : cells)
    if (isAbsorbing(c) || c == startCell || c == endCell || outgoingTransitionCount[c] > 1 || (strategy & KeepGapsOpen) != 0 ||
        (strategy & CollapseChains) == 0) {
      profStateIndex[c] = prof.state.size();
      prof.state.push_back(ProfileState());
      if (isAbsorbing(c)) switch (c.state) {
          case PairHMM::IMM:
            initAbsorbScratch(c.xpos, c.ypos);
            prof.state.back().lpAbsorb = absorbScratch;
            break;
          case PairHMM::IMD: prof.state.back().lpAbsorb = subx.state[c.xpos].lpAbsorb; break;
          case PairHMM::IDM: prof.state.back().lpAbsorb = suby.state[c.ypos].lpAbsorb; break;
          default: break;
        }
      prof.state.back().alignPath = cellAlignPath(c);
      prof.state.back().seqCoords = cellSeqCoords(c);
      prof.state.back().name = cellName(c);
      prof.state.back().meta["fwdLogProb"] = std::to_string(c.state == PairHMM::EEE ? lpEnd : cell(c.xpos, c.ypos, c.state));
    }

  if (strategy & KeepGapsOpen)
    for (const auto& c : cells)
      if (!isAbsorbing(c) && profStateIndex.count(c)) {
        const auto equiv = equivAbsorbCells(c);
        if (equiv.size() && profStateIndex.count(equiv.front())) prof.equivAbsorbState[profStateIndex[c]] = profStateIndex[equiv.front()];
      }

  // effective transitions from cells to retained cells, eliminated cells summed out
  map<CellCoords, map<ProfileStateIndex, EffectiveTransition> > effTrans;
  for (auto iter = cells.crbegin(); iter != cells.crend(); ++iter) {
    const CellCoords& iterCell = *iter;
    const map<CellCoords, LogProb> slp = sourceTransitionsWithoutEmitOrAbsorb(iterCell);
    const LogProb cellLogProbInsert = eliminatedLogProbInsert(iterCell);
    if (profStateIndex.find(iterCell) != profStateIndex.end()) {
      const ProfileStateIndex cellIdx = profStateIndex[iterCell];
      for (const auto& slpIter : slp) {
        const CellCoords& src = slpIter.first;
        EffectiveTransition& eff = effTrans[src][cellIdx];
        eff.lpPath = eff.lpBestAlignPath = slpIter.second + cellLogProbInsert;
        eff.bestAlignPath = transitionAlignPath(src, iterCell);
        ProfileState::assertSeqCoordsConsistent(cellSeqCoords(src), prof.state[cellIdx], eff.bestAlignPath);
      }
    } else {
      const map<ProfileStateIndex, EffectiveTransition> cellEffTrans = effTrans[iterCell];
      const AlignPath cap = cellAlignPath(iterCell);
      for (const auto& slpIter : slp) {
        const CellCoords& src = slpIter.first;
        const LogProb srcCellLogProbTrans = slpIter.second;
        auto& srcEffTrans = effTrans[src];
        for (const auto& cellEffTransIter : cellEffTrans) {
          const ProfileStateIndex destIdx = cellEffTransIter.first;
          const EffectiveTransition& cellDestEffTrans = cellEffTransIter.second;
          EffectiveTransition& srcDestEffTrans = srcEffTrans[destIdx];
          const LogProb lpPath = srcCellLogProbTrans + cellLogProbInsert + cellDestEffTrans.lpPath;
          log_accum_exp(srcDestEffTrans.lpPath, lpPath);
          const LogProb srcDestLogProbBestAlignPath = srcCellLogProbTrans + cellLogProbInsert + cellDestEffTrans.lpBestAlignPath;
          const AlignPath tap = transitionAlignPath(src, iterCell);
          if (srcDestLogProbBestAlignPath > srcDestEffTrans.lpBestAlignPath) {
            srcDestEffTrans.lpBestAlignPath = srcDestLogProbBestAlignPath;
            srcDestEffTrans.bestAlignPath = alignPathConcat(tap, cap, cellDestEffTrans.bestAlignPath);
          }
          ProfileState::assertSeqCoordsConsistent(cellSeqCoords(iterCell), prof.state[destIdx], cellDestEffTrans.bestAlignPath);
          ProfileState::assertSeqCoordsConsistent(cellSeqCoords(src), cellSeqCoords(iterCell), tap, cap);
          ProfileState::assertSeqCoordsConsistent(cellSeqCoords(src), prof.state[destIdx], srcDestEffTrans.bestAlignPath);
        }
      }
    }
  }

  for (const auto& profStateIter : profStateIndex) {
    const CellCoords& cellc = profStateIter.first;
    const ProfileStateIndex srcIdx = profStateIter.second;
    for (const auto& effTransIter : effTrans[cellc]) {
      const ProfileStateIndex destIdx = effTransIter.first;
      const EffectiveTransition& srcDestEffTrans = effTransIter.second;
      const ProfileTransitionIndex transIdx = prof.trans.size();
      ProfileTransition trans;
      trans.src = srcIdx;
      trans.dest = destIdx;
      trans.lpTrans = srcDestEffTrans.lpPath;
      trans.alignPath = srcDestEffTrans.bestAlignPath;
      prof.trans.push_back(trans);
      (prof.state[destIdx].isNull() ? prof.state[srcIdx].nullOut : prof.state[srcIdx].absorbOut).push_back(transIdx);
      prof.state[destIdx].in.push_back(transIdx);
    }
  }

  prof.seq = x.seq;
  prof.seq.insert(y.seq.begin(), y.seq.end());
  prof.assertTransitionsConsistent();
  prof.assertPathToEndExists();
  prof = prof.addReadyStates();
  prof.assertSeqCoordsConsistent();
  return prof;
}

// reference src/forward.cpp:845-889
Profile ForwardMatrix::sampleProfile(random_engine& generator, size_t profileSamples, size_t maxCells, ProfilingStrategy strategy,
                                     size_t minLen, size_t maxLen) {
  map<CellCoords, size_t> cellCount;
  Require((strategy & IncludeBestTrace) || profileSamples > 0, "Must allow at least one sample path in the profile");
  size_t nTraces = 0;
  if (strategy & IncludeBestTrace) {
    const Path best = bestTrace();
    for (auto& c : best) cellCount[c] = 2;
    ++nTraces;
  }
  size_t nAccepted = 0;
  for (size_t n = 0; nAccepted < profileSamples && (maxCells == 0 || cellCount.size() < maxCells); ++n) {
    const Path sampled = sampleTrace(generator);
    size_t ancLen = 0;
    for (auto& c : sampled) switch (c.state) {
        case PairHMM::IMM:
        case PairHMM::IDM:
        case PairHMM::IMD: ++ancLen;
        default: break;
      }
    if (ancLen < minLen || ancLen > maxLen) break;
    for (auto& c : sampled) ++cellCount[c];
    ++nTraces;
    ++nAccepted;
  }
  set<CellCoords> profCells;
  const size_t threshold = (nTraces > 1 && maxCells > 0 && cellCount.size() >= maxCells) ? 2 : 1;
  for (const auto& cc : cellCount)
    if (cc.second >= threshold) profCells.insert(cc.first);
  return makeProfile(profCells, strategy);
}

Profile ForwardMatrix::bestProfile(ProfilingStrategy strategy) {
  const Path best = bestTrace();
  const set<CellCoords> profCells(best.begin(), best.end());
  return makeProfile(profCells, strategy);
}

void ForwardMatrix::slowFillTest() {}

// ---- BackwardMatrix ---------------------------------------------------------------------------
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
  if (!handle->backwardDone) {      // one launch fills the Backward matrices of every job of the batch
    hxCheck(hx_batch_backward(batch, NULL), "hx_batch_backward");
    handle->backwardDone = true;
  }
  vguard<double> lpStarts((size_t)handle->nJobs, NEG_INF);
  hxCheck(hx_batch_lp_start(batch, lpStarts.data()), "hx_batch_lp_start");
  const double lpStartDev = lpStarts[(size_t)jobIndex];
  // |a-b| <= eps * 2^exponent(max(|a|,|b|)): gsl_fcmp (reference src/forward.cpp:1091)
  const double a = lpStartDev, b = fwd.lpEnd;
  int exponent;
  frexp(fabs(a) > fabs(b) ? a : b, &exponent);
  const double delta = ldexp(FWD_BACK_ERROR_TOLERANCE, exponent);
  if (!(fabs(a - b) <= delta)) {
    fwd.slowFillTest();
    slowFillTest();
    sourceDestTransTest();
    Warn("Forward log-likelihood is %g, Backward log-likelihood is %g", fwd.lpEnd, lpStartDev);
  }
}

double BackwardMatrix::cellPostProb(const CellCoords& c) const { return exp(fwd.cell(c) + cell(c) - fwd.lpEnd); }

double BackwardMatrix::transPostProb(const CellCoords& src, const CellCoords& dest) const {
  const auto srcTrans = fwd.sourceTransitions(dest);
  if (srcTrans.find(src) != srcTrans.end()) return exp(fwd.cell(src) + srcTrans.at(src) + cell(dest) - fwd.lpEnd);
  return 0;
}

map<DPMatrix::CellCoords, LogProb> BackwardMatrix::destCells(const CellCoords& srcCell) {
  map<CellCoords, LogProb> clp = destTransitions(srcCell);
  for (auto& c_lp : clp)
    if (c_lp.first.state != PairHMM::EEE) c_lp.second += cell(c_lp.first);
  return clp;
}

// reference src/forward.cpp:1224-1285
map<DPMatrix::CellCoords, LogProb> BackwardMatrix::destTransitions(const CellCoords& srcCell) {
  map<CellCoords, LogProb> clp;
  const ProfileState& xState = x.state[srcCell.xpos];
  const ProfileState& yState = y.state[srcCell.ypos];
  for (auto xt : xState.absorbOut) {
    const ProfileTransition& xTrans = x.trans[xt];
    for (auto yt : yState.absorbOut) {
      const ProfileTransition& yTrans = y.trans[yt];
      clp[CellCoords(xTrans.dest, yTrans.dest, PairHMM::IMM)] = hmm.lpTrans(srcCell.state, PairHMM::IMM) + xTrans.lpTrans + yTrans.lpTrans;
    }
  }
  if (yState.isReady() || yEmpty)
    for (auto xt : xState.absorbOut) {
      const ProfileTransition& xTrans = x.trans[xt];
      clp[CellCoords(xTrans.dest, srcCell.ypos, PairHMM::IMD)] = hmm.lpTrans(srcCell.state, PairHMM::IMD) + xTrans.lpTrans;
      clp[CellCoords(xTrans.dest, srcCell.ypos, PairHMM::IIW)] = hmm.lpTrans(srcCell.state, PairHMM::IIW) + xTrans.lpTrans;
    }
  if (xState.isReady() || xEmpty)
    for (auto yt : yState.absorbOut) {
      const ProfileTransition& yTrans = y.trans[yt];
      clp[CellCoords(srcCell.xpos, yTrans.dest, PairHMM::IDM)] = hmm.lpTrans(srcCell.state, PairHMM::IDM) + yTrans.lpTrans;
      clp[CellCoords(srcCell.xpos, yTrans.dest, PairHMM::IMI)] = hmm.lpTrans(srcCell.state, PairHMM::IMI) + yTrans.lpTrans;
    }
  if ((yState.isReady() || yEmpty) && (srcCell.state == PairHMM::IMD || srcCell.state == PairHMM::IIW || srcCell.state == PairHMM::IMM))
    for (auto xt : xState.nullOut) {
      const ProfileTransition& xTrans = x.trans[xt];
      if (xTrans.dest != xSize - 1) clp[CellCoords(xTrans.dest, srcCell.ypos, srcCell.state)] = xTrans.lpTrans;
    }
  if (srcCell.state == PairHMM::IDM || srcCell.state == PairHMM::IMI || (xState.isEmitOrStart() && srcCell.state == PairHMM::IMM))
    for (auto yt : yState.nullOut) {
      const ProfileTransition& yTrans = y.trans[yt];
      if (yTrans.dest != ySize - 1) clp[CellCoords(srcCell.xpos, yTrans.dest, srcCell.state)] = yTrans.lpTrans;
    }
  for (auto xt : xState.nullOut) {
    const ProfileTransition& xTrans = x.trans[xt];
    if (xTrans.dest == xSize - 1)
      for (auto yt : yState.nullOut) {
        const ProfileTransition& yTrans = y.trans[yt];
        if (yTrans.dest == ySize - 1)
          clp[CellCoords(xTrans.dest, yTrans.dest, PairHMM::EEE)] = xTrans.lpTrans + yTrans.lpTrans + hmm.lpTrans(srcCell.state, PairHMM::EEE);
      }
  }
  for (auto& dest_lp : clp) dest_lp.second += lpCellEmitOrAbsorb(dest_lp.first);
  return clp;
}

BackwardMatrix::Path BackwardMatrix::bestTrace(const CellCoords& traceStart) {
  Path path;
  CellCoords current = traceStart;
  while (current.xpos < xSize - 1 && current.ypos < ySize - 1) {
    map<CellCoords, LogProb> clp = destCells(current);
    current = bestCell(clp);
    path.push_back(current);
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
  hxCheck(hx_batch_posterior_scan(batch, jobIndex, minPostProb, NULL, 0, &n), "hx_batch_posterior_scan");
  vguard<hx_cell> found((size_t)n);
  if (n > 0) {
    int64_t n2 = 0;
    hxCheck(hx_batch_posterior_scan(batch, jobIndex, minPostProb, found.data(), n, &n2), "hx_batch_posterior_scan");
    found.resize((size_t)std::min(n, n2));
  }
  std::sort(found.begin(), found.end(), [](const hx_cell& a, const hx_cell& b) {
    if (a.xpos != b.xpos) return a.xpos > b.xpos;     // i descending, j descending, state ascending
    if (a.ypos != b.ypos) return a.ypos > b.ypos;
    return a.state < b.state;
  });
  for (const auto& c : found) bc.push(CellPostProb(c.xpos, c.ypos, (PairHMM::State)c.state, c.log_post_prob));
  return bc;
}

Profile BackwardMatrix::bestProfile(ProfilingStrategy strategy) {
  set<CellCoords> cells;
  addTrace(endCell, cells, 0, (strategy & KeepGapsOpen) != 0);
  return fwd.makeProfile(cells, strategy);
}

Profile BackwardMatrix::postProbProfile(double minPostProb, size_t maxCells, ProfilingStrategy strategy) {
  std::priority_queue<CellPostProb> bc = cellsAbovePostProbThreshold(minPostProb);
  set<CellCoords> cells;
  if (bc.empty() || (strategy & IncludeBestTrace)) addCells(cells, 0, fwd.bestTrace(), list<CellCoords>(), (strategy & KeepGapsOpen) != 0);
  while ((maxCells == 0 || cells.size() < maxCells) && !bc.empty()) {
    const CellCoords best = bc.top();
    if (cells.count(best))
      bc.pop();
    else if (!addTrace(best, cells, maxCells, (strategy & KeepGapsOpen) != 0))
      break;
  }
  return fwd.makeProfile(cells, strategy);
}

bool BackwardMatrix::addCells(set<CellCoords>& cells, size_t maxCells, const list<CellCoords>& fwdTrace, const list<CellCoords>& backTrace, bool keepGapsOpen) {
  list<CellCoords> newCells;
  for (auto cellIter = fwdTrace.rbegin(); cellIter != fwdTrace.rend(); ++cellIter)
    if (cells.count(*cellIter))
      break;
    else
      newCells.push_back(*cellIter);
  for (const auto& c : backTrace)
    if (cells.count(c))
      break;
    else
      newCells.push_back(c);
  if (maxCells > 0 && cells.size() > 0 && cells.size() + newCells.size() > maxCells) return false;
  cells.insert(newCells.begin(), newCells.end());
  if (keepGapsOpen)
    for (const auto& newCell : newCells) {
      const list<CellCoords> eqvCells = equivAbsorbCells(newCell);
      for (auto& eqvCell : eqvCells)
        if (!cells.count(eqvCell) && cellPostProb(eqvCell) > 0 && inEnvelope(eqvCell.xpos, eqvCell.ypos)) addTrace(eqvCell, cells, maxCells, false);
    }
  return true;
}

bool BackwardMatrix::addTrace(const CellCoords& cell, set<CellCoords>& cells, size_t maxCells, bool keepGapsOpen) {
  const list<CellCoords> fwdTrace = fwd.bestTrace(cell), backTrace = bestTrace(cell);
  return addCells(cells, maxCells, fwdTrace, backTrace, keepGapsOpen);
}

void BackwardMatrix::slowFillTest() {}
void BackwardMatrix::sourceDestTransTest() {}

}  // namespace historian
