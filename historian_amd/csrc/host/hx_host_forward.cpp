// DPMatrix / ForwardMatrix of the host mirror (see hx_host.h), device side: flattening the reference's objects into the
// C ABI's POD images, the fills on the GPU (one job or a batch of jobs), and reading cells back (whole matrices into
// pooled page-locked buffers, or gathers of individual cells).  The O(path) host logic over the filled matrix -
// tracebacks, posterior decoding, profile construction - is hx_host_walk.cpp.
#include "hx_host.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <iomanip>
#include <sstream>
#include <thread>

#include "../../../include/historian_hip.h"

namespace historian {

static const double NEG_INF = -std::numeric_limits<double>::infinity();

static unsigned g_fillMode = HX_LSE_TRUNC;    // the library's default policy; HX_FILL_MODE=exact / setFillMode(HX_LSE_EXACT): bit-identical
static std::mutex g_deviceMutex;                  // device initialisation and the pinned-buffer pool
static bool g_deviceReady[32] = {false};
static bool g_modeRead = false;
static thread_local int t_device = -1;

void DPMatrix::setFillMode(unsigned hxFlags) { g_fillMode = hxFlags & (HX_LSE_TRUNC | HX_LSE_EXACT); }
static int g_deviceTraceback = -1;
void DPMatrix::setDeviceTraceback(bool on) { g_deviceTraceback = on ? 1 : 0; }
bool DPMatrix::deviceTraceback() {
  if (g_deviceTraceback < 0) g_deviceTraceback = getenv("HX_HOST_TRACEBACK") ? 0 : 1;
  return g_deviceTraceback != 0;
}
unsigned DPMatrix::fillMode() { return g_fillMode; }
// HX_TIE_REFILL=1: opt-in, because near ties are common - the walks of 15 of the 31 nodes of a 32-leaf, 1000-residue tree meet one,
// 40 of 63 at 64 leaves x 5000 residues - and a second, exact fill of each of those pairs costs a third to a half of the run
bool ForwardMatrix::refillAtNearTies() {
  static const bool on = getenv("HX_TIE_REFILL") != NULL;
  return on && g_fillMode != HX_LSE_EXACT;
}

static void hxCheck(int rc, const char* what) {
  if (rc != HX_OK) Abort("%s failed (%d): %s", what, rc, hx_last_error());
}

static int threadDevice() {
  if (t_device < 0) {
    const char* dev = getenv("HX_DEVICE");
    t_device = dev ? atoi(dev) : 0;
  }
  return t_device;
}

static void ensureDevice(int ordinal) {
  std::lock_guard<std::mutex> lock(g_deviceMutex);
  Require(ordinal >= 0 && ordinal < 32, "device ordinal %d out of range", ordinal);
  if (!g_modeRead) {
    const char* mode = getenv("HX_FILL_MODE");   // exact | fast | linear | trunc (default): the arithmetic policy of this process's fills
    if (mode && string(mode) == "exact") g_fillMode = HX_LSE_EXACT;
    if (mode && string(mode) == "fast") g_fillMode = HX_LSE_FAST;
    if (mode && string(mode) == "linear") g_fillMode = HX_LSE_LINEAR;
    if (mode && string(mode) == "trunc") g_fillMode = HX_LSE_TRUNC;
    g_modeRead = true;
  }
  if (g_deviceReady[ordinal]) return;
  const double t0 = wallSeconds();
  // the device gets the table the host built with its own libm (reference src/logsumexp.cpp:6-16)
  hxCheck(hx_init(ordinal, logSumExpLookupTable.lookup, HX_LSE_TABLE_ENTRIES), "hx_init");
  g_deviceReady[ordinal] = true;
  fillTiming.deviceInit += wallSeconds() - t0;
}
static void ensureDevice() { ensureDevice(threadDevice()); }

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

// subx / suby: the reference multiplies a copy of the whole profile (src/profile.cpp:78-91 - states with their
// alignment paths, sequence coordinates, names).  All a matrix ever reads of them is lpAbsorb, so here they are
// shells: as many states, emitting where the profile's are, nothing else copied (building and tearing down the
// full copies was 3-4 ms per internal node, tens of thousands of small allocations).
static Profile absorbShell(const Profile& p) {
  Profile s(p.components, p.alphSize, p.rootRowIndex);
  s.name = p.name;
  s.state.resize(p.size());
  for (ProfileStateIndex i = 0; i < p.size(); ++i)
    if (!p.state[i].isNull()) s.state[i].lpAbsorb.assign(p.components, vguard<LogProb>(p.alphSize, NEG_INF));
  return s;
}

DPMatrix::DPMatrix(const Profile& x, const Profile& y, const PairHMM& hmm, const GuideAlignmentEnvelope& env)
    : x(x), y(y), xEmpty(x.isEmpty()), yEmpty(y.isEmpty()), subx(absorbShell(x)), suby(absorbShell(y)), hmm(hmm),
      alphSize((AlphTok)hmm.alphabetSize()), xSize(x.size()), ySize(y.size()),
      startCell(0, 0, PairHMM::SSS), endCell(xSize - 1, ySize - 1, PairHMM::EEE),
      lpEnd(NEG_INF), envelope(env), xClosestLeafPos(xSize, 0), yClosestLeafPos(ySize, 0),
      xNearStart(xSize, false), yNearEnd(ySize, false),
      insx(x.size(), NEG_INF), insy(y.size(), NEG_INF), rootsubx(x.size(), NEG_INF), rootsuby(y.size(), NEG_INF),
      absorbScratch(hmm.components(), vguard<LogProb>(hmm.alphabetSize())),
      batch(NULL), jobIndex(0), which(0), hostCells(NULL), hostCellsCap(0), haveHostCells(false), hostCopyInFlight(false), stripStride(0), planeStride(0), blockStride(128), matrixDoubles(0) {
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
  std::thread warming;                                 // page-locks the first buffers while the first fill runs
  std::mutex warmMutex;
  bool hostReadsExpected = false;                      // warm() was called: the caller walks matrices on the host
  void settle() {
    std::lock_guard<std::mutex> lock(warmMutex);
    if (warming.joinable()) warming.join();
  }
  // Page-locking memory costs ~12 ms per 40 MB buffer: do it on a helper thread, under the first fill, not in front of
  // the first traceback.
  void warm(size_t doubles, int count) {
    settle();
    hostReadsExpected = true;
    std::lock_guard<std::mutex> lock(warmMutex);
    warming = std::thread([this, doubles, count]() {
      for (int k = 0; k < count; ++k) {
        void* p = NULL;
        if (hx_host_alloc(doubles * sizeof(double), &p) != HX_OK) return;
        give(static_cast<double*>(p), doubles);
      }
    });
  }
  ~PinnedPool() { settle(); }
  double* take(size_t n, size_t& cap) {
    settle();
    std::lock_guard<std::mutex> lock(g_deviceMutex);
    size_t best = freeList.size();
    for (size_t k = 0; k < freeList.size(); ++k)
      if (freeList[k].first >= n && (best == freeList.size() || freeList[k].first < freeList[best].first)) best = k;
    if (best < freeList.size()) {
      cap = freeList[best].first;
      double* p = freeList[best].second;
      freeList.erase(freeList.begin() + (long)best);
      return p;
    }
    if (freeList.size() >= 32) {                        // bound the pool: drop the smallest buffer
      size_t small = 0;
      for (size_t k = 1; k < freeList.size(); ++k)
        if (freeList[k].first < freeList[small].first) small = k;
      hx_host_free(freeList[small].second);
      freeList.erase(freeList.begin() + (long)small);
    }
    void* p = NULL;
    cap = n + n / 2;                                    // (profiles grow towards the root: leave room for the next few matrices)
    const double t0 = wallSeconds();
    hxCheck(hx_host_alloc(cap * sizeof(double), &p), "hx_host_alloc");
    fillTiming.pinnedAlloc += wallSeconds() - t0;
    fillTiming.pinnedAllocs += 1;
    return static_cast<double*>(p);
  }
  void give(double* p, size_t cap) {
    std::lock_guard<std::mutex> lock(g_deviceMutex);
    freeList.push_back(std::make_pair(cap, p));
  }
  // makes sure a free buffer of n doubles exists; called while a fill kernel runs, so that page-locking a buffer for a
  // matrix larger than any before it is not paid in front of its traceback
  void reserve(size_t n, int count = 1, bool certain = false) {
    if (!hostReadsExpected && !certain) return;
    settle();
    std::vector<std::pair<size_t, double*> > held;
    for (int k = 0; k < count; ++k) {              // (take() picks the smallest free buffer that fits, or locks a new one)
      size_t cap = 0;
      double* p = take(n, cap);
      held.push_back(std::make_pair(cap, p));
    }
    for (const auto& h : held) give(h.second, h.first);
  }
};
PinnedPool g_pinned;
}  // namespace

DPMatrix::~DPMatrix() {
  if (hostCopyInFlight && batch) (void)hx_batch_wait_read(batch, jobIndex, which);    // the copy writes into hostCells
  if (hostCells) g_pinned.give(hostCells, hostCellsCap);
}

namespace detail {
void warmHostBuffers(size_t doubles, int count) { g_pinned.warm(doubles, count); }
void ensureDevice() { historian::ensureDevice(); }
void ensureDevice(int ordinal) { historian::ensureDevice(ordinal); }
int threadDevice() { return historian::threadDevice(); }
void setThreadDevice(int ordinal) { t_device = ordinal; }
void mergeTiming(const FillTiming& a, FillTiming& b) {
  b.deviceInit += a.deviceInit; b.flattenAndUpload += a.flattenAndUpload; b.forwardWait += a.forwardWait;
  b.forwardKernel += a.forwardKernel; b.backwardWait += a.backwardWait; b.readMatrix += a.readMatrix;
  b.construct += a.construct; b.readPrepared += a.readPrepared;
  b.deviceTrace += a.deviceTrace; b.cellGather += a.cellGather; b.hostTraces += a.hostTraces; b.tieRefill += a.tieRefill; b.tieRefills += a.tieRefills; b.hostMakeProfile += a.hostMakeProfile; b.cellSets += a.cellSets; b.retain += a.retain; b.pinnedAlloc += a.pinnedAlloc; b.pinnedAllocs += a.pinnedAllocs;
  b.fills += a.fills; b.matrixReads += a.matrixReads; b.deviceTraces += a.deviceTraces; b.cellGathers += a.cellGathers; b.cells += a.cells;
}
double* pinnedTake(size_t doubles, size_t& capacity) { return g_pinned.take(doubles, capacity); }
void pinnedReserve(size_t doubles, int count) { g_pinned.reserve(doubles, count, true); }
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
  hxCheck(hx_batch_create_on(threadDevice(), &im.job, 1, g_fillMode | HX_SPARSE_ENVELOPE, &b), "hx_batch_create");
  std::shared_ptr<BatchHandle> h(new BatchHandle(b, 1));
  const double t1 = wallSeconds();
  hxCheck(hx_batch_forward(b, NULL), "hx_batch_forward");
  {
    // while the fill runs: a page-locked buffer for this matrix, if the pool has none large enough
    hx_layout lay;
    hxCheck(hx_batch_layout(b, 0, 0, &lay), "hx_batch_layout");
    g_pinned.reserve((size_t)lay.matrix_doubles);
  }
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

// ForwardMatrix::bestTrace at a near tie: this pair once more, under the exact policy, walked on the device
ForwardMatrix::Path ForwardMatrix::exactBestTrace() {
  ensureDevice();
  const double t0 = wallSeconds();
  JobImage im;
  buildJobImage(*this, xClosestLeafPos, yClosestLeafPos, im);
  hx_batch* b = NULL;
  hxCheck(hx_batch_create_on(threadDevice(), &im.job, 1, HX_LSE_EXACT | HX_SPARSE_ENVELOPE, &b), "hx_batch_create");
  std::shared_ptr<BatchHandle> h(new BatchHandle(b, 1));          // (destroys the batch)
  hxCheck(hx_batch_forward(b, NULL), "hx_batch_forward");
  double lp = NEG_INF;
  hxCheck(hx_batch_lp_end(b, &lp), "hx_batch_lp_end");
  const long long cap = (long long)xSize + ySize + 4;
  vguard<int32_t> cells(3 * (size_t)cap);
  int32_t len = 0;
  hxCheck(hx_batch_best_trace(b, reinterpret_cast<hx_trace_cell*>(cells.data()), cap, &len), "hx_batch_best_trace");
  Assert(len > 0, "traceback failure");
  const hx_trace_cell* tc = reinterpret_cast<const hx_trace_cell*>(cells.data());
  Path path;
  for (int k = 0; k < len; ++k) path.emplace_back(tc[k].xpos, tc[k].ypos, (PairHMM::State)tc[k].state);
  fillTiming.tieRefill += wallSeconds() - t0;
  fillTiming.tieRefills += 1;
  return path;
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

// n independent fills as device batches (not in the reference; see hx_host.h)
vguard<ForwardMatrix*> ForwardMatrix::fillBatch(const vguard<JobSpec>& jobs) {
  return fillBatch(jobs, vguard<int>(1, threadDevice()));
}

vguard<int> lptAssign(const vguard<double>& cost, int nDevices) {
  vguard<size_t> order(cost.size());
  for (size_t k = 0; k < order.size(); ++k) order[k] = k;
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return cost[a] > cost[b]; });
  vguard<double> load((size_t)std::max(nDevices, 1), 0.);
  vguard<int> device(cost.size(), 0);
  for (size_t k : order) {
    const size_t d = (size_t)(std::min_element(load.begin(), load.end()) - load.begin());
    device[k] = (int)d;
    load[d] += cost[k];
  }
  return device;
}

vguard<ForwardMatrix*> ForwardMatrix::fillBatch(const vguard<JobSpec>& jobs, const vguard<int>& devices) {
  vguard<ForwardMatrix*> out;
  if (jobs.empty()) return out;
  Require(!devices.empty(), "fillBatch needs at least one device");
  const double t0 = wallSeconds(), init0 = fillTiming.deviceInit;
  const size_t n = jobs.size();
  vguard<JobImage> images(n);          // sized once: hx_pair_job holds pointers into its elements
  vguard<double> cells(n);
  for (size_t k = 0; k < n; ++k) {
    const JobSpec& js = jobs[k];
    ForwardMatrix* f = new ForwardMatrix(*js.x, *js.y, *js.hmm, js.parentRowIndex, js.env, Deferred());
    out.push_back(f);
    buildJobImage(*f, f->xClosestLeafPos, f->yClosestLeafPos, images[k]);
    cells[k] = (double)(f->xSize - 1) * (double)(f->ySize - 1);
  }
  const double t0b = wallSeconds();
  // deal the jobs to the devices, longest first; one batch per device, every batch launched before any result is awaited
  const vguard<int> dealt = lptAssign(cells, (int)devices.size());
  vguard<std::shared_ptr<BatchHandle> > handles(devices.size());
  for (size_t d = 0; d < devices.size(); ++d) {
    vguard<hx_pair_job> pods;
    vguard<int> jobOf;
    for (size_t k = 0; k < n; ++k)
      if ((size_t)dealt[k] == d) { pods.push_back(images[k].job); jobOf.push_back((int)k); }
    if (pods.empty()) continue;
    ensureDevice(devices[d]);
    hx_batch* b = NULL;
    hxCheck(hx_batch_create_on(devices[d], pods.data(), (int32_t)pods.size(), g_fillMode | HX_SPARSE_ENVELOPE, &b), "hx_batch_create");
    handles[d].reset(new BatchHandle(b, (int)pods.size()));
    handles[d]->jobOf = jobOf;
  }
  const double t1 = wallSeconds();
  if (getenv("HX_TIMING_LEVELS"))
    fprintf(stderr, "timing: fill batch of %zu jobs: matrices and POD images %.4f s, hx_batch_create %.4f s\n", n, t0b - t0, t1 - t0b);
  for (auto& h : handles)
    if (h) hxCheck(hx_batch_forward(h->b, NULL), "hx_batch_forward");
  // while the fills run: a page-locked buffer for the largest matrix of the batch, if the pool has none
  {
    size_t largest = 0;
    for (auto& h : handles) {
      if (!h) continue;
      for (int j = 0; j < h->nJobs; ++j) {
        hx_layout lay;
        hxCheck(hx_batch_layout(h->b, j, 0, &lay), "hx_batch_layout");
        largest = std::max(largest, (size_t)lay.matrix_doubles);
      }
    }
    if (largest) g_pinned.reserve(largest);
  }
  for (auto& h : handles) {
    if (!h) continue;
    vguard<double> lp((size_t)h->nJobs, NEG_INF);
    hxCheck(hx_batch_lp_end(h->b, lp.data()), "hx_batch_lp_end");
    float kms = 0;
    if (hx_batch_last_kernel_ms(h->b, 0, &kms) == HX_OK) fillTiming.forwardKernel += 1e-3 * kms;
    for (int j = 0; j < h->nJobs; ++j) {
      ForwardMatrix* f = out[(size_t)h->jobOf[(size_t)j]];
      f->attach(h, j, lp[(size_t)j]);
      fillTiming.fills += 1;
      fillTiming.cells += (long long)(f->xSize - 1) * (long long)(f->ySize - 1);
    }
  }
  const double t2 = wallSeconds();
  fillTiming.flattenAndUpload += (t1 - t0) - (fillTiming.deviceInit - init0);     // (the one-off hx_init is accounted for on its own)
  fillTiming.forwardWait += t2 - t1;
  return out;
}

void DPMatrix::fetchPrepared() {
  const double tPrep = wallSeconds();
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
  fillTiming.readPrepared += wallSeconds() - tPrep;
}

void DPMatrix::startHostCopy() const {
  if (haveHostCells || hostCopyInFlight || !batch) return;
  const double t0 = wallSeconds();
  hostCells = g_pinned.take((size_t)matrixDoubles, hostCellsCap);
  hxCheck(hx_batch_read_matrix_async(batch, jobIndex, which, hostCells), "hx_batch_read_matrix_async");
  hostCopyInFlight = true;
  fillTiming.readMatrix += wallSeconds() - t0;
}

void DPMatrix::ensureHostCells() const {
  if (haveHostCells) return;
  const double t0 = wallSeconds();
  if (hostCopyInFlight) {
    hxCheck(hx_batch_wait_read(batch, jobIndex, which), "hx_batch_wait_read");
    hostCopyInFlight = false;
  } else {
    hostCells = g_pinned.take((size_t)matrixDoubles, hostCellsCap);
    hxCheck(hx_batch_read_matrix(batch, jobIndex, which, hostCells), "hx_batch_read_matrix");
  }
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

// Keeps the values of `cells` (all five states of each (x, y)) and lets the page-locked copy of the matrix go: what
// makeProfile reads afterwards is in sparseCells, so it needs neither the buffer nor the device.
void DPMatrix::retainCells(const set<CellCoords>& cells) const {
  if (!haveHostCells) return;
  const double t0 = wallSeconds();
  std::pair<ProfileStateIndex, ProfileStateIndex> last((ProfileStateIndex)-1, (ProfileStateIndex)-1);
  for (const auto& c : cells) {
    if (c.xpos + 1 >= xSize || c.ypos + 1 >= ySize) continue;
    const std::pair<ProfileStateIndex, ProfileStateIndex> key(c.xpos, c.ypos);
    if (key == last) continue;                     // the set is ordered by (x, y, state)
    last = key;
    sparseCells[key] = xyCell(c.xpos, c.ypos);
  }
  g_pinned.give(hostCells, hostCellsCap);
  hostCells = NULL;
  hostCellsCap = 0;
  haveHostCells = false;
  fillTiming.retain += wallSeconds() - t0;
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

// ---- ForwardMatrix: construction = the device fill (traceback, profiles, posteriors: hx_host_walk.cpp) ----------
ForwardMatrix::ForwardMatrix(const Profile& x, const Profile& y, const PairHMM& hmm, AlignRowIndex parentRowIndex,
                             const GuideAlignmentEnvelope& env, SumProduct* sumProd)
    : DPMatrix(x, y, hmm, env), sumProd(sumProd), parentRowIndex(parentRowIndex) {
  Require(sumProd == NULL, "substitution counts (SumProduct) are outside this build's scope");
  createBatchAndPrepare();
}

ForwardMatrix::ForwardMatrix(const Profile& x, const Profile& y, const PairHMM& hmm, AlignRowIndex parentRowIndex,
                             const GuideAlignmentEnvelope& env, Deferred)
    : DPMatrix(x, y, hmm, env), sumProd(NULL), parentRowIndex(parentRowIndex) {}

}  // namespace historian
