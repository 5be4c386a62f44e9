// C-ABI implementation (include/historian_hip.h): host-side flattening, device
// memory layout and kernel launches for batches of independent pair DPs.
#include <hip/hip_runtime.h>

#include <cmath>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <unordered_map>
#include <mutex>
#include <vector>

#include "../../include/historian_hip.h"
#include "hx_device.h"
#include "hx_kernels.h"

using namespace hx;

// The lone-pair launches (several workgroups per pair, hand-off through memory: hx_dag.hip, hx_daglin.hip) need every
// workgroup of the launch resident at once - a strip's wave polls the wave of the strip above.  Up to eight pairs, and never
// more workgroups than half the chip's CUs (one such workgroup's LDS fills a CU), so that a second stream cannot starve them.
#define HX_MULTI_MAX_PAIRS 8
#define HX_MULTI_MAX_GROUPS 128
// progress counters (256 ints per pair): the leaf-pair launch of hx_chain.hip takes up to 128 pairs; Forward fills use the
// first half of the buffer, Backward fills the second (a caller may run the two fills on two streams)
#define HX_MULTI_COUNTER_PAIRS 128

namespace {

thread_local char g_err[512] = "";
// Constant tables, one set per device hx_init was called for (the rate-model-independent part of the constant block of
// SURVEY 8e).  A batch is bound to one device at creation; a process may drive several devices (hx_batch_create_on),
// from one host thread or from one thread per device.
#define HX_MAX_DEVICES 16
struct DeviceTables {
  double* tab = nullptr;      // device copy of the host-built log_sum_exp table (8-byte entries)
  double* fast_tab = nullptr; // quadratic pieces of the fast fill mode (FastPiece, 16 bytes each)
  double* log_tab = nullptr;  // {c, -log c} entries of the scaled-probability fills' logarithm (hx_linear.hip)
  double* pair_tab = nullptr; // {lookup[n], lookup[n+1]-lookup[n]} pairs, 16-byte aligned (exact fill mode)
  // the stream of hx_batch_read_matrix_async, shared by the device's batches: a stream's queue is set up at its first use,
  // which costs a caller that makes a batch per fill (a reconstruction's internal nodes) a millisecond or two per matrix
  hipStream_t copy_stream = nullptr;
  std::mutex copy_mutex;
  // side stream of the fills: the two-pairs-per-wavefront banded sweep's edge kernel runs beside the sweep (hx_batch_forward)
  hipStream_t side_stream = nullptr;
  bool ready = false;
};
DeviceTables g_dev[HX_MAX_DEVICES];
int g_device = -1;            // the default device: the one hx_init was last called for

// Quadratic pieces of T(d) = log(1 + exp(-d)) on [0,10): piece k covers [k h, (k+1) h),
// h = 10/N, as c0 + c1 t + c2 t^2 with t = (d - k h)/h, interpolating T at the three Chebyshev
// nodes of the piece (long double arithmetic; |error| < 1e-11).  c1 and c2 are stored as fp32
// (their terms are < 1.3e-3 and < 8e-7), c0 absorbs the mean of their rounding; 16 bytes per
// piece = one ds_read_b128.  One extra all-zero piece implements T = 0 for d >= 10.
struct FastPieceHost { double c0; float c1, c2; };
void build_fast_table(std::vector<double>& out) {
  const int N = HX_FAST_INTERVALS;
  static_assert(sizeof(FastPieceHost) == 16, "fast table piece layout");
  out.assign((size_t)(N + 1) * 2, 0.0);
  FastPieceHost* tabp = reinterpret_cast<FastPieceHost*>(out.data());
  const long double h = 10.0L / N;
  const long double pi = 3.14159265358979323846264338327950288L;
  for (int k = 0; k < N; ++k) {
    long double t[3], y[3];
    for (int m = 0; m < 3; ++m) {
      t[m] = 0.5L + 0.5L * cosl((2 * m + 1) * pi / 6);
      y[m] = log1pl(expl(-(k + t[m]) * h));
    }
    const long double d01 = (y[1] - y[0]) / (t[1] - t[0]);
    const long double d12 = (y[2] - y[1]) / (t[2] - t[1]);
    const long double c2 = (d12 - d01) / (t[2] - t[0]);
    const long double c1 = d01 - c2 * (t[0] + t[1]);
    const long double c0 = y[0] - t[0] * (c1 + c2 * t[0]);
    const float c1f = (float)c1, c2f = (float)c2;
    tabp[k].c1 = c1f;
    tabp[k].c2 = c2f;
    tabp[k].c0 = (double)(c0 + 0.5L * (c1 - (long double)c1f) + (c2 - (long double)c2f) / 3.0L);
  }
  tabp[N].c0 = 0.0; tabp[N].c1 = 0.0f; tabp[N].c2 = 0.0f;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                 \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess)                                                             \
      return fail(e_ == hipErrorOutOfMemory ? HX_ERR_OUT_OF_MEMORY : HX_ERR_HIP,      \
                  "%s failed: %s", #expr, hipGetErrorString(e_));                     \
  } while (0)

constexpr double NEG_INF = -std::numeric_limits<double>::infinity();

// Grows a host staging image of the device arena; returns byte offsets.
struct Arena {
  std::vector<char> host;
  size_t put(const void* src, size_t bytes) {
    size_t off = (host.size() + 255) & ~size_t(255);
    host.resize(off + bytes);
    if (src && bytes) memcpy(host.data() + off, src, bytes);
    return off;
  }
  size_t reserve(size_t bytes) { return put(nullptr, bytes); }
};

// offsets into the arena for one profile
struct ProfOff {
  size_t flags, in_off, in_src, in_lp, ao_off, ao_dst, ao_lp, no_off, no_dst, no_lp;
  size_t lp_absorb, sub, ins, rootsub, env, cls, cls_rep, pack, ecls, subc, insc, rootsubc, fpack;
  bool has_env;
  int n, empty, n_cls, chain, interior_emit, lp_zero;
};

int check_csr(const int32_t* off, const int32_t* idx, int N, int T, const char* what) {
  if (!off) return fail(HX_ERR_INVALID_ARG, "%s offsets are null", what);
  if (off[0] != 0) return fail(HX_ERR_INVALID_ARG, "%s offsets do not start at 0", what);
  for (int i = 0; i < N; ++i)
    if (off[i + 1] < off[i]) return fail(HX_ERR_INVALID_ARG, "%s offsets decrease at state %d", what, i);
  if (off[N] > 0 && !idx) return fail(HX_ERR_INVALID_ARG, "%s index list is null", what);
  for (int k = 0; k < off[N]; ++k)
    if (idx[k] < 0 || idx[k] >= T) return fail(HX_ERR_RANGE, "%s transition index %d out of range", what, idx[k]);
  return HX_OK;
}

int flatten_profile(const hx_profile* p, int CA, bool need_env, bool is_y, Arena& ar, ProfOff& o) {
  if (!p) return fail(HX_ERR_INVALID_ARG, "null profile");
  const int N = p->n_states, T = p->n_trans;
  if (N < 2 || T < 0) return fail(HX_ERR_INVALID_ARG, "profile needs >= 2 states (got %d)", N);
  if (!p->is_null || (T > 0 && (!p->trans_src || !p->trans_dst || !p->trans_lp)))
    return fail(HX_ERR_INVALID_ARG, "null profile arrays");
  if (!p->is_null[0] || !p->is_null[N - 1])
    return fail(HX_ERR_INVALID_ARG, "START and END must be null states");
  int rc;
  if ((rc = check_csr(p->in_off, p->in_idx, N, T, "in")) != HX_OK) return rc;
  if ((rc = check_csr(p->aout_off, p->aout_idx, N, T, "absorbOut")) != HX_OK) return rc;
  if ((rc = check_csr(p->nout_off, p->nout_idx, N, T, "nullOut")) != HX_OK) return rc;
  for (int t = 0; t < T; ++t) {
    if (p->trans_src[t] < 0 || p->trans_dst[t] >= N) return fail(HX_ERR_RANGE, "transition %d out of range", t);
    if (p->trans_src[t] >= p->trans_dst[t])
      return fail(HX_ERR_NOT_TOPOSORTED, "transition #%d from %d -> %d is not toposorted", t, p->trans_src[t], p->trans_dst[t]);
  }
  bool any_emit = false;
  for (int i = 0; i < N; ++i) any_emit |= !p->is_null[i];
  if (any_emit && !p->lp_absorb) return fail(HX_ERR_INVALID_ARG, "lp_absorb is null");
  if (need_env && !p->env_pos) return fail(HX_ERR_INVALID_ARG, "env_pos is null but max_distance >= 0");

  std::vector<int32_t> in_src, ao_dst, no_dst;
  std::vector<double> in_lp, ao_lp, no_lp;
  for (int k = 0; k < p->in_off[N]; ++k) {
    const int t = p->in_idx[k];
    in_src.push_back(p->trans_src[t]);
    in_lp.push_back(p->trans_lp[t]);
  }
  for (int i = 0; i < N; ++i) {
    for (int k = p->in_off[i]; k < p->in_off[i + 1]; ++k)
      if (p->trans_dst[p->in_idx[k]] != i) return fail(HX_ERR_INVALID_ARG, "incoming transition destination doesn't match state %d", i);
    for (int k = p->aout_off[i]; k < p->aout_off[i + 1]; ++k) {
      const int t = p->aout_idx[k];
      if (p->trans_src[t] != i) return fail(HX_ERR_INVALID_ARG, "absorbing transition source doesn't match state %d", i);
    }
    for (int k = p->nout_off[i]; k < p->nout_off[i + 1]; ++k) {
      const int t = p->nout_idx[k];
      if (p->trans_src[t] != i) return fail(HX_ERR_INVALID_ARG, "null transition source doesn't match state %d", i);
    }
  }
  for (int k = 0; k < p->aout_off[N]; ++k) { ao_dst.push_back(p->trans_dst[p->aout_idx[k]]); ao_lp.push_back(p->trans_lp[p->aout_idx[k]]); }
  for (int k = 0; k < p->nout_off[N]; ++k) { no_dst.push_back(p->trans_dst[p->nout_idx[k]]); no_lp.push_back(p->trans_lp[p->nout_idx[k]]); }

  // flags (reference src/profile.h:32-37, src/forward.cpp:58-65)
  std::vector<uint8_t> flags(N, 0);
  for (int i = 0; i < N; ++i) {
    const bool isnull = p->is_null[i] != 0;
    if (isnull) flags[i] |= F_NULL;
    if (p->nout_off[i + 1] == p->nout_off[i]) flags[i] |= F_READY;
    if (!isnull || p->in_off[i + 1] == p->in_off[i]) flags[i] |= F_EMIT_OR_START;
  }
  for (int k = p->in_off[N - 1]; k < p->in_off[N]; ++k) flags[in_src[k]] |= F_TO_END;
  if (!is_y) {
    std::vector<char> near(N, 0);
    near[0] = 1;
    for (int i = 0; i < N; ++i)
      if (near[i])
        for (int k = p->nout_off[i]; k < p->nout_off[i + 1]; ++k) near[no_dst[k]] = 1;
    for (int i = 0; i < N; ++i)
      if (near[i]) flags[i] |= F_EDGE;
  } else {
    for (int k = p->in_off[N - 1]; k < p->in_off[N]; ++k) flags[in_src[k]] |= F_EDGE;
  }

  // emission classes: states with byte-identical lpAbsorb rows share one class
  std::vector<int32_t> cls(N, -1), cls_rep;
  {
    std::unordered_map<std::string, int> seen;
    for (int i = 0; i < N; ++i) {
      if (p->is_null[i]) continue;
      std::string key(reinterpret_cast<const char*>(p->lp_absorb + (size_t)i * CA), (size_t)CA * sizeof(double));
      auto it = seen.find(key);
      if (it == seen.end()) {
        it = seen.emplace(std::move(key), (int)cls_rep.size()).first;
        cls_rep.push_back(i);
      }
      cls[i] = it->second;
    }
  }
  int chain = 1;
  for (int i = 1; i < N && chain; ++i)
    if (p->in_off[i + 1] - p->in_off[i] != 1 || in_src[p->in_off[i]] != i - 1) chain = 0;
  if (p->in_off[1] != 0) chain = 0;

  o.n = N;
  o.empty = any_emit ? 0 : 1;
  o.n_cls = (int)cls_rep.size();
  o.chain = chain;
  o.flags = ar.put(flags.data(), N);
  o.in_off = ar.put(p->in_off, sizeof(int32_t) * (N + 1));
  o.in_src = ar.put(in_src.data(), sizeof(int32_t) * in_src.size());
  o.in_lp = ar.put(in_lp.data(), sizeof(double) * in_lp.size());
  o.ao_off = ar.put(p->aout_off, sizeof(int32_t) * (N + 1));
  o.ao_dst = ar.put(ao_dst.data(), sizeof(int32_t) * ao_dst.size());
  o.ao_lp = ar.put(ao_lp.data(), sizeof(double) * ao_lp.size());
  o.no_off = ar.put(p->nout_off, sizeof(int32_t) * (N + 1));
  o.no_dst = ar.put(no_dst.data(), sizeof(int32_t) * no_dst.size());
  o.no_lp = ar.put(no_lp.data(), sizeof(double) * no_lp.size());
  {
    // null rows are never read by the reference; give them a defined value
    std::vector<double> raw((size_t)N * CA, NEG_INF);
    for (int i = 0; i < N; ++i)
      if (!p->is_null[i]) memcpy(&raw[(size_t)i * CA], p->lp_absorb + (size_t)i * CA, sizeof(double) * CA);
    o.lp_absorb = ar.put(raw.data(), sizeof(double) * raw.size());
  }
  o.sub = ar.reserve(sizeof(double) * (size_t)N * CA);
  o.ins = ar.reserve(sizeof(double) * N);
  o.rootsub = ar.reserve(sizeof(double) * N);
  o.has_env = need_env;
  o.env = need_env ? ar.put(p->env_pos, sizeof(int32_t) * N) : 0;
  o.cls = ar.put(cls.data(), sizeof(int32_t) * N);
  o.cls_rep = ar.put(cls_rep.data(), sizeof(int32_t) * cls_rep.size());
  {
    std::vector<int32_t> ecls(N);
    for (int i = 0; i < N; ++i) ecls[i] = cls[i] < 0 ? (int32_t)cls_rep.size() : cls[i];
    o.ecls = ar.put(ecls.data(), sizeof(int32_t) * N);
  }
  o.pack = ar.reserve(sizeof(double) * 4 * (size_t)N);
  o.fpack = ar.reserve(sizeof(FwdPack) * (size_t)N);
  o.subc = ar.reserve(sizeof(double) * cls_rep.size() * (size_t)CA);
  o.insc = ar.reserve(sizeof(double) * cls_rep.size());
  o.rootsubc = ar.reserve(sizeof(double) * cls_rep.size());
  o.lp_zero = 1;
  for (int t = 0; t < T; ++t)
    if (p->trans_lp[t] != 0.0) o.lp_zero = 0;
  o.interior_emit = 1;
  for (int i = 1; i < N - 1; ++i)
    if (p->is_null[i]) o.interior_emit = 0;
  return HX_OK;
}

void bind_profile(DevProfile& d, const ProfOff& o, char* base) {
  d.n = o.n;
  d.empty = o.empty;
  d.flags = reinterpret_cast<uint8_t*>(base + o.flags);
  d.in_off = reinterpret_cast<int32_t*>(base + o.in_off);
  d.in_src = reinterpret_cast<int32_t*>(base + o.in_src);
  d.in_lp = reinterpret_cast<double*>(base + o.in_lp);
  d.ao_off = reinterpret_cast<int32_t*>(base + o.ao_off);
  d.ao_dst = reinterpret_cast<int32_t*>(base + o.ao_dst);
  d.ao_lp = reinterpret_cast<double*>(base + o.ao_lp);
  d.no_off = reinterpret_cast<int32_t*>(base + o.no_off);
  d.no_dst = reinterpret_cast<int32_t*>(base + o.no_dst);
  d.no_lp = reinterpret_cast<double*>(base + o.no_lp);
  d.lp_absorb = reinterpret_cast<double*>(base + o.lp_absorb);
  d.sub = reinterpret_cast<double*>(base + o.sub);
  d.ins = reinterpret_cast<double*>(base + o.ins);
  d.rootsub = reinterpret_cast<double*>(base + o.rootsub);
  d.env = o.has_env ? reinterpret_cast<int32_t*>(base + o.env) : nullptr;
  d.cls = reinterpret_cast<int32_t*>(base + o.cls);
  d.cls_rep = reinterpret_cast<int32_t*>(base + o.cls_rep);
  d.n_cls = o.n_cls;
  d.pad_ = 0;
  d.pack = reinterpret_cast<double*>(base + o.pack);
  d.fpack = reinterpret_cast<FwdPack*>(base + o.fpack);
  d.subc = reinterpret_cast<double*>(base + o.subc);
  d.insc = reinterpret_cast<double*>(base + o.insc);
  d.rootsubc = reinterpret_cast<double*>(base + o.rootsubc);
  d.ecls = reinterpret_cast<int32_t*>(base + o.ecls);
}

struct JobOff {
  ProfOff x, y;
  size_t log_root, log_sub_l, log_sub_r, log_ins_l, log_ins_r, log_cptw_l, log_cptw_r, emis, emis_pad;
  size_t fwd_windows, bwd_windows, strip_base, yword, yword_bwd, band_rows, band_rows_bwd;
  bool compressed;
  int64_t compact_plane;
  int64_t eplane_off;     // into hx_batch::d_eplane, or -1
  bool table_emission;
};

// Step windows of the general-profile strip pipeline (hx_dag.hip): for every 64-row strip of the sweep
// (mirrored for Backward) up to two half-open ranges of steps t = column + row-in-strip that together
// contain all of the strip's in-envelope cells (reference src/forward.h:92-98: a cell is in the envelope
// when it is at an edge or within max_distance of the guide alignment).  Supersets are harmless.
std::vector<int32_t> strip_windows(const uint8_t* xf, const int32_t* xenv, const uint8_t* yf, const int32_t* yenv,
                                   int R, int Cc, int band, bool mirrored) {
  const int n_strips = (R + HX_STRIP - 1) / HX_STRIP;
  const int nsteps = Cc + HX_STRIP - 1;
  std::vector<int32_t> w(4 * (size_t)n_strips, 0);
  int V = 0;
  for (int j = 0; j < Cc; ++j) V = std::max(V, (int)yenv[j]);
  for (int i = 0; i < R; ++i) V = std::max(V, (int)xenv[i]);
  const bool cheap = (int64_t)R * (2 * (int64_t)band + 1) <= (int64_t)1 << 26;
  std::vector<int> minj(V + 1, INT_MAX), maxj(V + 1, -1);
  for (int j = 0; j < Cc; ++j) {
    const int v = yenv[j] < 0 ? 0 : yenv[j];
    minj[v] = std::min(minj[v], j);
    maxj[v] = std::max(maxj[v], j);
  }
  int ejmin = INT_MAX, ejmax = -1;
  for (int j = 0; j < Cc; ++j)
    if (yf[j] & F_EDGE) { ejmin = std::min(ejmin, j); ejmax = std::max(ejmax, j); }
  for (int s = 0; s < n_strips; ++s) {
    int lo = INT_MAX, hi = -1;            // band window (inclusive steps)
    bool whole = !cheap;
    const int rows = std::min(HX_STRIP, R - s * HX_STRIP);
    for (int l = 0; l < rows && !whole; ++l) {
      const int im = s * HX_STRIP + l, i = mirrored ? R - 1 - im : im;
      if (xf[i] & F_EDGE) { whole = true; break; }
      int jmin = INT_MAX, jmax = -1;
      const int xe = xenv[i] < 0 ? 0 : xenv[i];
      for (int v = std::max(0, xe - band); v <= std::min(V, xe + band); ++v) {
        jmin = std::min(jmin, minj[v]);
        jmax = std::max(jmax, maxj[v]);
      }
      if (jmax < 0) continue;
      const int a = mirrored ? Cc - 1 - jmax : jmin, b = mirrored ? Cc - 1 - jmin : jmax;
      lo = std::min(lo, a + l);
      hi = std::max(hi, b + l);
    }
    int32_t* o = &w[4 * (size_t)s];
    if (whole) { o[0] = 0; o[1] = nsteps; continue; }
    int elo = INT_MAX, ehi = -1;          // edge-column window
    if (ejmax >= 0) {
      const int a = mirrored ? Cc - 1 - ejmax : ejmin, b = mirrored ? Cc - 1 - ejmin : ejmax;
      elo = a; ehi = b + rows - 1;
    }
    if (hi < 0 && ehi < 0) continue;      // nothing in the envelope: both windows empty
    if (hi < 0) { o[0] = elo; o[1] = ehi + 1; continue; }
    if (ehi < 0) { o[0] = lo; o[1] = hi + 1; continue; }
    if (elo <= hi + 1 && lo <= ehi + 1) { o[0] = std::min(lo, elo); o[1] = std::max(hi, ehi) + 1; continue; }
    if (lo < elo) { o[0] = lo; o[1] = hi + 1; o[2] = elo; o[3] = ehi + 1; }
    else { o[0] = elo; o[1] = ehi + 1; o[2] = lo; o[3] = hi + 1; }
  }
  return w;
}


// Rows of the banded rotating-row sweep (hx_band.hip): for every row of a leaf-like pair the anti-diagonal steps
// k = i + j it owns - its in-envelope span of columns (reference src/forward.h:92-98), widened to whole step pairs -
// and the base of its cells in a state plane.  Returns false when the pair does not satisfy what that kernel assumes:
//   * the envelope coordinates of both profiles are non-decreasing (true for leaf profiles under any guide alignment:
//     cumulative match counts, reference src/alignpath.cpp:282-310), so a row's band is one span and spans move right;
//   * the only always-in-envelope row is x START and the only such column the y state that feeds END (leaf profiles);
//   * a lane that finishes row i is free before row i + 63's span opens (rows i .. i + 63 are never alive together).
// Layout of a record: see BandRow in hx_band.hip.
bool build_band_rows(const int32_t* xenv, const int32_t* yenv, const uint8_t* xf, const uint8_t* yf, const int32_t* xecls,
                     bool x_empty, int R, int Cc, int band, int64_t ss, int blk, const int32_t* cwin, const int64_t* cbase,
                     std::vector<int32_t>& out, int& n_steps) {
  if (R < 2 || Cc < 2 || R + Cc > 60000) return false;
  for (int i = 1; i <= R; ++i) if (xenv[i] < xenv[i - 1]) return false;
  for (int j = 1; j <= Cc; ++j) if (yenv[j] < yenv[j - 1]) return false;
  for (int i = 0; i < R; ++i) if (((xf[i] & F_EDGE) != 0) != (i == 0)) return false;
  for (int j = 0; j < Cc; ++j) if (((yf[j] & F_EDGE) != 0) != (j == Cc - 1)) return false;
  std::vector<int> lo(R), hi(R);
  {
    int a = 0, b = -1;                 // two pointers over the columns: first with yenv >= xenv - band, last with yenv <= xenv + band
    for (int i = 0; i < R; ++i) {
      while (a < Cc && (int64_t)yenv[a] < (int64_t)xenv[i] - band) ++a;
      while (b + 1 < Cc && (int64_t)yenv[b + 1] <= (int64_t)xenv[i] + band) ++b;
      if (b < a) return false;         // an empty band row: not a guide-alignment envelope
      lo[i] = a; hi[i] = b;
    }
  }
  lo[0] = 0;
  hi[0] = std::max(hi[0], hi[1]);      // row 1 reads row 0 up to its own last column (row 0 is in the envelope throughout)
  for (int i = 0; i < R; ++i) if (hi[i] >= Cc - 2) hi[i] = Cc - 1;   // the cells next to the band in the always-in column
  const int n_strips = (R + HX_STRIP - 1) / HX_STRIP;
  out.assign(2 * (size_t)(R + 64) + (size_t)n_strips + 4, 0);
  int32_t* strip_store = &out[2 * (size_t)(R + 64)];
  std::vector<int> os(R), oe(R), as_(R), ae_(R);
  std::vector<char> have(n_strips, 0);
  n_steps = 0;
  for (int i = 0; i < R; ++i) {
    as_[i] = i + lo[i]; ae_[i] = i + hi[i];
    os[i] = as_[i] & ~1; oe[i] = ae_[i] | 1;
  }
  // Whole cache lines: a step pair of a state plane holds rows 4g .. 4g + 3 of a strip in one 64-byte line, and a line that
  // the sweep writes in part costs a read-modify-write (a build that simply dropped the partly written lines at both ends of
  // a store ran a quarter faster).  So the four rows of such a group own the same steps - from the first row's first to the
  // last row's last: the cells a row gains lie outside the envelope (the sweep computes them as zero cells and stores -inf,
  // which is what they hold in pre-filled planes and "anything" in the others), except that nothing is gained in the
  // always-in-envelope column: a row whose band does not reach it stops one column short (cell (1, Ny-2) there is the edge
  // kernel's, the others are its -inf).  A group whose spread does not fit the record's three-bit pads keeps its own spans.
  if (!getenv("HX_BAND_NO_LINE_GROUPS"))
    for (int g = 0; g < R; g += 4) {
      const int ge = std::min(g + 4, R);
      int gos = INT_MAX, goe = -1;
      for (int i = g; i < ge; ++i) { gos = std::min(gos, os[i]); goe = std::max(goe, oe[i]); }
      bool fits = true;
      std::vector<int> no(ge - g), ne(ge - g);
      for (int i = g; i < ge; ++i) {
        // (never past the pad cell of the last column - the plane ends there -, and a row whose band does not reach the
        // always-in-envelope column stops one column short of it)
        int e = std::min(goe, (i + Cc - 1) | 1);
        if (hi[i] < Cc - 1) { int lim = i + Cc - 2; if (!(lim & 1)) --lim; e = std::min(e, std::max(lim, oe[i])); }
        e = std::max(e, oe[i]);
        no[i - g] = gos; ne[i - g] = e;
        if (as_[i] - gos > 7 || e - ae_[i] > 7) fits = false;
      }
      if (!fits) continue;
      for (int i = g; i < ge; ++i) { os[i] = no[i - g]; oe[i] = ne[i - g]; }
    }
  for (int i = 0; i < R; ++i) {
    const int as = as_[i];
    int ae = ae_[i];
    // row 0's pad cells are inside the envelope (unless they are past the last column): the sweep computes them
    if (i == 0) ae = std::min(oe[0], Cc - 1);
    // where the strip's cells live: slot(i, k) = strip_store[q] + 2 (i % 64) + (k >> 1) blk + (k & 1)
    const int q = i >> 6;
    int64_t A;
    if (!cwin) {
      A = (int64_t)q * ss - (int64_t)32 * q * blk;
    } else {
      // band-compressed planes: the window of strip q that holds the row's steps t = k - 64 q (the same for all its rows)
      const int t0 = os[i] - 64 * q, t1 = oe[i] - 64 * q;
      int w = -1;
      for (int c = 0; c < 2; ++c)
        if (cwin[4 * q + 2 * c] <= t0 && t1 < cwin[4 * q + 2 * c + 1]) w = c;
      if (w < 0) return false;
      A = cbase[2 * q + w] - (int64_t)((64 * q + cwin[4 * q + 2 * w]) >> 1) * (2 * HX_STRIP);
    }
    if (A < INT32_MIN / 2 || A > INT32_MAX / 2) return false;
    if (have[q] && strip_store[q] != (int32_t)A) return false;
    have[q] = 1;
    strip_store[q] = (int32_t)A;
    const bool ready = (xf[i] & F_READY) || x_empty;
    int32_t* o = &out[2 * (size_t)i];
    if (oe[i] - os[i] > 0xFFFF || os[i] >= 0xFFFF) return false;
    o[0] = os[i] | ((oe[i] - os[i]) << 16);
    if (as - os[i] > 7 || oe[i] - ae > 7 || as < os[i] || ae > oe[i]) return false;
    o[1] = (xecls[i] & 0xFF) | (ready ? 0 : 0x100) | ((as - os[i]) << 9) | ((oe[i] - ae) << 12);
    n_steps = std::max(n_steps, oe[i] + 1);
  }
  // a lane must be idle for at least one whole step pair between two rows (its register window restarts from zero cells),
  // and the lane above must have left row i - 1 + 64's predecessor... i.e. rows i and i + 63 are never alive together
  for (int i = 0; i + 63 < R; ++i) if (os[i + 63] < oe[i] + 1) return false;
  for (int i = 0; i + 64 < R; ++i) if (os[i + 64] < oe[i] + 3) return false;
  // the largest slot must fit the converting wave's 32-bit slot
  if ((int64_t)n_steps / 2 * blk + INT32_MAX / 2 > INT32_MAX) return false;
  for (int i = R; i < R + 64; ++i) out[2 * (size_t)i] = 0xFFFF;    // sentinels: never owned
  n_steps = (n_steps + 1) & ~1;
  return true;
}

// The same for the Backward sweep of hx_band.hip, in mirrored coordinates i' = R-1-i, j' = Cc-1-j (the layout of the
// Backward matrix).  What is always inside the envelope is now the LAST row (x START) and the FIRST column (the y state
// feeding END): both are -inf away from the band (see below), written by the kernel's second wave; the sweep owns a row's band
// cells, widened to column 0 where the band touches it, and on the last row from where the row above's band begins.
//   x = first owned step | (owned steps - 1) << 16;  y = class of x state i + 1 | state i not ready << 8 | pads << 9, 12 (three bits each)
// followed by the per-strip store bases (dense planes only) and, last, one int: the first column the sweep owns on the last
// row (0: the whole row).
bool build_band_rows_bwd(const int32_t* xenv, const int32_t* yenv, const uint8_t* xf, const uint8_t* yf, const int32_t* xecls,
                         bool x_empty, bool y_empty, int R, int Cc, int band, int64_t ss, int blk, std::vector<int32_t>& out, int& n_steps) {
  if (R < 3 || Cc < 3 || R + Cc > 60000) return false;
  // the y state that feeds END must not be ready (it has the null transition to END): then no x-absorbing move leaves a
  // cell of its column (reference src/forward.cpp:1041-1049), and the column and the last row are -inf away from the band
  if ((yf[Cc - 1] & F_READY) || y_empty) return false;
  for (int i = 1; i <= R; ++i) if (xenv[i] < xenv[i - 1]) return false;
  for (int j = 1; j <= Cc; ++j) if (yenv[j] < yenv[j - 1]) return false;
  for (int i = 0; i < R; ++i) if (((xf[i] & F_EDGE) != 0) != (i == 0)) return false;
  for (int j = 0; j < Cc; ++j) if (((yf[j] & F_EDGE) != 0) != (j == Cc - 1)) return false;
  std::vector<int> lo(R), hi(R);          // mirrored rows, mirrored columns
  {
    int a = 0, b = -1;
    for (int i = 0; i < R; ++i) {
      while (a < Cc && (int64_t)yenv[a] < (int64_t)xenv[i] - band) ++a;
      while (b + 1 < Cc && (int64_t)yenv[b + 1] <= (int64_t)xenv[i] + band) ++b;
      if (b < a) return false;
      lo[R - 1 - i] = Cc - 1 - b; hi[R - 1 - i] = Cc - 1 - a;
    }
  }
  for (int i = 0; i < R; ++i) if (lo[i] <= 1) lo[i] = 0;            // the cells next to the band in the always-in column
  hi[R - 1] = Cc - 1;
  lo[R - 1] = std::min(lo[R - 1], lo[R - 2]);                       // the last row reads the row above from where its band begins
  const int n_strips = (R + HX_STRIP - 1) / HX_STRIP;
  out.assign(2 * (size_t)(R + 64) + (size_t)n_strips + 4, 0);
  int32_t* strip_store = &out[2 * (size_t)(R + 64)];
  std::vector<int> os(R), oe(R);
  n_steps = 0;
  for (int i = 0; i < R; ++i) { os[i] = (i + lo[i]) & ~1; oe[i] = (i + hi[i]) | 1; }
  // whole cache lines, as in build_band_rows: the four rows of a 64-byte group own the same steps.  What a row gains lies
  // outside the envelope or - first column, last row away from the band - is -inf by the argument above, which is what the
  // sweep stores for an owned cell outside its row's span.
  if (!getenv("HX_BAND_NO_LINE_GROUPS"))
    for (int g = 0; g < R; g += 4) {
      const int ge = std::min(g + 4, R);
      int gos = INT_MAX, goe = -1;
      for (int i = g; i < ge; ++i) { gos = std::min(gos, os[i]); goe = std::max(goe, oe[i]); }
      bool fits = true;
      for (int i = g; i < ge; ++i) {
        const int e = std::max(oe[i], std::min(goe, (i + Cc - 1) | 1));
        if (i + lo[i] - gos > 7 || e - (i + hi[i]) > 7) fits = false;
      }
      if (!fits) continue;
      for (int i = g; i < ge; ++i) { oe[i] = std::max(oe[i], std::min(goe, (i + Cc - 1) | 1)); os[i] = gos; }
    }
  for (int i = 0; i < R; ++i) {
    const int as = i + lo[i], ae = i + hi[i];
    const int q = i >> 6;
    const int64_t A = (int64_t)q * ss - (int64_t)32 * q * blk;
    if (A < INT32_MIN / 2 || A > INT32_MAX / 2) return false;
    strip_store[q] = (int32_t)A;
    const int ic = R - 1 - i;                                        // the actual x state of the row
    const bool ready = (xf[ic] & F_READY) || x_empty;
    int32_t* o = &out[2 * (size_t)i];
    if (oe[i] - os[i] > 0xFFFF || os[i] >= 0xFFFF) return false;
    o[0] = os[i] | ((oe[i] - os[i]) << 16);
    if (as - os[i] > 7 || oe[i] - ae > 7) return false;
    o[1] = (xecls[ic + 1] & 0xFF) | (ready ? 0 : 0x100) | ((as - os[i]) << 9) | ((oe[i] - ae) << 12);
    n_steps = std::max(n_steps, oe[i] + 1);
  }
  for (int i = 0; i + 63 < R; ++i) if (os[i + 63] < oe[i] + 1) return false;
  for (int i = 0; i + 64 < R; ++i) if (os[i + 64] < oe[i] + 3) return false;
  if ((int64_t)n_steps / 2 * blk + INT32_MAX / 2 > INT32_MAX) return false;
  for (int i = R; i < R + 64; ++i) out[2 * (size_t)i] = 0xFFFF;
  out[2 * (size_t)(R + 64) + n_strips] = lo[R - 1];
  n_steps = (n_steps + 1) & ~1;
  return true;
}

}  // namespace

// Kernel classes: the jobs of a batch are grouped by the fill kernel that suits them, each class is launched on its own
// (same stream, one after the other), so a tree level that mixes leaf-leaf and internal-node pairs does not fall to the
// slowest kernel as a whole.  Classes are contiguous in the class-ordered job table and in the matrix allocation.
enum KernelClass {
  KC_LEAF_LDS = 0,            // leaf-like pairs whose y side fits LDS: scaled-probability fills (HX_LSE_LINEAR) or k_fill_chain<YL>
  KC_LEAF_LDS_BANDED,
  KC_LEAF_ROT_BANDED,         // ... of them, those the banded rotating-row sweep takes (hx_band.hip), Forward only
  KC_LEAF,                    // other leaf-like pairs: k_fill_chain<LEAF>
  KC_LEAF_BANDED,
  KC_CHAIN,                   // other in-degree-1 profiles: k_fill_chain (Forward); Backward runs the general pipeline
  KC_CHAIN_BANDED,
  KC_DAG,                     // general profiles: the strip pipelines of hx_dag.hip
  KC_DAG_BANDED,
  KC_GENERIC,                 // HX_FORCE_GENERIC: the first general kernels (barrier per anti-diagonal)
  KC_COUNT
};
struct ClassRange {
  int64_t agg_begin = 0, agg_doubles = 0;   // the class's part of hx_batch::d_agg (general-profile classes)
  bool bwd_band = false;      // KC_LEAF_ROT_BANDED: every pair of the class also has the Backward sweep's row records
  int n_w32 = 0;              // KC_LEAF_ROT_BANDED: pairs of the class that may share a wavefront (DevJob::band_w32)
  int begin = 0, n = 0;       // positions in the class-ordered job table
  int max_rows = 0, max_cols = 0, max_cls = 0, yl_cols = 0, yl_emis = 0;
  int64_t mat_begin = 0, mat_doubles = 0;   // the class's matrices are contiguous: [mat_begin, mat_begin + mat_doubles)
};

struct hx_batch {
  int device = 0;
  int n_jobs = 0;
  uint32_t flags = 0;
  std::vector<DevJob> jobs;         // host copies in the caller's job order (device pointers inside)
  std::vector<hx_layout> layouts;
  std::vector<int> order;           // class-ordered table position -> caller's job index
  ClassRange cls[KC_COUNT];
  DevJob* d_jobs = nullptr;         // the caller's job order (readers, traceback, prep)
  DevJob* d_jobs_cls = nullptr;     // class order (fills)
  char* d_arena = nullptr;
  double* d_fwd = nullptr;
  double* d_bwd = nullptr;
  double* d_eplane = nullptr;       // per-cell emission planes of the jobs without a class-pair table
  double* d_agg = nullptr;          // outgoing-sum planes of the general-profile Forward pipeline (KC_DAG* jobs only)
  int64_t fwd_total = 0, max_eplane = 0;
  int max_states = 0, max_ca = 0, max_cls_pairs = 0, max_cls = 0;
  bool any_compressed = false;
  size_t lp_end_off = 0, lp_start_off = 0;   // [n_jobs] doubles each in the arena, caller's order: one copy per read
  int64_t total_cells = 0;
  bool forward_done = false, backward_done = false;
  bool used_multi[2] = {false, false};   // the last Forward / Backward launch dealt some pair to several workgroups
  bool no_multi = false;                 // ... which once failed (waves gave up waiting for one another): one workgroup per pair from now on
  int relaunches = 0;                    // fills repeated for that reason
  bool sub_scattered = false;        // subx / suby of table-emission jobs exist per state (k_scatter_sub, on demand)
  hipStream_t last_stream = nullptr;
  hipEvent_t ev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  hipEvent_t ev_side[2] = {nullptr, nullptr};   // fork / join of the device's side stream (banded sweep's edge kernel)
  bool dag_linear = false;                      // general-profile classes run the scaled-probability fill (hx_daglin.hip)
  hipStream_t copy_stream = nullptr;            // hx_batch_read_matrix_async: the device's copy stream (DeviceTables)
  int* d_multi = nullptr;                       // progress counters of a pair dealt to several workgroups (hx_dag.hip MULTI)
  std::vector<hipEvent_t> copied[2];            // per job: the event behind its asynchronous matrix copy, or null
  bool records_valid = false;        // the per-state records (FwdPack) of the device job tables are built (launch_prep / ensure_state_records)
  int32_t* d_trace = nullptr;        // hx_batch_best_trace: path buffers, kept between calls
  int64_t* d_trace_n = nullptr;
  void* h_trace = nullptr;
  int64_t trace_cap = 0;
  bool trace_ties_valid = false;     // the near-tie flags of the last hx_batch_best_trace are in d_trace_n's third block
  bool ev_valid[2] = {false, false};
};

namespace {
// (re)upload both job tables: the caller's order and the class order
int publish_jobs(hx_batch* b) {
  HIP_TRY(hipMemcpy(b->d_jobs, b->jobs.data(), sizeof(DevJob) * b->n_jobs, hipMemcpyHostToDevice));
  std::vector<DevJob> sorted((size_t)b->n_jobs);
  for (int p = 0; p < b->n_jobs; ++p) sorted[p] = b->jobs[b->order[p]];
  HIP_TRY(hipMemcpy(b->d_jobs_cls, sorted.data(), sizeof(DevJob) * b->n_jobs, hipMemcpyHostToDevice));
  return HX_OK;
}
// every entry point that touches a batch runs on the batch's device
int use_device(const hx_batch* b) {
  if (hipSetDevice(b->device) != hipSuccess) return fail(HX_ERR_HIP, "hipSetDevice(%d) failed", b->device);
  return HX_OK;
}
}  // namespace

namespace hx {
// hx_sumprod.hip: the 8-byte log_sum_exp table of an initialised device, or null
const double* device_lse_table(int device) {
  return device >= 0 && device < HX_MAX_DEVICES && g_dev[device].ready ? g_dev[device].tab : nullptr;
}
int api_fail(int code, const char* what) { return fail(code, "%s", what); }
}  // namespace hx

extern "C" {

int hx_version(void) { return 1; }

const char* hx_last_error(void) { return g_err; }

namespace {
// keep_stream: hx_init on an ordinal that is already initialised replaces the tables only - batches created before keep the
// device's copy stream (hx_batch::copy_stream) and their events on it
void free_tables(DeviceTables& t, bool keep_stream = false) {
  if (t.tab) (void)hipFree(t.tab);
  if (t.fast_tab) (void)hipFree(t.fast_tab);
  if (t.log_tab) (void)hipFree(t.log_tab);
  if (t.pair_tab) (void)hipFree(t.pair_tab);
  t.tab = t.fast_tab = t.log_tab = t.pair_tab = nullptr;
  if (!keep_stream) {
    if (t.copy_stream) (void)hipStreamDestroy(t.copy_stream);
    if (t.side_stream) (void)hipStreamDestroy(t.side_stream);
    t.copy_stream = t.side_stream = nullptr;
  }
  t.ready = false;
}
int upload(double** dst, const std::vector<double>& src) {
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(dst), src.size() * sizeof(double)));
  HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(double), hipMemcpyHostToDevice));
  return HX_OK;
}
}  // namespace

int hx_device_count(void) {
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess) return 0;
  return n_dev;
}

int hx_init(int device_ordinal, const double* lse_table, size_t n_entries) {
  if (!lse_table || n_entries != HX_LSE_TABLE_ENTRIES)
    return fail(HX_ERR_INVALID_ARG, "lse_table must hold %d doubles (got %zu)", HX_LSE_TABLE_ENTRIES, n_entries);
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return fail(HX_ERR_NO_DEVICE, "no HIP device available");
  if (device_ordinal < 0 || device_ordinal >= n_dev || device_ordinal >= HX_MAX_DEVICES)
    return fail(HX_ERR_NO_DEVICE, "device %d out of range (%d devices)", device_ordinal, n_dev);
  if (hipSetDevice(device_ordinal) != hipSuccess) return fail(HX_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device_ordinal);
  DeviceTables& D = g_dev[device_ordinal];
  free_tables(D, /*keep_stream*/ true);
  try {
    int rc;
    if ((rc = upload(&D.tab, std::vector<double>(lse_table, lse_table + n_entries))) != HX_OK) { free_tables(D); return rc; }
    std::vector<double> pairs(2 * n_entries, 0.0);
    for (size_t n = 0; n < n_entries; ++n) {
      pairs[2 * n] = lse_table[n];
      pairs[2 * n + 1] = n + 1 < n_entries ? lse_table[n + 1] - lse_table[n] : 0.0;
    }
    if ((rc = upload(&D.pair_tab, pairs)) != HX_OK) { free_tables(D); return rc; }
    std::vector<double> lt((size_t)log_table_doubles());
    build_log_table(lt.data());
    if ((rc = upload(&D.log_tab, lt)) != HX_OK) { free_tables(D); return rc; }
    std::vector<double> ft;
    build_fast_table(ft);
    if ((rc = upload(&D.fast_tab, ft)) != HX_OK) { free_tables(D); return rc; }
  } catch (const std::bad_alloc&) {
    free_tables(D);
    return fail(HX_ERR_OUT_OF_MEMORY, "host allocation failed");
  }
  D.ready = true;
  g_device = device_ordinal;
  return HX_OK;
}

int hx_shutdown(void) {
  for (int d = 0; d < HX_MAX_DEVICES; ++d)
    if (g_dev[d].ready || g_dev[d].tab) {
      (void)hipSetDevice(d);
      free_tables(g_dev[d]);
    }
  g_device = -1;
  return HX_OK;
}

static int batch_create_impl(int device, const hx_pair_job* jobs, int32_t n_jobs, uint32_t flags, hx_batch** out) {
  if (!out) return fail(HX_ERR_INVALID_ARG, "out is null");
  *out = nullptr;
  if (device < 0 || device >= HX_MAX_DEVICES || !g_dev[device].ready)
    return fail(HX_ERR_NOT_INITIALIZED, "hx_init has not been called for device %d", device);
  if (!jobs || n_jobs <= 0) return fail(HX_ERR_INVALID_ARG, "need at least one job");
  HIP_TRY(hipSetDevice(device));
  const bool timing = getenv("HX_TIMING_CREATE") != nullptr;
  const auto tick = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double tc0 = tick();

  hx_batch* b = new (std::nothrow) hx_batch;
  if (!b) return fail(HX_ERR_OUT_OF_MEMORY, "host allocation failed");
  b->device = device;
  b->n_jobs = n_jobs;
  b->flags = flags;
  b->jobs.resize(n_jobs);
  b->layouts.resize(n_jobs);

  Arena ar;
  std::vector<JobOff> offs(n_jobs);
  std::vector<int64_t> mat_off(n_jobs);
  std::vector<int> kclass(n_jobs);
  int64_t eplane_total = 0;
  int rc = HX_OK;
  static const bool force_dag = getenv("HX_FORCE_DAG") != nullptr;   // tuning hook: the general pipeline for chain profiles too
  const bool band_old = getenv("HX_BAND_OLD") != nullptr;             // tuning / test hook: banded leaf pairs on the strip pipelines
  const bool linear = (flags & HX_LSE_LINEAR) == HX_LSE_LINEAR;
  for (int k = 0; k < n_jobs && rc == HX_OK; ++k) {
    const hx_pair_job& pj = jobs[k];
    const hx_hmm* h = pj.hmm;
    if (!h || !pj.x || !pj.y) { rc = fail(HX_ERR_INVALID_ARG, "job %d has null members", k); break; }
    const int A = h->alph_size, C = h->components, CA = A * C;
    if (A <= 0 || C <= 0) { rc = fail(HX_ERR_INVALID_ARG, "job %d: bad alphabet/components", k); break; }
    if (!h->log_root || !h->log_sub_l || !h->log_sub_r || !h->log_ins_l || !h->log_ins_r || !h->log_cptw_l || !h->log_cptw_r) {
      rc = fail(HX_ERR_INVALID_ARG, "job %d: null hmm arrays", k);
      break;
    }
    const bool need_env = pj.max_distance >= 0;
    JobOff& jo = offs[k];
    if ((rc = flatten_profile(pj.x, CA, need_env, false, ar, jo.x)) != HX_OK) break;
    if ((rc = flatten_profile(pj.y, CA, need_env, true, ar, jo.y)) != HX_OK) break;
    jo.log_root = ar.put(h->log_root, sizeof(double) * CA);
    jo.log_sub_l = ar.put(h->log_sub_l, sizeof(double) * CA * A);
    jo.log_sub_r = ar.put(h->log_sub_r, sizeof(double) * CA * A);
    jo.log_ins_l = ar.put(h->log_ins_l, sizeof(double) * CA);
    jo.log_ins_r = ar.put(h->log_ins_r, sizeof(double) * CA);
    jo.log_cptw_l = ar.put(h->log_cptw_l, sizeof(double) * C);
    jo.log_cptw_r = ar.put(h->log_cptw_r, sizeof(double) * C);
    const int64_t pairs = (int64_t)jo.x.n_cls * jo.y.n_cls;
    jo.table_emission = pairs > 0 && pairs <= (1 << 16);
    jo.emis = jo.table_emission ? ar.reserve(sizeof(double) * pairs) : 0;
    jo.emis_pad = jo.table_emission ? ar.reserve(sizeof(double) * (jo.x.n_cls + 1) * (jo.y.n_cls + 1)) : 0;
    jo.fwd_windows = jo.bwd_windows = jo.strip_base = jo.yword = jo.yword_bwd = jo.band_rows = jo.band_rows_bwd = 0;
    jo.compressed = false;
    jo.compact_plane = 0;

    // ---- kernel class of the job ----
    const bool chain = jo.x.chain && jo.y.chain;
    const bool leaf_like = chain && jo.x.interior_emit && jo.y.interior_emit && jo.table_emission;
    // y side small enough for LDS (hx_chain.hip HX_YL_*, hx_linear.hip) and all its transitions have lpTrans 0
    const bool ylds = leaf_like && jo.y.lp_zero && jo.y.n <= 6144 && jo.y.n_cls + 1 <= 64 && jo.x.n_cls < 255 &&
                      (int64_t)(jo.x.n_cls + 1) * (jo.y.n_cls + 1) <= 1024;
    int kc;
    if (flags & HX_FORCE_GENERIC) kc = KC_GENERIC;
    else if (force_dag || !chain) kc = need_env ? KC_DAG_BANDED : KC_DAG;
    else if (ylds) kc = need_env ? KC_LEAF_LDS_BANDED : KC_LEAF_LDS;
    else if (leaf_like) kc = need_env ? KC_LEAF_BANDED : KC_LEAF;
    else kc = need_env ? KC_CHAIN_BANDED : KC_CHAIN;
    kclass[k] = kc;
    if ((flags & HX_BAND_COMPRESSED) && (kc >= KC_DAG || (flags & HX_KEEP_BACKWARD))) {
      rc = fail(HX_ERR_INVALID_ARG, "HX_BAND_COMPRESSED is implemented by the Forward fills of chain (leaf) profiles only: "
                                    "no general profiles (job %d), no HX_KEEP_BACKWARD / HX_FORCE_GENERIC", k);
      break;
    }

    if (linear && (kc == KC_LEAF_LDS || kc == KC_LEAF_LDS_BANDED) && jo.y.n >= 2) {
      // per-column words of the banded scaled-probability fill (hx_linear.hip): {emission class : 8, not ready : 1,
      // always in envelope : 1, envelope coordinate : 22}, with 64 words of padding on either side (the edge columns').
      const int Cc = jo.y.n - 1;
      const std::vector<uint8_t> yf(ar.host.data() + jo.y.flags, ar.host.data() + jo.y.flags + jo.y.n);
      const std::vector<int32_t> ecls(reinterpret_cast<const int32_t*>(ar.host.data() + jo.y.ecls),
                                      reinterpret_cast<const int32_t*>(ar.host.data() + jo.y.ecls) + jo.y.n);
      std::vector<uint32_t> yw((size_t)Cc + 328);     // (the kernel refills its word ring 64 words at a time, up to 256 ahead)
      for (int jp = 0; jp < Cc + 328; ++jp) {
        const int j = jp < 64 ? 0 : (jp - 64 >= Cc ? Cc - 1 : jp - 64);
        const bool ready = (yf[j] & F_READY) || jo.y.empty;
        const uint32_t env = (need_env && pj.y->env_pos) ? (uint32_t)pj.y->env_pos[j] : 0u;
        yw[jp] = (uint32_t)ecls[j] | (ready ? 0u : 0x100u) | ((yf[j] & F_EDGE) ? 0x200u : 0u) | (env << 10);
      }
      jo.yword = ar.put(yw.data(), sizeof(uint32_t) * yw.size());
      // Backward: sweep column j is y state jc = Cc-1-j; its own readiness, edge flag and envelope coordinate, and the
      // emission class of state jc+1
      for (int jp = 0; jp < Cc + 328; ++jp) {
        const int j = jp < 64 ? 0 : (jp - 64 >= Cc ? Cc - 1 : jp - 64), jc = Cc - 1 - j;
        const bool ready = (yf[jc] & F_READY) || jo.y.empty;
        const uint32_t env = (need_env && pj.y->env_pos) ? (uint32_t)pj.y->env_pos[jc] : 0u;
        yw[jp] = (uint32_t)ecls[jc + 1] | (ready ? 0u : 0x100u) | ((yf[jc] & F_EDGE) ? 0x200u : 0u) | (env << 10);
      }
      jo.yword_bwd = ar.put(yw.data(), sizeof(uint32_t) * yw.size());
    }
    std::vector<int32_t> cwin_keep;
    std::vector<int64_t> cbase_keep;
    if (need_env) {
      const int R = jo.x.n - 1, Cc = jo.y.n - 1;
      // (copies: put() may reallocate the staging image the pointers would point into)
      const std::vector<uint8_t> xf(ar.host.data() + jo.x.flags, ar.host.data() + jo.x.flags + jo.x.n);
      const std::vector<uint8_t> yf(ar.host.data() + jo.y.flags, ar.host.data() + jo.y.flags + jo.y.n);
      const std::vector<int32_t> wf = strip_windows(xf.data(), pj.x->env_pos, yf.data(), pj.y->env_pos, R, Cc, pj.max_distance, false);
      const std::vector<int32_t> wb = strip_windows(xf.data(), pj.x->env_pos, yf.data(), pj.y->env_pos, R, Cc, pj.max_distance, true);
      jo.bwd_windows = ar.put(wb.data(), sizeof(int32_t) * wb.size());
      if (flags & HX_BAND_COMPRESSED) {
        // Band-compressed storage: a strip keeps only the step windows it sweeps.  The windows are put into the
        // form the fill uses them in (whole step pairs, clipped, merged when they touch - k_fill_leaf_linear),
        // which that kernel's own widening leaves unchanged, and every window gets its offset in the state plane.
        std::vector<int32_t> we(wf);
        const int n_strips = (R + HX_STRIP - 1) / HX_STRIP, nsteps = (Cc + HX_STRIP) & ~1;
        std::vector<int64_t> sb(2 * (size_t)n_strips, 0);
        int64_t off = 0;
        for (int s = 0; s < n_strips; ++s) {
          int32_t* o = &we[4 * (size_t)s];
          int lo[2], hi[2];
          for (int w = 0; w < 2; ++w) {
            lo[w] = o[2 * w] & ~1;
            const int hh = (o[2 * w + 1] + 1) & ~1;
            hi[w] = hh < nsteps ? hh : nsteps;
            if (hi[w] < lo[w]) hi[w] = lo[w];
          }
          if (hi[1] > lo[1] && lo[1] <= hi[0]) { hi[0] = std::max(hi[0], hi[1]); lo[1] = hi[1] = 0; }
          for (int w = 0; w < 2; ++w) {
            o[2 * w] = lo[w]; o[2 * w + 1] = hi[w];
            sb[2 * (size_t)s + w] = off;
            off += (int64_t)((hi[w] - lo[w]) >> 1) * (2 * HX_STRIP);
          }
        }
        jo.fwd_windows = ar.put(we.data(), sizeof(int32_t) * we.size());
        jo.strip_base = ar.put(sb.data(), sizeof(int64_t) * sb.size());
        cwin_keep = we; cbase_keep = sb;
        jo.compressed = true;
        jo.compact_plane = (off + 1) & ~(int64_t)1;
      } else {
        jo.fwd_windows = ar.put(wf.data(), sizeof(int32_t) * wf.size());
      }
    }

    DevJob& J = b->jobs[k];
    memset(&J, 0, sizeof(J));
    for (int s = 0; s < 5; ++s)
      for (int d = 0; d < 6; ++d) J.T[s][d] = h->lp_trans[s][d];
    J.A = A; J.C = C; J.CA = CA;
    J.max_dist = pj.max_distance;
    J.n_rows = jo.x.n - 1;
    J.n_cols = jo.y.n - 1;
    J.n_strips = (J.n_rows + HX_STRIP - 1) / HX_STRIP;
    J.strip_stride = strip_stride_for(J.n_cols);
    J.plane = jo.compressed ? jo.compact_plane : J.n_strips * J.strip_stride;
    J.chain = chain;
    J.leaf_like = leaf_like;
    hx_layout& L = b->layouts[k];
    L.n_rows = J.n_rows; L.n_cols = J.n_cols; L.strip_rows = HX_STRIP; L.n_strips = J.n_strips;
    J.blk = 2 * HX_STRIP; J.matrix_doubles = 5 * J.plane;
    // Unbanded leaf pairs of the scaled-probability fills keep the five states of a step pair adjacent: a wavefront then
    // writes 5 KiB contiguous per iteration instead of 1 KiB into each of five planes (hx_linear.hip; same total size).
    // Only those kernels and the layout-aware readers ever see such a job.
    if (linear && kc == KC_LEAF_LDS && !getenv("HX_PLANAR_LAYOUT")) {
      J.strip_stride *= 5; J.plane = 2 * HX_STRIP; J.blk = 10 * HX_STRIP;
    }
    if (kc == KC_LEAF_LDS_BANDED && jo.x.lp_zero && !band_old) {
      // the banded rotating-row sweep (hx_band.hip), when the pair satisfies its assumptions and its sides fit LDS
      const int pol = linear ? ((flags & HX_LSE_TRUNC) == HX_LSE_TRUNC ? 3 : 0) : ((flags & HX_LSE_FAST) ? 1 : 2);
      std::vector<int32_t> rows;
      int n_steps = 0;
      const std::vector<uint8_t> xf(ar.host.data() + jo.x.flags, ar.host.data() + jo.x.flags + jo.x.n);
      const std::vector<uint8_t> yf(ar.host.data() + jo.y.flags, ar.host.data() + jo.y.flags + jo.y.n);
      const std::vector<int32_t> xecls(reinterpret_cast<const int32_t*>(ar.host.data() + jo.x.ecls),
                                       reinterpret_cast<const int32_t*>(ar.host.data() + jo.x.ecls) + jo.x.n);
      // Band-compressed planes of such a pair need not hold the always-in-envelope column (the y state feeding END) away
      // from the band: below row 1 those cells are -inf (no x-absorbing move enters that column from a cell that is not
      // itself in it, and the column's only finite cell off the band is (1, Ny-2): see hx_band.hip), and a cell that is not
      // stored reads as -inf (hx_batch_read_cells, the traceback, lpEnd).  A strip's second window is that column whenever
      // it is apart from the band's window, which also says that no row of the strip has a band that reaches it.  Without
      // those windows a pair's planes are about half the size, and the fill no longer writes 5 x 8 bytes of -inf into a
      // cache line of its own for every row (a fifth of the sweep's time at 4096 pairs).
      std::vector<int32_t> twin;
      std::vector<int64_t> tbase;
      int64_t tplane = 0;
      if (jo.compressed) {
        twin = cwin_keep;
        const int n_strips = (J.n_rows + HX_STRIP - 1) / HX_STRIP;
        tbase.assign(2 * (size_t)n_strips, 0);
        int64_t off = 0;
        for (int q = 0; q < n_strips; ++q) {
          int32_t* o = &twin[4 * (size_t)q];
          if (q >= 1 && o[3] > o[2]) o[2] = o[3] = 0;
          for (int w = 0; w < 2; ++w) {
            tbase[2 * (size_t)q + w] = off;
            off += (int64_t)((o[2 * w + 1] - o[2 * w]) >> 1) * (2 * HX_STRIP);
          }
        }
        tplane = (off + 1) & ~(int64_t)1;
      }
      const bool fits = band_kernel_fits(pol, J.n_rows, J.n_cols, std::max(jo.x.n_cls, jo.y.n_cls));
      bool trimmed = fits && jo.compressed && !getenv("HX_BAND_KEEP_EDGE_COLUMN") &&
                     build_band_rows(pj.x->env_pos, pj.y->env_pos, xf.data(), yf.data(), xecls.data(), jo.x.empty != 0, J.n_rows, J.n_cols,
                                     pj.max_distance, J.strip_stride, J.blk, twin.data(), tbase.data(), rows, n_steps);
      if (trimmed) {
        jo.fwd_windows = ar.put(twin.data(), sizeof(int32_t) * twin.size());
        jo.strip_base = ar.put(tbase.data(), sizeof(int64_t) * tbase.size());
        jo.compact_plane = tplane;
        J.plane = tplane;
        J.matrix_doubles = 5 * J.plane;
      }
      if (trimmed ||
          (fits &&
          build_band_rows(pj.x->env_pos, pj.y->env_pos, xf.data(), yf.data(), xecls.data(), jo.x.empty != 0, J.n_rows, J.n_cols,
                          pj.max_distance, J.strip_stride, J.blk, jo.compressed ? cwin_keep.data() : nullptr,
                          jo.compressed ? cbase_keep.data() : nullptr, rows, n_steps))) {
        // the Backward sweep of the same kernel (dense planes: band-compressed batches have no Backward).  A pair the
        // Backward sweep cannot take (fewer than three rows or columns) stays out of the class altogether, so that the
        // class never falls back to the strip pipeline because of one such pair.
        std::vector<int32_t> rows_b;
        int n_steps_b = 0;
        const bool bwd_ok = jo.compressed ||
            build_band_rows_bwd(pj.x->env_pos, pj.y->env_pos, xf.data(), yf.data(), xecls.data(), jo.x.empty != 0, jo.y.empty != 0, J.n_rows, J.n_cols,
                                pj.max_distance, J.strip_stride, J.blk, rows_b, n_steps_b);
        if (bwd_ok) {
          jo.band_rows = ar.put(rows.data(), sizeof(int32_t) * rows.size());
          J.band_steps = n_steps;
          kc = kclass[k] = KC_LEAF_ROT_BANDED;
          // two pairs per wavefront (hx_band2.hip): a lane is a row modulo 32 there, so rows i and i + 31 must never be alive
          // together and a lane must be idle for a whole step pair between rows i and i + 32 - the conditions build_band_rows
          // checks for 63 / 64, on both sweeps' records
          auto ring32 = [&](const std::vector<int32_t>& rec) {
            auto first = [&](int i) { return rec[2 * (size_t)i] & 0xFFFF; };
            auto last = [&](int i) { return (rec[2 * (size_t)i] & 0xFFFF) + ((rec[2 * (size_t)i] >> 16) & 0xFFFF); };
            for (int i = 0; i + 31 < J.n_rows; ++i) if (first(i + 31) < last(i) + 1) return false;
            for (int i = 0; i + 32 < J.n_rows; ++i) if (first(i + 32) < last(i) + 3) return false;
            return true;
          };
          J.band_w32 = (J.blk == 2 * HX_STRIP && ring32(rows) && (jo.compressed || ring32(rows_b)) &&
                        band2_kernel_fits(J.n_rows, J.n_cols, std::max(jo.x.n_cls, jo.y.n_cls))) ? 1 : 0;
          J.band_steps_bwd = 0;
          if (!jo.compressed) {
            jo.band_rows_bwd = ar.put(rows_b.data(), sizeof(int32_t) * rows_b.size());
            J.band_steps_bwd = n_steps_b;
          }
        }
      }
    }
    L.strip_stride = J.strip_stride; L.plane_stride = J.plane;
    L.block_stride = J.blk; L.matrix_doubles = J.matrix_doubles;
    L.mirrored = 0; L.compressed = jo.compressed ? 1 : 0;
    b->any_compressed = b->any_compressed || jo.compressed;
    jo.eplane_off = -1;
    if (!jo.table_emission) {   // (also for a profile without emitting states: the pipeline's loads are unconditional)
      jo.eplane_off = eplane_total;
      eplane_total += J.plane;
      if (J.plane > b->max_eplane) b->max_eplane = J.plane;
    }
    b->total_cells += (int64_t)J.n_rows * J.n_cols;
    if (jo.x.n > b->max_states) b->max_states = jo.x.n;
    if (jo.y.n > b->max_states) b->max_states = jo.y.n;
    if (jo.x.n_cls > b->max_cls) b->max_cls = jo.x.n_cls;
    if (jo.y.n_cls > b->max_cls) b->max_cls = jo.y.n_cls;
    if (CA > b->max_ca) b->max_ca = CA;
    if (jo.table_emission && pairs > b->max_cls_pairs) b->max_cls_pairs = (int)pairs;
    ClassRange& cr = b->cls[kc];
    cr.n++;
    if (kc == KC_LEAF_ROT_BANDED && J.band_w32) cr.n_w32++;
    if (J.n_rows > cr.max_rows) cr.max_rows = J.n_rows;
    if (J.n_cols > cr.max_cols) cr.max_cols = J.n_cols;
    if (jo.x.n_cls > cr.max_cls) cr.max_cls = jo.x.n_cls;
    if (jo.y.n_cls > cr.max_cls) cr.max_cls = jo.y.n_cls;
    if (((jo.y.n + 3) & ~3) > cr.yl_cols) cr.yl_cols = (jo.y.n + 3) & ~3;
    const int ep = ((jo.x.n_cls + 1) * (jo.y.n_cls + 1) + 1) & ~1;
    if (ep > cr.yl_emis) cr.yl_emis = ep;
  }
  if (rc != HX_OK) { delete b; return rc; }

  // class-ordered job table and matrix allocation (stable within a class)
  // The general-profile classes need scratch planes next to the matrices: the five outgoing sums of every cell
  // (hx_dag.hip), or - HX_LSE_LINEAR, when every such job's planes fit 32-bit byte offsets - the cells in the
  // scaled-probability fill's own format (hx_daglin.hip).
  // (General profiles under the truncating policy run as HX_LSE_FAST, which truncates as the reference does.  The scaled-
  // probability pipeline of hx_daglin.hip with truncating sums was built and withdrawn in round 3: it sums a state's fourth and
  // further in-transitions out of the reference's order, and with truncation the order of a sum matters at the 4.5e-5 level -
  // cells downstream of such a state moved by up to 6e-5 where the leaf kernels hold 1e-7.)
  b->dag_linear = (flags & HX_LSE_LINEAR) == HX_LSE_LINEAR && (flags & HX_LSE_TRUNC) != HX_LSE_TRUNC && !(flags & HX_FORCE_GENERIC);
  for (int k = 0; k < n_jobs && b->dag_linear; ++k)
    if ((kclass[k] == KC_DAG || kclass[k] == KC_DAG_BANDED) && !dag_linear_fits(b->jobs[k].plane)) b->dag_linear = false;
  int64_t mat_total = 0, agg_total = 0;
  std::vector<int64_t> agg_off(n_jobs, -1);
  {
    int pos = 0;
    b->order.resize(n_jobs);
    for (int c = 0; c < KC_COUNT; ++c) {
      ClassRange& cr = b->cls[c];
      cr.begin = pos;
      cr.mat_begin = mat_total;
      cr.agg_begin = agg_total;
      for (int k = 0; k < n_jobs; ++k)
        if (kclass[k] == c) {
          b->order[pos++] = k;
          mat_off[k] = mat_total;
          mat_total += b->jobs[k].matrix_doubles;
          if (c == KC_DAG || c == KC_DAG_BANDED) {
            agg_off[k] = agg_total;
            agg_total += b->dag_linear ? dag_linear_scratch_doubles(b->jobs[k].plane, offs[k].x.n, offs[k].y.n, jobs[k].x->in_off[jobs[k].x->n_states], jobs[k].y->in_off[jobs[k].y->n_states]) : b->jobs[k].matrix_doubles;
          }
        }
      cr.mat_doubles = mat_total - cr.mat_begin;
      cr.agg_doubles = agg_total - cr.agg_begin;
    }
  }
  // lpEnd / lpStart of all jobs, contiguous (caller's order): one copy back per read
  b->lp_end_off = ar.reserve(sizeof(double) * n_jobs);
  b->lp_start_off = ar.reserve(sizeof(double) * n_jobs);

  auto cleanup = [&](int code) { hx_batch_destroy(b); return code; };
  const double tc1 = tick();
  if (hipMalloc(reinterpret_cast<void**>(&b->d_arena), ar.host.size() + 256) != hipSuccess)
    return cleanup(fail(HX_ERR_OUT_OF_MEMORY, "hipMalloc of %zu input bytes failed", ar.host.size()));
  if (hipMemcpy(b->d_arena, ar.host.data(), ar.host.size(), hipMemcpyHostToDevice) != hipSuccess)
    return cleanup(fail(HX_ERR_HIP, "input upload failed"));
  if (hipMalloc(reinterpret_cast<void**>(&b->d_fwd), sizeof(double) * (size_t)mat_total) != hipSuccess)
    return cleanup(fail(HX_ERR_OUT_OF_MEMORY, "hipMalloc of %lld Forward-matrix bytes failed", (long long)(mat_total * 8)));
  b->fwd_total = mat_total;
  if (eplane_total > 0)
    if (hipMalloc(reinterpret_cast<void**>(&b->d_eplane), sizeof(double) * (size_t)eplane_total) != hipSuccess)
      return cleanup(fail(HX_ERR_OUT_OF_MEMORY, "hipMalloc of %lld emission-plane bytes failed", (long long)(eplane_total * 8)));
  if (agg_total > 0)
    if (hipMalloc(reinterpret_cast<void**>(&b->d_agg), sizeof(double) * (size_t)agg_total) != hipSuccess)
      return cleanup(fail(HX_ERR_OUT_OF_MEMORY, "hipMalloc of %lld scratch bytes failed", (long long)(agg_total * 8)));
  if (flags & HX_KEEP_BACKWARD)
    if (hipMalloc(reinterpret_cast<void**>(&b->d_bwd), sizeof(double) * (size_t)mat_total) != hipSuccess)
      return cleanup(fail(HX_ERR_OUT_OF_MEMORY, "hipMalloc of %lld Backward-matrix bytes failed", (long long)(mat_total * 8)));

  b->cls[KC_LEAF_ROT_BANDED].bwd_band = b->cls[KC_LEAF_ROT_BANDED].n > 0 && !getenv("HX_BAND_BWD_OLD");
  for (int k = 0; k < n_jobs; ++k)
    if (kclass[k] == KC_LEAF_ROT_BANDED && !offs[k].band_rows_bwd) b->cls[KC_LEAF_ROT_BANDED].bwd_band = false;
  for (int k = 0; k < n_jobs; ++k) {
    DevJob& J = b->jobs[k];
    const JobOff& jo = offs[k];
    char* base = b->d_arena;
    bind_profile(J.x, jo.x, base);
    bind_profile(J.y, jo.y, base);
    J.log_root = reinterpret_cast<double*>(base + jo.log_root);
    J.log_sub_l = reinterpret_cast<double*>(base + jo.log_sub_l);
    J.log_sub_r = reinterpret_cast<double*>(base + jo.log_sub_r);
    J.log_ins_l = reinterpret_cast<double*>(base + jo.log_ins_l);
    J.log_ins_r = reinterpret_cast<double*>(base + jo.log_ins_r);
    J.log_cptw_l = reinterpret_cast<double*>(base + jo.log_cptw_l);
    J.log_cptw_r = reinterpret_cast<double*>(base + jo.log_cptw_r);
    J.emis = jo.table_emission ? reinterpret_cast<double*>(base + jo.emis) : nullptr;
    J.emis_pad = jo.table_emission ? reinterpret_cast<double*>(base + jo.emis_pad) : nullptr;
    J.emis_plane = jo.eplane_off >= 0 ? b->d_eplane + jo.eplane_off : nullptr;
    J.agg = agg_off[k] >= 0 ? b->d_agg + agg_off[k] : nullptr;
    J.fwd_windows = J.max_dist >= 0 ? reinterpret_cast<int32_t*>(base + jo.fwd_windows) : nullptr;
    J.bwd_windows = J.max_dist >= 0 ? reinterpret_cast<int32_t*>(base + jo.bwd_windows) : nullptr;
    J.strip_base = jo.compressed ? reinterpret_cast<int64_t*>(base + jo.strip_base) : nullptr;
    J.yword = jo.yword ? reinterpret_cast<uint32_t*>(base + jo.yword) : nullptr;
    J.yword_bwd = jo.yword_bwd ? reinterpret_cast<uint32_t*>(base + jo.yword_bwd) : nullptr;
    J.band_rows = jo.band_rows ? base + jo.band_rows : nullptr;
    J.band_rows_bwd = jo.band_rows_bwd ? base + jo.band_rows_bwd : nullptr;
    J.lp_end = reinterpret_cast<double*>(base + b->lp_end_off) + k;
    J.lp_start = reinterpret_cast<double*>(base + b->lp_start_off) + k;
    J.fwd = b->d_fwd + mat_off[k];
    J.bwd = b->d_bwd ? b->d_bwd + mat_off[k] : nullptr;
  }
  if (hipMalloc(reinterpret_cast<void**>(&b->d_jobs), sizeof(DevJob) * n_jobs) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&b->d_jobs_cls), sizeof(DevJob) * n_jobs) != hipSuccess)
    return cleanup(fail(HX_ERR_OUT_OF_MEMORY, "hipMalloc of job table failed"));
  const double tc2 = tick();
  if ((rc = publish_jobs(b)) != HX_OK) return cleanup(rc);
  if (timing)
    fprintf(stderr, "timing: hx_batch_create of %d jobs: host images and plans %.4f s, allocations and upload of %zu bytes %.4f s, job tables %.4f s\n",
            n_jobs, tc1 - tc0, ar.host.size(), tc2 - tc1, tick() - tc2);
  for (int w = 0; w < 2; ++w)
    for (int e = 0; e < 2; ++e)
      if (hipEventCreate(&b->ev[w][e]) != hipSuccess) return cleanup(fail(HX_ERR_HIP, "hipEventCreate failed"));
  *out = b;
  return HX_OK;
}

int hx_batch_create_on(int device, const hx_pair_job* jobs, int32_t n_jobs, uint32_t flags, hx_batch** out) {
  // the arithmetic policy: HX_LSE_EXACT overrides the other policy bits; no policy bit at all = the default, HX_LSE_TRUNC.
  // Inside the library "exact" is the absence of the fast / linear / truncating bits.
  if (flags & HX_LSE_EXACT) flags &= ~(HX_LSE_EXACT | HX_LSE_TRUNC);
  else if (!(flags & HX_LSE_TRUNC)) flags |= HX_LSE_TRUNC;
  try {
    return batch_create_impl(device, jobs, n_jobs, flags, out);
  } catch (const std::bad_alloc&) {         // (std::vector / Arena growth: nothing throws across the ABI)
    if (out) *out = nullptr;
    return fail(HX_ERR_OUT_OF_MEMORY, "host allocation failed while building the batch");
  }
}

int hx_batch_create(const hx_pair_job* jobs, int32_t n_jobs, uint32_t flags, hx_batch** out) {
  if (g_device < 0) { if (out) *out = nullptr; return fail(HX_ERR_NOT_INITIALIZED, "hx_init has not been called"); }
  return hx_batch_create_on(g_device, jobs, n_jobs, flags, out);
}

int hx_batch_device(const hx_batch* b) { return b ? b->device : -1; }

int hx_batch_destroy(hx_batch* b) {
  if (!b) return HX_OK;
  (void)hipSetDevice(b->device);
  (void)hipDeviceSynchronize();
  for (int w = 0; w < 2; ++w)
    for (int e = 0; e < 2; ++e)
      if (b->ev[w][e]) (void)hipEventDestroy(b->ev[w][e]);
  for (int e = 0; e < 2; ++e)
    if (b->ev_side[e]) (void)hipEventDestroy(b->ev_side[e]);
  for (int w = 0; w < 2; ++w)
    for (hipEvent_t e : b->copied[w])
      if (e) (void)hipEventDestroy(e);
  if (b->d_multi) (void)hipFree(b->d_multi);
  if (b->d_jobs) (void)hipFree(b->d_jobs);
  if (b->d_jobs_cls) (void)hipFree(b->d_jobs_cls);
  if (b->d_arena) (void)hipFree(b->d_arena);
  if (b->d_fwd) (void)hipFree(b->d_fwd);
  if (b->d_bwd) (void)hipFree(b->d_bwd);
  if (b->d_eplane) (void)hipFree(b->d_eplane);
  if (b->d_agg) (void)hipFree(b->d_agg);
  if (b->d_trace) (void)hipFree(b->d_trace);
  if (b->d_trace_n) (void)hipFree(b->d_trace_n);
  if (b->h_trace) (void)hipHostFree(b->h_trace);
  delete b;
  return HX_OK;
}

// a launcher refused the shape (LDS plan over budget, grid too large ...): an error code, never a faulting launch
#define LAUNCH_TRY(expr)                                                                             \
  do {                                                                                               \
    if ((expr) != 0) return fail(HX_ERR_INVALID_ARG, "unsupported batch shape: %s", launch_error()); \
  } while (0)

// Two pairs per wavefront (hx_band2.hip) for a class of banded leaf pairs: scaled-probability policies only, every pair of
// the class admitted (DevJob::band_w32), and enough pairs that halving the wavefronts matters - below HX_BAND2_MIN_PAIRS a pair
// per wavefront with a converting wavefront beside it (hx_band.hip) has the shorter critical path.  HX_BAND2 = 0 / 1: never /
// whenever admissible (tests).
#define HX_BAND2_MIN_PAIRS 1024
static bool band2_wanted(bool linear, int n, int n_w32) {
  if (!linear || n_w32 != n) return false;
  if (const char* v = getenv("HX_BAND2")) return atoi(v) != 0;
  return n > HX_BAND2_MIN_PAIRS;
}

// The two-pairs-per-wavefront banded sweep leaves a CU's memory pipeline mostly idle (it is bound by vector issue), while its
// edge kernel - row 0 beyond the band: one 16-byte cell per kilobyte of a plane - is bound by the number of cache lines it
// touches.  The edge kernel therefore runs BESIDE the sweep, on the device's side stream: fork behind the preparation
// kernels, join before anything that follows the sweep on the caller's stream.  Without a side stream (creation failed) the
// two run one after the other on the caller's stream.
static hipStream_t side_fork(hx_batch* b, hipStream_t st) {
  DeviceTables& D = g_dev[b->device];
  if (getenv("HX_NO_SIDE_STREAM")) return st;
  if (!D.side_stream && hipStreamCreateWithFlags(&D.side_stream, hipStreamNonBlocking) != hipSuccess) { D.side_stream = nullptr; return st; }
  for (int e = 0; e < 2; ++e)
    if (!b->ev_side[e] && hipEventCreateWithFlags(&b->ev_side[e], hipEventDisableTiming) != hipSuccess) { b->ev_side[e] = nullptr; return st; }
  if (hipEventRecord(b->ev_side[0], st) != hipSuccess || hipStreamWaitEvent(D.side_stream, b->ev_side[0], 0) != hipSuccess) return st;
  return D.side_stream;
}
static int side_join(hx_batch* b, hipStream_t st, hipStream_t side) {
  if (side == st) return HX_OK;
  HIP_TRY(hipEventRecord(b->ev_side[1], side));
  HIP_TRY(hipStreamWaitEvent(st, b->ev_side[1], 0));
  return HX_OK;
}

// The 80-byte state records that the traceback and counting kernels read (and the general-profile fills, which get them with
// the preparation): built on first demand for a batch of leaf pairs, and with every preparation from then on.
static void ensure_state_records(hx_batch* b, hipStream_t st) {
  if (b->records_valid) return;
  launch_state_records(b->d_jobs, b->n_jobs, b->max_states, st);
  b->records_valid = true;
}

int hx_batch_forward(hx_batch* b, void* stream) {
  if (!b) return fail(HX_ERR_INVALID_ARG, "batch is null");
  int rc;
  if ((rc = use_device(b)) != HX_OK) return rc;
  const DeviceTables& D = g_dev[b->device];
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool fast = (b->flags & HX_LSE_FAST) != 0, linear = (b->flags & HX_LSE_LINEAR) == HX_LSE_LINEAR;
  const bool trunc = (b->flags & HX_LSE_TRUNC) == HX_LSE_TRUNC;
  const Tab16 lse_tab{fast ? D.fast_tab : D.pair_tab};    // FastPiece table, or the exact mode's {f0, df} pairs
  // the state records: with the preparation when a general-profile class will read them in the fill, else when a traceback or
  // counting kernel first asks (ensure_state_records)
  const bool records_now = b->cls[KC_DAG].n > 0 || b->cls[KC_DAG_BANDED].n > 0 || b->cls[KC_GENERIC].n > 0 || b->records_valid ||
                           (b->flags & HX_FORCE_GENERIC) != 0 || getenv("HX_FORCE_DAG") != nullptr;
  LAUNCH_TRY(launch_prep(b->d_jobs, b->n_jobs, b->max_states, b->max_cls, b->max_ca, b->max_cls_pairs, Tab8{D.tab}, records_now, st));
  b->records_valid = records_now;
  b->used_multi[0] = false;
  HIP_TRY(hipEventRecord(b->ev[0][0], st));
  // per-cell emission terms of the jobs without a class-pair table (general profiles).  Part of the fill:
  // the reference evaluates them inside its fill loop, so the launch is inside the timed region.
  if (!(b->flags & HX_FORCE_GENERIC)) launch_emission_plane(b->d_jobs, b->n_jobs, b->max_eplane, Tab8{D.tab}, Tab16{D.fast_tab}, fast && !getenv("HX_EXACT_EMISSION"), st);
  for (int c = 0; c < KC_COUNT; ++c) {
    const ClassRange& cr = b->cls[c];
    if (cr.n == 0) continue;
    const DevJob* jobs = b->d_jobs_cls + cr.begin;
    const bool banded = c == KC_LEAF_LDS_BANDED || c == KC_LEAF_BANDED || c == KC_CHAIN_BANDED || c == KC_DAG_BANDED;
    switch (c) {
      case KC_LEAF_ROT_BANDED:
        if (!(b->flags & (HX_SPARSE_ENVELOPE | HX_BAND_COMPRESSED))) launch_fill_neg_inf(b->d_fwd + cr.mat_begin, cr.mat_doubles, st);
        if (band2_wanted(linear, cr.n, cr.n_w32)) {
          const hipStream_t side = side_fork(b, st);
          LAUNCH_TRY(launch_forward_band2(jobs, cr.n, trunc, cr.max_rows, cr.max_cols, cr.max_cls, Tab8{D.tab}, Tab16{D.log_tab},
                                          (b->flags & (HX_SPARSE_ENVELOPE | HX_BAND_COMPRESSED)) != 0, st, side));
          if ((rc = side_join(b, st, side)) != HX_OK) return rc;
          launch_band2_result(jobs, cr.n, 0, Tab8{D.tab}, st);
        } else
        LAUNCH_TRY(launch_forward_band(jobs, cr.n, linear ? (trunc ? 3 : 0) : (fast ? 1 : 2), cr.max_rows, cr.max_cols, cr.max_cls, Tab8{D.tab},
                                       linear ? Tab16{D.log_tab} : lse_tab, (b->flags & (HX_SPARSE_ENVELOPE | HX_BAND_COMPRESSED)) != 0, st));
        break;
      case KC_LEAF_LDS: case KC_LEAF_LDS_BANDED: case KC_LEAF: case KC_LEAF_BANDED: case KC_CHAIN: case KC_CHAIN_BANDED: {
        // with a band the strip pipelines only visit in-envelope windows; everything else is -inf
        // (band-compressed matrices hold nothing but the swept windows, which the fill writes completely)
        if (banded && !(b->flags & (HX_SPARSE_ENVELOPE | HX_BAND_COMPRESSED)))
          launch_fill_neg_inf(b->d_fwd + cr.mat_begin, cr.mat_doubles, st);
        const int leaf = (c == KC_LEAF_LDS || c == KC_LEAF_LDS_BANDED) ? 2 : ((c == KC_LEAF || c == KC_LEAF_BANDED) ? 1 : 0);
        // HX_LSE_LINEAR on leaf pairs whose y side fits LDS: the recursion runs on scaled probabilities instead of
        // table log-sum-exps (hx_linear.hip)
        // a small batch of unbanded leaf pairs (one rank's share of a strong-scaling run): several workgroups per pair
        int multi = (leaf == 2 && !banded && cr.n <= HX_MULTI_COUNTER_PAIRS && !b->no_multi) ? chain_multi_groups(cr.n, cr.max_rows, linear ? 64 : 128) : 1;
        if (multi > 1) {
          b->used_multi[0] = true;
          if (!b->d_multi && hipMalloc(reinterpret_cast<void**>(&b->d_multi), 2 * HX_MULTI_COUNTER_PAIRS * 256 * sizeof(int)) != hipSuccess)
            return fail(HX_ERR_OUT_OF_MEMORY, "hipMalloc of the progress counters failed");
          HIP_TRY(hipMemsetAsync(b->d_multi, 0, (size_t)cr.n * 256 * sizeof(int), st));
        }
        if (linear && leaf == 2)
          LAUNCH_TRY(launch_forward_leaf_linear(jobs, cr.n, cr.max_rows, banded, Tab8{D.tab}, Tab16{D.log_tab}, cr.yl_cols, cr.yl_emis, cr.max_cls + 1,
                                                multi, b->d_multi, trunc, st));
        else
          LAUNCH_TRY(launch_forward_chain(jobs, cr.n, cr.max_rows, Tab8{D.tab}, lse_tab, fast, leaf, banded, cr.yl_cols, cr.yl_emis, multi, b->d_multi, st));
        break;
      }
      case KC_DAG: case KC_DAG_BANDED:
        // general profiles: the strip pipeline; with a band it only visits in-envelope windows, the rest is -inf
        if (banded) {
          launch_fill_neg_inf(b->d_fwd + cr.mat_begin, cr.mat_doubles, st);
          if (b->dag_linear) launch_dag_linear_clear(jobs, cr.n, st);
          else launch_fill_neg_inf(b->d_agg + cr.agg_begin, cr.agg_doubles, st);
        }
        {
          // a lone pair (or two) of more than sixteen strips: dealt to several workgroups, as the Backward fill is (below)
          int multi = 1, multi_waves = 4;
          const char* min_strips = getenv("HX_DAG_MULTI_MIN_STRIPS");
          if (cr.n <= HX_MULTI_MAX_PAIRS && cr.max_rows > (min_strips ? atoi(min_strips) : 16) * HX_STRIP && !getenv("HX_DAG_FWD_SINGLE") && !b->no_multi) {
            b->used_multi[0] = true;
            const int strips = (cr.max_rows + HX_STRIP - 1) / HX_STRIP;
            if (const char* e = getenv("HX_DAG_MULTI_WAVES")) multi_waves = atoi(e);
            // (the table policies' kernel is built for at most four waves per workgroup in this launch: 512 registers per lane)
            if ((multi_waves != 8 || !b->dag_linear) && multi_waves != 2) multi_waves = 4;
            multi = std::min(std::min(32, HX_MULTI_MAX_GROUPS / cr.n), (strips + multi_waves - 1) / multi_waves);
            if (!b->d_multi && hipMalloc(reinterpret_cast<void**>(&b->d_multi), 2 * HX_MULTI_COUNTER_PAIRS * 256 * sizeof(int)) != hipSuccess)
              return fail(HX_ERR_OUT_OF_MEMORY, "hipMalloc of the progress counters failed");
            HIP_TRY(hipMemsetAsync(b->d_multi, 0, HX_MULTI_MAX_PAIRS * 256 * sizeof(int), st));
          }
          if (b->dag_linear)
            LAUNCH_TRY(launch_forward_dag_linear(jobs, cr.n, cr.max_rows, Tab8{D.tab}, Tab16{D.log_tab}, multi, multi_waves, b->d_multi, st));
          else
            LAUNCH_TRY(launch_forward_dag_pipe(jobs, cr.n, cr.max_rows, Tab8{D.tab}, lse_tab, fast, multi, multi_waves, b->d_multi, st));
        }
        break;
      default:
        LAUNCH_TRY(launch_forward_dag(jobs, cr.n, cr.max_rows, Tab8{D.tab}, st));
    }
  }
  HIP_TRY(hipEventRecord(b->ev[0][1], st));
  HIP_TRY(hipGetLastError());
  b->ev_valid[0] = true;
  b->forward_done = true;
  b->last_stream = st;
  return HX_OK;
}

int hx_batch_backward(hx_batch* b, void* stream) {
  if (!b) return fail(HX_ERR_INVALID_ARG, "batch is null");
  if (!b->forward_done) return fail(HX_ERR_STATE, "hx_batch_backward needs a previous hx_batch_forward");
  if (b->any_compressed) return fail(HX_ERR_STATE, "the Backward fill does not support HX_BAND_COMPRESSED batches");
  int rc;
  if ((rc = use_device(b)) != HX_OK) return rc;
  const DeviceTables& D = g_dev[b->device];
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool fast = (b->flags & HX_LSE_FAST) != 0, linear = (b->flags & HX_LSE_LINEAR) == HX_LSE_LINEAR;
  const bool trunc = (b->flags & HX_LSE_TRUNC) == HX_LSE_TRUNC;
  const Tab16 lse_tab{fast ? D.fast_tab : D.pair_tab};
  ensure_state_records(b, st);                    // (the Backward fill of chain profiles runs the general pipeline, which reads them)
  if (!b->d_bwd) {
    // not pre-allocated with HX_KEEP_BACKWARD: allocate the Backward matrices now and re-publish the job tables
    HIP_TRY(hipStreamSynchronize(b->last_stream));
    if (hipMalloc(reinterpret_cast<void**>(&b->d_bwd), sizeof(double) * (size_t)b->fwd_total) != hipSuccess)
      return fail(HX_ERR_OUT_OF_MEMORY, "hipMalloc of %lld Backward-matrix bytes failed", (long long)(b->fwd_total * 8));
    for (int k = 0; k < b->n_jobs; ++k) b->jobs[k].bwd = b->d_bwd + (b->jobs[k].fwd - b->d_fwd);
    if ((rc = publish_jobs(b)) != HX_OK) return rc;
  } else if (st != b->last_stream) {
    // the Backward fill reads what the Forward launch prepared (and, for the posterior, follows it): order the streams
    HIP_TRY(hipStreamWaitEvent(st, b->ev[0][1], 0));
  }
  b->used_multi[1] = false;
  HIP_TRY(hipEventRecord(b->ev[1][0], st));
  for (int c = 0; c < KC_COUNT; ++c) {
    const ClassRange& cr = b->cls[c];
    if (cr.n == 0) continue;
    const DevJob* jobs = b->d_jobs_cls + cr.begin;
    const bool banded = c == KC_LEAF_LDS_BANDED || c == KC_LEAF_ROT_BANDED || c == KC_LEAF_BANDED || c == KC_CHAIN_BANDED || c == KC_DAG_BANDED;
    switch (c) {
      case KC_LEAF_LDS: case KC_LEAF_LDS_BANDED: case KC_LEAF_ROT_BANDED: case KC_LEAF: case KC_LEAF_BANDED: {
        if (banded && !(b->flags & HX_SPARSE_ENVELOPE)) launch_fill_neg_inf(b->d_bwd + cr.mat_begin, cr.mat_doubles, st);
        const int leaf = (c == KC_LEAF_LDS || c == KC_LEAF_LDS_BANDED || c == KC_LEAF_ROT_BANDED) ? 2 : 1;
        if (c == KC_LEAF_ROT_BANDED && cr.bwd_band && band2_wanted(linear, cr.n, cr.n_w32)) {
          const hipStream_t side = side_fork(b, st);
          LAUNCH_TRY(launch_backward_band2(jobs, cr.n, trunc, cr.max_rows, cr.max_cols, cr.max_cls, Tab8{D.tab}, Tab16{D.log_tab},
                                           (b->flags & HX_SPARSE_ENVELOPE) != 0, st, side));
          if ((rc = side_join(b, st, side)) != HX_OK) return rc;
          launch_band2_result(jobs, cr.n, 1, Tab8{D.tab}, st);
        } else if (c == KC_LEAF_ROT_BANDED && cr.bwd_band)
          // the rotating-row sweep in mirrored coordinates (hx_band.hip)
          LAUNCH_TRY(launch_backward_band(jobs, cr.n, linear ? (trunc ? 3 : 0) : (fast ? 1 : 2), cr.max_rows, cr.max_cols, cr.max_cls, Tab8{D.tab},
                                          linear ? Tab16{D.log_tab} : lse_tab, (b->flags & HX_SPARSE_ENVELOPE) != 0, st));
        else {
          int multi = (leaf == 2 && !banded && cr.n <= HX_MULTI_COUNTER_PAIRS && !b->no_multi) ? chain_multi_groups(cr.n, cr.max_rows, linear ? 64 : 128) : 1;
          int* counters = nullptr;
          if (multi > 1) {
            b->used_multi[1] = true;
            if (!b->d_multi && hipMalloc(reinterpret_cast<void**>(&b->d_multi), 2 * HX_MULTI_COUNTER_PAIRS * 256 * sizeof(int)) != hipSuccess)
              return fail(HX_ERR_OUT_OF_MEMORY, "hipMalloc of the progress counters failed");
            counters = b->d_multi + HX_MULTI_COUNTER_PAIRS * 256;
            HIP_TRY(hipMemsetAsync(counters, 0, (size_t)cr.n * 256 * sizeof(int), st));
          }
          if (linear && leaf == 2)
            LAUNCH_TRY(launch_backward_leaf_linear(jobs, cr.n, cr.max_rows, banded, Tab8{D.tab}, Tab16{D.log_tab}, cr.yl_cols, cr.yl_emis, cr.max_cls + 1,
                                                   multi, counters, trunc, st));
          else
            LAUNCH_TRY(launch_backward_chain(jobs, cr.n, cr.max_rows, Tab8{D.tab}, lse_tab, fast, leaf, banded, cr.yl_cols, cr.yl_emis, multi, counters, st));
        }
        break;
      }
      case KC_CHAIN: case KC_CHAIN_BANDED: case KC_DAG: case KC_DAG_BANDED:
        if (banded) launch_fill_neg_inf(b->d_bwd + cr.mat_begin, cr.mat_doubles, st);
        {
          // the state-record formulation needs the pair's scratch planes (general-profile classes have them: five planes
          // per pair - twelve in the scaled-probability mode - of no use once the Forward fill is done: a later Forward
          // launch rebuilds its per-state packs (k_lin_pack) and, with a band, clears the planes again (k_lin_clear))
          const bool records = (c == KC_DAG || c == KC_DAG_BANDED) && !getenv("HX_DAG_BWD_OLD");
          // a lone pair (or two) of more than sixteen strips: its strips dealt to two to four workgroups (hx_dag.hip
          // k_backward_dag_multi); their progress counters - the last 256 ints of each pair's scratch planes - start at zero
          int multi = 1, multi_waves = 4;
          const char* min_strips = getenv("HX_DAG_MULTI_MIN_STRIPS");      // tuning hook
          if (records && cr.n <= HX_MULTI_MAX_PAIRS && cr.max_rows > (min_strips ? atoi(min_strips) : 16) * HX_STRIP && !getenv("HX_DAG_BWD_SINGLE") && !b->no_multi) {
            b->used_multi[1] = true;
            const int strips = (cr.max_rows + HX_STRIP - 1) / HX_STRIP;
            if (const char* e = getenv("HX_DAG_MULTI_WAVES")) multi_waves = atoi(e);      // tuning hook: waves per workgroup (default 4: measured 1.24 / 1.13 / 1.02 / 1.01 s at one workgroup / 16 / 8 / 4 waves)
            if (multi_waves != 8 && multi_waves != 2) multi_waves = 4;      // (the kernel is built for at most eight: 256 registers per lane)
            multi = std::min(256 / multi_waves, (strips + multi_waves - 1) / multi_waves);   // (progress counters: 256 per pair)
            multi = std::min(multi, std::min(32, HX_MULTI_MAX_GROUPS / cr.n));
            for (int q = 0; q < cr.n && multi > 1; ++q) {
              const DevJob& Jh = b->jobs[b->order[cr.begin + q]];
              HIP_TRY(hipMemsetAsync(reinterpret_cast<int*>(Jh.agg + 5 * Jh.plane) - 256, 0, 256 * sizeof(int), st));
            }
          }
          LAUNCH_TRY(launch_backward_dag_pipe(jobs, cr.n, cr.max_rows, Tab8{D.tab}, lse_tab, fast, records, multi, multi_waves, st));
        }
        break;
      default:
        LAUNCH_TRY(launch_backward_dag(jobs, cr.n, cr.max_rows, Tab8{D.tab}, st));
    }
  }
  HIP_TRY(hipEventRecord(b->ev[1][1], st));
  HIP_TRY(hipGetLastError());
  b->ev_valid[1] = true;
  b->backward_done = true;
  b->last_stream = st;
  return HX_OK;
}

int hx_batch_sync(hx_batch* b) {
  if (!b) return fail(HX_ERR_INVALID_ARG, "batch is null");
  { const int rc_ = use_device(b); if (rc_ != HX_OK) return rc_; }
  HIP_TRY(hipStreamSynchronize(b->last_stream));
  return HX_OK;
}

int hx_batch_last_kernel_ms(hx_batch* b, int32_t which, float* ms) {
  if (!b || !ms || which < 0 || which > 1) return fail(HX_ERR_INVALID_ARG, "bad arguments");
  if (!b->ev_valid[which]) return fail(HX_ERR_STATE, "no such fill has been launched");
  { const int rc_ = use_device(b); if (rc_ != HX_OK) return rc_; }
  HIP_TRY(hipEventSynchronize(b->ev[which][1]));
  HIP_TRY(hipEventElapsedTime(ms, b->ev[which][0], b->ev[which][1]));
  return HX_OK;
}

static int read_scalars(hx_batch* b, double* out, int which) {
  if (!b || !out) return fail(HX_ERR_INVALID_ARG, "bad arguments");
  if (which == 0 ? !b->forward_done : !b->backward_done) return fail(HX_ERR_STATE, "fill has not been launched");
  int rc;
  if ((rc = use_device(b)) != HX_OK) return rc;
  HIP_TRY(hipStreamSynchronize(b->last_stream));
  HIP_TRY(hipMemcpy(out, b->d_arena + (which == 0 ? b->lp_end_off : b->lp_start_off), sizeof(double) * b->n_jobs, hipMemcpyDeviceToHost));
  return HX_OK;
}

// A fill dealt to several workgroups per pair marks a pair whose waves gave up waiting for one another (a bounded number of
// polls, never a hang: hx_chain.hip / hx_linear.hip / hx_dag.hip / hx_daglin.hip MULTI) with NaN; no fill produces NaN
// otherwise.  That is a scheduling accident - not every workgroup of the launch was resident - and not the caller's problem:
// the batch's fills are launched again with one workgroup per pair (from then on), once; only if NaN remains is it an error.
static int first_nan(const hx_batch* b, const double* v) {
  for (int k = 0; k < b->n_jobs; ++k)
    if (v[k] != v[k]) return k;
  return -1;
}
static int relaunch_single(hx_batch* b, bool backward_too) {
  if (getenv("HX_NO_RELAUNCH")) return HX_ERR_HIP;      // (tests: see the raw outcome)
  b->no_multi = true;
  b->relaunches++;
  int rc = hx_batch_forward(b, b->last_stream);
  if (rc == HX_OK && backward_too) rc = hx_batch_backward(b, b->last_stream);
  return rc;
}
int hx_batch_lp_end(hx_batch* b, double* out) {
  int rc = read_scalars(b, out, 0);
  if (rc != HX_OK) return rc;
  int k = first_nan(b, out);
  if (k >= 0 && b->used_multi[0] && !b->no_multi) {
    if ((rc = relaunch_single(b, b->backward_done)) != HX_OK) return fail(HX_ERR_HIP, "the Forward fill of pair %d did not complete (workgroups lost one another) and could not be launched again", k);
    if ((rc = read_scalars(b, out, 0)) != HX_OK) return rc;
    k = first_nan(b, out);
  }
  if (k >= 0) return fail(HX_ERR_HIP, "the Forward fill of pair %d did not complete (workgroups lost one another)", k);
  return HX_OK;
}
int hx_batch_lp_start(hx_batch* b, double* out) {
  int rc = read_scalars(b, out, 1);
  if (rc != HX_OK) return rc;
  int k = first_nan(b, out);
  if (k >= 0 && b->used_multi[1] && !b->no_multi) {
    if ((rc = relaunch_single(b, true)) != HX_OK) return fail(HX_ERR_HIP, "the Backward fill of pair %d did not complete (workgroups lost one another) and could not be launched again", k);
    if ((rc = read_scalars(b, out, 1)) != HX_OK) return rc;
    k = first_nan(b, out);
  }
  if (k >= 0) return fail(HX_ERR_HIP, "the Backward fill of pair %d did not complete (workgroups lost one another)", k);
  return HX_OK;
}

int hx_batch_relaunches(const hx_batch* b) {
  if (!b) return fail(HX_ERR_INVALID_ARG, "bad arguments");
  return b->relaunches;
}

int hx_batch_layout(const hx_batch* b, int32_t job, int32_t which, hx_layout* out) {
  if (!b || !out || which < 0 || which > 1) return fail(HX_ERR_INVALID_ARG, "bad arguments");
  if (job < 0 || job >= b->n_jobs) return fail(HX_ERR_RANGE, "job %d out of range", job);
  *out = b->layouts[job];
  out->mirrored = which;      // the Backward matrix is stored in mirrored coordinates
  return HX_OK;
}

int hx_batch_strip_windows(const hx_batch* b, int32_t job, int32_t* windows, int64_t* bases) {
  if (!b || !windows || !bases) return fail(HX_ERR_INVALID_ARG, "bad arguments");
  if (job < 0 || job >= b->n_jobs) return fail(HX_ERR_RANGE, "job %d out of range", job);
  const DevJob& J = b->jobs[job];
  if (!J.strip_base) return fail(HX_ERR_STATE, "job %d is not stored band-compressed", job);
  { const int rc_ = use_device(b); if (rc_ != HX_OK) return rc_; }
  HIP_TRY(hipMemcpy(windows, J.fwd_windows, sizeof(int32_t) * 4 * J.n_strips, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(bases, J.strip_base, sizeof(int64_t) * 2 * J.n_strips, hipMemcpyDeviceToHost));
  return HX_OK;
}

int64_t hx_batch_total_cells(const hx_batch* b) { return b ? b->total_cells : 0; }

int hx_batch_job_kernel(const hx_batch* b, int32_t job, int32_t* forward_class, int32_t* backward_sweep) {
  if (!b) return fail(HX_ERR_INVALID_ARG, "bad arguments");
  if (job < 0 || job >= b->n_jobs) return fail(HX_ERR_RANGE, "job %d out of range", job);
  for (int c = 0; c < KC_COUNT; ++c) {
    const ClassRange& cr = b->cls[c];
    for (int p = cr.begin; p < cr.begin + cr.n; ++p)
      if (b->order[p] == job) {
        if (forward_class) *forward_class = c;
        if (backward_sweep) *backward_sweep = (c == KC_LEAF_ROT_BANDED && cr.bwd_band) ? 1 : 0;
        return HX_OK;
      }
  }
  return fail(HX_ERR_INVALID_ARG, "job %d is in no kernel class", job);
}

int hx_batch_shared_wavefront_pairs(const hx_batch* b) {
  if (!b) return fail(HX_ERR_INVALID_ARG, "bad arguments");
  const ClassRange& cr = b->cls[KC_LEAF_ROT_BANDED];
  return band2_wanted((b->flags & HX_LSE_LINEAR) == HX_LSE_LINEAR, cr.n, cr.n_w32) ? cr.n : 0;
}

static const double* matrix_of(hx_batch* b, int job, int which) {
  return which == 0 ? b->jobs[job].fwd : b->jobs[job].bwd;
}

int hx_batch_read_matrix(hx_batch* b, int32_t job, int32_t which, double* out) {
  if (!b || !out || which < 0 || which > 1) return fail(HX_ERR_INVALID_ARG, "bad arguments");
  if (job < 0 || job >= b->n_jobs) return fail(HX_ERR_RANGE, "job %d out of range", job);
  if (which == 0 ? !b->forward_done : !b->backward_done) return fail(HX_ERR_STATE, "fill has not been launched");
  { const int rc_ = use_device(b); if (rc_ != HX_OK) return rc_; }
  HIP_TRY(hipStreamSynchronize(b->last_stream));
  HIP_TRY(hipMemcpy(out, matrix_of(b, job, which), sizeof(double) * (size_t)b->jobs[job].matrix_doubles, hipMemcpyDeviceToHost));
  return HX_OK;
}

int hx_batch_read_matrix_async(hx_batch* b, int32_t job, int32_t which, double* out) {
  if (!b || !out || which < 0 || which > 1) return fail(HX_ERR_INVALID_ARG, "bad arguments");
  if (job < 0 || job >= b->n_jobs) return fail(HX_ERR_RANGE, "job %d out of range", job);
  if (which == 0 ? !b->forward_done : !b->backward_done) return fail(HX_ERR_STATE, "fill has not been launched");
  { const int rc_ = use_device(b); if (rc_ != HX_OK) return rc_; }
  HIP_TRY(hipStreamSynchronize(b->last_stream));
  if (!b->copy_stream) {
    DeviceTables& D = g_dev[b->device];
    std::lock_guard<std::mutex> lock(D.copy_mutex);
    if (!D.copy_stream) HIP_TRY(hipStreamCreateWithFlags(&D.copy_stream, hipStreamNonBlocking));
    b->copy_stream = D.copy_stream;
  }
  if (b->copied[which].empty()) b->copied[which].assign((size_t)b->n_jobs, nullptr);
  hipEvent_t& e = b->copied[which][job];
  if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIP_TRY(hipMemcpyAsync(out, matrix_of(b, job, which), sizeof(double) * (size_t)b->jobs[job].matrix_doubles, hipMemcpyDeviceToHost, b->copy_stream));
  HIP_TRY(hipEventRecord(e, b->copy_stream));
  return HX_OK;
}

int hx_batch_wait_read(hx_batch* b, int32_t job, int32_t which) {
  if (!b || which < 0 || which > 1) return fail(HX_ERR_INVALID_ARG, "bad arguments");
  if (job < 0 || job >= b->n_jobs) return fail(HX_ERR_RANGE, "job %d out of range", job);
  if (b->copied[which].empty() || !b->copied[which][job]) return fail(HX_ERR_STATE, "no asynchronous read of job %d was started", job);
  { const int rc_ = use_device(b); if (rc_ != HX_OK) return rc_; }
  HIP_TRY(hipEventSynchronize(b->copied[which][job]));
  return HX_OK;
}

// ---------------------------------------------------------------------------
// guide-alignment Viterbi batch (hx_quick.hip)
// ---------------------------------------------------------------------------
struct hx_quick_batch {
  int device = 0;
  int n_jobs = 0;
  std::vector<DevQuick> jobs;
  std::vector<hx_layout> layouts;
  DevQuick* d_jobs = nullptr;
  char* d_arena = nullptr;
  double* d_cells = nullptr;
  int max_rows = 0, max_cols = 0;
  bool all_full = true;
  bool done = false;
  int64_t total_cells = 0;
  hipStream_t last_stream = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};
  size_t res_off = 0, xy_off = 0;
};

static int quick_create_impl(int device, const hx_quick_job* jobs, int32_t n_jobs, hx_quick_batch** out) {
  if (!out) return fail(HX_ERR_INVALID_ARG, "out is null");
  *out = nullptr;
  if (device < 0 || device >= HX_MAX_DEVICES || !g_dev[device].ready)
    return fail(HX_ERR_NOT_INITIALIZED, "hx_init has not been called for device %d", device);
  if (!jobs || n_jobs <= 0) return fail(HX_ERR_INVALID_ARG, "no jobs");
  HIP_TRY(hipSetDevice(device));
  hx_quick_batch* b = new (std::nothrow) hx_quick_batch;
  if (!b) return fail(HX_ERR_OUT_OF_MEMORY, "out of host memory");
  b->device = device;
  b->n_jobs = n_jobs;
  b->jobs.resize(n_jobs);
  b->layouts.resize(n_jobs);
  struct Off { size_t xtok, ytok, submat, env, best, bestj, cols; bool has_env; };
  std::vector<Off> offs(n_jobs);
  std::vector<int64_t> cell_off(n_jobs);
  Arena ar;
  int64_t cells_total = 0;
  int rc = HX_OK;
  for (int k = 0; k < n_jobs && rc == HX_OK; ++k) {
    const hx_quick_job& q = jobs[k];
    if (q.x_len <= 0 || q.y_len <= 0 || !q.x_tok || !q.y_tok) { rc = fail(HX_ERR_INVALID_ARG, "job %d: empty sequence", k); break; }
    if (q.alph_size <= 0 || q.alph_size > 31 || !q.submat) { rc = fail(HX_ERR_INVALID_ARG, "job %d: alphabet size must be 1..31", k); break; }
    for (int p = 0; p < q.x_len && rc == HX_OK; ++p)
      if (q.x_tok[p] >= q.alph_size) rc = fail(HX_ERR_RANGE, "job %d: x token %d out of range", k, q.x_tok[p]);
    for (int p = 0; p < q.y_len && rc == HX_OK; ++p)
      if (q.y_tok[p] >= q.alph_size) rc = fail(HX_ERR_RANGE, "job %d: y token %d out of range", k, q.y_tok[p]);
    if (rc != HX_OK) break;
    Off& o = offs[k];
    o.xtok = ar.put(q.x_tok, sizeof(int32_t) * q.x_len);
    o.ytok = ar.put(q.y_tok, sizeof(int32_t) * q.y_len);
    o.submat = ar.put(q.submat, sizeof(double) * q.alph_size * q.alph_size);
    o.has_env = q.diagonals != nullptr;
    o.env = 0;
    if (o.has_env) {
      std::vector<uint8_t> bits((size_t)q.x_len + q.y_len + 1, 0);
      for (int d = 0; d < q.n_diagonals; ++d) {
        const int64_t idx = (int64_t)q.diagonals[d] + q.y_len;
        if (idx < 0 || idx >= (int64_t)bits.size()) { rc = fail(HX_ERR_RANGE, "job %d: diagonal %d out of range", k, q.diagonals[d]); break; }
        bits[(size_t)idx] = 1;
      }
      if (rc != HX_OK) break;
      o.env = ar.put(bits.data(), bits.size());
      b->all_full = false;
    }
    o.best = ar.reserve(sizeof(double) * q.x_len);
    o.bestj = ar.reserve(sizeof(int32_t) * q.x_len);
    DevQuick& J = b->jobs[k];
    memset(&J, 0, sizeof(J));
    J.xlen = q.x_len; J.ylen = q.y_len; J.alph = q.alph_size;
    J.n_strips = (q.x_len + HX_STRIP - 1) / HX_STRIP;
    // the three states of a step pair adjacent (a wavefront writes 3 KiB contiguous per iteration), unless
    // HX_PLANAR_LAYOUT asks for separate state planes
    const bool planar = getenv("HX_PLANAR_LAYOUT") != nullptr;
    J.strip_stride = strip_stride_for(q.y_len) * (planar ? 1 : 3);
    J.plane = planar ? J.n_strips * J.strip_stride : 2 * HX_STRIP;
    J.blk = planar ? 2 * HX_STRIP : 6 * HX_STRIP;
    const int64_t matrix_doubles = (planar ? 3 : 1) * (int64_t)J.n_strips * J.strip_stride;
    for (int s = 0; s < 11; ++s) J.sc[s] = q.scores[s];
    hx_layout& L = b->layouts[k];
    L.n_rows = q.x_len; L.n_cols = q.y_len; L.strip_rows = HX_STRIP; L.n_strips = J.n_strips;
    L.strip_stride = J.strip_stride; L.plane_stride = J.plane; L.mirrored = 0; L.compressed = 0;
    L.block_stride = J.blk; L.matrix_doubles = matrix_doubles;
    cell_off[k] = cells_total;
    cells_total += matrix_doubles;
    b->total_cells += (int64_t)q.x_len * q.y_len;
    if (q.x_len > b->max_rows) b->max_rows = q.x_len;
    if (q.y_len > b->max_cols) b->max_cols = q.y_len;
  }
  if (rc != HX_OK) { delete b; return rc; }
  // per-column constants of the pairs in global memory when the longest y does not fit the kernel's LDS tables
  for (int k = 0; k < n_jobs; ++k)
    offs[k].cols = b->max_cols > 7000 ? ar.reserve((sizeof(double) * 2 + sizeof(int32_t)) * (size_t)jobs[k].y_len + 16) : 0;
  // scores and end coordinates of all pairs, contiguous: one copy back per batch
  b->res_off = ar.reserve(sizeof(double) * n_jobs);
  b->xy_off = ar.reserve(sizeof(int32_t) * 2 * n_jobs);
  auto cleanup = [&](int code) { hx_quick_batch_destroy(b); return code; };
  if (hipMalloc(reinterpret_cast<void**>(&b->d_arena), ar.host.size() + 256) != hipSuccess)
    return cleanup(fail(HX_ERR_OUT_OF_MEMORY, "hipMalloc of %zu input bytes failed", ar.host.size()));
  if (hipMemcpy(b->d_arena, ar.host.data(), ar.host.size(), hipMemcpyHostToDevice) != hipSuccess)
    return cleanup(fail(HX_ERR_HIP, "input upload failed"));
  if (hipMalloc(reinterpret_cast<void**>(&b->d_cells), sizeof(double) * (size_t)cells_total) != hipSuccess)
    return cleanup(fail(HX_ERR_OUT_OF_MEMORY, "hipMalloc of %lld matrix bytes failed", (long long)(cells_total * 8)));
  for (int k = 0; k < n_jobs; ++k) {
    DevQuick& J = b->jobs[k];
    const Off& o = offs[k];
    char* base = b->d_arena;
    J.xtok = reinterpret_cast<int32_t*>(base + o.xtok);
    J.ytok = reinterpret_cast<int32_t*>(base + o.ytok);
    J.submat = reinterpret_cast<double*>(base + o.submat);
    J.in_env = o.has_env ? reinterpret_cast<uint8_t*>(base + o.env) : nullptr;
    J.col_scratch = b->max_cols > 7000 ? reinterpret_cast<double*>(base + o.cols) : nullptr;
    J.best_score = reinterpret_cast<double*>(base + o.best);
    J.best_j = reinterpret_cast<int32_t*>(base + o.bestj);
    J.result = reinterpret_cast<double*>(base + b->res_off) + k;
    J.xy_end = reinterpret_cast<int32_t*>(base + b->xy_off) + 2 * k;
    J.cells = b->d_cells + cell_off[k];
  }
  if (hipMalloc(reinterpret_cast<void**>(&b->d_jobs), sizeof(DevQuick) * n_jobs) != hipSuccess)
    return cleanup(fail(HX_ERR_OUT_OF_MEMORY, "hipMalloc of job table failed"));
  if (hipMemcpy(b->d_jobs, b->jobs.data(), sizeof(DevQuick) * n_jobs, hipMemcpyHostToDevice) != hipSuccess)
    return cleanup(fail(HX_ERR_HIP, "job table upload failed"));
  for (int e = 0; e < 2; ++e)
    if (hipEventCreate(&b->ev[e]) != hipSuccess) return cleanup(fail(HX_ERR_HIP, "hipEventCreate failed"));
  *out = b;
  return HX_OK;
}

int hx_quick_batch_create_on(int device, const hx_quick_job* jobs, int32_t n_jobs, hx_quick_batch** out) {
  try {
    return quick_create_impl(device, jobs, n_jobs, out);
  } catch (const std::bad_alloc&) {
    if (out) *out = nullptr;
    return fail(HX_ERR_OUT_OF_MEMORY, "host allocation failed while building the batch");
  }
}

int hx_quick_batch_create(const hx_quick_job* jobs, int32_t n_jobs, hx_quick_batch** out) {
  if (g_device < 0) { if (out) *out = nullptr; return fail(HX_ERR_NOT_INITIALIZED, "hx_init has not been called"); }
  return hx_quick_batch_create_on(g_device, jobs, n_jobs, out);
}

int hx_quick_batch_destroy(hx_quick_batch* b) {
  if (!b) return HX_OK;
  (void)hipSetDevice(b->device);
  (void)hipDeviceSynchronize();
  for (int e = 0; e < 2; ++e)
    if (b->ev[e]) (void)hipEventDestroy(b->ev[e]);
  if (b->d_jobs) (void)hipFree(b->d_jobs);
  if (b->d_arena) (void)hipFree(b->d_arena);
  if (b->d_cells) (void)hipFree(b->d_cells);
  delete b;
  return HX_OK;
}

int hx_quick_batch_run(hx_quick_batch* b, void* stream) {
  if (!b) return fail(HX_ERR_INVALID_ARG, "batch is null");
  if (hipSetDevice(b->device) != hipSuccess) return fail(HX_ERR_HIP, "hipSetDevice(%d) failed", b->device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  HIP_TRY(hipEventRecord(b->ev[0], st));
  LAUNCH_TRY(launch_quickalign(b->d_jobs, b->n_jobs, b->max_rows, b->max_cols, b->all_full, st));
  HIP_TRY(hipEventRecord(b->ev[1], st));
  HIP_TRY(hipGetLastError());
  b->done = true;
  b->last_stream = st;
  return HX_OK;
}

int hx_quick_batch_results(hx_quick_batch* b, double* score, int32_t* x_end, int32_t* y_end) {
  if (!b) return fail(HX_ERR_INVALID_ARG, "batch is null");
  if (!b->done) return fail(HX_ERR_STATE, "hx_quick_batch_run has not been launched");
  if (hipSetDevice(b->device) != hipSuccess) return fail(HX_ERR_HIP, "hipSetDevice(%d) failed", b->device);
  HIP_TRY(hipStreamSynchronize(b->last_stream));
  if (score) HIP_TRY(hipMemcpy(score, b->d_arena + b->res_off, sizeof(double) * b->n_jobs, hipMemcpyDeviceToHost));
  std::vector<int32_t> xy(2 * (size_t)b->n_jobs);
  HIP_TRY(hipMemcpy(xy.data(), b->d_arena + b->xy_off, sizeof(int32_t) * xy.size(), hipMemcpyDeviceToHost));
  for (int k = 0; k < b->n_jobs; ++k) {
    if (x_end) x_end[k] = xy[2 * k];
    if (y_end) y_end[k] = xy[2 * k + 1];
  }
  return HX_OK;
}

int hx_quick_batch_layout(const hx_quick_batch* b, int32_t job, hx_layout* out) {
  if (!b || !out) return fail(HX_ERR_INVALID_ARG, "bad arguments");
  if (job < 0 || job >= b->n_jobs) return fail(HX_ERR_RANGE, "job %d out of range", job);
  *out = b->layouts[job];
  return HX_OK;
}

int hx_quick_batch_read_matrix(hx_quick_batch* b, int32_t job, double* out) {
  if (!b || !out) return fail(HX_ERR_INVALID_ARG, "bad arguments");
  if (job < 0 || job >= b->n_jobs) return fail(HX_ERR_RANGE, "job %d out of range", job);
  if (!b->done) return fail(HX_ERR_STATE, "hx_quick_batch_run has not been launched");
  if (hipSetDevice(b->device) != hipSuccess) return fail(HX_ERR_HIP, "hipSetDevice(%d) failed", b->device);
  HIP_TRY(hipStreamSynchronize(b->last_stream));
  HIP_TRY(hipMemcpy(out, b->jobs[job].cells, sizeof(double) * (size_t)b->layouts[job].matrix_doubles, hipMemcpyDeviceToHost));
  return HX_OK;
}

int64_t hx_quick_batch_total_cells(const hx_quick_batch* b) { return b ? b->total_cells : 0; }

int hx_quick_batch_last_kernel_ms(hx_quick_batch* b, float* ms) {
  if (!b || !ms) return fail(HX_ERR_INVALID_ARG, "bad arguments");
  if (!b->done) return fail(HX_ERR_STATE, "hx_quick_batch_run has not been launched");
  if (hipSetDevice(b->device) != hipSuccess) return fail(HX_ERR_HIP, "hipSetDevice(%d) failed", b->device);
  HIP_TRY(hipEventSynchronize(b->ev[1]));
  HIP_TRY(hipEventElapsedTime(ms, b->ev[0], b->ev[1]));
  return HX_OK;
}

int hx_host_alloc(size_t bytes, void** out) {
  if (!out) return fail(HX_ERR_INVALID_ARG, "out is null");
  *out = nullptr;
  if (bytes == 0) return HX_OK;
  if (hipHostMalloc(out, bytes, hipHostMallocPortable) != hipSuccess) {     // (usable from every device of the process)
    *out = nullptr;
    return fail(HX_ERR_OUT_OF_MEMORY, "hipHostMalloc of %zu bytes failed", bytes);
  }
  return HX_OK;
}

int hx_host_free(void* p) {
  if (p) HIP_TRY(hipHostFree(p));
  return HX_OK;
}

int hx_batch_read_cells(hx_batch* b, int32_t job, int32_t which, const int32_t* ij, int64_t n, double* out) {
  if (!b || !out || !ij || n < 0 || which < 0 || which > 1) return fail(HX_ERR_INVALID_ARG, "bad arguments");
  if (job < 0 || job >= b->n_jobs) return fail(HX_ERR_RANGE, "job %d out of range", job);
  if (which == 0 ? !b->forward_done : !b->backward_done) return fail(HX_ERR_STATE, "fill has not been launched");
  if (n == 0) return HX_OK;
  { const int rc_ = use_device(b); if (rc_ != HX_OK) return rc_; }
  int32_t* d_ij = nullptr;
  double* d_out = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_ij), sizeof(int32_t) * 2 * n));
  if (hipMalloc(reinterpret_cast<void**>(&d_out), sizeof(double) * 5 * n) != hipSuccess) {
    (void)hipFree(d_ij);
    return fail(HX_ERR_OUT_OF_MEMORY, "hipMalloc failed");
  }
  int rc = HX_OK;
  if (hipMemcpy(d_ij, ij, sizeof(int32_t) * 2 * n, hipMemcpyHostToDevice) != hipSuccess) rc = fail(HX_ERR_HIP, "upload failed");
  if (rc == HX_OK) {
    launch_gather_cells(b->d_jobs, job, matrix_of(b, job, which), which, d_ij, n, d_out, b->last_stream);
    if (hipStreamSynchronize(b->last_stream) != hipSuccess ||
        hipMemcpy(out, d_out, sizeof(double) * 5 * n, hipMemcpyDeviceToHost) != hipSuccess)
      rc = fail(HX_ERR_HIP, "cell gather failed");
  }
  (void)hipFree(d_ij);
  (void)hipFree(d_out);
  return rc;
}

// Device-side ForwardMatrix::bestTrace for every job of the batch (hx_trace.hip).
int hx_batch_best_trace(hx_batch* b, hx_trace_cell* cells, int64_t cap, int32_t* n_cells) {
  if (!b || !cells || !n_cells || cap < 1) return fail(HX_ERR_INVALID_ARG, "bad arguments");
  if (!b->forward_done) return fail(HX_ERR_STATE, "hx_batch_best_trace needs a previous hx_batch_forward");
  { const int rc_ = use_device(b); if (rc_ != HX_OK) return rc_; }
  const int n = b->n_jobs;
  // device buffers and the page-locked staging buffer are kept with the batch (a host mirror asks once per fill batch,
  // a benchmark many times)
  if (cap > b->trace_cap) {
    if (b->d_trace) (void)hipFree(b->d_trace);
    if (b->d_trace_n) (void)hipFree(b->d_trace_n);
    if (b->h_trace) (void)hipHostFree(b->h_trace);
    b->d_trace = nullptr; b->d_trace_n = nullptr; b->h_trace = nullptr; b->trace_cap = 0;
    // [n][cap][3] as walked (END cell first) + the same amount for the compacted, start-first copy; [2n] lengths + offsets
    if (hipMalloc(reinterpret_cast<void**>(&b->d_trace), sizeof(int32_t) * 6 * (size_t)cap * n) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&b->d_trace_n), sizeof(int64_t) * 3 * (size_t)n) != hipSuccess ||      // lengths, offsets, near-tie flags
        hipHostMalloc(&b->h_trace, sizeof(int32_t) * 3 * (size_t)cap * n, hipHostMallocDefault) != hipSuccess)
      return fail(HX_ERR_OUT_OF_MEMORY, "allocating %zu path bytes failed", sizeof(int32_t) * 9 * (size_t)cap * n);
    b->trace_cap = cap;
  }
  int32_t* d_paths = b->d_trace;
  int32_t* d_out = b->d_trace + 3 * (size_t)cap * n;
  int32_t* d_n = reinterpret_cast<int32_t*>(b->d_trace_n);
  int64_t* d_off = b->d_trace_n + n;
  int32_t* d_ties = reinterpret_cast<int32_t*>(b->d_trace_n + 2 * (size_t)n);
  b->trace_ties_valid = false;
  hipStream_t st = b->last_stream;
  // the per-cell emission plane is only filled by the strip pipelines (hx_batch_forward)
  ensure_state_records(b, st);
  launch_best_trace(b->d_jobs, n, d_paths, cap, d_n, Tab8{g_dev[b->device].tab}, !(b->flags & HX_FORCE_GENERIC), d_ties, st);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess ||
      hipMemcpy(n_cells, d_n, sizeof(int32_t) * n, hipMemcpyDeviceToHost) != hipSuccess)
    return fail(HX_ERR_HIP, "best-trace kernel failed: %s", hipGetErrorString(hipGetLastError()));
  b->trace_ties_valid = true;
  std::vector<int64_t> off((size_t)n + 1, 0);
  for (int k = 0; k < n; ++k) {
    if (n_cells[k] == -3) return fail(HX_ERR_RANGE, "path of job %d does not fit %lld cells", k, (long long)cap);
    off[k + 1] = off[k] + (n_cells[k] > 0 ? n_cells[k] : 0);
  }
  if (off[n] == 0) return HX_OK;
  // the kernel walks from the END cell backwards; the reference's Path starts at the start cell: reverse and compact on
  // the device, one copy over PCIe, scatter on the host
  HIP_TRY(hipMemcpy(d_off, off.data(), sizeof(int64_t) * n, hipMemcpyHostToDevice));
  launch_reverse_paths(d_paths, cap, d_n, d_off, d_out, n, st);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(b->h_trace, d_out, sizeof(int32_t) * 3 * (size_t)off[n], hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  const hx_trace_cell* src = static_cast<const hx_trace_cell*>(b->h_trace);
  for (int k = 0; k < n; ++k)
    if (n_cells[k] > 0) memcpy(cells + (size_t)cap * k, src + off[k], sizeof(hx_trace_cell) * (size_t)n_cells[k]);
  return HX_OK;
}

int hx_batch_best_trace_ties(hx_batch* b, int32_t* near_tie) {
  if (!b || !near_tie) return fail(HX_ERR_INVALID_ARG, "bad arguments");
  if (!b->trace_ties_valid) return fail(HX_ERR_STATE, "hx_batch_best_trace_ties needs a previous hx_batch_best_trace");
  { const int rc_ = use_device(b); if (rc_ != HX_OK) return rc_; }
  HIP_TRY(hipMemcpy(near_tie, reinterpret_cast<const int32_t*>(b->d_trace_n + 2 * (size_t)b->n_jobs), sizeof(int32_t) * b->n_jobs, hipMemcpyDeviceToHost));
  return HX_OK;
}

int hx_batch_sample_traces(hx_batch* b, int32_t job, int32_t n_walks, const double* uniforms, int64_t n_uniforms,
                           hx_trace_cell* cells, int64_t cap, int32_t* n_cells, int64_t* draws_used) {
  if (!b || !uniforms || !cells || !n_cells || !draws_used || n_walks < 1 || n_uniforms < 0 || cap < 1)
    return fail(HX_ERR_INVALID_ARG, "bad arguments");
  if (job < 0 || job >= b->n_jobs) return fail(HX_ERR_RANGE, "job %d out of range", job);
  if (!b->forward_done) return fail(HX_ERR_STATE, "hx_batch_sample_traces needs a previous hx_batch_forward");
  { const int rc_ = use_device(b); if (rc_ != HX_OK) return rc_; }
  struct Dev {                                       // released on every way out
    void* p = nullptr;
    ~Dev() { if (p) (void)hipFree(p); }
  } d_u, d_paths, d_n, d_draws;
  const size_t path_ints = 3 * (size_t)cap * (size_t)n_walks;
  if (hipMalloc(&d_u.p, sizeof(double) * (size_t)(n_uniforms > 0 ? n_uniforms : 1)) != hipSuccess ||
      hipMalloc(&d_paths.p, sizeof(int32_t) * path_ints) != hipSuccess ||
      hipMalloc(&d_n.p, sizeof(int32_t) * (size_t)n_walks) != hipSuccess ||
      hipMalloc(&d_draws.p, sizeof(int64_t) * (size_t)n_walks) != hipSuccess)
    return fail(HX_ERR_OUT_OF_MEMORY, "allocating the buffers of %d sampled walks failed", n_walks);
  hipStream_t st = b->last_stream;
  HIP_TRY(hipMemcpyAsync(d_u.p, uniforms, sizeof(double) * (size_t)n_uniforms, hipMemcpyHostToDevice, st));
  ensure_state_records(b, st);
  launch_sample_traces(b->d_jobs, job, n_walks, static_cast<const double*>(d_u.p), n_uniforms, static_cast<int32_t*>(d_paths.p), cap,
                       static_cast<int32_t*>(d_n.p), static_cast<int64_t*>(d_draws.p), Tab8{g_dev[b->device].tab},
                       !(b->flags & HX_FORCE_GENERIC), st);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(n_cells, d_n.p, sizeof(int32_t) * (size_t)n_walks, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(draws_used, d_draws.p, sizeof(int64_t) * (size_t)n_walks, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  // the kernel walks from the END cell backwards; the reference's Path starts at the start cell
  std::vector<int32_t> raw;
  for (int w = 0; w < n_walks; ++w) {
    const int n = n_cells[w];
    if (n <= 0) continue;
    raw.resize(3 * (size_t)n);
    HIP_TRY(hipMemcpy(raw.data(), static_cast<const int32_t*>(d_paths.p) + 3 * (size_t)cap * w, sizeof(int32_t) * 3 * (size_t)n, hipMemcpyDeviceToHost));
    hx_trace_cell* dst = cells + (size_t)cap * w;
    for (int c = 0; c < n; ++c) {
      const int32_t* t = &raw[3 * (size_t)(n - 1 - c)];
      dst[c].xpos = t[0]; dst[c].ypos = t[1]; dst[c].state = t[2];
    }
  }
  return HX_OK;
}

int hx_batch_indel_counts(hx_batch* b, int32_t job, const double* branch_times, double* out) {
  if (!b || !branch_times || !out) return fail(HX_ERR_INVALID_ARG, "bad arguments");
  if (job < 0 || job >= b->n_jobs) return fail(HX_ERR_RANGE, "job %d out of range", job);
  if (!b->forward_done || !b->backward_done) return fail(HX_ERR_STATE, "hx_batch_indel_counts needs the Forward and the Backward fill");
  { const int rc_ = use_device(b); if (rc_ != HX_OK) return rc_; }
  hipStream_t st = b->last_stream;
  double* d = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&d), 12 * sizeof(double)) != hipSuccess) return fail(HX_ERR_OUT_OF_MEMORY, "device allocation failed");
  int rc = HX_OK;
  double host[12];
  for (int k = 0; k < 6; ++k) { host[k] = branch_times[k]; host[6 + k] = 0.; }
  if (hipMemcpyAsync(d, host, sizeof(host), hipMemcpyHostToDevice, st) != hipSuccess) rc = HX_ERR_HIP;
  if (rc == HX_OK) {
    const DevJob& J = b->jobs[job];
    ensure_state_records(b, st);
    launch_indel_counts(b->d_jobs, job, d, d + 6, (int64_t)J.n_rows * J.n_cols, Tab8{g_dev[b->device].tab}, !(b->flags & HX_FORCE_GENERIC), st);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(out, d + 6, 6 * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      rc = HX_ERR_HIP;
  }
  (void)hipFree(d);
  return rc == HX_OK ? HX_OK : fail(rc, "indel-count kernel failed: %s", hipGetErrorString(hipGetLastError()));
}

int hx_batch_read_prepared(hx_batch* b, int32_t job, double* subx, double* suby, double* insx, double* rootsubx,
                           double* insy, double* rootsuby) {
  if (!b) return fail(HX_ERR_INVALID_ARG, "batch is null");
  if (job < 0 || job >= b->n_jobs) return fail(HX_ERR_RANGE, "job %d out of range", job);
  if (!b->forward_done) return fail(HX_ERR_STATE, "hx_batch_forward has not been launched");
  { const int rc_ = use_device(b); if (rc_ != HX_OK) return rc_; }
  if ((subx || suby) && !b->sub_scattered) {
    // the fills use per-class tables; the per-state leftMultiply rows are produced when somebody asks for them
    launch_scatter_sub(b->d_jobs, b->n_jobs, b->max_states, b->last_stream);
    HIP_TRY(hipGetLastError());
    b->sub_scattered = true;
  }
  HIP_TRY(hipStreamSynchronize(b->last_stream));
  const DevJob& J = b->jobs[job];
  if (subx) HIP_TRY(hipMemcpy(subx, J.x.sub, sizeof(double) * (size_t)J.x.n * J.CA, hipMemcpyDeviceToHost));
  if (suby) HIP_TRY(hipMemcpy(suby, J.y.sub, sizeof(double) * (size_t)J.y.n * J.CA, hipMemcpyDeviceToHost));
  if (insx) HIP_TRY(hipMemcpy(insx, J.x.ins, sizeof(double) * J.x.n, hipMemcpyDeviceToHost));
  if (rootsubx) HIP_TRY(hipMemcpy(rootsubx, J.x.rootsub, sizeof(double) * J.x.n, hipMemcpyDeviceToHost));
  if (insy) HIP_TRY(hipMemcpy(insy, J.y.ins, sizeof(double) * J.y.n, hipMemcpyDeviceToHost));
  if (rootsuby) HIP_TRY(hipMemcpy(rootsuby, J.y.rootsub, sizeof(double) * J.y.n, hipMemcpyDeviceToHost));
  return HX_OK;
}

int hx_batch_posterior_scan(hx_batch* b, int32_t job, double min_post_prob, hx_cell* out, int64_t cap, int64_t* n_out) {
  if (!b || !n_out || cap < 0 || (cap > 0 && !out)) return fail(HX_ERR_INVALID_ARG, "bad arguments");
  if (job < 0 || job >= b->n_jobs) return fail(HX_ERR_RANGE, "job %d out of range", job);
  if (!b->forward_done || !b->backward_done) return fail(HX_ERR_STATE, "posterior scan needs Forward and Backward fills");
  { const int rc_ = use_device(b); if (rc_ != HX_OK) return rc_; }
  static_assert(sizeof(PostCell) == sizeof(hx_cell), "hx_cell layout");
  const double thr = std::log(min_post_prob);   // host libm, as reference src/forward.cpp:1304
  PostCell* d_out = nullptr;
  unsigned long long* d_cnt = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_cnt), sizeof(unsigned long long)));
  if (cap > 0 && hipMalloc(reinterpret_cast<void**>(&d_out), sizeof(PostCell) * cap) != hipSuccess) {
    (void)hipFree(d_cnt);
    return fail(HX_ERR_OUT_OF_MEMORY, "hipMalloc failed");
  }
  int rc = HX_OK;
  unsigned long long cnt = 0;
  if (hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), b->last_stream) != hipSuccess) rc = fail(HX_ERR_HIP, "memset failed");
  if (rc == HX_OK) {
    launch_posterior_scan(b->d_jobs, job, thr, d_out, (unsigned long long)cap, d_cnt, b->last_stream);
    if (hipStreamSynchronize(b->last_stream) != hipSuccess ||
        hipMemcpy(&cnt, d_cnt, sizeof(cnt), hipMemcpyDeviceToHost) != hipSuccess)
      rc = fail(HX_ERR_HIP, "posterior scan failed: %s", hipGetErrorString(hipGetLastError()));
  }
  if (rc == HX_OK) {
    *n_out = (int64_t)cnt;
    const int64_t n_copy = (int64_t)cnt < cap ? (int64_t)cnt : cap;
    if (n_copy > 0 && hipMemcpy(out, d_out, sizeof(PostCell) * n_copy, hipMemcpyDeviceToHost) != hipSuccess)
      rc = fail(HX_ERR_HIP, "result download failed");
  }
  (void)hipFree(d_cnt);
  if (d_out) (void)hipFree(d_out);
  return rc;
}

}  // extern "C"
