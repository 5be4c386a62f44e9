// Per-branch pair DPs: the three-state (Match / Insert / Delete) alignment of a parent sequence
// profile with a child sequence profile across one tree branch, inside a GuideAlignmentEnvelope.
//
// Reference: Refiner::BranchMatrix::BranchMatrix (src/refiner.cpp:10-60: Viterbi, what `historian reconstruct -refine` runs on
// every branch) and Sampler::BranchMatrix::BranchMatrix (src/sampler.cpp:1034-1084: the same lattice with log_sum_exp, the
// MCMC sampler's branch move); cell storage and envelope of TreeAlignFuncs::SparseDPMatrix<3> (src/sampler.h:66-166), the
// per-cell emission BranchMatrixBase::logMatch (src/sampler.h:207-209).
//
// Two kernels.  k_branch_emission evaluates logMatch(i, j) = logInnerProduct(xSeq[i-1], ySub[j-1]) for every in-envelope cell
// up front - it does not depend on DP values, is fully parallel, and uses the reference's table log_sum_exp bit for bit.
// k_branch_fill sweeps a pair with ONE wavefront: 64-row strips one after the other, lane <-> row, step <-> anti-diagonal; a
// cell's left source is the lane's own previous cell, up and diagonal are the previous lane's cells of one and two steps ago
// (DPP wave_shr:1); lane 0's come from the strip above's last row, which the same wavefront stored earlier: 64 columns of it
// are block-loaded every 64 steps and handed out by v_readlane.  The batch supplies the parallelism: a refinement sweep aligns
// every branch of a tree (2 N - 2 pairs), the sampler many moves.  Max-plus is exact in any order, the log_sum_exp form uses
// the reference's operator in the reference's left-nested order: cells and lpEnd are bit-identical to the restatement
// (oracle/branch_oracle.py) in both forms.
//
// Storage: three state planes per pair, strip-skewed like the Forward matrices (hx_device.h cell_slot), -inf outside the
// envelope; hx_branch_batch_read_matrix returns the dense [x_len + 1][y_len + 1][3] array.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>
#include "hx_device.h"
#include "hx_lse.h"
#include "hx_common.h"
#include "hx_policy.h"
#include "hx_kernels.h"
#include "../../include/historian_hip.h"

namespace hx {

int api_fail(int code, const char* what);                  // hx_api.hip: sets hx_last_error()
const double* device_lse_table(int device);               // hx_api.hip: the table hx_init uploaded, or nullptr

namespace {

struct DevBranch {
  int32_t X, Y;                 // positions 0 .. x_len, 0 .. y_len
  int32_t CA, C;                // components * alphabet; components
  int32_t max_dist;             // < 0: no band
  const double* x_pwm;          // [x_len][CA]
  const double* y_sub;          // [y_len][CA]
  const double* y_emit;         // [y_len]
  const int32_t* x_env;         // [X] or nullptr
  const int32_t* y_env;         // [Y]
  double T[3][4];
  double* cells;                // [3][plane]
  double* emis;                 // [plane]: logMatch in the matrix layout
  int64_t plane, strip_stride;
  double* lp_end;
  const int32_t* win;           // banded: [n_strips][3][2] step windows of the strips (half-open, merged, in order; empty ones last), or nullptr
};

__device__ __forceinline__ bool branch_in_env(const DevBranch& J, const int i, const int j) {
  // TreeAlignFuncs::SparseDPMatrix::inEnvelope (src/sampler.h:146-149)
  if (i == 0 || j == 0 || i == J.X - 1 || j == J.Y - 1 || J.max_dist < 0) return true;
  int d = J.x_env[i] - J.y_env[j];
  d = d < 0 ? -d : d;
  return d <= J.max_dist;
}

// logMatch for every in-envelope cell with i, j >= 1: the nested logInnerProduct of src/logsumexp.h:132-151 - over the
// components, of the sum over the residues - in the reference's table arithmetic.  grid (jobs, row slices)
__global__ void k_branch_emission(const DevBranch* __restrict__ jobs, const double* __restrict__ tab) {
  const DevBranch& J = jobs[blockIdx.x];
  const int C = J.C, A = J.CA / C;
  const int64_t n = (int64_t)J.X * J.Y;
  for (int64_t c = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; c < n; c += (int64_t)gridDim.y * blockDim.x) {
    const int i = (int)(c / J.Y), j = (int)(c % J.Y);
    if (i == 0 || j == 0 || !branch_in_env(J, i, j)) continue;
    const double* xs = J.x_pwm + (size_t)(i - 1) * J.CA;
    const double* ys = J.y_sub + (size_t)(j - 1) * J.CA;
    double lip = HX_NEG_INF;
    for (int cpt = 0; cpt < C; ++cpt) {
      double inner = HX_NEG_INF;
      for (int a = 0; a < A; ++a) inner = lse(inner, xs[cpt * A + a] + ys[cpt * A + a], tab);
      lip = lse(lip, inner, tab);
    }
    J.emis[cell_slot(J.strip_stride, i, j)] = lip;
  }
}

__global__ void k_branch_clear(const DevBranch* __restrict__ jobs) {
  const DevBranch& J = jobs[blockIdx.x];
  const int64_t n = 3 * J.plane;
  for (int64_t c = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; c < n; c += (int64_t)gridDim.y * blockDim.x) J.cells[c] = HX_NEG_INF;
}

typedef double d2v __attribute__((ext_vector_type(2)));
struct B3 { double m, i, d; };

// value of lane `src` (wave-uniform)
__device__ __forceinline__ double read_lane64(const double v, const int src) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}

template <bool VITERBI>
__device__ __forceinline__ double combine(const double a, const double b, const double* __restrict__ tab) {
  return VITERBI ? vmax(a, b) : lse(a, b, tab);
}

// One workgroup per branch; its 64-row strips are dealt to the workgroup's wavefronts round-robin, each wavefront sweeps its
// strip on its own clock (lane <-> row, step <-> anti-diagonal, up / diagonal from the lane above by DPP), and what a strip
// needs of the strip above - that strip's last row - it reads from the matrix, HXBR_BLK columns at a time, once the wavefront
// above has said that those columns are stored: a monotonic column count per strip in LDS, published behind a drain of the
// producer's stores (the consumer's loads are agent-scope: served by L2, where the stores are by then).  A strip therefore
// starts ~HXBR_BLK + 64 steps behind the one above, and a branch of S strips takes columns + ~80 (S - 1) steps instead of the
// S (columns + 63) of one wavefront per branch.  Waits cannot form a cycle: strip s waits for strip s - 1 only.
#define HXBR_BLK 16
#define HXBR_MAX_STRIPS 1024
// YL: the child side of a step - insertion score and envelope coordinate of its column - out of LDS (staged once per
// workgroup; the launcher checks that the longest child profile of the launch fits), not fetched from memory inside the step;
// the step's emission term is fetched one step ahead either way.
template <bool VITERBI, bool YL>
__global__ void __launch_bounds__(1024) k_branch_fill(const DevBranch* __restrict__ jobs, const double* __restrict__ tab, const int y_cap) {
  __shared__ int progress[HXBR_MAX_STRIPS];         // columns of the strip's last row that are stored
  extern __shared__ __attribute__((aligned(16))) unsigned char ydyn[];
  double* yemitL = reinterpret_cast<double*>(ydyn);                 // [y_cap]
  int* yenvL = reinterpret_cast<int*>(ydyn + 8 * (size_t)y_cap);    // [y_cap]
  const DevBranch& J = jobs[blockIdx.x];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), n_waves = (int)(blockDim.x >> 6);
  const int X = J.X, Y = J.Y;
  const int64_t plane = J.plane, ss = J.strip_stride;
  HX_GLOBAL double* __restrict__ M = as_global(J.cells);
  const HX_GLOBAL double* __restrict__ E = as_global((const double*)J.emis);
  const double mm = J.T[0][0], mi = J.T[0][1], md = J.T[0][2], im = J.T[1][0], ii = J.T[1][1], id = J.T[1][2], dm = J.T[2][0],
               dd = J.T[2][2];
  const int n_strips = (X + 63) >> 6;
  for (int q = threadIdx.x; q < n_strips && q < HXBR_MAX_STRIPS; q += blockDim.x) progress[q] = 0;
  if (YL)
    for (int j = threadIdx.x; j < Y; j += blockDim.x) {
      yemitL[j] = j > 0 ? J.y_emit[j - 1] : 0.0;    // (the score of entering column j: yEmit of child position j - 1)
      yenvL[j] = J.max_dist >= 0 ? J.y_env[j] : 0;
    }
  __syncthreads();
  volatile int* prog = progress;
  const B3 none{HX_NEG_INF, HX_NEG_INF, HX_NEG_INF};
  for (int s = wave; s < n_strips; s += n_waves) {
    const int i = (s << 6) + lane;
    const bool rvalid = i < X;
    const int xe = (rvalid && J.max_dist >= 0) ? J.x_env[i] : 0;
    const bool xedge = i == 0 || i == X - 1;
    const bool feeds = s + 1 < n_strips;            // a strip below reads this strip's last row
    B3 left = none, up = none, diag = none;        // (i, j-1); (i-1, j) and (i-1, j-1) of the step being computed
    B3 bnd = none;                                  // lane l < HXBR_BLK: cell (row above the strip, column c0 + l) of the current block of columns
    B3 held = none;                                 // the lane's cell of the even step of the current step pair
    bool held_in = false;
    int seen = 0;
    // logMatch of the lane's cell of the NEXT step (column t + 1 - lane), fetched a step ahead
    auto emis_at = [&](const int jj) -> double {
      return (rvalid && i > 0 && jj > 0 && jj < Y) ? E[cell_slot(ss, i, jj)] : 0.0;
    };
    // a banded strip sweeps its step windows only (hx_branch_batch_create: branch_windows); between them nothing of the
    // strip is inside the envelope, so a window starts from -inf registers, and the strip below is told that the columns up
    // to the next window are final (they hold the -inf the planes were cleared to)
    const int32_t* wn = J.win ? J.win + 6 * s : nullptr;
    for (int wi = 0; wi < (wn ? 3 : 1); ++wi) {
    const int t0 = wn ? wn[2 * wi] : 0, t1 = wn ? wn[2 * wi + 1] : (Y + 63 + 1) & ~1;      // (whole step pairs)
    if (t1 <= t0) break;
    left = none; up = none; diag = none;
    if (s > 0 && t0 >= 1 && t0 - 1 < Y) {
      // ... except lane 0's diagonal source of the window's first step: cell (row above, column t0 - 1) belongs to the strip
      // above, whose band may well hold it
      while (seen < t0) {
        seen = __builtin_amdgcn_readfirstlane(prog[s - 1]);
        if (seen < t0) __builtin_amdgcn_s_sleep(2);
      }
      asm volatile("" ::: "memory");
      if (lane == 0) {
        const int64_t sl = cell_slot(ss, (s << 6) - 1, t0 - 1);
        diag.m = __hip_atomic_load(M + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        diag.i = __hip_atomic_load(M + plane + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        diag.d = __hip_atomic_load(M + 2 * plane + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    double e_next = emis_at(t0 - lane);
    for (int t = t0; t < t1; ++t) {
      const double e_now = e_next;
      e_next = emis_at(t + 1 - lane);
      if (s > 0 && ((t & (HXBR_BLK - 1)) == 0 || t == t0) && t < Y) {
        // the strip above's last row, HXBR_BLK columns at a time
        const int c0 = t & ~(HXBR_BLK - 1);
        const int need = c0 + HXBR_BLK < Y ? c0 + HXBR_BLK : Y;
        while (seen < need) {
          seen = __builtin_amdgcn_readfirstlane(prog[s - 1]);
          if (seen < need) __builtin_amdgcn_s_sleep(2);
        }
        asm volatile("" ::: "memory");
        const int c = c0 + lane;
        bnd = none;
        if (lane < HXBR_BLK && c < Y) {
          const int64_t sl = cell_slot(ss, (s << 6) - 1, c);
          bnd.m = __hip_atomic_load(M + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bnd.i = __hip_atomic_load(M + plane + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bnd.d = __hip_atomic_load(M + 2 * plane + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      // lane 0's upper neighbour of this step is column t of the row above: lane t mod HXBR_BLK of the block
      if (s > 0) {
        const int src = t & (HXBR_BLK - 1);
        const double bm = read_lane64(bnd.m, src), bi = read_lane64(bnd.i, src), bd = read_lane64(bnd.d, src);
        if (lane == 0) up = t < Y ? B3{bm, bi, bd} : none;
      }
      // The cell, straight-line: sources that do not exist (row / column -1, cells outside the envelope) are -inf in the
      // registers they come from, and -inf through the sums is what the reference's unassigned cell is
      // (src/refiner.cpp:24-50 / src/sampler.cpp:1049-1072); only the stores are conditional.
      const int j = t - lane;
      const bool jv = rvalid && j >= 0 && j < Y;
      const int jc = j < 0 ? 0 : (j < Y ? j : Y - 1);
      const int ye = J.max_dist < 0 ? 0 : (YL ? yenvL[jc] : J.y_env[jc]);
      const double yem = YL ? yemitL[jc] : (jc > 0 ? J.y_emit[jc - 1] : 0.0);
      const int dxy = xe - ye;
      const bool in = jv && (xedge || j == 0 || j == Y - 1 || J.max_dist < 0 || (dxy <= J.max_dist && -dxy <= J.max_dist));
      B3 now;
      now.d = combine<VITERBI>(combine<VITERBI>(up.m + md, up.i + id, tab), up.d + dd, tab);
      now.i = yem + combine<VITERBI>(left.m + mi, left.i + ii, tab);
      now.m = e_now + combine<VITERBI>(combine<VITERBI>(diag.m + mm, diag.i + im, tab), diag.d + dm, tab);
      if (i == 0 && j == 0) now.m = 0.0;            // lpStart() = 0
      if (!in) now = none;
      // The cells of steps 2m and 2m + 1 of a row lie side by side in a plane: stored together, 16 bytes per lane and
      // plane, a wavefront's store is whole 64-byte lines (stored one by one, every line was written in two halves - two
      // read-modify-writes; a build without stores ran 45 % faster).  A cell of the pair that is outside the envelope is
      // written as the -inf the plane was cleared to; windows are whole step pairs (branch_windows).
      if (!(t & 1)) { held = now; held_in = in; }
      else if (rvalid && (in || held_in)) {
        HX_GLOBAL d2v* P2 = (HX_GLOBAL d2v*)(M + cell_slot(ss, i, j - 1));
        const int64_t plane2 = plane >> 1;
        P2[0] = d2v{held.m, now.m}; P2[plane2] = d2v{held.i, now.i}; P2[2 * plane2] = d2v{held.d, now.d};
      }
      // next step: the lane's own cell is its left source; the previous lane's cell of this step its upper, of the last its diagonal
      diag = up;
      left = now;
      up = B3{wave_shr1(now.m), wave_shr1(now.i), wave_shr1(now.d)};
      if (lane == 0) up = none;                     // (row 0 has no row above; strips below take it from the block)
      // the last row's columns 0 .. t - 63 are computed; say so once their stores have left the wavefront
      if (feeds && (t & 1)) {                       // (behind the store of a step pair)
        const int done = t - 63 + 1;                // columns of lane 63's row computed and stored so far (odd)
        if (done > 0 && ((done & (HXBR_BLK - 1)) == 1 || done >= Y)) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0) prog[s] = done < Y ? done : Y;
        }
      }
    }
    // behind a window: the last row is final up to where the next window takes it up
    {
      const int nt0 = (wn && wi + 1 < 3 && wn[2 * wi + 3] > wn[2 * wi + 2]) ? wn[2 * wi + 2] : Y + 63;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      int fin = nt0 - 63;
      fin = fin < 0 ? 0 : (fin > Y ? Y : fin);
      if (feeds && lane == 0 && fin > 0) prog[s] = fin;
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (feeds && lane == 0) prog[s] = Y;
    if (s == n_strips - 1 && lane == 0) {
      const int64_t sl = cell_slot(ss, X - 1, Y - 1);
      const double em = __hip_atomic_load(M + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const double ei = __hip_atomic_load(M + plane + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const double ed = __hip_atomic_load(M + 2 * plane + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *J.lp_end = combine<VITERBI>(combine<VITERBI>(em + J.T[0][3], ei + J.T[1][3], tab), ed + J.T[2][3], tab);
    }
  }
}

// the skewed planes of one pair -> dense [X][Y][3]
__global__ void k_branch_dense(const DevBranch* __restrict__ jobs, const int job, double* __restrict__ out) {
  const DevBranch& J = jobs[job];
  const int64_t n = (int64_t)J.X * J.Y;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (int64_t)gridDim.x * blockDim.x) {
    const int64_t sl = cell_slot(J.strip_stride, (int)(c / J.Y), (int)(c % J.Y));
    for (int s = 0; s < 3; ++s) out[3 * c + s] = J.cells[s * J.plane + sl];
  }
}

}  // namespace
}  // namespace hx

using namespace hx;

struct hx_branch_batch {
  int device = 0, n_jobs = 0;
  std::vector<DevBranch> jobs;
  DevBranch* d_jobs = nullptr;
  char* d_arena = nullptr;          // inputs + lpEnd
  double* d_cells = nullptr;        // matrices + emission planes
  size_t lp_off = 0;
  int64_t max_cells = 0;
  int max_x = 0, max_y = 0;         // rows / columns of the longest branch
  hipEvent_t ev[2] = {nullptr, nullptr};
  hipStream_t last_stream = nullptr;
  bool done = false;
};

namespace {
// Step windows of a banded branch's strips (steps t = column + row-in-strip): what is always inside the envelope - the first
// and the last column (SparseDPMatrix::inEnvelope, src/sampler.h:146-149) - and the band, as up to three half-open ranges that
// together hold every in-envelope cell of the strip's rows (supersets are harmless: a cell is tested again).  The strips of
// the first and the last row sweep everything.  Any envelope coordinates (not only non-decreasing ones): the columns of a
// coordinate value are bracketed once, a row takes the brackets of the values within max_distance of its own.
std::vector<int32_t> branch_windows(const int32_t* xenv, const int32_t* yenv, int X, int Y, int band) {
  const int n_strips = (X + HX_STRIP - 1) / HX_STRIP, nsteps = Y + HX_STRIP - 1;
  std::vector<int32_t> w(6 * (size_t)n_strips, 0);
  int V = 0;
  for (int j = 0; j < Y; ++j) V = std::max(V, (int)yenv[j]);
  std::vector<int> minj(V + 1, INT_MAX), maxj(V + 1, -1);
  for (int j = 0; j < Y; ++j) {
    const int v = yenv[j] < 0 ? 0 : yenv[j];
    minj[v] = std::min(minj[v], j);
    maxj[v] = std::max(maxj[v], j);
  }
  // (prefix brackets would make a row O(1); bands are tens of values wide)
  for (int s = 0; s < n_strips; ++s) {
    int32_t* o = &w[6 * (size_t)s];
    const int rows = std::min(HX_STRIP, X - s * HX_STRIP);
    if (s == 0 || s == n_strips - 1) { o[0] = 0; o[1] = (nsteps + 1) & ~1; continue; }
    int lo = INT_MAX, hi = -1;
    for (int l = 0; l < rows; ++l) {
      const int xe = xenv[s * HX_STRIP + l] < 0 ? 0 : xenv[s * HX_STRIP + l];
      int jmin = INT_MAX, jmax = -1;
      for (int v = std::max(0, xe - band); v <= std::min(V, xe + band); ++v) {
        jmin = std::min(jmin, minj[v]);
        jmax = std::max(jmax, maxj[v]);
      }
      if (jmax < 0) continue;
      lo = std::min(lo, jmin + l);
      hi = std::max(hi, jmax + l);
    }
    std::pair<int, int> r[3] = {{0, rows}, {lo, hi + 1}, {Y - 1, Y - 1 + rows}};
    if (hi < 0) r[1] = {INT_MAX, INT_MAX};         // (no band cell in the strip)
    // in order, merged where they touch
    std::sort(r, r + 3);
    int n = 0;
    for (int k = 0; k < 3; ++k) {
      if (r[k].second <= r[k].first) continue;
      if (n > 0 && r[k].first <= o[2 * (n - 1) + 1]) o[2 * (n - 1) + 1] = std::max(o[2 * (n - 1) + 1], r[k].second);
      else { o[2 * n] = r[k].first; o[2 * n + 1] = r[k].second; ++n; }
    }
    // whole step pairs (the fill stores a row's cells of steps 2m, 2m + 1 together), merged again where they now touch
    int m = 0;
    for (int k = 0; k < n; ++k) {
      const int a = o[2 * k] & ~1, b = std::min((o[2 * k + 1] + 1) & ~1, (nsteps + 1) & ~1);
      if (m > 0 && a <= o[2 * (m - 1) + 1]) o[2 * (m - 1) + 1] = std::max(o[2 * (m - 1) + 1], b);
      else { o[2 * m] = a; o[2 * m + 1] = b; ++m; }
    }
    for (int k = m; k < 3; ++k) o[2 * k] = o[2 * k + 1] = 0;
  }
  return w;
}
}  // namespace

extern "C" {

int hx_branch_batch_destroy(hx_branch_batch* b) {
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

int hx_branch_batch_create(const hx_branch_job* jobs, int32_t n_jobs, hx_branch_batch** out) {
  if (out) *out = nullptr;
  if (!jobs || !out || n_jobs < 1) return api_fail(HX_ERR_INVALID_ARG, "hx_branch_batch_create: need at least one job");
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return api_fail(HX_ERR_NO_DEVICE, "no HIP device");
  if (!device_lse_table(device)) return api_fail(HX_ERR_NOT_INITIALIZED, "hx_init has not been called for the current device");
  hx_branch_batch* b = new (std::nothrow) hx_branch_batch;
  if (!b) return api_fail(HX_ERR_OUT_OF_MEMORY, "host allocation failed");
  b->device = device;
  b->n_jobs = n_jobs;
  std::vector<char> host;
  auto put = [&](const void* p, size_t bytes) -> size_t {
    const size_t off = (host.size() + 15) & ~(size_t)15;
    host.resize(off + bytes);
    if (bytes) memcpy(host.data() + off, p, bytes);
    return off;
  };
  struct Off { size_t x, y, e, xe, ye, win; bool env; };
  std::vector<Off> offs(n_jobs);
  int64_t cells_total = 0;
  try {
    b->jobs.resize(n_jobs);
    for (int k = 0; k < n_jobs; ++k) {
      const hx_branch_job& j = jobs[k];
      if (j.x_len >= 64 * HXBR_MAX_STRIPS) {
        hx_branch_batch_destroy(b);
        return api_fail(HX_ERR_RANGE, "hx_branch_batch_create: a parent profile of more than 65535 positions");
      }
      if (j.x_len < 0 || j.y_len < 0 || j.components < 1 || j.alphabet < 1 ||
          (j.x_len && !j.x_pwm) || (j.y_len && (!j.y_sub || !j.y_emit)) || (j.max_distance >= 0 && (!j.x_env || !j.y_env))) {
        hx_branch_batch_destroy(b);
        return api_fail(HX_ERR_INVALID_ARG, "hx_branch_batch_create: inconsistent job (lengths, components, missing arrays)");
      }
      DevBranch& J = b->jobs[k];
      memset(&J, 0, sizeof(J));
      J.X = j.x_len + 1; J.Y = j.y_len + 1;
      J.CA = j.components * j.alphabet;
      J.C = j.components;
      J.max_dist = j.max_distance;
      for (int s = 0; s < 3; ++s)
        for (int d = 0; d < 4; ++d) J.T[s][d] = j.trans[s][d];
      J.strip_stride = strip_stride_for(J.Y);
      J.plane = (int64_t)((J.X + HX_STRIP - 1) / HX_STRIP) * J.strip_stride;
      offs[k].x = put(j.x_pwm, sizeof(double) * (size_t)j.x_len * J.CA);
      offs[k].y = put(j.y_sub, sizeof(double) * (size_t)j.y_len * J.CA);
      offs[k].e = put(j.y_emit, sizeof(double) * (size_t)j.y_len);
      offs[k].env = j.max_distance >= 0;
      offs[k].xe = offs[k].env ? put(j.x_env, sizeof(int32_t) * (size_t)J.X) : 0;
      offs[k].ye = offs[k].env ? put(j.y_env, sizeof(int32_t) * (size_t)J.Y) : 0;
      offs[k].win = 0;
      if (offs[k].env && !getenv("HX_BRANCH_NO_WINDOWS")) {
        const std::vector<int32_t> w = branch_windows(j.x_env, j.y_env, J.X, J.Y, j.max_distance);
        offs[k].win = put(w.data(), sizeof(int32_t) * w.size()) + 1;      // (+1: 0 means none)
      }
      cells_total += 4 * J.plane;
      if ((int64_t)J.X * J.Y > b->max_cells) b->max_cells = (int64_t)J.X * J.Y;
      if (J.X > b->max_x) b->max_x = J.X;
      if (J.Y > b->max_y) b->max_y = J.Y;
    }
    b->lp_off = put(nullptr, 0);
    host.resize(b->lp_off + sizeof(double) * n_jobs);
  } catch (const std::bad_alloc&) {
    hx_branch_batch_destroy(b);
    return api_fail(HX_ERR_OUT_OF_MEMORY, "host allocation failed while building the batch");
  }
  if (hipMalloc(reinterpret_cast<void**>(&b->d_arena), host.size()) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&b->d_cells), sizeof(double) * (size_t)cells_total) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&b->d_jobs), sizeof(DevBranch) * n_jobs) != hipSuccess) {
    hx_branch_batch_destroy(b);
    return api_fail(HX_ERR_OUT_OF_MEMORY, "hx_branch_batch_create: device allocation failed");
  }
  int64_t at = 0;
  for (int k = 0; k < n_jobs; ++k) {
    DevBranch& J = b->jobs[k];
    J.x_pwm = reinterpret_cast<const double*>(b->d_arena + offs[k].x);
    J.y_sub = reinterpret_cast<const double*>(b->d_arena + offs[k].y);
    J.y_emit = reinterpret_cast<const double*>(b->d_arena + offs[k].e);
    J.x_env = offs[k].env ? reinterpret_cast<const int32_t*>(b->d_arena + offs[k].xe) : nullptr;
    J.y_env = offs[k].env ? reinterpret_cast<const int32_t*>(b->d_arena + offs[k].ye) : nullptr;
    J.win = offs[k].win ? reinterpret_cast<const int32_t*>(b->d_arena + (offs[k].win - 1)) : nullptr;
    J.cells = b->d_cells + at;
    J.emis = b->d_cells + at + 3 * J.plane;
    at += 4 * J.plane;
    J.lp_end = reinterpret_cast<double*>(b->d_arena + b->lp_off) + k;
  }
  if (hipMemcpy(b->d_arena, host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(b->d_jobs, b->jobs.data(), sizeof(DevBranch) * n_jobs, hipMemcpyHostToDevice) != hipSuccess ||
      hipEventCreate(&b->ev[0]) != hipSuccess || hipEventCreate(&b->ev[1]) != hipSuccess) {
    hx_branch_batch_destroy(b);
    return api_fail(HX_ERR_HIP, "hx_branch_batch_create: copy to the device failed");
  }
  *out = b;
  return HX_OK;
}

int hx_branch_batch_run(hx_branch_batch* b, int32_t viterbi, void* stream) {
  if (!b) return api_fail(HX_ERR_INVALID_ARG, "batch is null");
  if (hipSetDevice(b->device) != hipSuccess) return api_fail(HX_ERR_HIP, "hipSetDevice failed");
  const double* tab = device_lse_table(b->device);
  if (!tab) return api_fail(HX_ERR_NOT_INITIALIZED, "hx_init has not been called for the batch's device");
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int j0 = 0; j0 < b->n_jobs; j0 += 16384) {        // (grid.x of at most 16384 jobs per launch)
    const int n = b->n_jobs - j0 < 16384 ? b->n_jobs - j0 : 16384;
    hipLaunchKernelGGL(k_branch_clear, dim3(n, 16), dim3(256), 0, st, b->d_jobs + j0);
    hipLaunchKernelGGL(k_branch_emission, dim3(n, 16), dim3(256), 0, st, b->d_jobs + j0, tab);
  }
  if (hipEventRecord(b->ev[0], st) != hipSuccess) return api_fail(HX_ERR_HIP, "hipEventRecord failed");
  for (int j0 = 0; j0 < b->n_jobs; j0 += 65536) {
    const int n = b->n_jobs - j0 < 65536 ? b->n_jobs - j0 : 65536;
    // wavefronts per branch: as many as the longest branch has strips, at most 16 (a workgroup of 1024) - fewer when the
    // batch alone fills the chip (~8 wavefronts per SIMD)
    int waves = (b->max_x + 63) / 64;
    waves = waves < 1 ? 1 : (waves > 16 ? 16 : waves);
    if (const char* e = getenv("HX_BRANCH_WAVES")) { const int v = atoi(e); if (v >= 1 && v <= 16) waves = v; }
    else while (waves > 1 && (int64_t)n * waves > 8192 * 2) waves = (waves + 1) / 2;
    // the child sides in LDS when the longest one fits (12 bytes per position beside the progress counters)
    const bool yl = (size_t)b->max_y * 12 <= 96 * 1024;
    const size_t dyn = yl ? (size_t)b->max_y * 12 + 16 : 0;
    const int y_cap = yl ? (b->max_y + 1) & ~1 : 0;
    if (viterbi && yl) hipLaunchKernelGGL((k_branch_fill<true, true>), dim3(n), dim3(64 * waves), dyn, st, b->d_jobs + j0, tab, y_cap);
    else if (viterbi) hipLaunchKernelGGL((k_branch_fill<true, false>), dim3(n), dim3(64 * waves), 0, st, b->d_jobs + j0, tab, 0);
    else if (yl) hipLaunchKernelGGL((k_branch_fill<false, true>), dim3(n), dim3(64 * waves), dyn, st, b->d_jobs + j0, tab, y_cap);
    else hipLaunchKernelGGL((k_branch_fill<false, false>), dim3(n), dim3(64 * waves), 0, st, b->d_jobs + j0, tab, 0);
  }
  if (hipEventRecord(b->ev[1], st) != hipSuccess || hipGetLastError() != hipSuccess) return api_fail(HX_ERR_HIP, "hx_branch_batch_run: launch failed");
  b->done = true;
  b->last_stream = st;
  return HX_OK;
}

int hx_branch_batch_results(hx_branch_batch* b, double* lp_end) {
  if (!b || !lp_end) return api_fail(HX_ERR_INVALID_ARG, "bad arguments");
  if (!b->done) return api_fail(HX_ERR_STATE, "hx_branch_batch_run has not been launched");
  if (hipSetDevice(b->device) != hipSuccess || hipStreamSynchronize(b->last_stream) != hipSuccess ||
      hipMemcpy(lp_end, b->d_arena + b->lp_off, sizeof(double) * b->n_jobs, hipMemcpyDeviceToHost) != hipSuccess)
    return api_fail(HX_ERR_HIP, "hx_branch_batch_results: HIP call failed");
  return HX_OK;
}

int hx_branch_batch_read_matrix(hx_branch_batch* b, int32_t job, double* out) {
  if (!b || !out) return api_fail(HX_ERR_INVALID_ARG, "bad arguments");
  if (job < 0 || job >= b->n_jobs) return api_fail(HX_ERR_RANGE, "job out of range");
  if (!b->done) return api_fail(HX_ERR_STATE, "hx_branch_batch_run has not been launched");
  const DevBranch& J = b->jobs[job];
  const size_t bytes = sizeof(double) * 3 * (size_t)J.X * J.Y;
  double* dense = nullptr;
  if (hipSetDevice(b->device) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&dense), bytes) != hipSuccess)
    return api_fail(HX_ERR_OUT_OF_MEMORY, "hx_branch_batch_read_matrix: device allocation failed");
  hipLaunchKernelGGL(k_branch_dense, dim3(256), dim3(256), 0, b->last_stream, b->d_jobs, job, dense);
  const bool ok = hipStreamSynchronize(b->last_stream) == hipSuccess && hipMemcpy(out, dense, bytes, hipMemcpyDeviceToHost) == hipSuccess;
  (void)hipFree(dense);
  return ok ? HX_OK : api_fail(HX_ERR_HIP, "hx_branch_batch_read_matrix: HIP call failed");
}

int64_t hx_branch_batch_total_cells(const hx_branch_batch* b) {
  if (!b) return 0;
  int64_t n = 0;
  for (const DevBranch& J : b->jobs) n += (int64_t)J.X * J.Y;
  return n;
}

int hx_branch_batch_last_kernel_ms(hx_branch_batch* b, float* ms) {
  if (!b || !ms) return api_fail(HX_ERR_INVALID_ARG, "bad arguments");
  if (!b->done) return api_fail(HX_ERR_STATE, "hx_branch_batch_run has not been launched");
  if (hipSetDevice(b->device) != hipSuccess || hipEventSynchronize(b->ev[1]) != hipSuccess ||
      hipEventElapsedTime(ms, b->ev[0], b->ev[1]) != hipSuccess)
    return api_fail(HX_ERR_HIP, "hx_branch_batch_last_kernel_ms: HIP call failed");
  return HX_OK;
}

}  // extern "C"
