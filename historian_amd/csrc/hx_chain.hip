// Wavefront Forward fill for linear-chain profiles (leaves, and any profile whose
// state i has the single in-transition i-1 -> i): reference src/forward.cpp:68-223
// specialised to in-degree 1, with the ready/wait/null flags and the envelope kept.
//
// One workgroup per pair.  The matrix is swept in row passes of THREADS*RPT rows; every
// lane owns RPT consecutive rows and at step d computes its cells on anti-diagonal d.
// The three predecessor cells of a cell are
//   left  (i, j-1)   : the lane's own value from step d-1           (registers)
//   up    (i-1, j)   : the row above, step d-1  (registers, or lane-1 via DPP wave_shr)
//   diag  (i-1, j-1) : the row above, step d-2  (registers, or lane-1 via DPP wave_shr)
// so no DP cell is re-read from memory, except the one boundary row between two row
// passes (block-loaded 64 columns at a time).  The only cross-wave traffic is lane 63's
// last row -> next wave's lane 0 through a double-buffered LDS slot, one s_barrier per
// anti-diagonal.  Each step a wave stores RPT*64 consecutive doubles per state plane
// (strip-skewed layout, hx_device.h): fully coalesced, write-once 40 B/cell.
//
// Two log-sum-exp policies:
//   ExactLse  the reference's 100001-entry table + linear interpolation, bit for bit
//             (table in L2: the kernel is bound by L2 gather requests, not by HBM);
//   FastLse   same max + T(|a-b|) form and the same d >= 10 truncation, T from a
//             1024-interval cubic table held in LDS (|T - log1p(exp(-d))| < 1e-12).
//
// Compiled with -ffp-contract=off (see hx_lse.h); FastLse uses explicit FMAs.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include "hx_device.h"
#include "hx_lse.h"
#include "hx_common.h"
#include "hx_policy.h"
#include "hx_leafcell.h"
#include "hx_kernels.h"

namespace hx {

// dst lane l >= 1 receives src of lane l-1; lane 0 keeps `old` (DPP leaves lanes without a
// source untouched when bound_ctrl is off)
__device__ __forceinline__ double wave_shr1_keep0(double old, double v) {
  int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), 0x138, 0xf, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ C5 wave_shr1_keep0(const C5& o, const C5& c) {
  return C5{wave_shr1_keep0(o.imm, c.imm), wave_shr1_keep0(o.imd, c.imd), wave_shr1_keep0(o.idm, c.idm),
            wave_shr1_keep0(o.imi, c.imi), wave_shr1_keep0(o.iiw, c.iiw)};
}

__device__ __forceinline__ C5 wave_shr1(const C5& c) {
  return C5{wave_shr1(c.imm), wave_shr1(c.imd), wave_shr1(c.idm), wave_shr1(c.imi), wave_shr1(c.iiw)};
}

__device__ __forceinline__ double read_lane(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}

// per-row x-side constants
struct XRow {
  double lp, rootsub, ins;
  int env;
  int emis_off;    // cls * Ky, or -1
  uint8_t flags;
  bool valid;
};

// One cell of the chain recursion, written without data-dependent branches so that the
// 13 log-sum-exp evaluations of a cell (and those of the lane's other rows) are
// independent straight-line code the scheduler can interleave.  `j` is clamped to a
// valid column by the caller; `valid` says whether the cell exists.
template <class LSE>
__device__ __forceinline__ C5 chain_cell(const DevJob& J, const LSE& L, const XRow& X, int i, int j, bool valid,
                                         const C5& up, const C5& left, const C5& diag) {
  const uint8_t xf = X.flags, yf = J.y.flags[j];
  bool in_env = ((xf | yf) & F_EDGE) || J.max_dist < 0;
  if (J.max_dist >= 0) {
    int dd = X.env - J.y.env[j];
    dd = dd < 0 ? -dd : dd;
    in_env = in_env || dd <= J.max_dist;
  }
  in_env = in_env && valid;
  const bool xnull = xf & F_NULL, ynull = yf & F_NULL;
  const bool yok = (yf & F_READY) || J.y.empty;   // yState.isReady() || yEmpty
  const bool xok = (xf & F_READY) || J.x.empty;
  const double (*T)[6] = J.T;
  const bool hx_in = i > 0, hy_in = j > 0;
  const double lpx = X.lp;
  const double lpy = J.y.in_lp[hy_in ? j - 1 : 0];
  const double rsy = J.y.rootsub[j], insy = J.y.ins[j];

  // the five n-ary sums of reference src/forward.cpp:103-115,139-150,171-180 (left-nested)
  double a_imd = L(up.imm + T[0][1], up.imd + T[1][1]);
  double a_iiw = L(up.imm + T[0][4], up.imi + T[3][4]);
  double a_idm = L(left.imm + T[0][2], left.imd + T[1][2]);
  const double a_imi = L(left.imm + T[0][3], left.imi + T[3][3]);
  double a_imm = L(diag.imm + T[0][0], diag.imd + T[1][0]);
  a_imd = L(a_imd, up.idm + T[2][1]);
  a_iiw = L(a_iiw, up.iiw + T[4][4]);
  a_idm = L(a_idm, left.idm + T[2][2]);
  a_imm = L(a_imm, diag.idm + T[2][0]);
  a_imd = L(a_imd, up.imi + T[3][1]);
  a_idm = L(a_idm, left.iiw + T[4][2]);
  a_imm = L(a_imm, diag.imi + T[3][0]);
  a_imm = L(a_imm, diag.iiw + T[4][0]);

  // lse(-inf, v) == v exactly, so a single in-transition needs no accumulate step
  const double NI = HX_NEG_INF;
  C5 r;
  r.imd = (hx_in && yok) ? (xnull ? up.imd + lpx : (a_imd + lpx) + X.rootsub) : NI;
  r.iiw = (hx_in && yok) ? (xnull ? up.iiw + lpx : (a_iiw + lpx) + X.ins) : NI;
  r.idm = hy_in ? (ynull ? left.idm + lpy : (xok ? (a_idm + lpy) + rsy : NI)) : NI;
  r.imi = hy_in ? (ynull ? left.imi + lpy : (xok ? (a_imi + lpy) + insy : NI)) : NI;
  const double init = (i == 0 && j == 0) ? 0.0 : NI;
  if (!xnull && !ynull) {
    double e;
    if (J.emis) {
      const int cy = J.y.cls[j];
      e = J.emis[(X.emis_off < 0 || cy < 0) ? 0 : X.emis_off + cy];
    } else {
      e = emission_rows(J, J.x.sub + (size_t)i * J.CA, J.y.sub + (size_t)j * J.CA, L);
    }
    r.imm = ((hx_in && hy_in) ? a_imm + lpx + lpy : init) + e;
  } else if (ynull && (xf & F_EMIT_OR_START)) {
    r.imm = hy_in ? left.imm + lpy : init;
  } else if (yok) {
    r.imm = hx_in ? up.imm + lpx : init;
  } else {
    r.imm = init;
  }
  if (!in_env) r = c5_neg_inf();
  return r;
}

// Strip pipeline.  A workgroup of W waves owns one pair; 64*RPT-row strips are dealt to
// its waves round-robin and every wave sweeps its strip left to right on its own clock
// (no workgroup barrier in the loop).  A strip's only input from the strip above is that
// strip's last row, which the wave above has already written to the matrix: the consumer
// block-loads 64 columns of it at a time (L1-bypassing loads) once the producer's
// progress counter (LDS, monotonic) says those columns are complete and drained.
#define HX_PUBLISH_LAG 16
#define HX_CHAIN_MULTI_WAVES 4    // waves per workgroup of the several-workgroups-per-pair launch (MULTI)
#define HX_EXACT_LDS 15000      // table entries (d < 1.5) the exact chain kernel keeps in LDS: 117 KB (the y side may need 33 KB more)
#define HX_YL_MAX_COLS 6144
#define HX_YL_MAX_CLS 64
#define HX_YL_MAX_EMIS 1024

// PPW > 1: a workgroup holds PPW pairs, W waves each, sharing the LDS log-sum-exp table.  Used for
// banded batches: the strips of a banded pair run almost one after the other (a strip's window opens
// when the strip above has all but finished its own), so a pair keeps one wave busy, and the GPU is
// filled by putting many pairs, not many strips, in flight.
//
// MULTI: few pairs of many strips (one rank's share of a strong-scaling run): a pair's strips are dealt to `groups`
// workgroups of W waves, so that its waves spread over several CUs instead of sharing one CU's four SIMDs.  What a strip
// takes from the strip above already travels through the matrix; across workgroups (and XCDs) the stores are write-through
// (`sc1`), the progress counters sit in memory (`counters`, 256 zeroed ints per pair: one per wave, [255] = a poll gave up)
// and are written / polled with `sc1` accesses behind the storing wave's own s_waitcnt - the hand-off of hx_dag.hip's
// lone-pair launches.  A poll gives up after HX_CHAIN_PATIENCE rounds and the pair's lpEnd / lpStart becomes NaN.
#define HX_CHAIN_PATIENCE (1 << 22)
template <int DIR, int RPT, int W, class LSE, bool FAST, bool LEAF, bool YL, bool BANDED, int MINW = 1, int PPW = 1, bool MULTI = false>
__global__ void __launch_bounds__(W * PPW * 64, MINW) k_fill_chain(const DevJob* __restrict__ jobs,
                                                                      const double* __restrict__ exact_tab,
                                                                      const double* __restrict__ fast_tab, const int n_jobs, const int yl_emis,
                                                                      const int groups = 1, int* const counters = nullptr) {
  static_assert(!MULTI || (PPW == 1 && !BANDED && RPT == 1), "several workgroups per pair: unbanded pairs, one pair per workgroup");
  constexpr int THREADS = W * PPW * 64;
  constexpr int SR = 64 * RPT;                      // rows per strip
  static_assert(PPW == 1 || !YL, "the LDS-resident y side belongs to one pair");
  __shared__ volatile int prog[W * PPW];
  // FAST: the 4096-piece table.  Exact, one pair per workgroup: the head of the reference's table (differences
  // below HX_EXACT_LDS * 1e-4), see ExactLse3.  (Not for the one-wave-per-pair launches of banded batches: there
  // 117 KB of LDS per workgroup would leave two waves per CU; measured 16.4 -> 24.9 ms.)
  // Only with 16 waves per pair, where one workgroup fills the CU's register file anyway: for smaller pairs the
  // LDS would cap the workgroups per CU.
  constexpr int XH = (!FAST && PPW == 1 && W == 16) ? HX_EXACT_LDS : 0;
  __shared__ __attribute__((aligned(16))) double ftab[FAST ? (HX_FAST_INTERVALS + 1) * 2 : XH + 2];
  if (FAST) {
    for (int k = threadIdx.x; k < (HX_FAST_INTERVALS + 1) * 2; k += THREADS) ftab[k] = fast_tab[k];
  } else {
    for (int k = threadIdx.x; k < XH + 2; k += THREADS) ftab[k] = exact_tab[k];
  }
  if (threadIdx.x < W * PPW) prog[threadIdx.x] = 0;
  const LSE L = FAST ? LSE::make((const double*)ftab) : LSE::make(fast_tab /* exact mode: the pair table */, ftab, XH);
  const ExactLse LX{exact_tab};

  // (wave-uniform by construction; said explicitly so that the job record is addressed with scalar loads)
  const int pair_in_wg = PPW == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) / W;
  const int G = MULTI ? groups : 1;
  const int job_index = MULTI ? (int)blockIdx.x / G : (int)blockIdx.x * PPW + pair_in_wg;
  const int grp = MULTI ? (int)blockIdx.x % G : 0;
  const bool live = PPW == 1 || job_index < n_jobs;   // (PPW == 1: the grid is exactly n_jobs [x groups])
  const DevJob& J = jobs[live ? job_index : 0];
  // YL: the whole y side lives in LDS (leaf-like y profile whose transitions all have
  // lpTrans 0): per column one word {emission class, not-ready bit}, per class
  // {rootsuby, insy}, and the padded class-pair emission table.  The step loop then
  // issues no vector-memory loads at all, only its write-once stores.
  // (column words and emission table in dynamic LDS, sized for the batch: with 2001 columns and 22x22
  // classes a workgroup needs 78 KB in all, so two workgroups fit a CU)
  extern __shared__ __attribute__((aligned(16))) unsigned char yl_dyn[];
  double* elds = reinterpret_cast<double*>(yl_dyn);
  unsigned* ycol = reinterpret_cast<unsigned*>(yl_dyn + sizeof(double) * (size_t)(YL ? yl_emis : 0));
  __shared__ __attribute__((aligned(16))) double yclass[YL ? 2 * HX_YL_MAX_CLS : 2];
  if (YL) {
    const int Ky1 = J.y.n_cls + 1, Kx1 = J.x.n_cls + 1;
    for (int j = threadIdx.x; j < J.y.n; j += THREADS)
      ycol[j] = (unsigned)J.y.ecls[j] | (J.y.pack[4 * (size_t)j + 3] < 0.0 ? 0x10000u : 0u);
    for (int c = threadIdx.x; c < Ky1; c += THREADS) {
      const bool real = c < J.y.n_cls;
      const int rep = real ? J.y.cls_rep[c] : 0;
      yclass[2 * c] = real ? J.y.pack[4 * (size_t)rep + 1] : HX_NEG_INF;
      yclass[2 * c + 1] = real ? J.y.pack[4 * (size_t)rep + 2] : HX_NEG_INF;
    }
    for (int e = threadIdx.x; e < Kx1 * Ky1; e += THREADS) elds[e] = J.emis_pad[e];
  }
  __syncthreads();
  const int R = J.n_rows, Cc = J.n_cols;
  const int lane = threadIdx.x & 63, wave = PPW == 1 ? (int)(threadIdx.x >> 6) : (int)(threadIdx.x >> 6) % W;   // wave within its pair
  const int64_t plane = J.plane, ss = J.strip_stride;
  static_assert(DIR == 0 || LEAF, "the Backward strip pipeline exists for leaf-like profiles only");
  HX_GLOBAL double* __restrict__ M = as_global(DIR ? J.bwd : J.fwd);
  const HX_GLOBAL d4v* ypack = (const HX_GLOBAL d4v*)as_global(J.y.pack);
  const HX_GLOBAL d4v* xpack = (const HX_GLOBAL d4v*)as_global(J.x.pack);
  const HX_GLOBAL int32_t* yecls = as_global(J.y.ecls);
  const HX_GLOBAL double* epad = as_global(J.emis_pad);
  const HX_GLOBAL uint8_t* yflags = as_global(J.y.flags);
  const HX_GLOBAL int32_t* yenv = as_global(J.y.env);
  // the 30 transition weights, pinned in scalar registers for the whole kernel: left to itself the
  // compiler re-loads some of them inside the step loop (s_load + lgkmcnt(0) stalls, -3 % measured)
  double Tk[5][6];
#pragma unroll
  for (int a = 0; a < 5; ++a)
#pragma unroll
    for (int d = 0; d < 6; ++d) {
      double v = J.T[a][d];
      asm volatile("" : "+s"(v));
      Tk[a][d] = v;
    }
  volatile HX_LDS int* progp = (volatile HX_LDS int*)prog + pair_in_wg * W;   // keep the LDS address space through the lambdas
  const int n_strips = live ? (R + SR - 1) / SR : 0;
  const int WT = W * G, gw = grp * W + wave;        // the pair's waves, and this one among them
  const int prev_wave = (gw + WT - 1) % WT;
  HX_GLOBAL int* gprog = MULTI ? (HX_GLOBAL int*)as_global(counters + 256 * job_index) : nullptr;
  bool dead = false;                                // MULTI: a poll ran out of patience
  const auto wait_above = [&](const int need) {
    if (!MULTI) {
      while (progp[prev_wave] < need) __builtin_amdgcn_s_sleep(1);
      return;
    }
    if (dead) return;
    int polls = 0;
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(gprog + prev_wave, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < need) {
      if (++polls > HX_CHAIN_PATIENCE) {
        dead = true;
        if (lane == 0) __hip_atomic_store(gprog + 255, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
  };
  const auto publish = [&](const int value) {      // (behind the caller's s_waitcnt: the columns up to `value` are out)
    if (lane != 0) return;
    if (MULTI) __hip_atomic_store(gprog + gw, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else progp[wave] = value;
  };

  for (int s = gw; s < n_strips; s += WT) {
    const int row0 = s * SR;
    const int i0 = row0 + lane * RPT;              // first row of this lane
    XRow X[RPT];
    XLeaf XL[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const int i = i0 + k;                      // row in sweep coordinates (mirrored for Backward)
      X[k].valid = i < R;
      const int ic = X[k].valid ? (DIR ? R - 1 - i : i) : 0;   // actual x state
      if (LEAF) {
        const d4v p = xpack[ic];
        if (DIR == 0) {
          XL[k].lp = p.x; XL[k].rootsub = p.y; XL[k].ins = p.z; XL[k].pen = p.w;
          XL[k].eoff = (unsigned)J.x.ecls[ic] * (unsigned)(J.y.n_cls + 1);
        } else {
          const d4v q = xpack[ic + 1];             // the state the absorbing transition leads to
          XL[k].lp = q.x; XL[k].rootsub = q.y; XL[k].ins = q.z; XL[k].pen = p.w;
          XL[k].eoff = (unsigned)J.x.ecls[ic + 1] * (unsigned)(J.y.n_cls + 1);
        }
        XL[k].valid = X[k].valid;
      }
      X[k].flags = J.x.flags[ic];
      X[k].lp = ic > 0 ? J.x.in_lp[ic - 1] : 0.0;      // (chain_cell path, Forward only)
      X[k].rootsub = J.x.rootsub[ic];
      X[k].ins = J.x.ins[ic];
      X[k].env = (J.max_dist >= 0) ? J.x.env[ic] : 0;
      const int cx = J.x.cls[ic];
      X[k].emis_off = cx < 0 ? -1 : cx * J.y.n_cls;
    }
    // Register window, ping-ponged between two sets so that no value is ever copied: at an even step
    // the lane's previous cells are in cb[] (and those of two steps ago in ca[], which the new cells
    // overwrite); the previous lane's last row of one / two steps ago is in ua / ub, and the new
    // shifted-in row overwrites ub.  The next (odd) step swaps the roles.
    C5 ca[RPT], cb[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) { ca[k] = c5_neg_inf(); cb[k] = c5_neg_inf(); }
    C5 ua = c5_neg_inf(), ub = c5_neg_inf();
    C5 bnd = c5_neg_inf();                         // 64 columns of the strip above's last row
    const bool has_above = s > 0;
    const int above_base = ((s - 1) / WT) * Cc;    // columns the producer wave published in earlier strips
    const int my_base = (s / WT) * Cc;
    // strip-skewed store base (hx_device.h cell_slot): a lane's rows are adjacent pairs
    const int strip64 = i0 >> 6;
    const int64_t store_base2 = (int64_t)strip64 * ss + ((i0 & 63) << 1);
    const int t_off = row0 - (strip64 << 6) + 0;   // t64 = j + (i & 63) = step + t_off  (see below)
    const bool store_rows = i0 < ((R + 63) & ~63);
    // band-compressed planes (HX_BAND_COMPRESSED, Forward): a window's cells are stored from the window's own offset
    const int64_t* sbase = (BANDED && RPT == 1 && DIR == 0) ? J.strip_base : nullptr;
    int64_t cstore_base = 0;
    int cstore_t0 = 0;

    const int nsteps = Cc + SR - 1;
    // With a band, a strip only sweeps the step windows that hold its in-envelope cells (computed on
    // the host, hx_api.hip strip_windows; the matrix is pre-filled with -inf).  Windows are widened to
    // even bounds: a row's two cells of steps 2m, 2m+1 are stored together.
    constexpr bool WIN = BANDED && RPT == 1;       // (without it everything below folds to one full sweep)
    int wlo[2] = {0, 0}, whi[2] = {nsteps, 0};
    if (WIN) {
      const HX_GLOBAL int32_t* win = as_global(DIR ? J.bwd_windows : J.fwd_windows);
      if (win) {
        for (int w = 0; w < 2; ++w) {
          wlo[w] = win[4 * s + 2 * w] & ~1;
          const int h = (win[4 * s + 2 * w + 1] + 1) & ~1;
          whi[w] = h < nsteps ? h : nsteps;
        }
        if (whi[1] > wlo[1] && wlo[1] <= whi[0]) { whi[0] = whi[1] > whi[0] ? whi[1] : whi[0]; wlo[1] = whi[1] = 0; }
      }
    }
    int wstart = 0, published = 0;
    // is cell (row0-1, sweep column jj) of the strip above inside the envelope?  (src/forward.h:92-98)
    auto above_in_envelope = [&](const int jj) -> bool {
      if (J.max_dist < 0 || row0 == 0 || jj < 0 || jj >= Cc) return true;
      const int ia = DIR ? R - row0 : row0 - 1;          // actual x state of sweep row row0-1
      const int ja = DIR ? Cc - 1 - jj : jj;
      if ((J.x.flags[ia] | yflags[ja]) & F_EDGE) return true;
      int dd = J.x.env[ia] - yenv[ja];
      dd = dd < 0 ? -dd : dd;
      return dd <= J.max_dist;
    };
    // one anti-diagonal step of the strip; the lane's RPT new cells are returned in out[]
    auto step = [&](const int t, const C5 (&left)[RPT], C5 (&out)[RPT], C5& u1, C5& u2, const d4v (&Yp)[RPT],
                    const double (&ep)[RPT]) {
      if (has_above) {
        if (((t & 63) == 0 || (WIN && t == wstart)) && t < Cc) {
          // wait until the strip above has finished (and drained) the 64-column block that holds column t
          const int tb = t & ~63;
          const int hi = (tb + 64 < Cc) ? tb + 64 : Cc;
          const int need = above_base + hi;
          wait_above(need);
          const int jj = tb + lane;
          bnd = c5_neg_inf();
          const int64_t sl = jj < Cc ? (sbase ? stored_slot(J, row0 - 1, jj) : cell_slot(ss, row0 - 1, jj)) : -1;
          if (sl >= 0) {                           // (not stored: outside the envelope, -inf)
            // agent-scope relaxed loads (global_load ... sc1): served by L2, never by a stale L1 line
            bnd.imm = __hip_atomic_load(M + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bnd.imd = __hip_atomic_load(M + plane + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bnd.idm = __hip_atomic_load(M + 2 * plane + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bnd.imi = __hip_atomic_load(M + 3 * plane + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bnd.iiw = __hip_atomic_load(M + 4 * plane + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          // consume the loads here, so that the wait for them sits in this once-per-64-steps
          // block and not at the merge point every step passes
          asm volatile("" : "+v"(bnd.imm), "+v"(bnd.imd), "+v"(bnd.idm), "+v"(bnd.imi), "+v"(bnd.iiw));
          // with a band the strip above only wrote its in-envelope cells: anything else reads as -inf
          if (BANDED && !above_in_envelope(jj)) bnd = c5_neg_inf();
        }
        const int sel = t & 63;
        C5 a = C5{read_lane(bnd.imm, sel), read_lane(bnd.imd, sel), read_lane(bnd.idm, sel),
                  read_lane(bnd.imi, sel), read_lane(bnd.iiw, sel)};
        if (t >= Cc) a = c5_neg_inf();
        if (lane == 0) u1 = a;                     // (row0-1, t); u2 already holds (row0-1, t-1)
      }
      // rows in descending order: row k reads row k-1's old values, so each row's window can
      // be rotated as soon as its new cell is known
      C5 last = c5_neg_inf();
#pragma unroll
      for (int k = RPT - 1; k >= 0; --k) {
        const int i = i0 + k;
        const int j = t - (lane * RPT + k);
        const bool valid = X[k].valid && j >= 0 && j < Cc;
        const int jm = j < 0 ? 0 : (j >= Cc ? Cc - 1 : j);
        const int jc = DIR ? Cc - 1 - jm : jm;       // actual y state
        const C5& up = (k == 0) ? u1 : left[k - 1];
        const C5& dg = (k == 0) ? u2 : out[k - 1];   // still the value of two steps ago: rows go in descending order
        C5 nw;
        if (LEAF) {
          bool ok = valid;
          if (BANDED || !LEAF) if (J.max_dist >= 0) {
            const uint8_t ef = X[k].flags | yflags[jc];
            int dd = X[k].env - yenv[jc];
            dd = dd < 0 ? -dd : dd;
            ok = ok && ((ef & F_EDGE) || dd <= J.max_dist);
          }
          const double pj = ok ? 0.0 : HX_NEG_INF;
          if (DIR == 0) {
            nw = leaf_cell(Tk, L, XL[k], Yp[k], ep[k], pj, up, left[k], dg);
            if (s == 0 && t == 0 && k == 0) {        // wave-uniform: only the very first step of strip 0
              if (lane == 0) nw.imm = 0.0;           // cell (0,0): lpStart() = 0 (reference src/forward.cpp:73)
            }
          } else {
            nw = leaf_cell_bwd(Tk, L, XL[k], Yp[k], ep[k], pj, up, left[k], dg);
            if (s == 0 && t == 0 && k == 0 && lane == 0) {
              // the cell feeding END is initialised by assignment (reference src/forward.cpp:981-995)
              const double lpe = J.x.pack[4 * (size_t)R] + J.y.pack[4 * (size_t)Cc];
              nw = C5{lpe + J.T[0][5], lpe + J.T[1][5], lpe + J.T[2][5], lpe + J.T[3][5], lpe + J.T[4][5]};
            }
          }
        } else {
          nw = chain_cell(J, L, X[k], X[k].valid ? i : 0, jc, valid, up, left[k], dg);
        }
        if (k == RPT - 1) last = nw;
        out[k] = nw;
      }
      // the previous lane's new last row lands in the u2 slot, which becomes next step's u1
      // (lane 0 keeps the old content; it is re-filled from the boundary block when there is a strip above)
      u2 = wave_shr1_keep0(u2, last);
    };

    // y-side constants and emission terms of step t, fetched ahead of use: vector-memory
    // operations retire in issue order, so a load issued after the previous steps' stores
    // would wait for those stores to reach memory; issued before them it does not.
    // YL: per-row column word of the next step (and, for Backward, the word of the column the previous
    // step visited: in mirrored order that is actual column j+1, whose class the absorbing move needs)
    unsigned wnext[RPT], wprev1[RPT];
    auto init_words = [&](const int t0) {
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const int j0 = t0 - (lane * RPT + k);
        const int jm0 = j0 < 0 ? 0 : (j0 >= Cc ? Cc - 1 : j0);
        wnext[k] = YL ? ycol[DIR ? Cc - 1 - jm0 : jm0] : 0u;
        // Backward: the word of the column visited one step earlier (actual column j+1)
        wprev1[k] = YL ? ycol[DIR ? Cc - jm0 : 0] : 0u;
      }
    };
    auto prefetch = [&](const int t, d4v (&Yp)[RPT], double (&ep)[RPT]) {
      if (!LEAF) return;
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const int j = t - (lane * RPT + k);
        const int jm = j < 0 ? 0 : (j >= Cc ? Cc - 1 : j);
        const int jc = DIR ? Cc - 1 - jm : jm;       // actual y state
        if (YL) {
          // the column word was fetched one step ahead (wnext); fetch the next one now
          const unsigned w = wnext[k];
          {
            const int jn = j + 1;
            const int jnm = jn < 0 ? 0 : (jn >= Cc ? Cc - 1 : jn);
            wnext[k] = ycol[DIR ? Cc - 1 - jnm : jnm];
          }
          const unsigned c = (DIR ? wprev1[k] : w) & 0xFFFFu;
          if (DIR && j >= 0) wprev1[k] = w;    // (before the row starts, the clamped column is not a visit)
          const double2 rc = reinterpret_cast<const double2*>(yclass)[c];
          Yp[k] = d4v{0.0, rc.x, rc.y, __hiloint2double((w & 0x10000u) ? (int)0xFFF00000 : 0, 0)};
          ep[k] = elds[XL[k].eoff + c];
          continue;
        }
        if (DIR) {
          const d4v y0 = ypack[(unsigned)jc], y1 = ypack[(unsigned)jc + 1];
          Yp[k] = d4v{y1.x, y1.y, y1.z, y0.w};
          ep[k] = epad[XL[k].eoff + (unsigned)yecls[(unsigned)jc + 1]];
          continue;
        }
        Yp[k] = ypack[(unsigned)jc];
        ep[k] = epad[XL[k].eoff + (unsigned)yecls[(unsigned)jc]];
      }
    };
    d4v Ya[RPT], Yb[RPT];
    double ea[RPT], eb[RPT];
    for (int w = 0; w < (WIN ? 2 : 1); ++w) {
    if (WIN && whi[w] <= wlo[w]) continue;
    wstart = WIN ? wlo[w] : 0;
    if (WIN && w > 0) {
      // cells left of a window are outside the envelope: the register window restarts from -inf
#pragma unroll
      for (int k = 0; k < RPT; ++k) { ca[k] = c5_neg_inf(); cb[k] = c5_neg_inf(); }
      ua = c5_neg_inf(); ub = c5_neg_inf(); bnd = c5_neg_inf();
    }
    if (WIN && has_above && wstart > 0 && wstart <= Cc) {
      // lane 0's diagonal source at the window's first step, (row0-1, wstart-1), belongs to the strip
      // above and may well be inside the envelope: fetch it (in the steady state it is the boundary
      // value of the previous step)
      const int need = above_base + wstart;
      wait_above(need);
      const int64_t sl = sbase ? stored_slot(J, row0 - 1, wstart - 1) : cell_slot(ss, row0 - 1, wstart - 1);
      if (lane == 0 && sl >= 0) {
        ub.imm = __hip_atomic_load(M + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ub.imd = __hip_atomic_load(M + plane + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ub.idm = __hip_atomic_load(M + 2 * plane + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ub.imi = __hip_atomic_load(M + 3 * plane + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ub.iiw = __hip_atomic_load(M + 4 * plane + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!above_in_envelope(wstart - 1)) ub = c5_neg_inf();
      }
    }
    if (sbase) { cstore_base = sbase[2 * s + w] + ((i0 & 63) << 1); cstore_t0 = wstart; }
    init_words(wstart);
    if (!YL) {
      prefetch(wstart, Ya, ea);
      prefetch(wstart + 1, Yb, eb);
    }

    // Two steps per iteration: in the strip-skewed layout the two cells a row produces on
    // consecutive anti-diagonals are adjacent, so a lane stores RPT*16 contiguous bytes per
    // state plane every second step (a wave: RPT KiB, fully coalesced).
    const int wend = WIN ? whi[w] : nsteps;
    for (int t = wstart; t < wend; t += 2) {
      C5 (&oa)[RPT] = ca;
      C5 (&ob)[RPT] = cb;
      if (YL) prefetch(t, Ya, ea);
      step(t, cb, ca, ua, ub, Ya, ea);
      if (t + 1 < wend) {
        if (YL) prefetch(t + 1, Yb, eb);
        step(t + 1, ca, cb, ub, ua, Yb, eb);
      }
      if (!YL) {
        prefetch(t + 2, Ya, ea);                   // before this pair's stores (see above)
        prefetch(t + 3, Yb, eb);
      }
      const int t64 = t + t_off;                   // even: j + (i & 63) of the first of the two steps
      if (t64 >= 0 && t64 < Cc + 63 && store_rows) {
        typedef double d2v __attribute__((ext_vector_type(2)));
        const int64_t sl = sbase ? cstore_base + ((int64_t)((t - cstore_t0) >> 1) << 7) : store_base2 + ((int64_t)(t64 >> 1) << 7);
        HX_GLOBAL d2v* M2 = (HX_GLOBAL d2v*)(M + sl);
        const int64_t plane2 = plane >> 1;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
          if (MULTI) {
            // the strip below may run on another XCD: write-through stores (never `nt`, which stays in this XCD's L2)
            const d2v v0{oa[k].imm, ob[k].imm}, v1{oa[k].imd, ob[k].imd}, v2{oa[k].idm, ob[k].idm}, v3{oa[k].imi, ob[k].imi}, v4{oa[k].iiw, ob[k].iiw};
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(&M2[k]), "v"(v0) : "memory");
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(&M2[plane2 + k]), "v"(v1) : "memory");
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(&M2[2 * plane2 + k]), "v"(v2) : "memory");
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(&M2[3 * plane2 + k]), "v"(v3) : "memory");
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(&M2[4 * plane2 + k]), "v"(v4) : "memory");
            continue;
          }
          // write-once data: non-temporal stores (the 1/64 of it that the strip below reads back comes from L2 or memory)
          __builtin_nontemporal_store(d2v{oa[k].imm, ob[k].imm}, &M2[k]);
          __builtin_nontemporal_store(d2v{oa[k].imd, ob[k].imd}, &M2[plane2 + k]);
          __builtin_nontemporal_store(d2v{oa[k].idm, ob[k].idm}, &M2[2 * plane2 + k]);
          __builtin_nontemporal_store(d2v{oa[k].imi, ob[k].imi}, &M2[3 * plane2 + k]);
          __builtin_nontemporal_store(d2v{oa[k].iiw, ob[k].iiw}, &M2[4 * plane2 + k]);
        }
      }
      // publish progress.  The strip's last row has finished column (t + 1) - (SR - 1); a
      // column counts as published once its stores have left the wave.  Vector-memory
      // operations retire in issue order and every iteration (two steps) issues at least its
      // 5 * RPT stores (the LDS-resident-y path issues nothing else), so everything stored
      // HX_PUBLISH_LAG steps = HX_PUBLISH_LAG / 2 iterations ago is older than the wave's
      // PUBLISH_WAIT youngest operations.
      constexpr int STORES_PER_ITER = 5 * RPT;
      constexpr int PUBLISH_WAIT = STORES_PER_ITER * (HX_PUBLISH_LAG / 2) < 63 ? STORES_PER_ITER * (HX_PUBLISH_LAG / 2) : 63;
      static_assert(PUBLISH_WAIT <= STORES_PER_ITER * (HX_PUBLISH_LAG / 2) && PUBLISH_WAIT <= 63 && HX_PUBLISH_LAG % 2 == 0,
                    "the wait count must not exceed the operations issued since the published column's stores");
      const int fin = (t + 2 < nsteps ? t + 2 : nsteps) - (SR - 1);
      if (fin >= Cc) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (WIN) published = Cc;
        publish(my_base + Cc);
      } else {
        const int done = fin - HX_PUBLISH_LAG;
        if (done > (WIN ? published : 0) && ((done >> 6) != ((done - 2) >> 6))) {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PUBLISH_WAIT) : "memory");
          if (WIN) published = done;
          publish(my_base + done);
        }
      }
    }
    // end of a window: what lies between it and the next one (or the end of the strip) holds no
    // in-envelope cell, so those columns are complete as soon as the window's stores have drained
    if (WIN) {
      const int upto = (w == 0 && whi[1] > wlo[1]) ? wlo[1] - (SR - 1) : Cc;
      const int done = upto > Cc ? Cc : upto;
      if (done > published) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        published = done;
        publish(my_base + done);
      }
    }
    }
    if (WIN && published < Cc) {   // (a strip without any window still releases the strip below)
      published = Cc;
      publish(my_base + Cc);
    }
    if (MULTI && s == n_strips - 1) {
      // the last strip finishes last (every strip follows the one above) and its own stores are out (vmcnt(0) in front of
      // its last publish); what END reads may lie in other workgroups' strips: drop this CU's L1 first
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const bool gave_up = __hip_atomic_load(gprog + 255, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
      if (lane == 0) {
        if (DIR == 0) *J.lp_end = gave_up ? __builtin_nan("") : forward_lp_end(J, LX);
        else *J.lp_start = gave_up ? __builtin_nan("") : J.bwd[cell_slot(ss, R - 1, Cc - 1)];
      }
    }
  }
  if (MULTI) return;
  __syncthreads();
  if (live && wave == 0 && lane == 0) {
    if (DIR == 0) *J.lp_end = forward_lp_end(J, LX);
    else *J.lp_start = J.bwd[cell_slot(ss, R - 1, Cc - 1)];   // B(0,0).IMM in mirrored coordinates
  }
}

template <int DIR, int RPT, int W, int MINW = 1>
static int launch_variant(const DevJob* d_jobs, int n_jobs, const double* tab, const double* fast_tab, bool fast,
                           int leaf, bool banded, int yl_cols, int yl_emis, hipStream_t st) {
  const dim3 g(n_jobs), b(W * 64);
  const size_t dyn = leaf == 2 ? sizeof(double) * (size_t)yl_emis + sizeof(unsigned) * (size_t)yl_cols : 0;
  // the kernels' fixed-size y-side tables (HX_YL_*): hx_api.hip admits only jobs within them to this path
  if (leaf == 2 && (yl_cols > HX_YL_MAX_COLS || yl_emis > HX_YL_MAX_EMIS + 2))
    return launch_fail("LDS-resident y side of %d columns / %d class pairs exceeds the chain kernel's tables", yl_cols, yl_emis);
#define HX_LAUNCH(LSE_, FAST_, LEAF_, YL_, BANDED_) do { \
  HX_CHECK_LDS((k_fill_chain<DIR, RPT, W, LSE_, FAST_, LEAF_, YL_, BANDED_, MINW>), dyn, "k_fill_chain"); \
  hipLaunchKernelGGL((k_fill_chain<DIR, RPT, W, LSE_, FAST_, LEAF_, YL_, BANDED_, MINW>), g, b, dyn, st, d_jobs, tab, fast_tab, n_jobs, yl_emis); } while (0)
  if (leaf == 2 && !banded) {             // the headline configuration: unbanded leaf pairs, y side in LDS
    if (fast) HX_LAUNCH(FastLse, true, true, true, false); else HX_LAUNCH(ExactLse3, false, true, true, false);
  } else if (leaf == 2) {
    if (fast) HX_LAUNCH(FastLse, true, true, true, true); else HX_LAUNCH(ExactLse3, false, true, true, true);
  } else if (leaf == 1) {
    if (fast) HX_LAUNCH(FastLse, true, true, false, true); else HX_LAUNCH(ExactLse3, false, true, false, true);
  } else if (DIR == 0) {
    if (fast)
      hipLaunchKernelGGL((k_fill_chain<0, RPT, W, FastLse, true, false, false, true, MINW>), g, b, 0, st, d_jobs, tab, fast_tab, n_jobs, 0);
    else
      hipLaunchKernelGGL((k_fill_chain<0, RPT, W, ExactLse3, false, false, false, true, MINW>), g, b, 0, st, d_jobs, tab, fast_tab, n_jobs, 0);
  } else {
    return launch_fail("the Backward strip pipeline exists for leaf-like profiles only");
  }
#undef HX_LAUNCH
  return 0;
}

// banded leaf-like batches: one wave per pair, PPW pairs per workgroup (see k_fill_chain, PPW); few pairs
// per workgroup while that still leaves a workgroup for every CU
template <int DIR, int PPW>
static void launch_banded_leaf_ppw(const DevJob* d_jobs, int n_jobs, const double* tab, const double* fast_tab, bool fast,
                                   hipStream_t st) {
  const dim3 g((n_jobs + PPW - 1) / PPW), b(PPW * 64);
  if (fast)
    hipLaunchKernelGGL((k_fill_chain<DIR, 1, 1, FastLse, true, true, false, true, 1, PPW>), g, b, 0, st, d_jobs, tab, fast_tab, n_jobs, 0);
  else
    hipLaunchKernelGGL((k_fill_chain<DIR, 1, 1, ExactLse3, false, true, false, true, 1, PPW>), g, b, 0, st, d_jobs, tab, fast_tab, n_jobs, 0);
}
template <int DIR>
static void launch_banded_leaf(const DevJob* d_jobs, int n_jobs, const double* tab, const double* fast_tab, bool fast,
                               hipStream_t st) {
  if (n_jobs <= 1024) launch_banded_leaf_ppw<DIR, 2>(d_jobs, n_jobs, tab, fast_tab, fast, st);
  else launch_banded_leaf_ppw<DIR, 8>(d_jobs, n_jobs, tab, fast_tab, fast, st);
}

// few unbanded leaf pairs of many strips: `multi` workgroups of four waves per pair (k_fill_chain, MULTI)
template <int DIR>
static int launch_chain_multi(const DevJob* d_jobs, int n_jobs, const double* tab, const double* fast_tab, bool fast,
                              int yl_cols, int yl_emis, int multi, int* counters, hipStream_t st) {
  constexpr int W = HX_CHAIN_MULTI_WAVES;
  const dim3 g(n_jobs * multi), b(W * 64);
  const size_t dyn = sizeof(double) * (size_t)yl_emis + sizeof(unsigned) * (size_t)yl_cols;
  if (yl_cols > HX_YL_MAX_COLS || yl_emis > HX_YL_MAX_EMIS + 2)
    return launch_fail("LDS-resident y side of %d columns / %d class pairs exceeds the chain kernel's tables", yl_cols, yl_emis);
  if (fast) {
    HX_CHECK_LDS((k_fill_chain<DIR, 1, W, FastLse, true, true, true, false, 1, 1, true>), dyn, "k_fill_chain<multi>");
    hipLaunchKernelGGL((k_fill_chain<DIR, 1, W, FastLse, true, true, true, false, 1, 1, true>), g, b, dyn, st, d_jobs, tab, fast_tab, n_jobs, yl_emis, multi, counters);
  } else {
    HX_CHECK_LDS((k_fill_chain<DIR, 1, W, ExactLse3, false, true, true, false, 1, 1, true>), dyn, "k_fill_chain<multi>");
    hipLaunchKernelGGL((k_fill_chain<DIR, 1, W, ExactLse3, false, true, true, false, 1, 1, true>), g, b, dyn, st, d_jobs, tab, fast_tab, n_jobs, yl_emis, multi, counters);
  }
  return 0;
}

template <int DIR>
static int launch_chain(const DevJob* d_jobs, int n_jobs, int max_rows, const double* tab, const double* fast_tab,
                         bool fast, int leaf, bool banded, int yl_cols, int yl_emis, int multi, int* counters, hipStream_t st) {
  if (multi > 1 && leaf == 2 && !banded)
    return launch_chain_multi<DIR>(d_jobs, n_jobs, tab, fast_tab, fast, yl_cols, yl_emis, multi, counters, st);
  const char* v = getenv("HX_CHAIN_VARIANT");   // tuning hook: override for long profiles
  const int vi = v ? atoi(v) : 0;
  if (banded && leaf >= 1 && vi == 0 && n_jobs >= 64) {
    launch_banded_leaf<DIR>(d_jobs, n_jobs, tab, fast_tab, fast, st);
    return 0;
  }
  if (max_rows <= 64)
    return launch_variant<DIR, 1, 1>(d_jobs, n_jobs, tab, fast_tab, fast, leaf, banded, yl_cols, yl_emis, st);
  if (max_rows <= 128)
    return launch_variant<DIR, 1, 2>(d_jobs, n_jobs, tab, fast_tab, fast, leaf, banded, yl_cols, yl_emis, st);
  if (max_rows <= 256)
    return launch_variant<DIR, 1, 4>(d_jobs, n_jobs, tab, fast_tab, fast, leaf, banded, yl_cols, yl_emis, st);
  if (vi == 2 || (vi == 0 && max_rows <= 512))
    return launch_variant<DIR, 1, 8>(d_jobs, n_jobs, tab, fast_tab, fast, leaf, banded, yl_cols, yl_emis, st);
  if (vi == 4)
    return launch_variant<DIR, 2, 4>(d_jobs, n_jobs, tab, fast_tab, fast, leaf, banded, yl_cols, yl_emis, st);
  return launch_variant<DIR, 1, 16>(d_jobs, n_jobs, tab, fast_tab, fast, leaf, banded, yl_cols, yl_emis, st);   // measured fastest on 2x2000
}

// Workgroups per pair for a batch of n_jobs unbanded leaf pairs (y side in LDS) of up to max_rows rows: 1 = the ordinary launch.
// Several only when the ordinary launch (one workgroup of sixteen waves per pair) would leave most CUs idle - at most 128
// pairs - and never more than 256 workgroups, one per CU (two fit: every workgroup of the launch has to be resident).
// HX_CHAIN_MULTI: 0 = never, n > 1 = that many workgroups per pair (tuning / test hook).
// max_pairs: 128 for the table policies' kernel, 64 for the scaled-probability one (k_fill_leaf_linear: 128 pairs measured
// 5.1 ms in the ordinary launch, 5.5 ms dealt out).
// A pair owns 256 progress counters: one per wave of its workgroups (slot groups * W + wave, k_fill_chain) or one per workgroup
// (k_fill_leaf_linear), and slot 255 is the pair's "a poll gave up" flag - so at most 255 / W workgroups per pair, however
// few pairs and however many strips (a pair of more than ~16 000 rows would otherwise reach slot 255 and beyond).
int chain_multi_groups(int n_jobs, int max_rows, int max_pairs) {
  const int strips = (max_rows + 63) / 64;
  const int cap = std::min((strips + HX_CHAIN_MULTI_WAVES - 1) / HX_CHAIN_MULTI_WAVES, 255 / HX_CHAIN_MULTI_WAVES);
  static_assert((255 / HX_CHAIN_MULTI_WAVES) * HX_CHAIN_MULTI_WAVES <= 255, "progress slots of a pair end below its gave-up flag");
  if (const char* e = getenv("HX_CHAIN_MULTI")) {
    const int forced = atoi(e);
    if (forced <= 1) return 1;
    return std::max(1, std::min(forced, cap));
  }
  if (n_jobs > max_pairs || strips <= 16) return 1;
  const int groups = std::min(cap, 256 / n_jobs);
  return groups >= 2 ? groups : 1;
}

// leaf: 0 = general chain profiles, 1 = leaf-like, 2 = leaf-like with the y side in LDS
// multi > 1 (unbanded pairs with the y side in LDS only, see chain_multi_groups): workgroups per pair; counters: 256 zeroed ints per pair
int launch_forward_chain(const DevJob* d_jobs, int n_jobs, int max_rows, Tab8 tab8, Tab16 tab16,
                         bool fast, int leaf, bool banded, int yl_cols, int yl_emis, int multi, int* counters, hipStream_t st) {
  const double* tab = tab8.p;
  const double* fast_tab = tab16.p;
  return launch_chain<0>(d_jobs, n_jobs, max_rows, tab, fast_tab, fast, leaf, banded, yl_cols, yl_emis, multi, counters, st);
}

// leaf-like profiles only (leaf >= 1)
int launch_backward_chain(const DevJob* d_jobs, int n_jobs, int max_rows, Tab8 tab8, Tab16 tab16,
                          bool fast, int leaf, bool banded, int yl_cols, int yl_emis, int multi, int* counters, hipStream_t st) {
  const double* tab = tab8.p;
  const double* fast_tab = tab16.p;
  if (leaf < 1) return launch_fail("the Backward strip pipeline exists for leaf-like profiles only");
  return launch_chain<1>(d_jobs, n_jobs, max_rows, tab, fast_tab, fast, leaf, banded, yl_cols, yl_emis, multi, counters, st);
}

}  // namespace hx
