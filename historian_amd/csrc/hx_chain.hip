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
#include <cstdlib>
#include "hx_device.h"
#include "hx_lse.h"
#include "hx_common.h"
#include "hx_kernels.h"

namespace hx {

struct FastLse {
  const double* lds;   // [HX_FAST_INTERVALS + 1][4] cubic coefficients in t in [0,1); last piece all zero
  // Branch- and select-free.  d = |a - b| is scaled to table units and clamped to the
  // all-zero guard piece, which yields T = 0 for d >= 10, d = +inf and d = NaN (-inf - -inf):
  // exactly the reference's truncation (v_min_f64 returns the non-NaN operand).
  __device__ __forceinline__ double operator()(double a, double b) const {
    const double mx = __builtin_fmax(a, b);
    const double d = a - b;
    const double s = __builtin_fmin(__builtin_fabs(d) * (HX_FAST_INTERVALS / 10.0), (double)HX_FAST_INTERVALS);
    const int k = (int)s;
    const double t = __builtin_amdgcn_fract(s);
    const double2* c = reinterpret_cast<const double2*>(lds) + 2 * k;
    const double2 c01 = c[0], c23 = c[1];
    const double r = __builtin_fma(__builtin_fma(__builtin_fma(c23.y, t, c23.x), t, c01.y), t, c01.x);
    return mx + r;
  }
};

struct C5 { double imm, imd, idm, imi, iiw; };

__device__ __forceinline__ C5 c5_neg_inf() { return C5{HX_NEG_INF, HX_NEG_INF, HX_NEG_INF, HX_NEG_INF, HX_NEG_INF}; }

// value held by the previous lane (lane 0 keeps its own): one v_mov_b32_dpp per dword
__device__ __forceinline__ double wave_shr1(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
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

// Leaf-like profiles (chain, every interior state emits): the reference's per-cell case
// analysis collapses to one formula when the row above / column to the left of the lattice
// read as -inf, and the ready/wait tests become additive 0/-inf penalties (x + 0.0 == x and
// x + -inf == -inf exactly, so exact mode stays bit-identical).  No compares, no selects.
struct XLeaf {
  double lp, rootsub, ins, pen;
  unsigned eoff;      // ecls * (Ky + 1)
  bool valid;
};

template <class LSE>
__device__ __forceinline__ C5 leaf_cell(const double (*T)[6], const LSE& L, const XLeaf& X, const double4& Y,
                                        double e, double pj, const C5& up, const C5& left, const C5& diag) {
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
  const double p1 = Y.w + pj;      // y state ready (or y empty), and the cell exists
  const double p2 = X.pen + pj;    // x state ready (or x empty), and the cell exists
  C5 r;
  r.imd = ((a_imd + X.lp) + X.rootsub) + p1;
  r.iiw = ((a_iiw + X.lp) + X.ins) + p1;
  r.idm = ((a_idm + Y.x) + Y.y) + p2;
  r.imi = ((a_imi + Y.x) + Y.z) + p2;
  r.imm = ((a_imm + X.lp + Y.x) + e) + pj;
  return r;
}

// Strip pipeline.  A workgroup of W waves owns one pair; 64*RPT-row strips are dealt to
// its waves round-robin and every wave sweeps its strip left to right on its own clock
// (no workgroup barrier in the loop).  A strip's only input from the strip above is that
// strip's last row, which the wave above has already written to the matrix: the consumer
// block-loads 64 columns of it at a time (L1-bypassing loads) once the producer's
// progress counter (LDS, monotonic) says those columns are complete and drained.
template <int RPT, int W, class LSE, bool FAST, bool LEAF, int MINW = 1>
__global__ void __launch_bounds__(W * 64, MINW) k_forward_chain(const DevJob* __restrict__ jobs,
                                                                const double* __restrict__ exact_tab,
                                                                const double* __restrict__ fast_tab) {
  constexpr int THREADS = W * 64;
  constexpr int SR = 64 * RPT;                      // rows per strip
  __shared__ int prog[W];
  __shared__ __attribute__((aligned(16))) double ftab[FAST ? (HX_FAST_INTERVALS + 1) * 4 : 2];
  if (FAST) {
    for (int k = threadIdx.x; k < (HX_FAST_INTERVALS + 1) * 4; k += THREADS) ftab[k] = fast_tab[k];
  }
  if (threadIdx.x < W) prog[threadIdx.x] = 0;
  __syncthreads();
  LSE L{FAST ? (const double*)ftab : exact_tab};
  const ExactLse LX{exact_tab};

  const DevJob& J = jobs[blockIdx.x];
  const int R = J.n_rows, Cc = J.n_cols;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t plane = J.plane, ss = J.strip_stride;
  double* __restrict__ M = J.fwd;
  const int n_strips = (R + SR - 1) / SR;
  const int prev_wave = (wave + W - 1) % W;
  volatile int* vprog = prog;

  for (int s = wave; s < n_strips; s += W) {
    const int row0 = s * SR;
    const int i0 = row0 + lane * RPT;              // first row of this lane
    XRow X[RPT];
    XLeaf XL[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const int i = i0 + k;
      X[k].valid = i < R;
      const int ic = X[k].valid ? i : 0;
      if (LEAF) {
        const double4 p = reinterpret_cast<const double4*>(J.x.pack)[ic];
        XL[k].lp = p.x; XL[k].rootsub = p.y; XL[k].ins = p.z; XL[k].pen = p.w;
        XL[k].eoff = (unsigned)J.x.ecls[ic] * (unsigned)(J.y.n_cls + 1);
        XL[k].valid = X[k].valid;
      }
      X[k].flags = J.x.flags[ic];
      X[k].lp = ic > 0 ? J.x.in_lp[ic - 1] : 0.0;
      X[k].rootsub = J.x.rootsub[ic];
      X[k].ins = J.x.ins[ic];
      X[k].env = (J.max_dist >= 0) ? J.x.env[ic] : 0;
      const int cx = J.x.cls[ic];
      X[k].emis_off = cx < 0 ? -1 : cx * J.y.n_cls;
    }
    C5 v1[RPT], v2[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) { v1[k] = c5_neg_inf(); v2[k] = c5_neg_inf(); }
    C5 u1 = c5_neg_inf(), u2 = c5_neg_inf();
    C5 bnd = c5_neg_inf();                         // 64 columns of the strip above's last row
    const bool has_above = s > 0;
    const int above_base = ((s - 1) / W) * Cc;     // columns the producer wave published in earlier strips
    const int my_base = (s / W) * Cc;
    // strip-skewed store base: slot = strip64 * ss + ((j + l) << 6) + l, the lane's rows are contiguous
    const int strip64 = i0 >> 6;
    const int64_t store_base = (int64_t)strip64 * ss + (i0 & 63);
    const int t_off = row0 - (strip64 << 6) + 0;   // t64 = j + (i & 63) = step + t_off  (see below)
    const bool store_rows = i0 < ((R + 63) & ~63);

    const int nsteps = Cc + SR - 1;
    for (int t = 0; t < nsteps; ++t) {
      if (has_above) {
        if ((t & 63) == 0 && t < Cc) {
          // wait until the strip above has finished (and drained) columns t .. t+63
          const int hi = (t + 64 < Cc) ? t + 64 : Cc;
          const int need = above_base + hi;
          while (vprog[prev_wave] < need) __builtin_amdgcn_s_sleep(1);
          const int jj = t + lane;
          bnd = c5_neg_inf();
          if (jj < Cc) {
            const int64_t sl = cell_slot(ss, row0 - 1, jj);
            bnd.imm = __hip_atomic_load(M + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bnd.imd = __hip_atomic_load(M + plane + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bnd.idm = __hip_atomic_load(M + 2 * plane + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bnd.imi = __hip_atomic_load(M + 3 * plane + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bnd.iiw = __hip_atomic_load(M + 4 * plane + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
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
      C5 out[RPT];
#pragma unroll
      for (int k = RPT - 1; k >= 0; --k) {
        const int i = i0 + k;
        const int j = t - (lane * RPT + k);
        const bool valid = X[k].valid && j >= 0 && j < Cc;
        const int jc = j < 0 ? 0 : (j >= Cc ? Cc - 1 : j);
        const C5& up = (k == 0) ? u1 : v1[k - 1];
        const C5& dg = (k == 0) ? u2 : v2[k - 1];
        C5 nw;
        if (LEAF) {
          bool ok = valid;
          if (J.max_dist >= 0) {
            const uint8_t ef = X[k].flags | J.y.flags[jc];
            int dd = X[k].env - J.y.env[jc];
            dd = dd < 0 ? -dd : dd;
            ok = ok && ((ef & F_EDGE) || dd <= J.max_dist);
          }
          const double pj = ok ? 0.0 : HX_NEG_INF;
          const double4 Y = reinterpret_cast<const double4*>(J.y.pack)[(unsigned)jc];
          const double e = J.emis_pad[XL[k].eoff + (unsigned)J.y.ecls[(unsigned)jc]];
          nw = leaf_cell(J.T, L, XL[k], Y, e, pj, up, v1[k], dg);
          if (i == 0 && j == 0) nw.imm = 0.0;
        } else {
          nw = chain_cell(J, L, X[k], X[k].valid ? i : 0, jc, valid, up, v1[k], dg);
        }
        if (k == RPT - 1) last = nw;
        v2[k] = v1[k];
        v1[k] = nw;
        out[k] = nw;
      }
      // store RPT consecutive doubles per plane; t64 = j + (i & 63) is the same for all rows
      // of the lane: j + (i & 63) = t - (i - row0) + (i & 63) = t + row0 - 64 * strip64
      {
        const int t64 = t + t_off;
#if HX_ABLATE == 3 || HX_ABLATE == 4
        if (t64 == 12345678) {
#else
        if (t64 >= 0 && t64 < Cc + 63 && store_rows) {
#endif
          const int64_t sl = store_base + ((int64_t)t64 << 6);
#pragma unroll
          for (int k = 0; k < RPT; ++k) {
            M[sl + k] = out[k].imm;
            M[plane + sl + k] = out[k].imd;
            M[2 * plane + sl + k] = out[k].idm;
            M[3 * plane + sl + k] = out[k].imi;
            M[4 * plane + sl + k] = out[k].iiw;
          }
        }
      }
      u2 = u1;
      const C5 sh = wave_shr1(last);
      if (lane != 0) u1 = sh;
      // publish progress: the strip's last row has finished column t - (SR - 1)
      const int done = t - (SR - 1) + 1;
      if (done > 0 && ((done & 63) == 0 || done == Cc)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's stores have reached L2
        if (lane == 0) vprog[wave] = my_base + done;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) *J.lp_end = forward_lp_end(J, LX);
}

template <int RPT, int W, int MINW = 1>
static void launch_variant(const DevJob* d_jobs, int n_jobs, const double* tab, const double* fast_tab, bool fast,
                           bool leaf, hipStream_t st) {
  const dim3 g(n_jobs), b(W * 64);
  if (fast && leaf)
    hipLaunchKernelGGL((k_forward_chain<RPT, W, FastLse, true, true, MINW>), g, b, 0, st, d_jobs, tab, fast_tab);
  else if (fast)
    hipLaunchKernelGGL((k_forward_chain<RPT, W, FastLse, true, false, MINW>), g, b, 0, st, d_jobs, tab, fast_tab);
  else if (leaf)
    hipLaunchKernelGGL((k_forward_chain<RPT, W, ExactLse, false, true, MINW>), g, b, 0, st, d_jobs, tab, fast_tab);
  else
    hipLaunchKernelGGL((k_forward_chain<RPT, W, ExactLse, false, false, MINW>), g, b, 0, st, d_jobs, tab, fast_tab);
}

void launch_forward_chain(const DevJob* d_jobs, int n_jobs, int max_rows, const double* tab, const double* fast_tab,
                          bool fast, bool leaf, hipStream_t st) {
  const char* v = getenv("HX_CHAIN_VARIANT");   // tuning hook
  const int vi = v ? atoi(v) : 0;
  if (max_rows <= 64)
    launch_variant<1, 1>(d_jobs, n_jobs, tab, fast_tab, fast, leaf, st);
  else if (max_rows <= 128)
    launch_variant<1, 2>(d_jobs, n_jobs, tab, fast_tab, fast, leaf, st);
  else if (max_rows <= 256)
    launch_variant<1, 4>(d_jobs, n_jobs, tab, fast_tab, fast, leaf, st);
  else if (vi == 1) launch_variant<2, 8>(d_jobs, n_jobs, tab, fast_tab, fast, leaf, st);
  else if (vi == 2) launch_variant<1, 8>(d_jobs, n_jobs, tab, fast_tab, fast, leaf, st);
  else if (vi == 3) launch_variant<1, 16>(d_jobs, n_jobs, tab, fast_tab, fast, leaf, st);
  else if (vi == 4) launch_variant<4, 4>(d_jobs, n_jobs, tab, fast_tab, fast, leaf, st);
  else launch_variant<2, 4>(d_jobs, n_jobs, tab, fast_tab, fast, leaf, st);
}

}  // namespace hx
