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
#include "hx_device.h"
#include "hx_lse.h"
#include "hx_common.h"
#include "hx_kernels.h"

namespace hx {

struct FastLse {
  const double* lds;   // [HX_FAST_INTERVALS + 1][4] cubic coefficients in t in [0,1)
  __device__ __forceinline__ double operator()(double a, double b) const {
    const double mx = (a < b) ? b : a;
    const double mn = (a < b) ? a : b;
    const double d = (a == b) ? 0.0 : (mx - mn);
    double r = 0.0;
    if (d < 10.0) {
      const double s = d * (HX_FAST_INTERVALS / 10.0);
      const int k = (int)s;
      const double t = s - (double)k;
      const double2* c = reinterpret_cast<const double2*>(lds) + 2 * k;
      const double2 c01 = c[0], c23 = c[1];
      r = __builtin_fma(__builtin_fma(__builtin_fma(c23.y, t, c23.x), t, c01.y), t, c01.x);
    }
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

template <class LSE>
__device__ __forceinline__ C5 chain_cell(const DevJob& J, const LSE& L, const XRow& X, int i, int j,
                                         const C5& up, const C5& left, const C5& diag) {
  C5 r = c5_neg_inf();
  const uint8_t yf = J.y.flags[j];
  bool in_env = ((X.flags | yf) & F_EDGE) || J.max_dist < 0;
  if (!in_env) {
    int dd = X.env - J.y.env[j];
    dd = dd < 0 ? -dd : dd;
    in_env = dd <= J.max_dist;
  }
  if (!in_env) return r;
  if (i == 0 && j == 0) r.imm = 0.0;
  const uint8_t xf = X.flags;
  const bool xnull = xf & F_NULL, ynull = yf & F_NULL;
  const bool yok = (yf & F_READY) || J.y.empty;
  const bool xok = (xf & F_READY) || J.x.empty;
  const double (*T)[6] = J.T;
  const bool hx_in = i > 0, hy_in = j > 0;
  // lse(-inf, v) == v exactly, so the single in-transition needs no accumulate step.
  if (hx_in) {
    if (!xnull) {
      if (yok) {
        double a = L(up.imm + T[0][1], up.imd + T[1][1]);
        a = L(a, up.idm + T[2][1]);
        a = L(a, up.imi + T[3][1]);
        r.imd = (a + X.lp) + X.rootsub;
        double b = L(up.imm + T[0][4], up.imi + T[3][4]);
        b = L(b, up.iiw + T[4][4]);
        r.iiw = (b + X.lp) + X.ins;
      }
    } else if (yok) {
      r.imd = up.imd + X.lp;
      r.iiw = up.iiw + X.lp;
    }
  } else if (!xnull && yok) {   // emit state without in-transition: -inf + constants
    r.imd += X.rootsub;
    r.iiw += X.ins;
  }
  const double lpy = hy_in ? J.y.in_lp[j - 1] : 0.0;
  if (hy_in) {
    if (!ynull) {
      if (xok) {
        double a = L(left.imm + T[0][2], left.imd + T[1][2]);
        a = L(a, left.idm + T[2][2]);
        a = L(a, left.iiw + T[4][2]);
        r.idm = (a + lpy) + J.y.rootsub[j];
        const double b = L(left.imm + T[0][3], left.imi + T[3][3]);
        r.imi = (b + lpy) + J.y.ins[j];
      }
    } else {
      r.idm = left.idm + lpy;
      r.imi = left.imi + lpy;
    }
  } else if (!ynull && xok) {
    r.idm += J.y.rootsub[j];
    r.imi += J.y.ins[j];
  }
  if (!xnull && !ynull) {
    if (hx_in && hy_in) {
      double a = L(diag.imm + T[0][0], diag.imd + T[1][0]);
      a = L(a, diag.idm + T[2][0]);
      a = L(a, diag.imi + T[3][0]);
      a = L(a, diag.iiw + T[4][0]);
      r.imm = a + X.lp + lpy;     // accumulator was -inf: (0,0) is null x null
    }
    double e;
    if (J.emis) {
      const int cy = J.y.cls[j];
      e = (X.emis_off < 0 || cy < 0) ? HX_NEG_INF : J.emis[X.emis_off + cy];
    } else {
      e = emission_rows(J, J.x.sub + (size_t)i * J.CA, J.y.sub + (size_t)j * J.CA, L);
    }
    r.imm += e;
  } else if (ynull && (xf & F_EMIT_OR_START)) {
    if (hy_in) r.imm = left.imm + lpy;
  } else if (yok) {
    if (hx_in) r.imm = up.imm + X.lp;
  }
  return r;
}

template <int RPT, int THREADS, class LSE, bool FAST>
__global__ void __launch_bounds__(THREADS) k_forward_chain(const DevJob* __restrict__ jobs,
                                                           const double* __restrict__ exact_tab,
                                                           const double* __restrict__ fast_tab) {
  constexpr int NW = THREADS / 64;
  __shared__ double xchg[2][NW][5];
  __shared__ __attribute__((aligned(16))) double ftab[FAST ? (HX_FAST_INTERVALS + 1) * 4 : 2];
  if (FAST) {
    for (int k = threadIdx.x; k < (HX_FAST_INTERVALS + 1) * 4; k += THREADS) ftab[k] = fast_tab[k];
  }
  LSE L{FAST ? (const double*)ftab : exact_tab};
  const ExactLse LX{exact_tab};

  const DevJob& J = jobs[blockIdx.x];
  const int R = J.n_rows, Cc = J.n_cols;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int rows_per_pass = THREADS * RPT;
  const int64_t plane = J.plane, ss = J.strip_stride;
  double* __restrict__ M = J.fwd;

  for (int row0 = 0; row0 < R; row0 += rows_per_pass) {
    const int i0 = row0 + threadIdx.x * RPT;     // first row of this lane in this pass
    XRow X[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const int i = i0 + k;
      X[k].valid = i < R;
      const int ic = X[k].valid ? i : 0;
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
    C5 bnd = c5_neg_inf();                         // wave 0: 64 columns of the previous pass's last row
    if (threadIdx.x < 2 * NW * 5) (&xchg[0][0][0])[threadIdx.x] = HX_NEG_INF;
    __syncthreads();   // also orders the previous pass's stores before this pass's boundary loads

    const int pass_rows = (R - row0 < rows_per_pass) ? (R - row0) : rows_per_pass;
    const int nd = pass_rows + Cc - 1;           // anti-diagonals of this pass, d relative to row0
    const int wrow0 = wave * 64 * RPT;           // first row of the wave, relative to row0
    const bool has_prev_pass = row0 > 0;
    for (int d = 0; d < nd; ++d) {
      // rows [wrow0, wrow0 + 64*RPT) are active when some 0 <= d - (i - row0) < Cc
      const bool wave_active = (d >= wrow0) && (d - (wrow0 + 64 * RPT - 1) < Cc) && (wrow0 < pass_rows);
      if (wave_active) {
        if (wave == 0) {
          if (has_prev_pass) {
            // row row0-1, columns d..d+63, fetched once per 64 steps (lane l holds column (d&~63)+l)
            if ((d & 63) == 0) {
              const int jj = d + lane;
              bnd = c5_neg_inf();
              if (jj < Cc) {
                const int64_t sl = cell_slot(ss, row0 - 1, jj);
                bnd = C5{M[sl], M[plane + sl], M[2 * plane + sl], M[3 * plane + sl], M[4 * plane + sl]};
              }
            }
            const int sel = d & 63;
            const C5 a = C5{read_lane(bnd.imm, sel), read_lane(bnd.imd, sel), read_lane(bnd.idm, sel),
                            read_lane(bnd.imi, sel), read_lane(bnd.iiw, sel)};
            if (lane == 0) u1 = a;    // (row0-1, d); u2 already holds (row0-1, d-1)
          }
        } else if (lane == 0) {
          const double* s = xchg[(d - 1) & 1][wave - 1];
          u1 = C5{s[0], s[1], s[2], s[3], s[4]};
        }
        // rows in descending order: row k reads row k-1's old values, so each row's
        // window can be rotated as soon as its new cell is known
        C5 last = c5_neg_inf();
        C5 out[RPT];
#pragma unroll
        for (int k = RPT - 1; k >= 0; --k) {
          const int i = i0 + k;
          const int j = d - (i - row0);
          C5 nw = c5_neg_inf();
          if (X[k].valid && j >= 0 && j < Cc) {
            const C5& up = (k == 0) ? u1 : v1[k - 1];
            const C5& dg = (k == 0) ? u2 : v2[k - 1];
            nw = chain_cell(J, L, X[k], i, j, up, v1[k], dg);
          }
          if (k == RPT - 1) last = nw;
          v2[k] = v1[k];
          v1[k] = nw;
          out[k] = nw;
        }
        // store: RPT consecutive doubles per plane; the lanes of a strip are contiguous
        {
          const int strip = i0 >> 6;
          const int t = d + row0 - (strip << 6);       // = j + (i & 63), same for all rows of the lane
          if (t >= 0 && t < Cc + 63 && i0 < ((R + 63) & ~63)) {
            const int64_t sl = (int64_t)strip * ss + ((int64_t)t << 6) + (i0 & 63);
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
        if (lane == 63 && wave + 1 < NW) {
          double* s = xchg[d & 1][wave];
          s[0] = last.imm; s[1] = last.imd; s[2] = last.idm; s[3] = last.imi; s[4] = last.iiw;
        }
        u2 = u1;
        const C5 sh = wave_shr1(last);
        if (lane != 0) u1 = sh;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) *J.lp_end = forward_lp_end(J, LX);
}

template <int RPT, int THREADS>
static void launch_variant(const DevJob* d_jobs, int n_jobs, const double* tab, const double* fast_tab, bool fast,
                           hipStream_t st) {
  if (fast)
    hipLaunchKernelGGL((k_forward_chain<RPT, THREADS, FastLse, true>), dim3(n_jobs), dim3(THREADS), 0, st, d_jobs, tab, fast_tab);
  else
    hipLaunchKernelGGL((k_forward_chain<RPT, THREADS, ExactLse, false>), dim3(n_jobs), dim3(THREADS), 0, st, d_jobs, tab, fast_tab);
}

void launch_forward_chain(const DevJob* d_jobs, int n_jobs, int max_rows, const double* tab, const double* fast_tab,
                          bool fast, hipStream_t st) {
  if (max_rows <= 64)
    launch_variant<1, 64>(d_jobs, n_jobs, tab, fast_tab, fast, st);
  else if (max_rows <= 128)
    launch_variant<1, 128>(d_jobs, n_jobs, tab, fast_tab, fast, st);
  else if (max_rows <= 256)
    launch_variant<1, 256>(d_jobs, n_jobs, tab, fast_tab, fast, st);
  else
    launch_variant<2, 256>(d_jobs, n_jobs, tab, fast_tab, fast, st);
}

}  // namespace hx
