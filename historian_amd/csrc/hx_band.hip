// Banded Forward fill of leaf-like pairs: the rotating-row anti-diagonal sweep.
//
// Reference: ForwardMatrix::ForwardMatrix (src/forward.cpp:68-223) inside a GuideAlignmentEnvelope
// (src/forward.h:92-98, src/alignpath.cpp:282-310) - the reference's default mode (band 20).
//
// Inside a band an anti-diagonal holds only ~band+1 cells, so the strip pipelines of hx_chain.hip / hx_linear.hip
// (lane <-> row of a 64-row strip) keep a quarter of their lane-steps inside the envelope, restart their register window
// at every strip and hand every strip boundary through memory.  Here there are no strips: ONE wavefront sweeps the
// anti-diagonals k = i + j of the whole pair, lane = i mod 64.  The envelope coordinates of leaf profiles are
// non-decreasing, so the in-envelope cells of a row are one span of columns and the spans move down and right; the rows
// alive on an anti-diagonal are consecutive, far fewer than 64, and a lane that has finished row i takes row i + 64 well
// before the band reaches it (checked on the host: hx_api.hip build_band_rows; pairs that fail the check keep the strip
// pipelines).  A cell's three sources are the lane's own previous cell (left) and the previous lane's cells of one and two
// steps ago (up, diagonal), handed over by one DPP wave_ror:1 per register - lane 63 feeds lane 0, so nothing ever goes
// through memory or LDS between rows.  A lane outside its row's span produces the zero cell (-inf), which is exactly what
// the reference reads for cells outside the envelope.  No envelope arithmetic is left in the step: "inside the envelope"
// is `first step <= k <= last step` of the lane's row.
//
// The two envelope edges that are NOT part of the band - the rest of row 0 (x START: every column is in the envelope) and
// of column Ny-2 (the y state that feeds END) - are one-dimensional: the row-0 cells beyond the band form a chain
// IDM(0,j) = (IDM(0,j-1) + T[IDM][IDM]) + rootsuby[j], IMI likewise (every other term of the reference's sums is -inf,
// and log_sum_exp(-inf, v) = v exactly), and the column's cells away from the band are -inf (y state not ready: no IMD /
// IIW; the cells to their left are outside the envelope).  A second wavefront of the workgroup writes them while the
// first sweeps.
//
// Three arithmetic policies, as everywhere: scaled probabilities (HX_LSE_LINEAR: the step of hx_linear.hip), the
// LDS-table log-sum-exp (HX_LSE_FAST) and the reference's table bit for bit (HX_LSE_EXACT) - the latter two through the
// same leaf_cell as the strip pipeline, so exact mode stays bit-identical to the reference recursion.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "hx_device.h"
#include "hx_lse.h"
#include "hx_common.h"
#include "hx_policy.h"
#include "hx_leafcell.h"
#include "hx_kernels.h"

namespace hx {

namespace {

typedef double d2v __attribute__((ext_vector_type(2)));
typedef int i4v __attribute__((ext_vector_type(4)));

#define HXB_EMIN (-(1 << 28))
#define HXB_LOG_ENTRIES 1536       // the logarithm table of hx_linear.hip (build_log_table)

// value of the previous lane, lane 0 receives lane 63's: one v_mov_b32_dpp wave_ror:1 per dword
__device__ __forceinline__ int ror1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x13C /* wave_ror:1 */, 0xf, 0xf, false); }
__device__ __forceinline__ double ror1(double v) {
  return __hiloint2double(ror1(__double2hiint(v)), ror1(__double2loint(v)));
}

struct L5 { double imm, imd, idm, imi, iiw; int e; };
__device__ __forceinline__ L5 l5_zero() { return L5{0., 0., 0., 0., 0., HXB_EMIN}; }
__device__ __forceinline__ L5 ror1(const L5& c) { return L5{ror1(c.imm), ror1(c.imd), ror1(c.idm), ror1(c.imi), ror1(c.iiw), ror1(c.e)}; }
__device__ __forceinline__ C5 ror1(const C5& c) { return C5{ror1(c.imm), ror1(c.imd), ror1(c.idm), ror1(c.imi), ror1(c.iiw)}; }

// log(m * 2^e), m >= 0 (see hx_linear.hip log_scaled: frexp, one 16-byte table entry, a cubic)
__device__ __forceinline__ double log_scaled(double m, int e, const HX_LDS double* ltab) {
  const double f = __builtin_amdgcn_frexp_mant(m);
  const int k = __builtin_amdgcn_frexp_exp(m);
  const unsigned byte_off = ((unsigned)__double2hiint(f) >> 7) & 0x7FF0u;
  const d2v ce = *(const HX_LDS d2v*)((const HX_LDS char*)ltab + byte_off);
  const double r = __builtin_fma(f, ce.x, -1.0);
  double p = __builtin_fma(r, 1.0 / 3.0, -0.5);
  p = __builtin_fma(p, r, 1.0);
  const double lf = __builtin_fma(p, r, ce.y);
  return __builtin_fma((double)(e + k), 0.693147180559945309417, lf);
}

__device__ __forceinline__ double read_lane(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

enum { POL_LINEAR = 0, POL_FAST = 1, POL_EXACT = 2 };

// LDS plan (bytes), computed on the host (plan_band): the arithmetic's table first, then one block per pair
struct BandPlan { int table, xrec, ycol, yclass, xclass, elds, stride, total, max_rows, max_cols; };

// One row of the sweep (built by hx_api.hip build_band_rows), 16 bytes:
//   x = first owned step | (owned steps - 1) << 16     (owned: an even first and an odd last step - the two cells a row
//       produces on steps 2m, 2m+1 are stored together; the lane is busy with the row from its first to its last owned step)
//   y = emission class | not ready << 8 | lead pad << 9 | tail pad << 10   (pads: owned steps that lie outside the envelope)
//   z = store base A: the cell of step k lives at  A + (k >> 1) * blk + (k & 1)  in a state plane
struct BandRow { int32_t steps, meta, store, pad_; };

template <int POL, int PPW>
__global__ void __launch_bounds__(2 * PPW * 64)
k_fill_band(const DevJob* __restrict__ jobs, const double* __restrict__ exact_tab, const double* __restrict__ pol_tab,
            const BandPlan plan, const int n_jobs, const int write_edges) {
  constexpr int THREADS = 2 * PPW * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const bool helper = wave >= PPW;
  const int pair = helper ? wave - PPW : wave;
  const int job = (int)blockIdx.x * PPW + pair;
  const bool live = job < n_jobs;
  const DevJob& J = jobs[live ? job : 0];
  unsigned char* blkp = lds + plan.table + pair * plan.stride;
  i4v* xrec = reinterpret_cast<i4v*>(blkp + plan.xrec);
  unsigned* ycol = reinterpret_cast<unsigned*>(blkp + plan.ycol);
  double* yclass = reinterpret_cast<double*>(blkp + plan.yclass);
  double* xclass = reinterpret_cast<double*>(blkp + plan.xclass);
  double* elds = reinterpret_cast<double*>(blkp + plan.elds);
  double* ptab = reinterpret_cast<double*>(lds);

  // ---- stage the shared table and the pair's two sides ----
  if (POL == POL_LINEAR) {
    // (entries 1..1023 of the logarithm table are never addressed)
    for (int k = threadIdx.x; k < 2; k += THREADS) ptab[k] = pol_tab[k];
    for (int k = 2048 + threadIdx.x; k < 2 * HXB_LOG_ENTRIES; k += THREADS) ptab[k] = pol_tab[k];
  } else if (POL == POL_FAST) {
    for (int k = threadIdx.x; k < (HX_FAST_INTERVALS + 1) * 2; k += THREADS) ptab[k] = pol_tab[k];
  }
  const int R = J.n_rows, Cc = J.n_cols;
  {
    const int pt = (int)(threadIdx.x & 63) + (helper ? 64 : 0);       // the pair's two waves stage together
    const int Ky1 = J.y.n_cls + 1, Kx1 = J.x.n_cls + 1;
    const i4v* rows = reinterpret_cast<const i4v*>(J.band_rows);
    for (int i = pt; i < R + 64; i += 128) xrec[i] = rows[i];           // (64 sentinel rows past the end: never owned)
    for (int j = pt; j < Cc; j += 128)
      ycol[j] = (unsigned)J.y.ecls[j] | (J.y.pack[4 * (size_t)j + 3] < 0.0 ? 0x100u : 0u);
    for (int c = pt; c < Ky1; c += 128) {
      const bool real = c < J.y.n_cls;
      const int rep = real ? J.y.cls_rep[c] : 0;
      const double rs = real ? J.y.pack[4 * (size_t)rep + 1] : HX_NEG_INF, in = real ? J.y.pack[4 * (size_t)rep + 2] : HX_NEG_INF;
      yclass[2 * c] = POL == POL_LINEAR ? exp(rs) : rs;
      yclass[2 * c + 1] = POL == POL_LINEAR ? exp(in) : in;
    }
    for (int c = pt; c < Kx1; c += 128) {
      const bool real = c < J.x.n_cls;
      const int rep = real ? J.x.cls_rep[c] : 0;
      const double rs = real ? J.x.pack[4 * (size_t)rep + 1] : HX_NEG_INF, in = real ? J.x.pack[4 * (size_t)rep + 2] : HX_NEG_INF;
      xclass[2 * c] = POL == POL_LINEAR ? exp(rs) : rs;
      xclass[2 * c + 1] = POL == POL_LINEAR ? exp(in) : in;
    }
    for (int e = pt; e < Kx1 * Ky1; e += 128) elds[e] = POL == POL_LINEAR ? exp(J.emis_pad[e]) : J.emis_pad[e];
  }
  __syncthreads();
  const int64_t plane = J.plane;
  const int blk = J.blk;
  HX_GLOBAL double* __restrict__ M = as_global(J.fwd);
  const int Ky1 = J.y.n_cls + 1;

  if (helper) {
    // =====================================================================================================
    // the envelope's one-dimensional edges (see the header comment).  Nothing here is read by the sweep.
    // =====================================================================================================
    if (!live) return;
    // row 0 beyond what the sweep owns: the chain in log space (every policy stores log-probabilities)
    const int own0 = (xrec[0].x >> 16) & 0xFFFF;                       // row 0 is owned from step 0 to this step = column
    const double T02 = J.T[0][2], T03 = J.T[0][3], T22 = J.T[2][2], T33 = J.T[3][3];
    const double pen0 = J.x.pack[3];                                   // x START ready (or x empty): 0, else -inf
    if (own0 < Cc - 1) {
      double idm = HX_NEG_INF, imi = HX_NEG_INF;                       // (the chain's value in every lane)
      for (int j0 = 0; j0 < Cc; j0 += 64) {
        const int jl = j0 + lane < Cc ? j0 + lane : Cc - 1;
        const unsigned w = ycol[jl];
        const double lrs = POL == POL_LINEAR ? J.y.pack[4 * (size_t)jl + 1] : yclass[2 * (w & 0xFFu)];
        const double lin = POL == POL_LINEAR ? J.y.pack[4 * (size_t)jl + 2] : yclass[2 * (w & 0xFFu) + 1];
        double kidm = HX_NEG_INF, kimi = HX_NEG_INF;
#pragma unroll
        for (int m = 0; m < 64; ++m) {
          const int j = j0 + m;                                         // (wave-uniform)
          const double r = read_lane(lrs, m), n = read_lane(lin, m);
          if (j >= 1) {
            // leaf_cell for x START: ((a + lpTrans) + rootsuby) + ready-penalty, a = the only finite term of the sum
            idm = (((j == 1 ? T02 : idm + T22) + 0.0) + r) + pen0;
            imi = (((j == 1 ? T03 : imi + T33) + 0.0) + n) + pen0;
          }
          if (lane == m) { kidm = idm; kimi = imi; }
        }
        const int j = j0 + lane;
        if (j > own0 && j < Cc) {
          const int64_t sl = stored_slot(J, 0, j);
          if (sl >= 0) {
            M[sl] = HX_NEG_INF; M[plane + sl] = HX_NEG_INF; M[2 * plane + sl] = kidm; M[3 * plane + sl] = kimi; M[4 * plane + sl] = HX_NEG_INF;
          }
        }
      }
    }
    // column Ny-2 away from the band: -inf (only where the matrix was not pre-filled)
    if (write_edges)
      for (int i = 1 + lane; i < R; i += 64) {
        const i4v rec = xrec[i];
        const int last_col = (rec.x & 0xFFFF) + ((rec.x >> 16) & 0xFFFF) - i;   // column of the row's last owned step
        if (last_col < Cc - 1) {
          const int64_t sl = stored_slot(J, i, Cc - 1);
          if (sl >= 0) {
            M[sl] = HX_NEG_INF; M[plane + sl] = HX_NEG_INF; M[2 * plane + sl] = HX_NEG_INF; M[3 * plane + sl] = HX_NEG_INF; M[4 * plane + sl] = HX_NEG_INF;
          }
        }
      }
    return;
  }

  // =======================================================================================================
  // the sweep
  // =======================================================================================================
  const int n_steps = live ? J.band_steps : 0;
  // the lane's row, decoded; and the raw record of the row it takes next (i + 64), fetched a whole row ahead
  int i = lane, os, oe, as, ae, store;
  unsigned eoff;
  double xc_rs, xc_in;         // exp(rootsubx), exp(insx) (scaled probabilities) or rootsubx, insx
  int x_wait;                  // x state not ready: 2^29 (an exponent shift) / the 0 or -inf penalty lives in xpen
  double xpen;
  i4v nrec;
  d2v nxc = d2v{0., 0.};
  auto decode = [&](const i4v r, const d2v xc) {
    os = r.x & 0xFFFF; oe = os + ((r.x >> 16) & 0xFFFF);
    as = os + ((r.y >> 9) & 1); ae = oe - ((r.y >> 10) & 1);
    if ((r.x & 0xFFFF) == 0xFFFF) { os = 0x7FFFFFF0; oe = 0x7FFFFFF1; as = os; ae = oe; }     // sentinel: never owned
    store = r.z;
    eoff = (unsigned)(r.y & 0xFF) * (unsigned)Ky1;
    x_wait = (r.y & 0x100) ? (1 << 29) : 0;
    xpen = (r.y & 0x100) ? HX_NEG_INF : 0.0;
    xc_rs = xc.x; xc_in = xc.y;
  };
  {
    const i4v r0 = xrec[lane < R ? lane : R];
    decode(r0, reinterpret_cast<const d2v*>(xclass)[r0.y & 0xFF]);
    nrec = xrec[lane + 64 < R ? lane + 64 : R];
  }
  // the 18 transition weights, pinned in scalar registers: probabilities or log-probabilities
  double P[5][6];
#pragma unroll
  for (int a = 0; a < 5; ++a)
#pragma unroll
    for (int d = 0; d < 6; ++d) {
      double v = J.T[a][d];
      if (POL == POL_LINEAR) {
        const bool used = d == 0 || (d == 1 && a != 4) || (d == 2 && a != 3) || (d == 3 && (a == 0 || a == 3)) ||
                          (d == 4 && (a == 0 || a == 3 || a == 4));
        const double pv = used ? exp(v) : 0.;
        v = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(pv)), __builtin_amdgcn_readfirstlane(__double2loint(pv)));
      }
      asm volatile("" : "+s"(v));
      P[a][d] = v;
    }
  const HX_LDS double* lt = (const HX_LDS double*)ptab;
  const FastLse LF = FastLse::make(ptab);
  const ExactLse3 LE = ExactLse3::make(pol_tab);

  // cell registers, ping-ponged: at an even step the lane's previous cell is in cb (the one before in ca, which the new
  // cell overwrites), the previous lane's cells of one / two steps ago in ua / ub
  L5 la = l5_zero(), lb = l5_zero(), lua = l5_zero(), lub = l5_zero();
  C5 ca = c5_neg_inf(), cb = c5_neg_inf(), cua = c5_neg_inf(), cub = c5_neg_inf();

  // the y word of the lane's column, fetched one step ahead
  auto word_at = [&](const int col) -> unsigned {
    const int c = col < 0 ? 0 : (col >= Cc ? Cc - 1 : col);
    return ycol[c];
  };
  unsigned wnext = word_at(0 - i);

  // beginning of a step: a lane whose row ended with the previous step takes row i + 64; a lane whose row ends with this
  // step fetches the class constants of its next row now (so that neither LDS round trip is ever waited for)
  auto roll = [&](const int k) {
    if (k + 1 > oe) {
      if (k > oe) {
        i += 64;
        decode(nrec, nxc);
        nrec = xrec[i + 64 < R ? i + 64 : R];
      } else {
        nxc = reinterpret_cast<const d2v*>(xclass)[nrec.y & 0xFF];
      }
    }
  };
  // the word of step k + 1: of the next row's column when the lane is about to change rows
  auto next_word = [&](const int k) -> unsigned {
    const unsigned w = wnext;
    const int inext = (k + 1 > oe) ? i + 64 : i;
    wnext = word_at(k + 1 - inext);
    return w;
  };

  auto step_linear = [&](const int k, const L5& left, L5& out, L5& u1, L5& u2, const unsigned w) {
    const unsigned c = w & 0xFFu;
    const d2v rc = reinterpret_cast<const d2v*>(yclass)[c];
    const int y_wait = (int)((w & 0x100u) << 21);           // y state not ready: 2^29, else 0
    const double em = elds[eoff + c];
    // the five sums of src/forward.cpp:103-115,139-150,171-180 on probabilities
    double s_imd = u1.imm * P[0][1];
    double s_iiw = u1.imm * P[0][4];
    double s_idm = left.imm * P[0][2];
    double s_imi = left.imm * P[0][3];
    double s_imm = u2.imm * P[0][0];
    s_imd = __builtin_fma(u1.imd, P[1][1], s_imd);
    s_iiw = __builtin_fma(u1.imi, P[3][4], s_iiw);
    s_idm = __builtin_fma(left.imd, P[1][2], s_idm);
    s_imi = __builtin_fma(left.imi, P[3][3], s_imi);
    s_imm = __builtin_fma(u2.imd, P[1][0], s_imm);
    s_imd = __builtin_fma(u1.idm, P[2][1], s_imd);
    s_iiw = __builtin_fma(u1.iiw, P[4][4], s_iiw);
    s_idm = __builtin_fma(left.idm, P[2][2], s_idm);
    s_imm = __builtin_fma(u2.idm, P[2][0], s_imm);
    s_imd = __builtin_fma(u1.imi, P[3][1], s_imd);
    s_idm = __builtin_fma(left.iiw, P[4][2], s_idm);
    s_imm = __builtin_fma(u2.imi, P[3][0], s_imm);
    s_imm = __builtin_fma(u2.iiw, P[4][0], s_imm);
    // common exponent of the new cell, the three source groups brought to it; a state that may not be entered
    // (y or x state not ready: src/forward.cpp:97,133) and a cell outside the envelope are shifted out of range: zero
    int E = left.e > u1.e ? left.e : u1.e;
    E = E > u2.e ? E : u2.e;
    const int outside = (k >= as && k <= ae) ? 0 : (1 << 29);
    const int du = ((u1.e - E) - y_wait) - outside, dl = ((left.e - E) - x_wait) - outside, dd = (u2.e - E) - outside;
    out.imd = __builtin_ldexp(s_imd * xc_rs, du);
    out.iiw = __builtin_ldexp(s_iiw * xc_in, du);
    out.idm = __builtin_ldexp(s_idm * rc.x, dl);
    out.imi = __builtin_ldexp(s_imi * rc.y, dl);
    out.imm = __builtin_ldexp(s_imm * em, dd);
    out.e = E;
    if ((k & 6) == 0) {                            // wave-uniform: renormalise every 8th step
      if (k == 0 && lane == 0) { out.imm = 1.0; out.e = 0; }     // cell (0,0): lpStart() = 0 (src/forward.cpp:73)
      const double mx = vmax(vmax(vmax(out.imm, out.imd), vmax(out.idm, out.imi)), out.iiw);
      const int kk = __builtin_amdgcn_frexp_exp(mx);
      out.imm = __builtin_ldexp(out.imm, -kk);
      out.imd = __builtin_ldexp(out.imd, -kk);
      out.idm = __builtin_ldexp(out.idm, -kk);
      out.imi = __builtin_ldexp(out.imi, -kk);
      out.iiw = __builtin_ldexp(out.iiw, -kk);
      out.e = mx > 0. ? out.e + kk : HXB_EMIN;
    }
    u2 = ror1(out);                                // the previous lane's new cell: next step's upper neighbour
  };

  auto step_log = [&](const int k, const C5& left, C5& out, C5& u1, C5& u2, const unsigned w) {
    const unsigned c = w & 0xFFu;
    const d2v rc = reinterpret_cast<const d2v*>(yclass)[c];
    const double em = elds[eoff + c];
    const double ypen = __hiloint2double((w & 0x100u) ? (int)0xFFF00000 : 0, 0);     // y state not ready: -inf
    const double pj = (k >= as && k <= ae) ? 0.0 : HX_NEG_INF;
    XLeaf X;
    X.lp = 0.0; X.rootsub = xc_rs; X.ins = xc_in; X.pen = xpen; X.eoff = eoff; X.valid = true;
    const d4v Y = d4v{0.0, rc.x, rc.y, ypen};
    C5 nw;
    if (POL == POL_FAST) nw = leaf_cell(P, LF, X, Y, em, pj, u1, left, u2);
    else nw = leaf_cell(P, LE, X, Y, em, pj, u1, left, u2);
    if (k == 0 && lane == 0) nw.imm = 0.0;         // cell (0,0): lpStart() = 0 (src/forward.cpp:73)
    out = nw;
    u2 = ror1(nw);
  };

  for (int k = 0; k < n_steps; k += 2) {
    // a pair of steps: in the strip-skewed layout the two cells are adjacent, 16 bytes per lane and state plane
    double s0[5], s1[5];
    roll(k);
    const bool own = k >= os && k <= oe;           // (owned spans are whole step pairs)
    const int64_t sl = (int64_t)store + (int64_t)(k >> 1) * blk;
    if (POL == POL_LINEAR) {
      step_linear(k, lb, la, lua, lub, next_word(k));
      s0[0] = log_scaled(la.imm, la.e, lt); s0[1] = log_scaled(la.imd, la.e, lt); s0[2] = log_scaled(la.idm, la.e, lt);
      s0[3] = log_scaled(la.imi, la.e, lt); s0[4] = log_scaled(la.iiw, la.e, lt);
      roll(k + 1);
      step_linear(k + 1, la, lb, lub, lua, next_word(k + 1));
      s1[0] = log_scaled(lb.imm, lb.e, lt); s1[1] = log_scaled(lb.imd, lb.e, lt); s1[2] = log_scaled(lb.idm, lb.e, lt);
      s1[3] = log_scaled(lb.imi, lb.e, lt); s1[4] = log_scaled(lb.iiw, lb.e, lt);
    } else {
      step_log(k, cb, ca, cua, cub, next_word(k));
      s0[0] = ca.imm; s0[1] = ca.imd; s0[2] = ca.idm; s0[3] = ca.imi; s0[4] = ca.iiw;
      roll(k + 1);
      step_log(k + 1, ca, cb, cub, cua, next_word(k + 1));
      s1[0] = cb.imm; s1[1] = cb.imd; s1[2] = cb.idm; s1[3] = cb.imi; s1[4] = cb.iiw;
    }
    if (own) {
      HX_GLOBAL d2v* M2 = (HX_GLOBAL d2v*)(M + sl);
      const int64_t plane2 = plane >> 1;
      // write-once data: non-temporal stores
      __builtin_nontemporal_store(d2v{s0[0], s1[0]}, &M2[0]);
      __builtin_nontemporal_store(d2v{s0[1], s1[1]}, &M2[plane2]);
      __builtin_nontemporal_store(d2v{s0[2], s1[2]}, &M2[2 * plane2]);
      __builtin_nontemporal_store(d2v{s0[3], s1[3]}, &M2[3 * plane2]);
      __builtin_nontemporal_store(d2v{s0[4], s1[4]}, &M2[4 * plane2]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (live && lane == 0) *J.lp_end = forward_lp_end(J, ExactLse{exact_tab});
}

BandPlan plan_band(int pol, int ppw, int max_rows, int max_cols, int max_cls) {
  BandPlan p;
  p.table = pol == POL_LINEAR ? 16 * HXB_LOG_ENTRIES : (pol == POL_FAST ? 16 * (HX_FAST_INTERVALS + 1) : 16);
  int a = 0;
  p.xrec = a; a += 16 * (max_rows + 64);
  p.ycol = a; a += (4 * max_cols + 15) & ~15;
  p.yclass = a; a += 16 * (max_cls + 1);
  p.xclass = a; a += 16 * (max_cls + 1);
  p.elds = a; a += (8 * (max_cls + 1) * (max_cls + 1) + 15) & ~15;
  p.stride = a;
  p.total = p.table + ppw * a;
  p.max_rows = max_rows; p.max_cols = max_cols;
  return p;
}

template <int POL, int PPW>
int launch_pol(const DevJob* d_jobs, int n_jobs, const BandPlan& p, const double* tab, const double* pol_tab, int write_edges, hipStream_t st) {
  if (p.total > HX_LDS_LIMIT) return launch_fail("k_fill_band<%d, %d> needs %d bytes of LDS (limit %d)", POL, PPW, p.total, HX_LDS_LIMIT);
  hipLaunchKernelGGL((k_fill_band<POL, PPW>), dim3((n_jobs + PPW - 1) / PPW), dim3(2 * PPW * 64), p.total, st, d_jobs, tab, pol_tab, p,
                     n_jobs, write_edges);
  return 0;
}

}  // namespace

// Largest pair (rows, columns, emission classes) whose two sides fit LDS next to the policy's table with one pair per
// workgroup: hx_api.hip admits a pair to this kernel's class only below it
bool band_kernel_fits(int pol, int rows, int cols, int cls) { return plan_band(pol, 1, rows, cols, cls).total <= HX_LDS_LIMIT; }

int launch_forward_band(const DevJob* d_jobs, int n_jobs, int pol, int max_rows, int max_cols, int max_cls, const double* tab,
                        const double* pol_tab, bool write_edges, hipStream_t st) {
  // pairs per workgroup (they share the policy's table): as many as keep two workgroups on a CU while the batch still
  // gives every CU a workgroup
  const char* v = getenv("HX_BAND_PPW");           // tuning / test hook
  int ppw = v ? atoi(v) : 0;
  if (ppw <= 0) {
    ppw = 1;
    const int half = HX_LDS_LIMIT / 2 - 1024;
    if (n_jobs > 512) {
      for (int c = 2; c <= 4; c *= 2)
        if (plan_band(pol, c, max_rows, max_cols, max_cls).total <= half && n_jobs >= 256 * c) ppw = c;
    }
    if (plan_band(pol, ppw, max_rows, max_cols, max_cls).total > HX_LDS_LIMIT) ppw = 1;
    // the fast table alone is 64 KB: two pairs share it when that lets a CU hold four pairs instead of one
    if (pol == POL_FAST && ppw == 1 && plan_band(pol, 2, max_rows, max_cols, max_cls).total <= HX_LDS_LIMIT &&
        plan_band(pol, 1, max_rows, max_cols, max_cls).total > half)
      ppw = 2;
  }
  const int we = write_edges ? 1 : 0;
#define HXB_GO(POL_) do { \
    if (ppw >= 4) return launch_pol<POL_, 4>(d_jobs, n_jobs, plan_band(POL_, 4, max_rows, max_cols, max_cls), tab, pol_tab, we, st); \
    if (ppw >= 2) return launch_pol<POL_, 2>(d_jobs, n_jobs, plan_band(POL_, 2, max_rows, max_cols, max_cls), tab, pol_tab, we, st); \
    return launch_pol<POL_, 1>(d_jobs, n_jobs, plan_band(POL_, 1, max_rows, max_cols, max_cls), tab, pol_tab, we, st); } while (0)
  if (pol == POL_LINEAR) HXB_GO(POL_LINEAR);
  if (pol == POL_FAST) HXB_GO(POL_FAST);
  HXB_GO(POL_EXACT);
#undef HXB_GO
}

}  // namespace hx
