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
// A pair's critical path is its Nx + Ny anti-diagonals, and with 512 pairs on 1024 SIMDs nothing else runs on the sweep's
// SIMD: the step's instruction count is the fill's speed.  The workgroup therefore gives every pair a SECOND wavefront:
//   * scaled probabilities (HX_LSE_LINEAR): the sweep only runs the recursion (18 multiply-adds, exponent bookkeeping,
//     the DPP hand-over) and drops each cell - five mantissas, the exponent, the store slot - into an LDS ring; the second
//     wave takes the cells out, turns them into the reference's storage format (five logarithms per cell, 60 of the
//     step's ~170 vector instructions) and stores them.  Monotonic step counters in LDS in both directions.
//   * all policies: the two envelope edges that are NOT part of the band - the rest of row 0 (x START: every column is
//     in the envelope) and of column Ny-2 (the y state that feeds END) - are one-dimensional: the row-0 cells beyond the
//     band form a chain IDM(0,j) = (IDM(0,j-1) + T[IDM][IDM]) + rootsuby[j], IMI likewise (every other term of the
//     reference's sums is -inf, and log_sum_exp(-inf, v) = v exactly); the column's cells away from the band are -inf
//     (y state not ready: no IMD / IIW; the cells to their left are outside the envelope) except (1, Ny-2), whose diagonal
//     source lies in row 0.  The second wave writes them.
//
// Three arithmetic policies, as everywhere: scaled probabilities (the step of hx_linear.hip), the LDS-table log-sum-exp
// (HX_LSE_FAST) and the reference's table bit for bit (HX_LSE_EXACT) - the latter two through the same leaf_cell as the
// strip pipeline, so exact mode stays bit-identical to the reference recursion.
//
// Storage: the strip-skewed planes of hx_device.h, dense or band-compressed, written only where a row owns steps
// (its span widened to whole step pairs; the pad cells are outside the envelope and get -inf).  Cells outside the
// envelope that a dense plane holds are -inf from the pre-fill; with HX_SPARSE_ENVELOPE or HX_BAND_COMPRESSED they
// are undefined (readers test the envelope, as with the reference's sparse cell map).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "hx_device.h"
#include "hx_lse.h"
#include "hx_common.h"
#include "hx_policy.h"
#include "hx_leafcell.h"
#include "hx_kernels.h"
#include "hx_bandedge.h"

namespace hx {

namespace {

typedef double d2v __attribute__((ext_vector_type(2)));
typedef int i2v __attribute__((ext_vector_type(2)));

#define HXB_EMIN (-(1 << 28))
#define HXB_LOG_ENTRIES 1536       // the logarithm table of hx_linear.hip (build_log_table)
#define HXB_RING 8                 // steps of cells in flight between the sweep and the converting wave
#define HXB_RING_LEAN 4

// value of the previous lane, lane 0 receives lane 63's: one v_mov_b32_dpp wave_ror:1 per dword
__device__ __forceinline__ int ror1(int v) { return __builtin_amdgcn_mov_dpp(v, 0x13C /* wave_ror:1 */, 0xf, 0xf, false); }   // (every lane is written: no tied "old" operand, no copy)
__device__ __forceinline__ double ror1(double v) {
  return __hiloint2double(ror1(__double2hiint(v)), ror1(__double2loint(v)));
}

struct L5 { double imm, imd, idm, imi, iiw; int e; };
__device__ __forceinline__ L5 l5_zero() { return L5{0., 0., 0., 0., 0., HXB_EMIN}; }
// (hx_linear.hip) the reference's pairwise log_sum_exp on probabilities: the smaller term is dropped when it is at most e^-10 of the larger
__device__ __forceinline__ double trunc_sum(double a, double b) {
  const double hi = fmax_plain(a, b), lo = fmin_plain(a, b);
  // (a dropped term keeps its low word: a number below 2^-1042 that no sum of mantissas scaled to the cell's exponent feels - one select instead of two)
  const int keep = lo > hi * 4.5399929762484854e-05 ? __double2hiint(lo) : 0;
  return hi + __hiloint2double(keep, __double2loint(lo));
}
template <bool TRUNC> __device__ __forceinline__ double lin_acc(double m, double p, double acc) {
  if (TRUNC) return trunc_sum(acc, m * p);
  return __builtin_fma(m, p, acc);
}
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

enum { POL_LINEAR = 0, POL_FAST = 1, POL_EXACT = 2, POL_TRUNC = 3 };   // POL_TRUNC: scaled probabilities with the reference's truncation (hx_linear.hip trunc_sum)
#define HXB_IS_LIN(POL_) ((POL_) == POL_LINEAR || (POL_) == POL_TRUNC)

// LDS plan (bytes), computed on the host (plan_band): the arithmetic's table first, then one block per pair
struct BandPlan { int table, xrec, sbase, ycol, yclass, xclass, elds, ring, flags, stride, total; };

// One row of the sweep (built by hx_api.hip build_band_rows), 8 bytes:
//   x = first owned step | (owned steps - 1) << 16     (owned: an even first and an odd last step - the two cells a row
//       produces on steps 2m, 2m+1 are stored together; the lane is busy with the row from its first to its last owned step;
//       0xFFFF in the low half: a sentinel row past the end, never owned)
//   y = emission class | not ready << 8 | lead pad << 9 (3 bits) | tail pad << 12 (3 bits)
//       (pads: owned steps that lie outside the envelope - the widening to whole step pairs and, since round 3, to the steps
//        of the row's group of four: hx_api.hip build_band_rows, "whole cache lines")
// followed, after the Nx - 1 + 64 records, by one int32 per 64-row strip: the cell of row i, step k lives at
//   strip_store[i / 64] + 2 * (i % 64) + (k >> 1) * blk + (k & 1)   in a state plane.

// LEAN: the row records stay in memory (a lane fetches its next row's record a whole row ahead) and the ring between the
// sweep and the converting wave is four steps deep instead of eight: 25 KB of LDS per pair instead of 54 (scaled
// probabilities), 12 instead of 29 (table policies), so that large batches put five to seven pairs on a CU.
// DIR = 1: the Backward fill (reference src/forward.cpp:975-1088 for leaf-like profiles) as the same sweep in mirrored
// coordinates (row i' = R-1-i, column j' = Cc-1-j: the layout of the Backward matrix), every policy.
// What is always inside the envelope is then the last row (x START) and the first column (the y state feeding END).  That
// y state is never ready (it has the null transition to END), so away from the band both are -inf: the second wave only
// writes them where the matrix was not pre-filled.
template <int POL, int PPW, bool LEAN, int DIR = 0>
__global__ void __launch_bounds__(2 * PPW * 64)
k_fill_band(const DevJob* __restrict__ jobs, const double* __restrict__ exact_tab, const double* __restrict__ pol_tab,
            const BandPlan plan, const int n_jobs, const int write_edges) {
  constexpr int THREADS = 2 * PPW * 64;
  constexpr int RING = LEAN ? HXB_RING_LEAN : HXB_RING;
  constexpr bool LIN = HXB_IS_LIN(POL), TRUNC = POL == POL_TRUNC;
  constexpr bool OFFLOAD = LIN;        // the second wave converts and stores the sweep's cells
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const bool helper = wave >= PPW;
  const int pair = helper ? wave - PPW : wave;
  const int job = (int)blockIdx.x * PPW + pair;
  const bool live = job < n_jobs;
  const DevJob& J = jobs[live ? job : 0];
  unsigned char* blkp = lds + plan.table + pair * plan.stride;
  double* ptab = reinterpret_cast<double*>(lds);
  // LDS views in their own address space (generic pointers would turn every access into a flat load, which also waits
  // for the vector-memory queue, i.e. for the stores)
  HX_LDS i2v* xrecL = (HX_LDS i2v*)(blkp + plan.xrec);
  HX_LDS int* sbaseL = (HX_LDS int*)(blkp + plan.sbase);
  HX_LDS unsigned* ycolL = (HX_LDS unsigned*)(blkp + plan.ycol);
  HX_LDS d2v* yclassL = (HX_LDS d2v*)(blkp + plan.yclass);
  HX_LDS d2v* xclassL = (HX_LDS d2v*)(blkp + plan.xclass);
  HX_LDS double* eldsL = (HX_LDS double*)(blkp + plan.elds);
  HX_LDS d2v* ringL = (HX_LDS d2v*)(blkp + plan.ring);           // [RING][3][64]
  volatile HX_LDS int* progL = (volatile HX_LDS int*)(blkp + plan.flags);   // steps the sweep has put into the ring
  volatile HX_LDS int* consL = progL + 1;                                   // steps the converting wave has taken out
  const HX_GLOBAL i2v* xrecG = (const HX_GLOBAL i2v*)as_global(reinterpret_cast<const i2v*>(DIR ? J.band_rows_bwd : J.band_rows));
  auto xrec_at = [&](const int i) -> i2v { return LEAN ? xrecG[i] : xrecL[i]; };

  // ---- stage the shared table and the pair's two sides ----
  if (LIN) {
    // (entries 1..1023 of the logarithm table are never addressed)
    for (int k = threadIdx.x; k < 2; k += THREADS) ptab[k] = pol_tab[k];
    for (int k = 2048 + threadIdx.x; k < 2 * HXB_LOG_ENTRIES; k += THREADS) ptab[k] = pol_tab[k];
  } else if (POL == POL_FAST) {
    for (int k = threadIdx.x; k < (HX_FAST_INTERVALS + 1) * 2; k += THREADS) ptab[k] = pol_tab[k];
  }
  const int R = J.n_rows, Cc = J.n_cols;
  const int n_strips = (R + 63) >> 6;
  {
    const int pt = lane + (helper ? 64 : 0);                            // the pair's two waves stage together
    const int Ky1 = J.y.n_cls + 1, Kx1 = J.x.n_cls + 1;
    const i2v* rows = reinterpret_cast<const i2v*>(DIR ? J.band_rows_bwd : J.band_rows);
    const int* sb = reinterpret_cast<const int*>(rows + (R + 64));
    if (!LEAN)
      for (int i = pt; i < R + 64; i += 128) xrecL[i] = rows[i];        // (64 sentinel rows past the end: never owned)
    for (int q = pt; q < n_strips; q += 128) sbaseL[q] = sb[q];
    for (int j = pt; j < Cc; j += 128) {
      // Backward: sweep column j is y state Cc-1-j; the class is that of the state an absorbing move leads to, the
      // ready bit that of the state itself
      const int jc = DIR ? Cc - 1 - j : j;
      ycolL[j] = (unsigned)J.y.ecls[DIR ? jc + 1 : jc] | (J.y.pack[4 * (size_t)jc + 3] < 0.0 ? 0x100u : 0u);
    }
    for (int c = pt; c < Ky1; c += 128) {
      const bool real = c < J.y.n_cls;
      const int rep = real ? J.y.cls_rep[c] : 0;
      const double rs = real ? J.y.pack[4 * (size_t)rep + 1] : HX_NEG_INF, in = real ? J.y.pack[4 * (size_t)rep + 2] : HX_NEG_INF;
      yclassL[c] = LIN ? d2v{exp(rs), exp(in)} : d2v{rs, in};
    }
    for (int c = pt; c < Kx1; c += 128) {
      const bool real = c < J.x.n_cls;
      const int rep = real ? J.x.cls_rep[c] : 0;
      const double rs = real ? J.x.pack[4 * (size_t)rep + 1] : HX_NEG_INF, in = real ? J.x.pack[4 * (size_t)rep + 2] : HX_NEG_INF;
      xclassL[c] = LIN ? d2v{exp(rs), exp(in)} : d2v{rs, in};
    }
    for (int e = pt; e < Kx1 * Ky1; e += 128) eldsL[e] = LIN ? exp(J.emis_pad[e]) : J.emis_pad[e];
    if (pt < 2) progL[pt] = 0;
  }
  __syncthreads();
  const int64_t plane = J.plane;
  const int blk = J.blk;
  HX_GLOBAL double* __restrict__ M = as_global(DIR ? J.bwd : J.fwd);
  const int Ky1 = J.y.n_cls + 1;
  // anti-diagonal steps of the sweep, in whole blocks of eight (the extra steps own nothing)
  const int n_steps = live ? ((DIR ? J.band_steps_bwd : J.band_steps) + 7) & ~7 : 0;
  const HX_LDS double* lt = (const HX_LDS double*)ptab;

  if (helper) {
    // =====================================================================================================
    // the second wave.  (1) scaled probabilities: cells out of the ring, logarithms, stores.
    // =====================================================================================================
    if (OFFLOAD) {
      int seen = 0;
      for (int k = 0; k < n_steps; k += 2) {
        while (seen < k + 2) {
          seen = __builtin_amdgcn_readfirstlane(progL[0]);
          if (seen < k + 2) __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        const HX_LDS d2v* s0 = ringL + (size_t)(k & (RING - 1)) * 192 + lane;
        const HX_LDS d2v* s1 = ringL + (size_t)((k + 1) & (RING - 1)) * 192 + lane;
        const d2v a0 = s0[0], b0 = s0[64], c0 = s0[128], a1 = s1[0], b1 = s1[64], c1 = s1[128];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) consL[0] = k + 2;               // (the slots may be overwritten: their contents are in registers)
        const int e0 = __double2loint(c0.y), e1 = __double2loint(c1.y);
        const int sl = __double2hiint(c0.y);           // the pair's slot in a state plane, or -1: the lane does not own it
        const double l0 = log_scaled(a0.x, e0, lt), l1 = log_scaled(a0.y, e0, lt), l2 = log_scaled(b0.x, e0, lt),
                     l3 = log_scaled(b0.y, e0, lt), l4 = log_scaled(c0.x, e0, lt);
        const double h0 = log_scaled(a1.x, e1, lt), h1 = log_scaled(a1.y, e1, lt), h2 = log_scaled(b1.x, e1, lt),
                     h3 = log_scaled(b1.y, e1, lt), h4 = log_scaled(c1.x, e1, lt);
        if (sl >= 0) {
          HX_GLOBAL d2v* M2 = (HX_GLOBAL d2v*)(M + sl);
          const int64_t plane2 = plane >> 1;
          // write-once data: non-temporal stores
          __builtin_nontemporal_store(d2v{l0, h0}, &M2[0]);
          __builtin_nontemporal_store(d2v{l1, h1}, &M2[plane2]);
          __builtin_nontemporal_store(d2v{l2, h2}, &M2[2 * plane2]);
          __builtin_nontemporal_store(d2v{l3, h3}, &M2[3 * plane2]);
          __builtin_nontemporal_store(d2v{l4, h4}, &M2[4 * plane2]);
        }
      }
    }
    // =====================================================================================================
    // (2) the envelope's one-dimensional edges (see the header comment).  Nothing here is read by the sweep.
    // =====================================================================================================
    if (DIR == 1 && live) {
      // Backward.  In mirrored coordinates the always-inside parts of the envelope are column 0 (the y state that feeds
      // END) and the last row (x START).  The y state that feeds END has a null transition (to END), so it is not
      // ready and no x-absorbing move leaves a cell of its column (reference src/forward.cpp:1041-1049; hx_api.hip admits
      // a pair to this sweep only then): below the END-feeding cell the column is -inf, and so is the last row up to
      // where the row above's band begins.  Those cells are written here; the sweep's idle lanes read them as -inf anyway.
      const int64_t ssd = J.strip_stride;
      const int lo_last = reinterpret_cast<const int*>(reinterpret_cast<const i2v*>(J.band_rows_bwd) + (R + 64))[n_strips];   // first column the sweep owns on the last row
      auto put_inf = [&](const int ip, const int jp) {
        const int64_t sl = cell_slot_blk(ssd, blk, ip, jp);
        M[sl] = HX_NEG_INF; M[plane + sl] = HX_NEG_INF; M[2 * plane + sl] = HX_NEG_INF; M[3 * plane + sl] = HX_NEG_INF;
        M[4 * plane + sl] = HX_NEG_INF;
      };
      if (write_edges) {
        for (int ip = 1 + lane; ip < R; ip += 64) {
          const i2v rec = xrec_at(ip);
          if ((rec.x & 0xFFFF) + ((rec.y >> 9) & 7) - ip > 0) put_inf(ip, 0);      // (the sweep owns the row from a later column on)
        }
        for (int jp = 1 + lane; jp < lo_last; jp += 64) put_inf(R - 1, jp);
      }
      {
        // a first row whose band does not reach column 0: the END-feeding cell itself (src/forward.cpp:981-995)
        const i2v rec = xrec_at(0);
        if (lane == 0 && (rec.x & 0xFFFF) + ((rec.y >> 9) & 7) > 0) {
          const double lpe = J.x.pack[4 * (size_t)R] + J.y.pack[4 * (size_t)Cc];
          const int64_t sl = cell_slot_blk(ssd, blk, 0, 0);
          for (int st = 0; st < 5; ++st) M[st * plane + sl] = lpe + J.T[st][5];
        }
      }
    }
    if (DIR == 0 && live) {
    // row 0 beyond what the sweep owns: the chain in log space (every policy stores log-probabilities)
    const int own0 = (xrec_at(0).x >> 16) & 0xFFFF;                      // row 0 is owned from step 0 to this step = column
    const i2v rec1 = xrec_at(1);
    const bool row1_edge = ((rec1.x & 0xFFFF) + ((rec1.x >> 16) & 0xFFFF) - 1) < Cc - 1;   // row 1 does not own column Ny-2
    const double T02 = J.T[0][2], T03 = J.T[0][3], T22 = J.T[2][2], T33 = J.T[3][3];
    const double pen0 = J.x.pack[3];                                   // x START ready (or x empty): 0, else -inf
    double d_idm = HX_NEG_INF, d_imi = HX_NEG_INF;                     // cell (0, Ny-3): the diagonal source of (1, Ny-2)
    if (own0 < Cc - 1 || row1_edge) {
      double idm = HX_NEG_INF, imi = HX_NEG_INF;                       // (the chain's value in every lane)
      double carry_idm = 0., carry_imi = 0.;                            // (scaled-probability policies: the prefix sums' carries)
      for (int j0 = 0; j0 < Cc; j0 += 64) {
        const int jl = j0 + lane < Cc ? j0 + lane : Cc - 1;
        const unsigned w = ycolL[jl];
        const double lrs = LIN ? J.y.pack[4 * (size_t)jl + 1] : yclassL[w & 0xFFu].x;
        const double lin = LIN ? J.y.pack[4 * (size_t)jl + 2] : yclassL[w & 0xFFu].y;
        double kidm = HX_NEG_INF, kimi = HX_NEG_INF;
        if (LIN) {
          // as a prefix sum (hx_bandedge.h; the same function as the two-pairs-per-wavefront sweep's edge kernel: same bits)
          row0_chain_block(j0, lane, Cc, lrs, lin, T02, T03, T22, T33, pen0, carry_idm, carry_imi, kidm, kimi);
          if (Cc - 2 >= j0 && Cc - 2 < j0 + 64) { d_idm = read_lane(kidm, Cc - 2 - j0); d_imi = read_lane(kimi, Cc - 2 - j0); }
        } else
#pragma unroll 2
        for (int m = 0; m < 64; ++m) {
          const int j = j0 + m;                                         // (wave-uniform)
          const double r = read_lane(lrs, m), n = read_lane(lin, m);
          if (j >= 1) {
            // leaf_cell for x START: ((a + lpTrans) + rootsuby) + ready-penalty, a = the only finite term of the sum
            idm = (((j == 1 ? T02 : idm + T22) + 0.0) + r) + pen0;
            imi = (((j == 1 ? T03 : imi + T33) + 0.0) + n) + pen0;
          }
          if (j == Cc - 2) { d_idm = idm; d_imi = imi; }
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
    // Cell (1, Ny-2) when row 1's band does not reach it: always in the envelope (column Ny-2), and its diagonal source
    // (0, Ny-3) is too (row 0) - the one cell of that column away from the band that is not -inf.  The same leaf_cell
    // (log policies; the scaled-probability policy: the same sum with libm, written as a log-probability) with -inf above
    // and to the left.
    if (row1_edge) {
      C5 diag = c5_neg_inf();
      if (Cc - 2 == 0) diag.imm = 0.0; else { diag.idm = d_idm; diag.imi = d_imi; }
      const unsigned w = ycolL[Cc - 1];
      const unsigned c = w & 0xFFu;
      const unsigned eo = (unsigned)(rec1.y & 0xFF) * (unsigned)Ky1;
      C5 nw = c5_neg_inf();
      if (LIN) {
        double sum = 0.;
        const double dv[5] = {diag.imm, diag.imd, diag.idm, diag.imi, diag.iiw};
        double mx = HX_NEG_INF;
        for (int q = 0; q < 5; ++q) mx = vmax(mx, dv[q] + J.T[q][0]);
        if (TRUNC) {
          // the reference's left-nested sum with its truncation, in libm arithmetic
          double acc = HX_NEG_INF;
          for (int q = 0; q < 5; ++q) {
            const double t = dv[q] + J.T[q][0];
            const double hi = vmax(acc, t), lo = vmin(acc, t);
            acc = (hi > HX_NEG_INF && hi - lo < 10.0) ? hi + log1p(exp(lo - hi)) : hi;
          }
          nw.imm = acc > HX_NEG_INF ? acc + J.emis_pad[eo + c] : HX_NEG_INF;
        } else {
        for (int q = 0; q < 5; ++q) sum += (dv[q] + J.T[q][0] > HX_NEG_INF) ? exp(dv[q] + J.T[q][0] - mx) : 0.;
        nw.imm = mx > HX_NEG_INF ? mx + log(sum) + J.emis_pad[eo + c] : HX_NEG_INF;
        }
      } else {
        XLeaf X;
        const d2v xc = xclassL[rec1.y & 0xFF];
        X.lp = 0.0; X.rootsub = xc.x; X.ins = xc.y; X.pen = (rec1.y & 0x100) ? HX_NEG_INF : 0.0; X.eoff = eo; X.valid = true;
        const d2v rc = yclassL[c];
        const d4v Y = d4v{0.0, rc.x, rc.y, __hiloint2double((w & 0x100u) ? (int)0xFFF00000 : 0, 0)};
        double Tk[5][6];
        for (int a = 0; a < 5; ++a) for (int d = 0; d < 6; ++d) Tk[a][d] = J.T[a][d];
        const C5 ninf = c5_neg_inf();
        if (POL == POL_FAST) nw = leaf_cell(Tk, FastLse::make(ptab), X, Y, eldsL[eo + c], 0.0, ninf, ninf, diag);
        else nw = leaf_cell(Tk, ExactLse3::make(pol_tab), X, Y, eldsL[eo + c], 0.0, ninf, ninf, diag);
      }
      const int64_t sl = stored_slot(J, 1, Cc - 1);
      if (lane == 0 && sl >= 0) {
        M[sl] = nw.imm; M[plane + sl] = nw.imd; M[2 * plane + sl] = nw.idm; M[3 * plane + sl] = nw.imi; M[4 * plane + sl] = nw.iiw;
      }
    }
    // the rest of column Ny-2 away from the band: -inf (only where the matrix was not pre-filled)
    if (write_edges)
      for (int i = 2 + lane; i < R; i += 64) {
        const i2v rec = xrec_at(i);
        const int last_col = (rec.x & 0xFFFF) + ((rec.x >> 16) & 0xFFFF) - i;   // column of the row's last owned step
        if (last_col < Cc - 1) {
          const int64_t sl = stored_slot(J, i, Cc - 1);
          if (sl >= 0) {
            M[sl] = HX_NEG_INF; M[plane + sl] = HX_NEG_INF; M[2 * plane + sl] = HX_NEG_INF; M[3 * plane + sl] = HX_NEG_INF; M[4 * plane + sl] = HX_NEG_INF;
          }
        }
      }
    }
  } else {
  // =======================================================================================================
  // the sweep
  // =======================================================================================================
  // the lane's row, decoded; and the raw record of the row it takes next (i + 64), fetched a whole row ahead
  int i = lane, os, oe, as, ae, store;
  unsigned eoff;
  double xc_rs, xc_in;         // exp(rootsubx), exp(insx) (scaled probabilities) or rootsubx, insx
  int x_wait;                  // x state not ready: 2^29 (an exponent shift) / the 0 or -inf penalty lives in xpen
  double xpen;
  i2v nrec;
  d2v nxc = d2v{0., 0.};
  int nstore = 0;
  auto decode = [&](const i2v r, const d2v xc, const int sb) {
    os = r.x & 0xFFFF; oe = os + ((r.x >> 16) & 0xFFFF);
    as = os + ((r.y >> 9) & 7); ae = oe - ((r.y >> 12) & 7);
    if ((r.x & 0xFFFF) == 0xFFFF) { os = 0x7FFFFFF0; oe = 0x7FFFFFF1; as = os; ae = oe; }     // sentinel: never owned
    store = sb;
    eoff = (unsigned)(r.y & 0xFF) * (unsigned)Ky1;
    x_wait = (r.y & 0x100) ? (1 << 29) : 0;
    xpen = (r.y & 0x100) ? HX_NEG_INF : 0.0;
    xc_rs = xc.x; xc_in = xc.y;
  };
  auto store_base = [&](const int row) -> int { return sbaseL[row < R ? row >> 6 : 0] + 2 * (row & 63); };
  {
    const i2v r0 = xrec_at(lane < R ? lane : R);
    decode(r0, xclassL[r0.y & 0xFF], store_base(lane));
    nrec = xrec_at(lane + 64 < R ? lane + 64 : R);
  }
  // the 18 transition weights the recursion reads, pinned in scalar registers: probabilities or log-probabilities
  // (dest 5 = EEE is only read by lpEnd)
  double P[5][6];
#pragma unroll
  for (int a = 0; a < 5; ++a)
#pragma unroll
    for (int d = 0; d < 6; ++d) {
      const bool used = d == 0 || (d == 1 && a != 4) || (d == 2 && a != 3) || (d == 3 && (a == 0 || a == 3)) ||
                        (d == 4 && (a == 0 || a == 3 || a == 4));
      double v = J.T[a][d];
      if (!used) { P[a][d] = LIN ? 0. : v; continue; }
      if (LIN) {
        const double pv = exp(v);
        v = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(pv)), __builtin_amdgcn_readfirstlane(__double2loint(pv)));
      }
      asm volatile("" : "+s"(v));
      P[a][d] = v;
    }
  const FastLse LF = FastLse::make(ptab);
  const ExactLse3 LE = ExactLse3::make(pol_tab);
  // Backward: the cell feeding END
  C5 end_cell = c5_neg_inf();
  if (DIR == 1) {
    const double lpe = J.x.pack[4 * (size_t)R] + J.y.pack[4 * (size_t)Cc];
    end_cell = C5{lpe + J.T[0][5], lpe + J.T[1][5], lpe + J.T[2][5], lpe + J.T[3][5], lpe + J.T[4][5]};
    if (LIN) end_cell = C5{exp(end_cell.imm), exp(end_cell.imd), exp(end_cell.idm), exp(end_cell.imi), exp(end_cell.iiw)};
  }

  // cell registers, ping-ponged: at an even step the lane's previous cell is in cb (the one before in ca, which the new
  // cell overwrites), the previous lane's cells of one / two steps ago in ua / ub
  L5 la = l5_zero(), lb = l5_zero(), lua = l5_zero(), lub = l5_zero();
  C5 ca = c5_neg_inf(), cb = c5_neg_inf(), cua = c5_neg_inf(), cub = c5_neg_inf();

  // The y side of a step, fetched ahead so that no LDS round trip is ever waited for: the column's word two steps
  // ahead, and - from the word fetched the step before - the column's class constants and the emission term one step
  // ahead.  A lane that is about to change rows looks up the NEXT row's column / emission row.
  auto word_at = [&](const int col) -> unsigned {
    const int c = col < 0 ? 0 : (col >= Cc ? Cc - 1 : col);
    return ycolL[c];
  };
  unsigned neoff = 0;
  unsigned w_cur = word_at(0 - i), w_nxt = word_at(1 - i);        // words of steps k, k + 1
  d2v rc_cur = yclassL[w_cur & 0xFFu];                            // class constants of step k
  double em_cur = eldsL[eoff + (w_cur & 0xFFu)];

  // Owned spans are whole step pairs, so a lane changes rows at even steps only: at the even step after its row's last
  // step it takes row i + 64 (record fetched a row ago, class constants one step ago), and at the odd step that ends a
  // row it fetches the next row's class constants and store base.
  auto roll_even = [&](const int k) {
    if (k > oe) {
      i += 64;
      decode(nrec, nxc, nstore);
      nrec = xrec_at(i + 64 < R ? i + 64 : R);
    }
  };
  auto roll_odd = [&](const int k) {
    if (k == oe) {
      nxc = xclassL[nrec.y & 0xFF];
      nstore = store_base(i + 64);
      neoff = (unsigned)(nrec.y & 0xFF) * (unsigned)Ky1;
    }
  };
  struct YSide { unsigned w; d2v rc; double em; };
  // hand out step k's y side, issue the fetches for steps k + 1 (constants) and k + 2 (word)
  auto y_side = [&](const int k) -> YSide {
    const YSide now{w_cur, rc_cur, em_cur};
    const unsigned c1 = w_nxt & 0xFFu;
    rc_cur = yclassL[c1];
    em_cur = eldsL[((k + 1 > oe) ? neoff : eoff) + c1];
    w_cur = w_nxt;
    w_nxt = word_at(k + 2 - ((oe <= k + 1) ? i + 64 : i));
    return now;
  };

  auto step_linear = [&](const int k, const bool renorm, const L5& left, L5& out, L5& u1, L5& u2, const YSide ys) {
    const d2v rc = ys.rc;
    const int y_wait = (int)((ys.w & 0x100u) << 21);        // y state not ready: 2^29, else 0
    const double em = ys.em;
    if (DIR == 1) {
      // Backward (src/forward.cpp:1018-1065 for leaf-like profiles): the five destination terms - the xy-absorbing move into
      // (i+1,j+1), the x-absorbing moves into (i+1,j) as IMD / IIW, the y-absorbing moves into (i,j+1) as IDM / IMI - brought
      // to the cell's exponent (a move that may not be made, a cell outside the envelope: shifted out of range), then 18
      // multiply-adds.  Same arithmetic as the strip pipeline's (hx_linear.hip).
      const double tD = u2.imm * em;
      const double t1x = u1.imd * xc_rs, t2x = u1.iiw * xc_in;
      const double t1y = left.idm * rc.x, t2y = left.imi * rc.y;
      int E = left.e > u1.e ? left.e : u1.e;
      E = E > u2.e ? E : u2.e;
      const int outside = (k >= as && k <= ae) ? 0 : (1 << 29);
      const int du = ((u1.e - E) - y_wait) - outside, dl = ((left.e - E) - x_wait) - outside, dd = (u2.e - E) - outside;
      const double D = __builtin_ldexp(tD, dd);
      const double d1x = __builtin_ldexp(t1x, du), d2x = __builtin_ldexp(t2x, du);
      const double d1y = __builtin_ldexp(t1y, dl), d2y = __builtin_ldexp(t2y, dl);
      out.imm = lin_acc<TRUNC>(P[0][3], d2y, lin_acc<TRUNC>(P[0][2], d1y, lin_acc<TRUNC>(P[0][4], d2x, lin_acc<TRUNC>(P[0][1], d1x, P[0][0] * D))));
      out.imd = lin_acc<TRUNC>(P[1][2], d1y, lin_acc<TRUNC>(P[1][1], d1x, P[1][0] * D));
      out.idm = lin_acc<TRUNC>(P[2][2], d1y, lin_acc<TRUNC>(P[2][1], d1x, P[2][0] * D));
      out.imi = lin_acc<TRUNC>(P[3][3], d2y, lin_acc<TRUNC>(P[3][4], d2x, lin_acc<TRUNC>(P[3][1], d1x, P[3][0] * D)));
      out.iiw = lin_acc<TRUNC>(P[4][2], d1y, lin_acc<TRUNC>(P[4][4], d2x, P[4][0] * D));
      out.e = E;
    } else {
    // the five sums of src/forward.cpp:103-115,139-150,171-180 on probabilities
    double s_imd = u1.imm * P[0][1];
    double s_iiw = u1.imm * P[0][4];
    double s_idm = left.imm * P[0][2];
    double s_imi = left.imm * P[0][3];
    double s_imm = u2.imm * P[0][0];
    s_imd = lin_acc<TRUNC>(u1.imd, P[1][1], s_imd);
    s_iiw = lin_acc<TRUNC>(u1.imi, P[3][4], s_iiw);
    s_idm = lin_acc<TRUNC>(left.imd, P[1][2], s_idm);
    s_imi = lin_acc<TRUNC>(left.imi, P[3][3], s_imi);
    s_imm = lin_acc<TRUNC>(u2.imd, P[1][0], s_imm);
    s_imd = lin_acc<TRUNC>(u1.idm, P[2][1], s_imd);
    s_iiw = lin_acc<TRUNC>(u1.iiw, P[4][4], s_iiw);
    s_idm = lin_acc<TRUNC>(left.idm, P[2][2], s_idm);
    s_imm = lin_acc<TRUNC>(u2.idm, P[2][0], s_imm);
    s_imd = lin_acc<TRUNC>(u1.imi, P[3][1], s_imd);
    s_idm = lin_acc<TRUNC>(left.iiw, P[4][2], s_idm);
    s_imm = lin_acc<TRUNC>(u2.imi, P[3][0], s_imm);
    s_imm = lin_acc<TRUNC>(u2.iiw, P[4][0], s_imm);
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
    }
    if (renorm) {                                  // (compile-time: the first two steps of every block of eight)
      if (DIR == 0) {
        if (k == 0 && lane == 0) { out.imm = 1.0; out.e = 0; }   // cell (0,0): lpStart() = 0 (src/forward.cpp:73)
      } else if (k == 0 && lane == 0) {
        // the cell feeding END is initialised by assignment (src/forward.cpp:981-995)
        out.imm = end_cell.imm; out.imd = end_cell.imd; out.idm = end_cell.idm; out.imi = end_cell.imi; out.iiw = end_cell.iiw;
        out.e = 0;
      }
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

  auto step_log = [&](const int k, const C5& left, C5& out, C5& u1, C5& u2, const YSide ys) {
    const d2v rc = ys.rc;
    const double em = ys.em;
    const double ypen = __hiloint2double((ys.w & 0x100u) ? (int)0xFFF00000 : 0, 0);  // y state not ready: -inf
    const double pj = (k >= as && k <= ae) ? 0.0 : HX_NEG_INF;
    XLeaf X;
    X.lp = 0.0; X.rootsub = xc_rs; X.ins = xc_in; X.pen = xpen; X.eoff = eoff; X.valid = true;
    const d4v Y = d4v{0.0, rc.x, rc.y, ypen};
    C5 nw;
    if (DIR == 0) {
      if (POL == POL_FAST) nw = leaf_cell(P, LF, X, Y, em, pj, u1, left, u2);
      else nw = leaf_cell(P, LE, X, Y, em, pj, u1, left, u2);
      if (k == 0 && lane == 0) nw.imm = 0.0;         // cell (0,0): lpStart() = 0 (src/forward.cpp:73)
    } else {
      if (POL == POL_FAST) nw = leaf_cell_bwd(P, LF, X, Y, em, pj, u1, left, u2);
      else nw = leaf_cell_bwd(P, LE, X, Y, em, pj, u1, left, u2);
      if (k == 0 && lane == 0) nw = end_cell;        // the cell feeding END (src/forward.cpp:981-995)
    }
    out = nw;
    u2 = ror1(nw);
  };

  // a pair of steps: in the strip-skewed layout the two cells are adjacent, 16 bytes per lane and state plane
  int cons_seen = 0;
  auto step_pair = [&](const int k, const bool renorm) {
    roll_even(k);
    const bool own = k >= os && k <= oe;           // (owned spans are whole step pairs)
    const int sl = store + (k >> 1) * blk;
    if (OFFLOAD) {
      // the ring slots of steps k, k + 1 were last used by steps k - 8, k - 7: the converting wave must be past them
      while (cons_seen < k + 2 - RING) {
        cons_seen = __builtin_amdgcn_readfirstlane(consL[0]);
        if (cons_seen < k + 2 - RING) __builtin_amdgcn_s_sleep(1);
      }
      step_linear(k, renorm, lb, la, lua, lub, y_side(k));
      HX_LDS d2v* s0 = ringL + (size_t)(k & (RING - 1)) * 192 + lane;
      s0[0] = d2v{la.imm, la.imd}; s0[64] = d2v{la.idm, la.imi}; s0[128] = d2v{la.iiw, __hiloint2double(own ? sl : -1, la.e)};
      roll_odd(k + 1);
      step_linear(k + 1, renorm, la, lb, lub, lua, y_side(k + 1));
      HX_LDS d2v* s1 = ringL + (size_t)((k + 1) & (RING - 1)) * 192 + lane;
      s1[0] = d2v{lb.imm, lb.imd}; s1[64] = d2v{lb.idm, lb.imi}; s1[128] = d2v{lb.iiw, __hiloint2double(0, lb.e)};
      asm volatile("" ::: "memory");               // data before flag (LDS operations of a wave complete in order)
      if (lane == 0) progL[0] = k + 2;
    } else {
      step_log(k, cb, ca, cua, cub, y_side(k));
      const C5 first = ca;
      roll_odd(k + 1);
      step_log(k + 1, ca, cb, cub, cua, y_side(k + 1));
      if (own) {
        HX_GLOBAL d2v* M2 = (HX_GLOBAL d2v*)(M + sl);
        const int64_t plane2 = plane >> 1;
        // write-once data: non-temporal stores
        __builtin_nontemporal_store(d2v{first.imm, cb.imm}, &M2[0]);
        __builtin_nontemporal_store(d2v{first.imd, cb.imd}, &M2[plane2]);
        __builtin_nontemporal_store(d2v{first.idm, cb.idm}, &M2[2 * plane2]);
        __builtin_nontemporal_store(d2v{first.imi, cb.imi}, &M2[3 * plane2]);
        __builtin_nontemporal_store(d2v{first.iiw, cb.iiw}, &M2[4 * plane2]);
      }
    }
  };
  for (int k = 0; k < n_steps; k += 8) {
    step_pair(k, true);                            // (scaled probabilities: mantissas renormalised every eighth step)
    step_pair(k + 2, false);
    step_pair(k + 4, false);
    step_pair(k + 6, false);
  }
  }
  // lpEnd reads cell (Nx-2, Ny-2): the sweep's last cell (stored by either wave), or - a one-row band - an edge cell
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (DIR == 0) {
    if (!helper && live && lane == 0) *J.lp_end = forward_lp_end(J, ExactLse{exact_tab});
  } else {
    // B(START, START).IMM: the sweep's last cell
    if (!helper && live && lane == 0) *J.lp_start = J.bwd[cell_slot_blk(J.strip_stride, J.blk, R - 1, Cc - 1)];
  }
}

BandPlan plan_band(int pol, int ppw, int max_rows, int max_cols, int max_cls, bool lean = false) {
  BandPlan p;
  p.table = HXB_IS_LIN(pol) ? 16 * HXB_LOG_ENTRIES : (pol == POL_FAST ? 16 * (HX_FAST_INTERVALS + 1) : 16);
  int a = 0;
  p.xrec = a; a += lean ? 0 : (8 * (max_rows + 64) + 15) & ~15;
  p.sbase = a; a += (4 * ((max_rows + 63) / 64 + 1) + 15) & ~15;
  p.ycol = a; a += (4 * max_cols + 15) & ~15;
  p.yclass = a; a += 16 * (max_cls + 1);
  p.xclass = a; a += 16 * (max_cls + 1);
  p.elds = a; a += (8 * (max_cls + 1) * (max_cls + 1) + 15) & ~15;
  p.ring = a; a += HXB_IS_LIN(pol) ? (lean ? HXB_RING_LEAN : HXB_RING) * 3 * 64 * 16 : 0;
  p.flags = a; a += 16;
  p.stride = a;
  p.total = p.table + ppw * a;
  return p;
}

template <int POL, int PPW, bool LEAN = false, int DIR = 0>
int launch_pol(const DevJob* d_jobs, int n_jobs, const BandPlan& p, const double* tab, const double* pol_tab, int write_edges, hipStream_t st) {
  if (p.total > HX_LDS_LIMIT) return launch_fail("k_fill_band<%d, %d> needs %d bytes of LDS (limit %d)", POL, PPW, p.total, HX_LDS_LIMIT);
  hipLaunchKernelGGL((k_fill_band<POL, PPW, LEAN, DIR>), dim3((n_jobs + PPW - 1) / PPW), dim3(2 * PPW * 64), p.total, st, d_jobs, tab, pol_tab, p,
                     n_jobs, write_edges);
  return 0;
}

}  // namespace

// Largest pair (rows, columns, emission classes) whose two sides fit LDS next to the policy's table with one pair per
// workgroup: hx_api.hip admits a pair to this kernel's class only below it
bool band_kernel_fits(int pol, int rows, int cols, int cls) { return plan_band(pol, 1, rows, cols, cls).total <= HX_LDS_LIMIT; }

namespace {
template <int DIR>
int launch_band(const DevJob* d_jobs, int n_jobs, int pol, int max_rows, int max_cols, int max_cls, Tab8 tab8,
                Tab16 tab16, bool write_edges, hipStream_t st) {
  const double* tab = tab8.p;
  const double* pol_tab = tab16.p;
  // Pairs per workgroup (they share the policy's table).  A CU holds 160 KB of LDS and four SIMDs; a pair is two waves.
  // One pair per workgroup while two such workgroups fit a CU; else two pairs share the table if that fits (the 64 KB
  // table of the fast policy); large batches put up to four pairs in a workgroup.
  const char* v = getenv("HX_BAND_PPW");           // tuning / test hook: > 0 pairs per workgroup, < 0 the same in the lean variant
  int ppw = v ? atoi(v) : 0;
  bool lean = ppw < 0;
  if (lean) ppw = -ppw;
  if (ppw <= 0) {
    const int half = 72 * 1024;                    // (two workgroups of exactly 80 KB were measured NOT to fit a CU)
    ppw = 1;
    if (plan_band(pol, 1, max_rows, max_cols, max_cls).total > half && plan_band(pol, 2, max_rows, max_cols, max_cls).total <= HX_LDS_LIMIT) ppw = 2;
    if (n_jobs > 512) {
      // more than two pairs per CU: the lean variant, as many pairs per workgroup as there are pairs per CU and its LDS
      // plan admits (up to six).  Measured at 2560 pairs (tools/band_ppw_sweep.sh): scaled probabilities 7.7 ms with two
      // pairs per workgroup, 5.1 ms with five; fast policy 14.4 ms -> 8.7 ms from four on (then bound by instruction issue).
      int want = (n_jobs + 255) / 256;
      want = want > 6 ? 6 : want;
      int fit = want;
      while (fit > 1 && plan_band(pol, fit, max_rows, max_cols, max_cls, true).total > HX_LDS_LIMIT) --fit;
      if (fit >= 3) { lean = true; ppw = fit; }
    }
  }
  if (ppw > 1 && plan_band(pol, ppw, max_rows, max_cols, max_cls, lean).total > HX_LDS_LIMIT) { ppw = 1; lean = false; }
  const int we = write_edges ? 1 : 0;
#define HXB_LEAN(POL_, N_) return launch_pol<POL_, N_, true, DIR>(d_jobs, n_jobs, plan_band(POL_, N_, max_rows, max_cols, max_cls, true), tab, pol_tab, we, st)
#define HXB_GO(POL_) do { \
    if (lean && ppw >= 6) HXB_LEAN(POL_, 6); \
    if (lean && ppw == 5) HXB_LEAN(POL_, 5); \
    if (lean && ppw == 4) HXB_LEAN(POL_, 4); \
    if (lean && ppw == 3) HXB_LEAN(POL_, 3); \
    if (lean && ppw == 2) HXB_LEAN(POL_, 2); \
    if (lean) HXB_LEAN(POL_, 1); \
    if (ppw >= 4) return launch_pol<POL_, 4, false, DIR>(d_jobs, n_jobs, plan_band(POL_, 4, max_rows, max_cols, max_cls), tab, pol_tab, we, st); \
    if (ppw >= 2) return launch_pol<POL_, 2, false, DIR>(d_jobs, n_jobs, plan_band(POL_, 2, max_rows, max_cols, max_cls), tab, pol_tab, we, st); \
    return launch_pol<POL_, 1, false, DIR>(d_jobs, n_jobs, plan_band(POL_, 1, max_rows, max_cols, max_cls), tab, pol_tab, we, st); } while (0)
  if (pol == POL_LINEAR) HXB_GO(POL_LINEAR);
  if (pol == POL_TRUNC) HXB_GO(POL_TRUNC);
  if (pol == POL_FAST) HXB_GO(POL_FAST);
  HXB_GO(POL_EXACT);
#undef HXB_LEAN
#undef HXB_GO
}
}  // namespace

int launch_forward_band(const DevJob* d_jobs, int n_jobs, int pol, int max_rows, int max_cols, int max_cls, Tab8 tab8,
                        Tab16 tab16, bool write_edges, hipStream_t st) {
  return launch_band<0>(d_jobs, n_jobs, pol, max_rows, max_cols, max_cls, tab8, tab16, write_edges, st);
}

// The Backward sweep (the class's pairs all carry band_rows_bwd: hx_api.hip ClassRange::bwd_band)
int launch_backward_band(const DevJob* d_jobs, int n_jobs, int pol, int max_rows, int max_cols, int max_cls, Tab8 tab8,
                         Tab16 tab16, bool write_edges, hipStream_t st) {
  return launch_band<1>(d_jobs, n_jobs, pol, max_rows, max_cols, max_cls, tab8, tab16, write_edges, st);
}

}  // namespace hx
