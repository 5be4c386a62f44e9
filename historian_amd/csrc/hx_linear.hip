// Forward and Backward fills of leaf-like pairs on SCALED PROBABILITIES (the HX_LSE_LINEAR policy).
//
// The reference (src/forward.cpp:68-223, 975-1088) works on log-probabilities and combines them with a table-driven
// log-sum-exp: 18 table look-ups per cell for a pair of leaf profiles.  The same recursion on the probabilities
// themselves is 18 multiply-adds.  This kernel keeps every cell as five fp64 mantissas plus ONE integer
// exponent per cell (p_state = m_state * 2^e), so nothing under- or overflows however long the sequences are,
// runs the recursion with fused multiply-adds, and converts each finished cell to the reference's storage format
// - five log-probabilities, 40 B/cell, same strip-skewed layout - with an fp64 table-plus-polynomial logarithm
// (512 intervals, |error| < 2.3e-13) just before the store.  Nothing is approximated beyond fp64 rounding, so the
// results do NOT carry the reference's truncation of log-sum-exp terms below e^-10: they differ from the reference's by
// the reference's own approximation error (lpEnd ~3e-6 relative, north_star allows 1e-4), which is
// why the policy is an explicit opt-in; the bit-exact policy is ExactLse3 in hx_chain.hip.
//
// Pipeline structure is that of k_fill_chain (hx_chain.hip): one workgroup per pair, 64-row strips dealt to the
// waves round-robin, lane <-> row, step <-> anti-diagonal, up/diag neighbours by DPP wave_shr:1, the whole y side in LDS,
// two steps per iteration so that a lane stores 16 contiguous bytes per state plane.  Unlike there, the strip above's
// last row travels between waves through LDS rings (mantissas + exponent); only the link from the last wave to the first
// wave's next strip goes through the matrix.  Variants: banded batches (one wavefront per pair over the strips' step
// windows, band-compressed planes, several pairs per workgroup) and the Backward fill (mirrored sweep).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "hx_device.h"
#include "hx_lse.h"
#include "hx_common.h"
#include "hx_policy.h"
#include "hx_kernels.h"

namespace hx {

namespace {

#define HXL_PUBLISH_LAG 16         // wrap-around link: a column is published 16 steps after its stores were issued
#define HX_YL_MAX_CLS_LINEAR 64   // (hx_api.hip admits y sides of at most 63 emission classes to the LDS-resident path)
#define HXL_EMIN (-(1 << 28))      // exponent of an all-zero cell: loses every max()
#define HXL_RENORM_MASK 6          // mantissas are renormalised on steps with (t & 6) == 0: every 8th step

struct L5 { double imm, imd, idm, imi, iiw; int e; };

// One pairwise sum of the reference's log_sum_exp, on probabilities: the reference returns max + T(|a - b|) with T = 0 once the
// difference reaches 10 (src/logsumexp.h:45), i.e. it DROPS the smaller term when it is at most e^-10 of the larger.  Keeping
// that - and the reference's left-nested order of the sums (src/logsumexp.h:86-100) - is what separates the truncating policy
// (HX_LSE_TRUNC) from plain multiply-adds: the dropped terms are the 4.5e-5-per-operation bias that moves near-tied best paths.
#define HXL_EXP_M10 4.5399929762484854e-05      // e^-10
__device__ __forceinline__ double trunc_sum(double a, double b) {
  const double hi = fmax_plain(a, b), lo = fmin_plain(a, b);
  // (a dropped term keeps its low word: a number below 2^-1042 that no sum of mantissas scaled to the cell's exponent feels - one select instead of two)
  const int keep = lo > hi * HXL_EXP_M10 ? __double2hiint(lo) : 0;
  return hi + __hiloint2double(keep, __double2loint(lo));
}
// acc (+) m * p
template <bool TRUNC> __device__ __forceinline__ double lin_acc(double m, double p, double acc) {
  if (TRUNC) return trunc_sum(acc, m * p);
  return __builtin_fma(m, p, acc);
}

__device__ __forceinline__ L5 l5_zero() { return L5{0., 0., 0., 0., 0., HXL_EMIN}; }

__device__ __forceinline__ int dpp_shr1_keep0(int old, int v) {
  return __builtin_amdgcn_update_dpp(old, v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ double dpp_shr1_keep0(double old, double v) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// lane l >= 1 receives c of lane l-1; lane 0 receives its own o (DPP wave_shr:1 leaves lanes without a source untouched)
__device__ __forceinline__ L5 dpp_shr1_old(const L5& o, const L5& c) {
  return L5{dpp_shr1_keep0(o.imm, c.imm), dpp_shr1_keep0(o.imd, c.imd), dpp_shr1_keep0(o.idm, c.idm),
            dpp_shr1_keep0(o.imi, c.imi), dpp_shr1_keep0(o.iiw, c.iiw), dpp_shr1_keep0(o.e, c.e)};
}

// log(m * 2^e) for m >= 0: frexp, one 16-byte LDS entry {c, -log(c)} with c ~ 1/centre of the mantissa's
// interval (512 intervals over [0.5, 1)), and log1p(f*c - 1) by a cubic (|f*c - 1| <= 2^-10: truncation error
// < 2.3e-13, below the spacing of fp64 numbers at the magnitude of these log-probabilities).  The table is indexed
// by the two low exponent bits and nine mantissa bits of the frexp mantissa, so that m == 0 (mantissa 0.0) lands on
// entry 0 = {0, -inf} and comes out as -inf without a compare: real entries are 1024..1535.
#define HXL_LOG_ENTRIES 1536
__device__ __forceinline__ double log_scaled(double m, int e, const HX_LDS double* ltab) {
  typedef double d2v __attribute__((ext_vector_type(2)));
  const double f = __builtin_amdgcn_frexp_mant(m);           // [0.5, 1), or 0
  const int k = __builtin_amdgcn_frexp_exp(m);
  const unsigned byte_off = ((unsigned)__double2hiint(f) >> 7) & 0x7FF0u;
  const d2v ce = *(const HX_LDS d2v*)((const HX_LDS char*)ltab + byte_off);
  const double r = __builtin_fma(f, ce.x, -1.0);
  double p = __builtin_fma(r, 1.0 / 3.0, -0.5);
  p = __builtin_fma(p, r, 1.0);
  const double lf = __builtin_fma(p, r, ce.y);
  return __builtin_fma((double)(e + k), 0.693147180559945309417, lf);
}

// a stored log cell -> mantissas with a common exponent (only on the wrap-around link, see the kernel)
__device__ __forceinline__ L5 from_logs(double a, double b, double c, double d, double g) {
  const double mx = vmax(vmax(vmax(a, b), vmax(c, d)), g);
  if (!(mx > HX_NEG_INF)) return l5_zero();
  const double n = __builtin_rint(mx * 1.44269504088896340736);
  const double hi = n * 0.693147180369123816490, lo = n * 1.90821492927058770002e-10;   // ln2 in two parts
  L5 r;
  r.imm = exp((a - hi) - lo);
  r.imd = exp((b - hi) - lo);
  r.idm = exp((c - hi) - lo);
  r.imi = exp((d - hi) - lo);
  r.iiw = exp((g - hi) - lo);
  r.e = (int)n;
  return r;
}

// LDS plan of a workgroup (byte offsets into the dynamic allocation, computed by plan_lds on the host).  The
// logarithm table's entries 1..1023 are never addressed: the y side is put into that hole when it fits.
// A pair's LDS is two blocks: A = {emission table, [column words], class constants, counters, zero entry} and
// B = {rings, staging ring}; offsets inside a block are the same for every pair of the workgroup, pair 0's blocks may sit
// in the table's hole.
struct LdsPlan { int elds, ycol, yclass, flags, zero;   // inside block A
                 int a0, b0, a1, stride, total; };    // block A / B of pair 0; block A of pair 1 (B follows A); bytes per further pair

#define HXL_RING 64                // columns of the strip above's last row in flight between two waves
#define HXL_STAGE 128              // wave 0's staging ring on the wrap-around link: two 64-column blocks

// BANDED (W == 1): one wavefront per pair, as k_fill_chain does for banded leaf batches - the strips of a banded pair
// run one after the other anyway.  A strip sweeps only the step windows that hold its in-envelope cells (hx_api.hip
// strip_windows), cells outside the envelope are shifted to zero, and every strip boundary is the wrap-around link.
// PPW > 1 (banded): PPW pairs per workgroup share the logarithm table; their column words come from memory
// (DevJob::yword), so that a pair needs 11 KB of LDS and twelve pairs fit a CU.
// DIR = 1: the Backward fill (reference src/forward.cpp:975-1088 for leaf-like profiles) - the same pipeline in mirrored
// coordinates (sweep row i' = R-1-i, sweep column j' = Cc-1-j, as k_fill_chain<1> and the stored layout), with the Backward
// recursion: B(i,j,s) = sum over the destination cells (i+1,j+1), (i+1,j), (i,j+1) of P[s][dest state] x emission x B(dest).
// MULTI: few pairs of many strips (one rank's share of a strong-scaling run): a pair's passes of W strips are dealt to
// `groups` workgroups, so that its waves spread over several CUs.  Inside a workgroup nothing changes (LDS rings); the
// wrap-around link - already a trip through the matrix - now crosses workgroups: write-through stores (`sc1`), its progress
// counter in memory (`counters`: 256 zeroed ints per pair, one per workgroup, [255] = a poll gave up), `sc1` polls that give
// up after HXL_PATIENCE rounds (the pair's lpEnd / lpStart becomes NaN).  As k_fill_chain's MULTI (hx_chain.hip).
#define HXL_PATIENCE (1 << 22)
template <int W, bool BANDED, int PPW, int DIR, bool MULTI = false, bool TRUNC = false>
__global__ void __launch_bounds__(W * PPW * 64, 4)   // four waves per SIMD: 128 vector registers
k_fill_leaf_linear(const DevJob* __restrict__ jobs, const double* __restrict__ exact_tab, const double* __restrict__ log_tab,
                      const LdsPlan plan, const int n_jobs, const int groups = 1, int* const counters = nullptr) {
  static_assert(!MULTI || (!BANDED && PPW == 1 && W > 1), "several workgroups per pair: unbanded pairs, one pair per workgroup");
  static_assert(!BANDED || W == 1, "banded batches run one wavefront per pair");
  static_assert(PPW == 1 || BANDED, "several pairs per workgroup: banded batches only");
  constexpr int THREADS = W * PPW * 64, PT = W * 64;       // threads of the workgroup / of a pair
  constexpr int RING_ENTRIES = BANDED ? 0 : W * HXL_RING;  // (a banded pair is one wave: every strip boundary is the wrap-around link)
  typedef double d2v __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  double* ltab = reinterpret_cast<double*>(lds);
  const int pair = PPW == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int tid = (int)threadIdx.x - pair * PT;              // thread within its pair
  const int G = MULTI ? groups : 1;
  const int job = MULTI ? (int)blockIdx.x / G : (int)blockIdx.x * PPW + pair;
  const int grp = MULTI ? (int)blockIdx.x % G : 0;
  const bool live = PPW == 1 || job < n_jobs;
  const DevJob& J = jobs[live ? job : 0];
  unsigned char* blk_a = lds + (pair == 0 ? plan.a0 : plan.a1 + (pair - 1) * plan.stride);
  unsigned char* blk_b = pair == 0 ? lds + plan.b0 : blk_a + (plan.stride - (RING_ENTRIES + HXL_STAGE) * 48);
  double* elds = reinterpret_cast<double*>(blk_a + plan.elds);
  unsigned* ycol = reinterpret_cast<unsigned*>(blk_a + plan.ycol);
  double* yclass = reinterpret_cast<double*>(blk_a + plan.yclass);
  volatile int* prog = reinterpret_cast<volatile int*>(blk_a + plan.flags);   // [W] columns written to the wave's ring (+ base)
  volatile int* cons = prog + W;                                             // [W] columns the wave has taken from the ring above
  volatile int* drain = cons + W;                                            // [1] wrap-around link: columns of wave W-1's strip that are in memory
  constexpr bool lds_words = !BANDED;                        // all column words in LDS, or (banded) a ring refilled from DevJob::yword
  // (the table first: the y side may sit in its hole)
  for (int k = threadIdx.x; k < 2; k += THREADS) ltab[k] = log_tab[k];
  for (int k = 2048 + threadIdx.x; k < 2 * HXL_LOG_ENTRIES; k += THREADS) ltab[k] = log_tab[k];
  if (tid < 2 * W + 1) prog[tid] = 0;
  if (tid == 0) {
    d2v* z = reinterpret_cast<d2v*>(blk_a + plan.zero);
    z[0] = d2v{0., 0.}; z[1] = d2v{0., 0.}; z[2] = d2v{0., __hiloint2double(0, HXL_EMIN)};
  }
  {
    // the y side, in linear space: per column {emission class, not ready ? 0xFFFF : 0}, per class
    // {exp(rootsuby), exp(insy)}, and exp() of the padded class-pair emission table
    const int Ky1 = J.y.n_cls + 1, Kx1 = J.x.n_cls + 1;
    // (64 words of padding on either side: a lane whose column is outside the lattice reads the edge column's word)
    for (int jp = tid; jp < (lds_words ? J.n_cols + 130 : 0); jp += PT) {   // (banded: see refill_words)
      const int j = jp < 64 ? 0 : (jp - 64 >= J.n_cols ? J.n_cols - 1 : jp - 64);
      if (BANDED)   // {class : 8, not ready : 1, always in envelope : 1, envelope coordinate : 22}
        ycol[jp] = (unsigned)J.y.ecls[j] | (J.y.pack[4 * (size_t)j + 3] < 0.0 ? 0x100u : 0u) |
                   ((J.y.flags[j] & F_EDGE) ? 0x200u : 0u) | (J.max_dist >= 0 ? (unsigned)J.y.env[j] << 10 : 0u);
      else if (DIR == 0)
        ycol[jp] = (unsigned)J.y.ecls[j] | (J.y.pack[4 * (size_t)j + 3] < 0.0 ? 0xFFFF0000u : 0u);
      else {
        // sweep column j is y state jc = Cc-1-j: its own readiness, and the emission class of the state jc+1 that an
        // absorbing move leads to
        const int jc = J.n_cols - 1 - j;
        ycol[jp] = (unsigned)J.y.ecls[jc + 1] | (J.y.pack[4 * (size_t)jc + 3] < 0.0 ? 0xFFFF0000u : 0u);
      }
    }
    for (int c = tid; c < Ky1; c += PT) {
      const bool real = c < J.y.n_cls;
      const int rep = real ? J.y.cls_rep[c] : 0;
      yclass[2 * c] = real ? exp(J.y.pack[4 * (size_t)rep + 1]) : 0.;
      yclass[2 * c + 1] = real ? exp(J.y.pack[4 * (size_t)rep + 2]) : 0.;
    }
    for (int e = tid; e < Kx1 * Ky1; e += PT) elds[e] = exp(J.emis_pad[e]);
  }
  __syncthreads();
  const int R = J.n_rows, Cc = J.n_cols;
  const int max_dist = J.max_dist;
  const int lane = threadIdx.x & 63, wave = PPW == 1 ? (int)(threadIdx.x >> 6) : 0;   // wave within its pair
  const int64_t plane = J.plane, ss = J.strip_stride;
  const int blk = J.blk;                                     // doubles per step-pair block (hx_device.h cell_slot_blk)
  HX_GLOBAL double* __restrict__ M = as_global(DIR ? J.bwd : J.fwd);
  const HX_GLOBAL d4v* xpack = (const HX_GLOBAL d4v*)as_global(J.x.pack);
  // the 18 transition probabilities (src/pairhmm.cpp:17-43), pinned in scalar registers
  double P[5][5];
#pragma unroll
  for (int a = 0; a < 5; ++a)
#pragma unroll
    for (int d = 0; d < 5; ++d) {
      const bool used = d == 0 || (d == 1 && a != 4) || (d == 2 && a != 3) || (d == 3 && (a == 0 || a == 3)) ||
                        (d == 4 && (a == 0 || a == 3 || a == 4));
      if (!used) { P[a][d] = 0.; continue; }
      const double pv = exp(J.T[a][d]);            // (wave-uniform, but computed by the vector unit)
      double v = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(pv)), __builtin_amdgcn_readfirstlane(__double2loint(pv)));
      asm volatile("" : "+s"(v));
      P[a][d] = v;
    }
  volatile HX_LDS int* progp = (volatile HX_LDS int*)prog;
  volatile HX_LDS int* consp = (volatile HX_LDS int*)cons;
  volatile HX_LDS int* drainp = (volatile HX_LDS int*)drain;
  const HX_LDS double* lt = (const HX_LDS double*)ltab;
  HX_LDS d2v* ring_mine = (HX_LDS d2v*)blk_b + (size_t)wave * (HXL_RING * 3);
  const int n_strips = live ? (R + 63) / 64 : 0;
  const int prev_wave = (wave + W - 1) % W, next_wave = (wave + 1) % W;
  const int WT = W * G, gw = grp * W + wave;        // the pair's waves, and this one among them
  HX_GLOBAL int* gdrain = MULTI ? (HX_GLOBAL int*)as_global(counters + 256 * job) : nullptr;
  const int prev_grp = (grp + G - 1) % G;
  bool dead = false;                                // MULTI: a poll ran out of patience
  const auto wait_drained = [&](const int need) {   // the wrap-around link: columns of the strip above that are in memory
    if (!MULTI) {
      while (drainp[0] < need) __builtin_amdgcn_s_sleep(1);
      asm volatile("" ::: "memory");
      return;
    }
    if (dead) return;
    int polls = 0;
    while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(gdrain + prev_grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < need) {
      if (++polls > HXL_PATIENCE) {
        dead = true;
        if (lane == 0) __hip_atomic_store(gdrain + 255, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
  };
  const auto publish_drained = [&](const int value) {   // (behind the caller's s_waitcnt)
    if (lane != 0) return;
    if (MULTI) __hip_atomic_store(gdrain + grp, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else drainp[0] = value;
  };
  // Strips are dealt round-robin, so with more strips than waves the last wave feeds the first one's NEXT strip, which
  // starts a whole sweep later: that link cannot be a bounded ring (the waves would wait for each other in a circle).
  // It goes through the matrix instead - wave W-1 publishes how many columns of its last row have reached memory, wave 0
  // block-loads them 64 at a time, converts them back to mantissas + exponent and stages them in a ring of its own.
  HX_LDS d2v* staging = (HX_LDS d2v*)blk_b + (size_t)RING_ENTRIES * 3;   // [HXL_STAGE] entries
  const HX_LDS d2v* ring_prev = (const HX_LDS d2v*)blk_b + (size_t)prev_wave * (HXL_RING * 3);

  for (int s = gw; s < n_strips; s += WT) {
    const int row0 = s * 64;
    const int i0 = row0 + lane;
    const bool row_valid = i0 < R;
    // x-side constants of the lane's row, in linear space; an absent row computes zeros
    double fx, f_imd, f_iiw;
    int x_wait;                                    // x state not ready: 2^29, else 0
    unsigned eoff;
    int env_x = 0;
    bool edge_x = true;
    {
      const int ic = row_valid ? (DIR ? R - 1 - i0 : i0) : 0;      // the row's x state
      const d4v p = xpack[ic];
      // Forward: the state's own in-transition, rootsubx, insx.  Backward: those of state ic+1, which an absorbing
      // move leads to; the readiness test is the row's own state's in both directions.
      const d4v q = DIR ? xpack[ic + 1] : p;
      fx = row_valid ? exp(q.x) : 0.;
      f_imd = row_valid ? exp(q.x + q.y) : 0.;
      f_iiw = row_valid ? exp(q.x + q.z) : 0.;
      x_wait = (row_valid && !(p.w < 0.0)) ? 0 : (1 << 29);
      eoff = (unsigned)J.x.ecls[DIR ? ic + 1 : ic] * (unsigned)(J.y.n_cls + 1);
      if (BANDED) {
        env_x = J.max_dist >= 0 ? J.x.env[ic] : 0;
        edge_x = (J.x.flags[ic] & F_EDGE) != 0 || J.max_dist < 0;
      }
    }
    L5 ca = l5_zero(), cb = l5_zero(), ua = l5_zero(), ub = l5_zero();
    const bool has_above = s > 0, has_below = s + 1 < n_strips;
    const bool wrap_in = has_above && wave == 0;             // the strip above went through memory (W > 1: s >= W)
    const bool wrap_out = has_below && wave == W - 1;        // this strip's last row is read back from memory
    const bool ring_out = has_below && !wrap_out;
    // column sequence numbers: column j of strip s is number (s / W) * Cc + j of its wave's ring
    const int above_base = ((s - 1) / WT) * Cc;
    const int my_base = (s / WT) * Cc;
    int64_t store_base2 = (int64_t)s * ss + (lane << 1);
    int store_t0 = 0;                              // (band-compressed storage: a window's cells are stored from its own offset)
    const int nsteps = (Cc + 64) & ~1;             // Cc + 63 anti-diagonals, rounded up to whole step pairs

    // The strip above's last row arrives through that wave's LDS ring, one cell {5 mantissas, exponent} per
    // column: the producer's lane 63 writes it the moment it is computed, lane 0 of this wave needs it 64 + a few
    // steps later.  The DPP shift that hands every lane its upper neighbour's new cell (wave_shr:1) leaves lane 0
    // untouched - with the ring entry as its `old` operand, lane 0 receives the boundary cell in the same instruction.
    auto wait_for = [&](const int cols) {          // columns of the strip above that must have been written
      const int need = above_base + (cols < Cc ? cols : Cc);
      while (progp[prev_wave] < need) __builtin_amdgcn_s_sleep(1);
      asm volatile("" ::: "memory");               // the ring reads below stay below
    };
    // (strip 0 reads an all-zero entry: the ring read below is unconditional)
    const HX_LDS d2v* ring_above = !has_above ? (const HX_LDS d2v*)(blk_a + plan.zero) : wrap_in ? (const HX_LDS d2v*)staging : ring_prev;
    const int ring_base = (wrap_in || !has_above) ? 0 : above_base;
    const int ring_mask = !has_above ? 0 : wrap_in ? HXL_STAGE - 1 : HXL_RING - 1;
    auto ring_entry = [&](const int col) -> L5 {   // (uniform address: a broadcast read)
      const HX_LDS d2v* q = ring_above + (size_t)((ring_base + col) & ring_mask) * 3;
      const d2v a = q[0], b = q[1], c = q[2];
      return L5{a.x, a.y, b.x, b.y, c.x, __double2loint(c.y)};
    };
    // wrap-around link: stage columns [c0, c0 + 64) of the strip above's last row
    auto stage_block = [&](const int c0) {
      const int hi = (c0 + 64 < Cc) ? c0 + 64 : Cc;
      const int need = above_base + hi;
      wait_drained(need);
      const int jj = c0 + lane;
      const int64_t sl = jj < Cc ? ((BANDED && DIR == 0) ? stored_slot(J, row0 - 1, jj) : cell_slot_blk(ss, blk, row0 - 1, jj)) : -1;
      if (jj < Cc && sl < 0) {                     // band-compressed storage: not stored = outside the envelope
        HX_LDS d2v* q = staging + (size_t)(jj & (HXL_STAGE - 1)) * 3;
        q[0] = d2v{0., 0.}; q[1] = d2v{0., 0.}; q[2] = d2v{0., __hiloint2double(0, HXL_EMIN)};
      }
      if (sl >= 0) {
        // agent-scope relaxed loads: served by L2, never by a stale L1 line
        const double a = __hip_atomic_load(M + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double b = __hip_atomic_load(M + plane + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double c = __hip_atomic_load(M + 2 * plane + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double d = __hip_atomic_load(M + 3 * plane + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double g = __hip_atomic_load(M + 4 * plane + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        L5 v = from_logs(a, b, c, d, g);
        if (BANDED && max_dist >= 0) {
          // the strip above only wrote its in-envelope cells (HX_SPARSE_ENVELOPE: anything else is undefined)
          const int ia = DIR ? R - row0 : row0 - 1, ja = DIR ? Cc - 1 - jj : jj;   // the cell's x and y states
          int dist = J.x.env[ia] - J.y.env[ja];
          dist = dist < 0 ? -dist : dist;
          if (!(((J.x.flags[ia] | J.y.flags[ja]) & F_EDGE) || dist <= max_dist)) v = l5_zero();
        }
        HX_LDS d2v* q = staging + (size_t)(jj & (HXL_STAGE - 1)) * 3;
        q[0] = d2v{v.imm, v.imd};
        q[1] = d2v{v.idm, v.imi};
        q[2] = d2v{v.iiw, __hiloint2double(0, v.e)};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (same wave reads them back: LDS operations complete in order)
    };
    // lane 0's upper and diagonal neighbours at the first step t0 of a sweep: (row0-1, t0) and (row0-1, t0-1)
    auto open_sweep = [&](const int t0) {
      if (!has_above) return;
      if (wrap_in) {
        if (t0 > 0 && (t0 & 63) == 0) stage_block(t0 - 64);     // (the block of column t0-1)
        if (t0 < Cc) stage_block(t0 & ~63);
      } else {
        wait_for(8);
      }
      const L5 b0 = ring_entry(t0), bd = ring_entry(t0 > 0 ? t0 - 1 : 0);
      if (lane == 0) {
        ua = b0;
        if (BANDED && t0 > 0) ub = bd;
      }
    };
    // Column words, fetched one step ahead.  Unbanded: the pair's whole y side is in LDS (ycol, padded by 64 words on either
    // side).  Banded: a 256-word LDS ring refilled from memory (DevJob::yword) every 64 steps - a vector-memory load
    // inside the step loop would make every step wait for the stores before it (they retire in issue order).
    unsigned wnext = 0;
    const HX_GLOBAL unsigned* ywords = (const HX_GLOBAL unsigned*)as_global(DIR ? J.yword_bwd : J.yword);
    auto refill_words = [&](const int i0) {             // word indices [i0, i0 + 64)
      const unsigned v = ywords[i0 + lane];
      ycol[(i0 + lane) & 255] = v;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    };
    auto first_words = [&](const int t0) {
      if (lds_words) { wnext = ycol[t0 + 64 - lane]; return; }
      // the steps [t0, next multiple of 64) read word indices t + 64 - lane: within [(t0 & ~63) + 1, (t0 & ~63) + 128)
      refill_words(t0 & ~63); refill_words((t0 & ~63) + 64); refill_words((t0 & ~63) + 128);
      wnext = ycol[(t0 + 64 - lane) & 255];
    };
    auto next_word = [&](const int t) -> unsigned {
      const unsigned w = wnext;
      wnext = lds_words ? ycol[t + 65 - lane] : ycol[(t + 65 - lane) & 255];
      return w;
    };
    // one anti-diagonal step: the lane's new cell from left (own previous), u1 = (i-1, j), u2 = (i-1, j-1)
    auto step = [&](const int t, const L5& left, L5& out, L5& u1, L5& u2, const unsigned w) {
      if (!lds_words && (t & 63) == 62) refill_words(t + 2 + 128);   // (steps from t+2 on read up to index t+2+127)
      if ((t & 7) == 7 && has_above && t + 1 < Cc) {
        if (wrap_in) {
          if ((t & 63) == 63) stage_block(t + 1);
        } else {
          if (lane == 0) consp[wave] = above_base + t;     // the ring's slots up to column t may be reused
          wait_for(t + 9);
        }
      }
      // y-side constants of column j = t - lane (cells outside the lattice only feed cells outside it)
      const unsigned c = BANDED ? (w & 0xFFu) : (w & 0xFFFFu);
      const d2v rc = reinterpret_cast<const d2v*>(yclass)[c];
      const int y_wait = BANDED ? (int)((w & 0x100u) << 21) : (int)((w >> 16) << 13);   // y state not ready: 2^29, else 0
      const double em = elds[eoff + c];
      int E;
      if (DIR == 0) {
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
        const int e_diag = u2.e;
        // (row0-1, t+1), lane 0's upper neighbour of the next step, into the registers the diagonal cell has just
        // vacated.  Unconditional: past the last column it returns a stale entry, which only feeds cells outside the lattice.
        u2 = ring_entry(t + 1);
        // common exponent of the new cell, and the three groups brought to it; a state that may not be entered
        // (y or x state not ready: src/forward.cpp:97,133) is shifted out of the fp64 range, i.e. to zero
        E = left.e > u1.e ? left.e : u1.e;
        E = E > e_diag ? E : e_diag;
        int du = (u1.e - E) - y_wait, dl = (left.e - E) - x_wait, dd = e_diag - E;
        if (BANDED) {
          // a cell outside the envelope (src/forward.h:92-98) is shifted to zero as a whole
          int dist = env_x - (int)(w >> 10);
          dist = dist < 0 ? -dist : dist;
          const int out_of_env = (edge_x || (w & 0x200u) || dist <= max_dist) ? 0 : (1 << 29);
          du -= out_of_env; dl -= out_of_env; dd -= out_of_env;
        }
        out.imd = __builtin_ldexp(s_imd * f_imd, du);
        out.iiw = __builtin_ldexp(s_iiw * f_iiw, du);
        out.idm = __builtin_ldexp(s_idm * rc.x, dl);
        out.imi = __builtin_ldexp(s_imi * rc.y, dl);
        out.imm = __builtin_ldexp(s_imm * (fx * em), dd);
      } else {
        // Backward (src/forward.cpp:1018-1065 for leaf-like profiles): the five destination terms - the xy-absorbing move
        // into (i+1,j+1), the x-absorbing moves into (i+1,j) as IMD / IIW, the y-absorbing moves into (i,j+1) as IDM / IMI -
        // brought to the cell's exponent (a move that may not be made is shifted out of range), then 18 multiply-adds
        const int e_diag = u2.e;
        const double tD = u2.imm * (fx * em);
        const double t1x = u1.imd * f_imd, t2x = u1.iiw * f_iiw;
        const double t1y = left.idm * rc.x, t2y = left.imi * rc.y;
        u2 = ring_entry(t + 1);
        E = left.e > u1.e ? left.e : u1.e;
        E = E > e_diag ? E : e_diag;
        int du = (u1.e - E) - y_wait, dl = (left.e - E) - x_wait, dd = e_diag - E;
        if (BANDED) {
          // a cell outside the envelope (src/forward.h:92-98) is shifted to zero as a whole
          int dist = env_x - (int)(w >> 10);
          dist = dist < 0 ? -dist : dist;
          const int out_of_env = (edge_x || (w & 0x200u) || dist <= max_dist) ? 0 : (1 << 29);
          du -= out_of_env; dl -= out_of_env; dd -= out_of_env;
        }
        const double D = __builtin_ldexp(tD, dd);
        const double d1x = __builtin_ldexp(t1x, du), d2x = __builtin_ldexp(t2x, du);
        const double d1y = __builtin_ldexp(t1y, dl), d2y = __builtin_ldexp(t2y, dl);
        // (the reference accumulates term by term, in this order: src/forward.cpp:1018-1065)
        out.imm = lin_acc<TRUNC>(P[0][3], d2y, lin_acc<TRUNC>(P[0][2], d1y, lin_acc<TRUNC>(P[0][4], d2x, lin_acc<TRUNC>(P[0][1], d1x, P[0][0] * D))));
        out.imd = lin_acc<TRUNC>(P[1][2], d1y, lin_acc<TRUNC>(P[1][1], d1x, P[1][0] * D));
        out.idm = lin_acc<TRUNC>(P[2][2], d1y, lin_acc<TRUNC>(P[2][1], d1x, P[2][0] * D));
        out.imi = lin_acc<TRUNC>(P[3][3], d2y, lin_acc<TRUNC>(P[3][4], d2x, lin_acc<TRUNC>(P[3][1], d1x, P[3][0] * D)));
        out.iiw = lin_acc<TRUNC>(P[4][2], d1y, lin_acc<TRUNC>(P[4][4], d2x, P[4][0] * D));
      }
      out.e = E;
      if ((t & HXL_RENORM_MASK) == 0) {            // wave-uniform
        if (s == 0 && t == 0) {
          if (DIR == 0) {                          // cell (0,0): lpStart() = 0 (src/forward.cpp:73)
            if (lane == 0) { out.imm = 1.0; out.e = 0; }
          } else if (lane == 0) {
            // the cell feeding END is initialised by assignment (src/forward.cpp:981-995)
            const double lpe = J.x.pack[4 * (size_t)R] + J.y.pack[4 * (size_t)Cc];
            out.imm = exp(lpe + J.T[0][5]); out.imd = exp(lpe + J.T[1][5]); out.idm = exp(lpe + J.T[2][5]);
            out.imi = exp(lpe + J.T[3][5]); out.iiw = exp(lpe + J.T[4][5]);
            out.e = 0;
          }
        }
        const double mx = vmax(vmax(vmax(out.imm, out.imd), vmax(out.idm, out.imi)), out.iiw);
        const int k = __builtin_amdgcn_frexp_exp(mx);
        out.imm = __builtin_ldexp(out.imm, -k);
        out.imd = __builtin_ldexp(out.imd, -k);
        out.idm = __builtin_ldexp(out.idm, -k);
        out.imi = __builtin_ldexp(out.imi, -k);
        out.iiw = __builtin_ldexp(out.iiw, -k);
        out.e = mx > 0. ? out.e + k : HXL_EMIN;
      }
      // the strip's last row goes to the wave below through the ring (lane 63, column t - 63)
      if (ring_out && t >= 63 && t - 63 < Cc) {
        const int col = t - 63;
        if ((col & 7) == 0) {
          // flow control: the slots of the next eight columns must have been consumed
          const int need = my_base + col + 8 - HXL_RING;
          while (consp[next_wave] < need) __builtin_amdgcn_s_sleep(1);
          asm volatile("" ::: "memory");
        }
        if (lane == 63) {
          HX_LDS d2v* q = ring_mine + (size_t)((my_base + col) & (HXL_RING - 1)) * 3;
          q[0] = d2v{out.imm, out.imd};
          q[1] = d2v{out.idm, out.imi};
          q[2] = d2v{out.iiw, __hiloint2double(0, out.e)};
        }
        if ((col & 7) == 7 || col == Cc - 1) {
          asm volatile("" ::: "memory");           // data before flag
          if (lane == 63) progp[wave] = my_base + col + 1;   // (LDS operations of a wave complete in order)
        }
      }
      u2 = dpp_shr1_old(u2, out);
    };

    // column words, fetched one step ahead (ycol is padded by 64 words on either side)
    // With a band, the strip sweeps only the (up to two) step windows that hold its in-envelope cells, widened to whole
    // step pairs; cells left of a window are outside the envelope, so the register window restarts from zero.
    int wlo[2] = {0, 0}, whi[2] = {nsteps, 0};
    if (BANDED) {
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
    for (int wi = 0; wi < (BANDED ? 2 : 1); ++wi) {
    if (BANDED && whi[wi] <= wlo[wi]) continue;
    const int wstart = BANDED ? wlo[wi] : 0, wend = BANDED ? whi[wi] : nsteps;
    if (BANDED) { ca = l5_zero(); cb = l5_zero(); ua = l5_zero(); ub = l5_zero(); }
    if (BANDED && DIR == 0 && J.strip_base) { store_base2 = J.strip_base[2 * s + wi] + (lane << 1); store_t0 = wstart; }
    open_sweep(wstart);
    first_words(wstart);
    // a pair of steps: in the strip-skewed layout its two cells per row are adjacent, 16 bytes per lane and state plane
    auto step_pair = [&](const int t) {
      step(t, cb, ca, ua, ub, next_word(t));
      const double l0 = log_scaled(ca.imm, ca.e, lt), l1 = log_scaled(ca.imd, ca.e, lt),
                   l2 = log_scaled(ca.idm, ca.e, lt), l3 = log_scaled(ca.imi, ca.e, lt),
                   l4 = log_scaled(ca.iiw, ca.e, lt);
      step(t + 1, ca, cb, ub, ua, next_word(t + 1));
      const double h0 = log_scaled(cb.imm, cb.e, lt), h1 = log_scaled(cb.imd, cb.e, lt),
                   h2 = log_scaled(cb.idm, cb.e, lt), h3 = log_scaled(cb.imi, cb.e, lt),
                   h4 = log_scaled(cb.iiw, cb.e, lt);
      {
        // t64 = j + (i & 63) = t
        const int64_t sl = store_base2 + (int64_t)((t - store_t0) >> 1) * blk;
        HX_GLOBAL d2v* M2 = (HX_GLOBAL d2v*)(M + sl);
        const int64_t plane2 = plane >> 1;
        // write-once data that this kernel never reads again (the wrap-around link reads 1/64 of it, from L2 or
        // memory): non-temporal stores - 17.4 -> 15.8 ms on the headline workload with separate state planes; with the
        // interleaved layout (5 KiB contiguous per iteration) they measure the same as plain stores
        if (MULTI) {
          // the strip below may run on another XCD: write-through stores (never `nt`, which stays in this XCD's L2)
          const d2v v0{l0, h0}, v1{l1, h1}, v2{l2, h2}, v3{l3, h3}, v4{l4, h4};
          asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(&M2[0]), "v"(v0) : "memory");
          asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(&M2[plane2]), "v"(v1) : "memory");
          asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(&M2[2 * plane2]), "v"(v2) : "memory");
          asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(&M2[3 * plane2]), "v"(v3) : "memory");
          asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(&M2[4 * plane2]), "v"(v4) : "memory");
        } else {
        __builtin_nontemporal_store(d2v{l0, h0}, &M2[0]);
        __builtin_nontemporal_store(d2v{l1, h1}, &M2[plane2]);
        __builtin_nontemporal_store(d2v{l2, h2}, &M2[2 * plane2]);
        __builtin_nontemporal_store(d2v{l3, h3}, &M2[3 * plane2]);
        __builtin_nontemporal_store(d2v{l4, h4}, &M2[4 * plane2]);
        }
      }
      if (wrap_out && !BANDED) {
        // wrap-around link: a column counts once its stores have left the wave.  Vector-memory operations retire in
        // issue order and every iteration issues five stores (and nothing else), so everything stored
        // HXL_PUBLISH_LAG steps = 8 iterations ago is older than the wave's 40 youngest operations.
        const int fin = (t + 2 < nsteps ? t + 2 : nsteps) - 63;
        if (fin >= Cc) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          publish_drained(my_base + Cc);
        } else {
          const int done = fin - HXL_PUBLISH_LAG;
          if (done > 0 && ((done >> 6) != ((done - 2) >> 6))) {
            // (five stores per iteration and nothing else: the HXL_PUBLISH_LAG / 2 iterations since hold exactly this many)
            constexpr int PUBLISH_WAIT = 5 * (HXL_PUBLISH_LAG / 2);
            static_assert(PUBLISH_WAIT <= 63 && HXL_PUBLISH_LAG % 2 == 0, "s_waitcnt vmcnt holds 6 bits");
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PUBLISH_WAIT) : "memory");
            publish_drained(my_base + done);
          }
        }
      }
    };
    for (int t = wstart; t < wend; t += 2) step_pair(t);
    }
    if (BANDED && wrap_out) {
      // (the windows need not reach the strip's last step: everything this strip will ever write is out)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      publish_drained(my_base + Cc);
    }
    if (has_above && !wrap_in && lane == 0) consp[wave] = above_base + Cc;
    if (MULTI && s == n_strips - 1) {
      // the last strip finishes last; what lpEnd / lpStart read lies in this strip - its stores are write-through
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const bool gave_up = __hip_atomic_load(gdrain + 255, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
      if (lane == 0) {
        if (DIR == 0) *J.lp_end = gave_up ? __builtin_nan("") : forward_lp_end(J, ExactLse{exact_tab});
        else *J.lp_start = gave_up ? __builtin_nan("") : J.bwd[cell_slot_blk(ss, blk, R - 1, Cc - 1)];
      }
    }
  }
  if (MULTI) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (live && wave == 0 && lane == 0) {
    if (DIR == 0) *J.lp_end = forward_lp_end(J, ExactLse{exact_tab});
    else *J.lp_start = J.bwd[cell_slot_blk(ss, blk, R - 1, Cc - 1)];   // B(0,0).IMM in mirrored coordinates
  }
}

}  // namespace

// see log_scaled: entry 0 = {0, -inf}; entries 1024 + k, k < 512: {c, -log(c)} with c = 1 / (centre of interval k of [0.5, 1))
int log_table_doubles() { return 2 * HXL_LOG_ENTRIES; }
void build_log_table(double* out) {
  for (int k = 0; k < 2 * HXL_LOG_ENTRIES; ++k) out[k] = 0.;
  out[1] = -HUGE_VAL;
  for (int k = 0; k < 512; ++k) {
    const double centre = 0.5 * (1.0 + (k + 0.5) / 512.0);
    const double c = 1.0 / centre;
    out[2 * (1024 + k)] = c;
    out[2 * (1024 + k) + 1] = -log(c);
  }
}

static LdsPlan plan_lds(int W, int PPW, bool banded, int yl_cols, int yl_emis, int yl_cls) {
  LdsPlan p;
  const int table = 16 * HXL_LOG_ENTRIES, hole = 16 * 1024;
  int a = 0;
  p.elds = a; a += (8 * yl_emis + 15) & ~15;
  p.ycol = a; a += banded ? 4 * 256 : (4 * (yl_cols + 132) + 15) & ~15;   // (banded: a ring of 256 words)
  p.yclass = a; a += 16 * yl_cls;
  p.flags = a; a += (4 * (2 * W + 1) + 15) & ~15;
  p.zero = a; a += 48;                           // an all-zero ring entry: what strip 0 reads as its row above
  const int b = ((banded ? 0 : W * HXL_RING) + HXL_STAGE) * 48;   // one ring per wave + wave 0's staging ring
  int end = table;
  // pair 0: as much as fits goes into the hole of the table (entries 1..1023 are never addressed)
  if (16 + a + b <= hole) { p.a0 = 16; p.b0 = 16 + a; }
  else if (16 + a <= hole) { p.a0 = 16; p.b0 = end; end += b; }
  else { p.a0 = end; p.b0 = end + a; end += a + b; }
  p.a1 = end;
  p.stride = a + b;
  end += (PPW - 1) * p.stride;
  p.total = (end + 15) & ~15;
  return p;
}

int launch_forward_leaf_linear(const DevJob* d_jobs, int n_jobs, int max_rows, bool banded, Tab8 tab8, Tab16 tab16,
                               int yl_cols, int yl_emis, int yl_cls, int multi, int* counters, bool trunc, hipStream_t st) {
  const double* tab = tab8.p;
  const double* log_tab = tab16.p;
  if (yl_cls > HX_YL_MAX_CLS_LINEAR + 1) return launch_fail("%d emission classes exceed the scaled-probability kernel's column words", yl_cls);
  if (multi > 1 && !banded) {     // a small batch: `multi` workgroups of four waves per pair (MULTI; see chain_multi_groups)
    const LdsPlan p = plan_lds(4, 1, false, yl_cols, yl_emis, yl_cls);
    if (p.total > HX_LDS_LIMIT) return launch_fail("k_fill_leaf_linear<multi> needs %d bytes of LDS (limit %d)", p.total, HX_LDS_LIMIT);
    if (trunc)
      hipLaunchKernelGGL((k_fill_leaf_linear<4, false, 1, 0, true, true>), dim3(n_jobs * multi), dim3(4 * 64), p.total, st, d_jobs, tab, log_tab, p, n_jobs,
                         multi, counters);
    else
      hipLaunchKernelGGL((k_fill_leaf_linear<4, false, 1, 0, true>), dim3(n_jobs * multi), dim3(4 * 64), p.total, st, d_jobs, tab, log_tab, p, n_jobs,
                         multi, counters);
    return 0;
  }
#define HXL_LAUNCH(W_, B_, PPW_) do { const LdsPlan p = plan_lds(W_, PPW_, B_, yl_cols, yl_emis, yl_cls); \
    if (p.total > HX_LDS_LIMIT) return launch_fail("k_fill_leaf_linear<%d> needs %d bytes of LDS (limit %d)", W_, p.total, HX_LDS_LIMIT); \
    if (trunc) hipLaunchKernelGGL((k_fill_leaf_linear<W_, B_, PPW_, 0, false, true>), dim3((n_jobs + PPW_ - 1) / PPW_), dim3(W_ * PPW_ * 64), p.total, st, \
                       d_jobs, tab, log_tab, p, n_jobs); \
    else hipLaunchKernelGGL((k_fill_leaf_linear<W_, B_, PPW_, 0>), dim3((n_jobs + PPW_ - 1) / PPW_), dim3(W_ * PPW_ * 64), p.total, st, \
                       d_jobs, tab, log_tab, p, n_jobs); } while (0)
  if (banded) {
    // one wavefront per pair; with more pairs than fit the CUs one by one (LDS: four workgroups of one pair), six pairs
    // per workgroup share the logarithm table: twelve waves per CU
    const char* v = getenv("HX_LINEAR_PPW");     // tuning / test hook
    const int forced = v ? atoi(v) : 0;
    if (forced > 1 || (forced == 0 && n_jobs > 1024)) {
      // two workgroups per CU (160 KB of LDS; a workgroup of exactly 80 KB was measured NOT to fit twice)
      if (plan_lds(1, 6, true, yl_cols, yl_emis, yl_cls).total <= 76 * 1024) HXL_LAUNCH(1, true, 6);
      else if (plan_lds(1, 5, true, yl_cols, yl_emis, yl_cls).total <= 76 * 1024) HXL_LAUNCH(1, true, 5);
      else HXL_LAUNCH(1, true, 4);
    } else HXL_LAUNCH(1, true, 1);
    return 0;
  }
  const char* v = getenv("HX_LINEAR_WAVES");     // tuning / test hook: waves per pair (any count works for any size)
  const int forced = v ? atoi(v) : 0;
  if (forced == 1) HXL_LAUNCH(1, false, 1);
  else if (forced == 2) HXL_LAUNCH(2, false, 1);
  else if (forced == 4) HXL_LAUNCH(4, false, 1);
  else if (forced == 8) HXL_LAUNCH(8, false, 1);
  else if (forced == 16) HXL_LAUNCH(16, false, 1);
  else if (max_rows <= 64) HXL_LAUNCH(1, false, 1);
  else if (max_rows <= 128) HXL_LAUNCH(2, false, 1);
  else if (max_rows <= 256) HXL_LAUNCH(4, false, 1);
  // more pairs than compute units: eight waves per pair, two pairs per CU (less pipeline fill per pair)
  else if (max_rows <= 512 || n_jobs > 256) HXL_LAUNCH(8, false, 1);
  else HXL_LAUNCH(16, false, 1);
#undef HXL_LAUNCH
  return 0;
}

// Backward fill of leaf batches on scaled probabilities (the same kernel, DIR = 1)
int launch_backward_leaf_linear(const DevJob* d_jobs, int n_jobs, int max_rows, bool banded, Tab8 tab8, Tab16 tab16,
                                int yl_cols, int yl_emis, int yl_cls, int multi, int* counters, bool trunc, hipStream_t st) {
  const double* tab = tab8.p;
  const double* log_tab = tab16.p;
  if (yl_cls > HX_YL_MAX_CLS_LINEAR + 1) return launch_fail("%d emission classes exceed the scaled-probability kernel's column words", yl_cls);
  if (multi > 1 && !banded) {     // a small batch: `multi` workgroups of four waves per pair (MULTI; see chain_multi_groups)
    const LdsPlan p = plan_lds(4, 1, false, yl_cols, yl_emis, yl_cls);
    if (p.total > HX_LDS_LIMIT) return launch_fail("k_fill_leaf_linear<multi> needs %d bytes of LDS (limit %d)", p.total, HX_LDS_LIMIT);
    if (trunc)
      hipLaunchKernelGGL((k_fill_leaf_linear<4, false, 1, 1, true, true>), dim3(n_jobs * multi), dim3(4 * 64), p.total, st, d_jobs, tab, log_tab, p, n_jobs,
                         multi, counters);
    else
      hipLaunchKernelGGL((k_fill_leaf_linear<4, false, 1, 1, true>), dim3(n_jobs * multi), dim3(4 * 64), p.total, st, d_jobs, tab, log_tab, p, n_jobs,
                         multi, counters);
    return 0;
  }
#define HXL_LAUNCH(W_, B_, PPW_) do { const LdsPlan p = plan_lds(W_, PPW_, B_, yl_cols, yl_emis, yl_cls); \
    if (p.total > HX_LDS_LIMIT) return launch_fail("k_fill_leaf_linear<%d, bwd> needs %d bytes of LDS (limit %d)", W_, p.total, HX_LDS_LIMIT); \
    if (trunc) hipLaunchKernelGGL((k_fill_leaf_linear<W_, B_, PPW_, 1, false, true>), dim3((n_jobs + PPW_ - 1) / PPW_), dim3(W_ * PPW_ * 64), p.total, st, \
                       d_jobs, tab, log_tab, p, n_jobs); \
    else hipLaunchKernelGGL((k_fill_leaf_linear<W_, B_, PPW_, 1>), dim3((n_jobs + PPW_ - 1) / PPW_), dim3(W_ * PPW_ * 64), p.total, st, \
                       d_jobs, tab, log_tab, p, n_jobs); } while (0)
  if (banded) {
    const char* v = getenv("HX_LINEAR_PPW");
    const int forced = v ? atoi(v) : 0;
    if (forced > 1 || (forced == 0 && n_jobs > 1024)) {
      if (plan_lds(1, 6, true, yl_cols, yl_emis, yl_cls).total <= 76 * 1024) HXL_LAUNCH(1, true, 6);
      else if (plan_lds(1, 5, true, yl_cols, yl_emis, yl_cls).total <= 76 * 1024) HXL_LAUNCH(1, true, 5);
      else HXL_LAUNCH(1, true, 4);
    } else HXL_LAUNCH(1, true, 1);
    return 0;
  }
  const char* v = getenv("HX_LINEAR_WAVES");
  const int forced = v ? atoi(v) : 0;
  if (forced == 1) HXL_LAUNCH(1, false, 1);
  else if (forced == 2) HXL_LAUNCH(2, false, 1);
  else if (forced == 8) HXL_LAUNCH(8, false, 1);
  else if (max_rows <= 64) HXL_LAUNCH(1, false, 1);
  else if (max_rows <= 128) HXL_LAUNCH(2, false, 1);
  else if (max_rows <= 256) HXL_LAUNCH(4, false, 1);
  else if (max_rows <= 512 || n_jobs > 256) HXL_LAUNCH(8, false, 1);
  else HXL_LAUNCH(16, false, 1);
#undef HXL_LAUNCH
  return 0;
}

}  // namespace hx
