// Forward fill of GENERAL profiles (state DAGs: internal tree nodes) on SCALED PROBABILITIES: the HX_LSE_LINEAR policy for
// the kernel classes KC_DAG / KC_DAG_BANDED.  Recursion and data flow are those of k_forward_dag_pipe (hx_dag.hip; reference
// src/forward.cpp:68-223): one workgroup per pair, 64-row strips dealt to the waves round-robin, lane <-> row,
// step <-> anti-diagonal, a cell's sources read back from memory (its own strip: program order of the wave's memory
// operations; strips above: published progress counters).
//
// What changes is the arithmetic.  In log space a step of the general pipeline is ~18 log-sum-exps (13 for the five
// outgoing sums of the cell, the rest for its incoming transitions) and, with two to three waves per SIMD, the kernel is
// bound by instruction issue: ~1300 instructions per step.  Here every cell is five fp64 mantissas with ONE integer
// exponent (p_state = m_state * 2^E), so the outgoing sums are 13 multiply-adds and an incoming transition is a multiply-add
// after aligning exponents (v_ldexp).  A cell is kept in two forms:
//   * the reference's: five log-probabilities in the Forward matrix (what hx_batch_read_matrix, tracebacks and the
//     posterior read), produced by a table-plus-cubic logarithm just before the store (log_scaled, as hx_linear.hip);
//   * the kernel's own: ten mantissas (the five states and the five outgoing sums) and the exponent, in scratch planes
//     (DevJob::agg: [10][plane] doubles + [plane] ints), which is what later cells read.
// Emission terms and the per-state absorption constants enter as (mantissa, exponent) pairs: the constants are split once
// per state (k_lin_pack), the emission of a cell by a table-plus-quartic exponential (exp_split).
// Like the scaled-probability fills of leaf pairs this drops the reference's truncation of log-sum-exp terms below e^-10,
// so it agrees with the reference to the reference's own approximation error, not bit for bit.
//
// Compiled with -ffp-contract=off (see hx_lse.h); multiply-adds are written as __builtin_fma where they are wanted.
#include <hip/hip_runtime.h>
#include "hx_device.h"
#include "hx_lse.h"
#include "hx_common.h"
#include "hx_policy.h"
#include "hx_kernels.h"

namespace hx {

namespace {

#define HXD_MAX_WAVES 8
#define HXD_EMIN (-(1 << 28))      // exponent of a zero: loses every max()
#define HXD_LOG_ENTRIES 1536       // the logarithm table of hx_linear.hip (build_log_table)
#define HXD_EXP_ENTRIES 512
#define HXD_EXTRA 2                // in-transitions beyond the inline three whose cells are fetched in one batch per step

typedef double d2v __attribute__((ext_vector_type(2)));

// Trace builds (-DHX_DAG_TRACE): per-strip cycle sums of the phases of a step, printed by lane 0 at the end of every strip
#ifdef HX_DAG_TRACE
#define HXD_TR(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const long long now_ = (long long)__builtin_readcyclecounter(); tr_sum[k] += now_ - tr_last; tr_last = now_; } while (0)
#else
#define HXD_TR(k) do { } while (0)
#endif

// planes of the kernel's own cell format
enum { V_IMM = 0, V_IMD = 1, V_IDM = 2, V_IMI = 3, V_IIW = 4, V_G0 = 5, V_G1 = 6, V_G2 = 7, V_G3 = 8, V_G4 = 9, V_PLANES = 10 };

// per-state constants in linear form (k_lin_pack)
struct alignas(16) LinPack {
  double w0, w1, w2;          // exp(lpTrans) of the first three in-transitions
  double rs_m, ins_m;         // rootsub / ins of the state as mantissa ...
  int32_t rs_e, ins_e;        // ... and exponent (0, HXD_EMIN for -inf)
};
static_assert(sizeof(LinPack) == 48, "LinPack layout");

__device__ __forceinline__ double ldexp_fast(double m, int e) { return __builtin_amdgcn_ldexp(m, e); }

// exp(lp) = m * 2^e, m in [1, 2): 2^frac by a 512-entry table and a quartic (|r| < ln2 / 512: truncation < 4e-17)
__device__ __forceinline__ void exp_split(const double lp, const HX_LDS double* etab, double& m, int& e) {
  const double x = lp * 1.44269504088896340736;
  const double n = __builtin_floor(x);
  const double f = x - n;
  int k = (int)(f * 512.0);
  k = k > 511 ? 511 : k;
  const double r = (f - (double)k * (1.0 / 512.0)) * 0.693147180559945309417;
  double p = __builtin_fma(r, 1.0 / 24.0, 1.0 / 6.0);
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  const bool ok = lp > -1e8;
  m = ok ? etab[k] * p : 0.0;
  e = ok ? (int)n : HXD_EMIN;
}

// log(m * 2^e), m >= 0 (see hx_linear.hip: frexp, {c, -log c} table entry, cubic log1p); m == 0 gives -inf
__device__ __forceinline__ double log_scaled(const double m, const int e, const HX_LDS double* ltab) {
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

// a running sum of terms with their own exponents: (s, E) += m * 2^e * w
struct Acc1 { double s; int E; };
struct Acc2 { double a, b; int E; };
__device__ __forceinline__ void add1(Acc1& A, const double m, const int e, const double w) {
  const int te = m > 0. ? e : HXD_EMIN;
  const int En = te > A.E ? te : A.E;
  A.s = __builtin_fma(ldexp_fast(m, te - En), w, ldexp_fast(A.s, A.E - En));
  A.E = En;
}
__device__ __forceinline__ void add2(Acc2& A, const double ma, const double mb, const int e, const double w) {
  const int te = (ma > 0. || mb > 0.) ? e : HXD_EMIN;
  const int En = te > A.E ? te : A.E;
  const int dn = te - En, dp = A.E - En;
  A.a = __builtin_fma(ldexp_fast(ma, dn), w, ldexp_fast(A.a, dp));
  A.b = __builtin_fma(ldexp_fast(mb, dn), w, ldexp_fast(A.b, dp));
  A.E = En;
}

// COH: the pair's strips are dealt to several workgroups (the MULTI launch, see hx_dag.hip k_backward_dag_multi): every
// scratch access is `sc1` - stores write through, loads bypass the L1
template <bool COH = false>
__device__ __forceinline__ double ldg(const HX_GLOBAL double* base, const unsigned byte_off) {
  HX_GLOBAL double* p = (HX_GLOBAL double*)((HX_GLOBAL char*)const_cast<HX_GLOBAL double*>(base) + byte_off);
  if (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}
template <bool COH = false>
__device__ __forceinline__ int ldgi(const HX_GLOBAL int* base, const unsigned byte_off) {
  HX_GLOBAL int* p = (HX_GLOBAL int*)((HX_GLOBAL char*)const_cast<HX_GLOBAL int*>(base) + byte_off);
  if (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}
template <bool COH = false>
__device__ __forceinline__ void stg(HX_GLOBAL double* base, const unsigned byte_off, const double v) {
  {
    HX_GLOBAL double* p = (HX_GLOBAL double*)((HX_GLOBAL char*)base + byte_off);
    if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
  }
}
template <bool COH = false>
__device__ __forceinline__ void stgi(HX_GLOBAL int* base, const unsigned byte_off, const int v) {
  HX_GLOBAL int* p = (HX_GLOBAL int*)((HX_GLOBAL char*)base + byte_off);
  if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

// the cell a lane computed in the previous step: stored one step late (behind the next step's loads, so that those do not
// queue behind fresh stores), and forwarded through registers to the two cells that may need it before it is in memory -
// (i, j+1) of the lane itself and (i+1, j) of the next lane
struct PrevCell { double v[V_PLANES]; int E; unsigned slot; };

__device__ __forceinline__ int wave_shr1_int(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false); }

// The scratch planes are private to this kernel and use their own layout inside a strip: cell (row lane l, skewed column
// t = column + l) at byte (t * 64 + l) * 8 - a wavefront's store or load of one anti-diagonal is 512 contiguous bytes,
// eight full 64-byte requests, where the matrix layout's step pairs (hx_device.h: (t / 2) * 128 + 2 l + t % 2) make it
// sixteen half-used ones.  The fill is bound by the CU's vector-memory pipeline, so this halves its largest term.
__device__ __forceinline__ unsigned col_part(const int t) { return (unsigned)t << 9; }
// the Forward matrix itself (the five logarithms, the emission plane) keeps the matrix layout
__device__ __forceinline__ unsigned m_col_part(const int t) { return ((unsigned)(t >> 1) << 10) + ((unsigned)(t & 1) << 3); }

struct ColRec {            // what a step needs of its column: FwdPack's integer half and the LinPack
  int s0, s1, s2, in_b, meta, env, cls;
  double w0, w1, w2, rs_m, ins_m;
  int rs_e, ins_e;
};

__device__ __forceinline__ ColRec load_col(const HX_GLOBAL FwdPack* fp, const HX_GLOBAL LinPack* lp, const int c) {
  const HX_GLOBAL d2v* q = (const HX_GLOBAL d2v*)(fp + c);
  const d2v c2 = q[2], c3 = q[3];
  const HX_GLOBAL d2v* l = (const HX_GLOBAL d2v*)(lp + c);
  const d2v l0 = l[0], l1 = l[1], l2 = l[2];
  ColRec r;
  r.s0 = __double2loint(c2.x); r.s1 = __double2hiint(c2.x);
  r.in_b = __double2loint(c2.y); r.meta = __double2hiint(c2.y);
  r.env = __double2loint(c3.x); r.cls = __double2hiint(c3.x);
  r.s2 = __double2loint(c3.y);
  r.w0 = l0.x; r.w1 = l0.y; r.w2 = l1.x; r.rs_m = l1.y; r.ins_m = l2.x;
  r.rs_e = __double2loint(l2.y); r.ins_e = __double2hiint(l2.y);
  return r;
}

// ---- per-state constants in linear form: one workgroup per job, a thread per state of either profile ----------------
__device__ __forceinline__ void split_log(const double lp, double& m, int& e) {
  if (!(lp > -1e8)) { m = 0.; e = HXD_EMIN; return; }
  const double n = __builtin_rint(lp * 1.44269504088896340736);
  m = exp((lp - n * 0.693147180369123816490) - n * 1.90821492927058770002e-10);     // ln2 in two parts
  e = (int)n;
}

__global__ void k_lin_pack(const DevJob* __restrict__ jobs) {
  const DevJob& J = jobs[blockIdx.x];
  const int nx = J.x.n, ny = J.y.n;
  LinPack* out = reinterpret_cast<LinPack*>(J.agg + (int64_t)(V_PLANES + 1) * J.plane);
  // exp(lpTrans) of every in-transition, in CSR order (read for the transitions beyond the three inline ones)
  double* wx = J.agg + (int64_t)(V_PLANES + 1) * J.plane + 6 * (int64_t)(nx + ny);
  const int tx = J.x.in_off[nx], ty = J.y.in_off[ny];
  for (int k = threadIdx.x; k < tx + ty; k += blockDim.x) wx[k] = exp(k < tx ? J.x.in_lp[k] : J.y.in_lp[k - tx]);
  for (int k = threadIdx.x; k < nx + ny; k += blockDim.x) {
    const FwdPack& f = k < nx ? J.x.fpack[k] : J.y.fpack[k - nx];
    LinPack p;
    p.w0 = exp(f.lp0); p.w1 = exp(f.lp1); p.w2 = exp(f.lp2);
    split_log(f.rootsub, p.rs_m, p.rs_e);
    split_log(f.ins, p.ins_m, p.ins_e);
    out[k] = p;
  }
}

// zero cells for the planes of banded jobs (cells outside the envelope are read but never written): mantissas 0,
// exponents the exponent of zero.  grid (jobs, slices)
__global__ void k_lin_clear(const DevJob* __restrict__ jobs) {
  const DevJob& J = jobs[blockIdx.x];
  double* lin = J.agg;
  int* ex = reinterpret_cast<int*>(J.agg + (int64_t)V_PLANES * J.plane);
  const int64_t nd = (int64_t)V_PLANES * J.plane, ne = J.plane;
  const int64_t stride = (int64_t)gridDim.y * blockDim.x, first = (int64_t)blockIdx.y * blockDim.x + threadIdx.x;
  for (int64_t k = first; k < nd; k += stride) lin[k] = 0.;
  for (int64_t k = first; k < ne; k += stride) ex[k] = HXD_EMIN;
}

template <bool MULTI>
__global__ void __launch_bounds__(HXD_MAX_WAVES * 64) k_forward_dag_linear(const DevJob* __restrict__ jobs, const double* __restrict__ exact_tab,
                                                                           const double* __restrict__ log_tab, const int groups, int* const counters,
                                                                           const int patience) {
  __shared__ volatile int prog[HXD_MAX_WAVES];
  __shared__ __attribute__((aligned(16))) double ltab_s[HXD_LOG_ENTRIES * 2];
  __shared__ double etab_s[HXD_EXP_ENTRIES];
  // per wave: the column records (FwdPack's integer half + LinPack = five 16-byte quarters) of the 128 columns around the
  // wave's position, refilled 64 columns at a time with one coalesced read: a step's record is five LDS reads instead of
  // five vector loads of 64 different cache lines each (the fill is bound by the CU's vector-memory pipeline)
  __shared__ d2v ycols[HXD_MAX_WAVES][5][128];
  const int threads = blockDim.x, W = threads >> 6;
  for (int k = threadIdx.x; k < HXD_LOG_ENTRIES * 2; k += threads) ltab_s[k] = log_tab[k];
  for (int k = threadIdx.x; k < HXD_EXP_ENTRIES; k += threads) etab_s[k] = exp2((double)k * (1.0 / HXD_EXP_ENTRIES));
  if (threadIdx.x < HXD_MAX_WAVES) prog[threadIdx.x] = 0;
  __syncthreads();
  const HX_LDS double* ltab = (const HX_LDS double*)ltab_s;
  const HX_LDS double* etab2 = (const HX_LDS double*)etab_s;
  volatile HX_LDS int* progp = (volatile HX_LDS int*)prog;

  const int G = MULTI ? groups : 1;
  const int job = MULTI ? (int)blockIdx.x / G : (int)blockIdx.x, grp = MULTI ? (int)blockIdx.x % G : 0;
  const DevJob& J = jobs[job];
  HX_GLOBAL int* gprog = MULTI ? (HX_GLOBAL int*)as_global(counters + 256 * job) : nullptr;
  const int R = J.n_rows, Cc = J.n_cols;
  const unsigned planeB = (unsigned)(J.plane * 8), ssB = (unsigned)(J.strip_stride * 8);
  HX_GLOBAL double* M = as_global(J.fwd);
  HX_GLOBAL double* LIN = as_global(J.agg);
  HX_GLOBAL int* EX = (HX_GLOBAL int*)as_global(J.agg + (int64_t)V_PLANES * J.plane);
  const HX_GLOBAL LinPack* xlp = (const HX_GLOBAL LinPack*)as_global(J.agg + (int64_t)(V_PLANES + 1) * J.plane);
  const HX_GLOBAL LinPack* ylp = xlp + J.x.n;
  const HX_GLOBAL double* xin_w = (const HX_GLOBAL double*)(ylp + J.y.n);        // exp(in_lp), CSR order
  const HX_GLOBAL double* yin_w = xin_w + J.x.in_off[J.x.n];
  const HX_GLOBAL double* etab = as_global((const double*)J.emis);
  const HX_GLOBAL double* eplane = as_global((const double*)J.emis_plane);
  const HX_GLOBAL FwdPack* xpk = as_global((const FwdPack*)J.x.fpack);
  const HX_GLOBAL FwdPack* ypk = as_global((const FwdPack*)J.y.fpack);
  const HX_GLOBAL int32_t* xin_src = as_global(J.x.in_src);
  const HX_GLOBAL int32_t* yin_src = as_global(J.y.in_src);
  const int Ky = J.y.n_cls;
  const bool xempty = J.x.empty, yempty = J.y.empty;
  const int max_dist = J.max_dist;
  const bool banded = max_dist >= 0;
  const HX_GLOBAL int32_t* win = as_global(J.fwd_windows);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_strips = (R + 63) >> 6;
  const int WT = W * G, gw = grp * W + wave;      // the pair's waves, and this one among them
  const int prev_wave = (gw + WT - 1) % WT;
  bool dead = false;                              // MULTI: a poll ran out of patience - run out without computing
  const auto read_progress = [&](const int of) -> int {
    if (MULTI) return __hip_atomic_load(gprog + of, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return progp[of];
  };
  const auto publish = [&](const int value) {
    if (lane != 0) return;
    // (a wave that gave up publishes the poison value: the waves below give up as well, down to the one that reports lpEnd)
    if (MULTI) __hip_atomic_store(gprog + gw, dead ? HX_MULTI_POISON : value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else progp[wave] = value;
  };
  // transition probabilities of the pair HMM: in LDS, read as broadcasts where the outgoing sums are formed (36 scalar
  // registers on top of the plane bases and the control flow's saved masks made the compiler spill ~100 scalars per step)
  __shared__ double Psh[5][5];
  if (threadIdx.x < 25) Psh[threadIdx.x / 5][threadIdx.x % 5] = exp(J.T[threadIdx.x / 5][threadIdx.x % 5]);
  __syncthreads();
  const HX_LDS double* PL = (const HX_LDS double*)&Psh[0][0];
#define P01 PL[0 * 5 + 1]
#define P11 PL[1 * 5 + 1]
#define P21 PL[2 * 5 + 1]
#define P31 PL[3 * 5 + 1]
#define P04 PL[0 * 5 + 4]
#define P34 PL[3 * 5 + 4]
#define P44 PL[4 * 5 + 4]
#define P02 PL[0 * 5 + 2]
#define P12 PL[1 * 5 + 2]
#define P22 PL[2 * 5 + 2]
#define P42 PL[4 * 5 + 2]
#define P03 PL[0 * 5 + 3]
#define P33 PL[3 * 5 + 3]
#define P00 PL[0 * 5 + 0]
#define P10 PL[1 * 5 + 0]
#define P20 PL[2 * 5 + 0]
#define P30 PL[3 * 5 + 0]
#define P40 PL[4 * 5 + 0]

  HX_LDS d2v* ring = (HX_LDS d2v*)&ycols[wave][0][0];
  auto stage = [&](const int c0) {       // columns c0 .. c0+63 (clamped into the profile) -> ring
    int c = c0 + lane;
    c = c < 0 ? 0 : (c >= Cc ? Cc - 1 : c);
    const HX_GLOBAL d2v* q = (const HX_GLOBAL d2v*)(ypk + c);
    const HX_GLOBAL d2v* l = (const HX_GLOBAL d2v*)(ylp + c);
    const d2v q2 = q[2], q3 = q[3], l0 = l[0], l1 = l[1], l2 = l[2];
    const int k = (c0 + lane) & 127;
    ring[k] = q2; ring[128 + k] = q3; ring[256 + k] = l0; ring[384 + k] = l1; ring[512 + k] = l2;
  };
  auto column = [&](const int j) -> ColRec {
    const int k = j & 127;
    const d2v c2 = ring[k], c3 = ring[128 + k], l0 = ring[256 + k], l1 = ring[384 + k], l2 = ring[512 + k];
    ColRec r;
    r.s0 = __double2loint(c2.x); r.s1 = __double2hiint(c2.x);
    r.in_b = __double2loint(c2.y); r.meta = __double2hiint(c2.y);
    r.env = __double2loint(c3.x); r.cls = __double2hiint(c3.x);
    r.s2 = __double2loint(c3.y);
    r.w0 = l0.x; r.w1 = l0.y; r.w2 = l1.x; r.rs_m = l1.y; r.ins_m = l2.x;
    r.rs_e = __double2loint(l2.y); r.ins_e = __double2hiint(l2.y);
    return r;
  };

  for (int s = gw; s < n_strips; s += WT) {
    const int i = (s << 6) + lane;
    const bool rvalid = i < R;
    const int ir = rvalid ? i : 0;
    const ColRec X = load_col(xpk, xlp, ir);
    const int xf = X.meta & 0xff, xdeg = X.meta >> 8;
    const bool xnull = xf & F_NULL, xok = (xf & F_READY) || xempty, xeos = xf & F_EMIT_OR_START;
    const unsigned ownB = (unsigned)s * ssB + ((unsigned)lane << 3), ownB_m = (unsigned)s * ssB + ((unsigned)lane << 4);
    // the rows of the first three in-transitions: byte offset of the row inside a plane, and the row's lane
    const unsigned xrB0 = (unsigned)(X.s0 >> 6) * ssB + ((unsigned)(X.s0 & 63) << 3);
    const unsigned xrB1 = (unsigned)(X.s1 >> 6) * ssB + ((unsigned)(X.s1 & 63) << 3);
    const unsigned xrB2 = (unsigned)(X.s2 >> 6) * ssB + ((unsigned)(X.s2 & 63) << 3);
    const int xl0 = X.s0 & 63, xl1 = X.s1 & 63, xl2 = X.s2 & 63;
    const unsigned vXa = (xnull ? V_IMD : V_G0) * planeB, vXb = (xnull ? V_IIW : V_G1) * planeB;
    // a source in the row directly above, inside this strip, is the previous lane's cell of the previous step
    const bool adjx0 = lane > 0 && xdeg > 0 && X.s0 == i - 1, adjx1 = lane > 0 && xdeg > 1 && X.s1 == i - 1, adjx2 = lane > 0 && xdeg > 2 && X.s2 == i - 1;
    // in-transitions 3 .. 7 of the row (rare: ~1 % of the states have more than three): kept for the whole strip, so that
    // a step fetches their cells in one batch
    unsigned xrx[HXD_EXTRA];
    int xlx[HXD_EXTRA];
    double xwx[HXD_EXTRA];
#pragma unroll
    for (int q = 0; q < HXD_EXTRA; ++q) {
      const int a = HX_DAG_INLINE + q;
      const int src = xdeg > a ? xin_src[X.in_b + a] : 0;
      xrx[q] = (unsigned)(src >> 6) * ssB + ((unsigned)(src & 63) << 3);
      xlx[q] = (src & 63) | ((lane > 0 && xdeg > a && src == i - 1) ? 64 : 0);     // bit 6: the row directly above
      xwx[q] = xdeg > a ? xin_w[X.in_b + a] : 0.;
    }
    const int above_base = ((s - 1) / WT) * Cc;
    const int my_base = (s / WT) * Cc;
    int seen = 0, published = 0;
#ifdef HX_DAG_TRACE
    long long tr_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tr_last = (long long)__builtin_readcyclecounter();
    int tr_steps = 0;
#endif
    int wlo[2] = {0, 0}, whi[2] = {Cc + 63, 0};
    if (banded) {
      wlo[0] = win[4 * s]; whi[0] = win[4 * s + 1];
      wlo[1] = win[4 * s + 2]; whi[1] = win[4 * s + 3];
    }
    for (int w = 0; w < 2; ++w) {
      if (whi[w] <= wlo[w]) continue;
      stage(wlo[w] - 64);
      stage(wlo[w]);
      PrevCell pv;
#pragma unroll
      for (int v = 0; v < V_PLANES; ++v) pv.v[v] = 0.;
      pv.E = HXD_EMIN;
      pv.slot = ownB + col_part(wlo[w]);      // (the first step's store goes to the slot that step's own cell overwrites one step later)
      double up_imm = 0., up_imd = 0., up_iiw = 0., up_g0 = 0., up_g1 = 0.;
      int up_E = HXD_EMIN;
      for (int t = wlo[w]; t < whi[w]; ++t) {
        if (s > 0) {
          const int need = above_base + (t + 1 < Cc ? t + 1 : Cc);
          if (seen < need && !dead) {
            int polls = 0;
            do {
              seen = __builtin_amdgcn_readfirstlane(read_progress(prev_wave));
              if (seen < need) {
                if (MULTI && ++polls > patience) { dead = true; break; }
                __builtin_amdgcn_s_sleep(1);
              }
            } while (seen < need);
            if (MULTI && seen == HX_MULTI_POISON) dead = true;      // the wave above (or one above it) gave up
            asm volatile("" ::: "memory");
          }
        }
        HXD_TR(0);     // progress wait
        if (t > wlo[w] && ((t - wlo[w]) & 63) == 0) stage(t);
        const int j = t - lane;
        const ColRec Y = column(j);
        const int yf = Y.meta & 0xff, ydeg = Y.meta >> 8;
        bool act = rvalid && j >= 0 && j < Cc && !dead;
        if (banded) {
          int dd = X.env - Y.env;
          dd = dd < 0 ? -dd : dd;
          act = act && (((xf | yf) & F_EDGE) || dd <= max_dist);
        }
        const bool ynull = yf & F_NULL, yok = (yf & F_READY) || yempty;
        const int mode = (!xnull && !ynull) ? 1 : ((ynull && xeos) ? 2 : (yok ? 3 : 0));
        const bool xgo = act && yok, ygo = act && (ynull || xok);
        const int jc = j < 0 ? 0 : (j >= Cc ? Cc - 1 : j);
        const unsigned vYa = (ynull ? V_IDM : V_G2) * planeB, vYb = (ynull ? V_IMI : V_G3) * planeB;
        // slots (byte offsets inside a plane) of the source cells
        const unsigned sx0 = xrB0 + col_part(jc + xl0), sx1 = xrB1 + col_part(jc + xl1), sx2 = xrB2 + col_part(jc + xl2);
        const unsigned sy0 = ownB + col_part(Y.s0 + lane), sy1 = ownB + col_part(Y.s1 + lane), sy2 = ownB + col_part(Y.s2 + lane);
        const unsigned own_slot = ownB + col_part(t), own_slot_m = ownB_m + m_col_part(t);

        // ---- emission (log) of the cell ----
        double elog = HX_NEG_INF;
        if (act && mode == 1) {
          if (etab) elog = etab[(int64_t)(X.cls < 0 ? 0 : X.cls) * Ky + (Y.cls < 0 ? 0 : Y.cls)];
          else elog = ldg<MULTI>(eplane, own_slot_m);
          if (X.cls < 0 || Y.cls < 0) elog = HX_NEG_INF;
        }

        HXD_TR(1);     // column record, flags, addresses, emission load
        double m_imm = 0., m_imd = 0., m_idm = 0., m_imi = 0., m_iiw = 0.;
        int e_imm = HXD_EMIN, e_imd = HXD_EMIN, e_idm = HXD_EMIN, e_imi = HXD_EMIN, e_iiw = HXD_EMIN;
        // ---- this step's loads, all issued before the first use: the first three transitions of the row and of the
        // column, and the IMM sources that hang on them (pairs when both states emit, else the y or the x sources) ----
        const unsigned sxs[3] = {sx0, sx1, sx2}, sys_[3] = {sy0, sy1, sy2};
        const unsigned xr[3] = {xrB0, xrB1, xrB2};
        const int xl[3] = {xl0, xl1, xl2};
        const double wxs[3] = {X.w0, X.w1, X.w2}, wys[3] = {Y.w0, Y.w1, Y.w2};
        const int ys[3] = {Y.s0, Y.s1, Y.s2};
        double xa[3], xb[3], ya[3], yb[3], mv[9];
        int xe[3], ye[3], me[9];
        int ysx[HXD_EXTRA];
        double ywx[HXD_EXTRA];
#ifdef HX_DAG_TRACE
        {   // latency probe: one source load by itself (the first transition of the row), then an old cell of the same row
          volatile double probe = ldg<MULTI>(LIN, vXa + sx0);
          (void)probe;
          HXD_TR(7);
          const int told = t - 40 < wlo[w] ? wlo[w] : t - 40;
          volatile double probe2 = ldg<MULTI>(LIN, V_G2 * planeB + ownB + col_part(told));     // written 40 steps ago by this lane
          (void)probe2;
          HXD_TR(6);
          volatile double probe3 = ldg<MULTI>(LIN, V_G2 * planeB + ownB + col_part(told));     // the same line again
          (void)probe3;
          HXD_TR(4);
        }
#endif
        if (act) {
          // (no initial values: every element is read under the predicate it was loaded under)
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            if (xgo && xdeg > k) { xa[k] = ldg<MULTI>(LIN, vXa + sxs[k]); xb[k] = ldg<MULTI>(LIN, vXb + sxs[k]); xe[k] = ldgi<MULTI>(EX, sxs[k] >> 1); }
            if (ygo && ydeg > k) { ya[k] = ldg<MULTI>(LIN, vYa + sys_[k]); yb[k] = ldg<MULTI>(LIN, vYb + sys_[k]); ye[k] = ldgi<MULTI>(EX, sys_[k] >> 1); }
          }
          if (mode == 1) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
              for (int b = 0; b < 3; ++b)
                if (xdeg > a && ydeg > b) {
                  const unsigned sl = xr[a] + col_part(ys[b] + xl[a]);
                  mv[a * 3 + b] = ldg<MULTI>(LIN, V_G4 * planeB + sl);
                  me[a * 3 + b] = ldgi<MULTI>(EX, sl >> 1);
                }
          } else if (mode == 2) {
#pragma unroll
            for (int b = 0; b < 3; ++b)
              if (ydeg > b) { mv[b] = ldg<MULTI>(LIN, sys_[b]); me[b] = ygo ? ye[b] : ldgi<MULTI>(EX, sys_[b] >> 1); }
          } else if (mode == 3) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
              if (xdeg > a) { mv[a] = ldg<MULTI>(LIN, sxs[a]); me[a] = xgo ? xe[a] : ldgi<MULTI>(EX, sxs[a] >> 1); }
          }
          // the column's in-transitions 3 .. 7 (CSR), fetched with the batch above
#pragma unroll
          for (int q = 0; q < HXD_EXTRA; ++q) {
            ysx[q] = 0; ywx[q] = 0.;
            if (ydeg > HX_DAG_INLINE + q) { ysx[q] = yin_src[Y.in_b + HX_DAG_INLINE + q]; ywx[q] = yin_w[Y.in_b + HX_DAG_INLINE + q]; }
          }
        }
        // ---- the previous step's cell goes to memory now, behind this step's loads (every lane: an idle lane's cell is
        // all zeros, which is what padding and cells outside the envelope hold) ----
#pragma unroll
        for (int v = 0; v < V_PLANES; ++v) stg<MULTI>(LIN, v * planeB + pv.slot, pv.v[v]);
        stgi<MULTI>(EX, pv.slot >> 1, pv.E);
        if (act) {
          // ---- values still on their way to memory: the previous step's cells of this lane and of the lane above ----
          {
            const bool ax3[3] = {adjx0, adjx1, adjx2};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              if (ax3[k]) { xa[k] = xnull ? up_imd : up_g0; xb[k] = xnull ? up_iiw : up_g1; xe[k] = up_E; }
              const bool ady = ydeg > k && ys[k] == j - 1;
              if (ady) { ya[k] = ynull ? pv.v[V_IDM] : pv.v[V_G2]; yb[k] = ynull ? pv.v[V_IMI] : pv.v[V_G3]; ye[k] = pv.E; }
              if (mode == 2 && ady) { mv[k] = pv.v[V_IMM]; me[k] = pv.E; }
              if (mode == 3 && ax3[k]) { mv[k] = up_imm; me[k] = up_E; }
            }
          }
          HXD_TR(5);     // (trace builds: the first batch of loads has landed)
          Acc2 ax = Acc2{0., 0., HXD_EMIN}, ay = Acc2{0., 0., HXD_EMIN};
          Acc1 am = Acc1{0., HXD_EMIN};                       // IMM
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            if (xgo && xdeg > k) add2(ax, xa[k], xb[k], xe[k], wxs[k]);      // x-absorbing (or x-null) moves: IMD, IIW
            if (ygo && ydeg > k) add2(ay, ya[k], yb[k], ye[k], wys[k]);      // y-absorbing (or y-null) moves: IDM, IMI
          }
          if (mode == 1) {
            // transition pairs (reference src/forward.cpp:98-116): the source's outgoing sum into IMM
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
              for (int b = 0; b < 3; ++b)
                if (xdeg > a && ydeg > b) add1(am, mv[a * 3 + b], me[a * 3 + b], wxs[a] * wys[b]);
          } else if (mode == 2) {
#pragma unroll
            for (int b = 0; b < 3; ++b)
              if (ydeg > b) add1(am, mv[b], me[b], wys[b]);
          } else if (mode == 3) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
              if (xdeg > a) add1(am, mv[a], me[a], wxs[a]);
          }
          HXD_TR(6);     // (the inline transitions are summed)
          // ---- second batch: the cells of in-transitions 3 .. 7 of the row and of the column, all loads first ----
          if (xdeg > HX_DAG_INLINE) {
            // the row's further transitions: their cells at this column, and their pairs with the column's inline transitions
            double xa2[HXD_EXTRA], xb2[HXD_EXTRA], mx2[HXD_EXTRA * 3];
            int xe2[HXD_EXTRA], mxe[HXD_EXTRA * 3];
#pragma unroll
            for (int q = 0; q < HXD_EXTRA; ++q) {
              xa2[q] = xb2[q] = 0.; xe2[q] = HXD_EMIN;
#pragma unroll
              for (int k = 0; k < 3; ++k) { mx2[q * 3 + k] = 0.; mxe[q * 3 + k] = HXD_EMIN; }
              const bool hx = xdeg > HX_DAG_INLINE + q;
              const unsigned slx = xrx[q] + col_part(jc + (xlx[q] & 63));
              if (hx && (xgo || mode == 3)) xe2[q] = ldgi<MULTI>(EX, slx >> 1);
              if (hx && xgo) { xa2[q] = ldg<MULTI>(LIN, vXa + slx); xb2[q] = ldg<MULTI>(LIN, vXb + slx); }
              if (hx && mode == 3) mx2[q * 3] = ldg<MULTI>(LIN, slx);
              if (mode == 1) {
#pragma unroll
                for (int k = 0; k < 3; ++k)
                  if (hx && ydeg > k) {                         // pair (row transition 3+q, column transition k)
                    const unsigned sl = xrx[q] + col_part(ys[k] + (xlx[q] & 63));
                    mx2[q * 3 + k] = ldg<MULTI>(LIN, V_G4 * planeB + sl); mxe[q * 3 + k] = ldgi<MULTI>(EX, sl >> 1);
                  }
              }
            }
#pragma unroll
            for (int q = 0; q < HXD_EXTRA; ++q) {
              const bool hx = xdeg > HX_DAG_INLINE + q;
              if (hx && (xlx[q] & 64)) {                        // the row directly above: still in registers
                xa2[q] = xnull ? up_imd : up_g0; xb2[q] = xnull ? up_iiw : up_g1; xe2[q] = up_E;
                if (mode == 3) mx2[q * 3] = up_imm;
              }
              if (hx && xgo) add2(ax, xa2[q], xb2[q], xe2[q], xwx[q]);
              if (hx && mode == 3) add1(am, mx2[q * 3], xe2[q], xwx[q]);
              if (mode == 1) {
#pragma unroll
                for (int k = 0; k < 3; ++k)
                  if (hx && ydeg > k) add1(am, mx2[q * 3 + k], mxe[q * 3 + k], xwx[q] * wys[k]);
              }
            }
          }
          if (ydeg > HX_DAG_INLINE) {
            // the column's further transitions: their cells in this row, and their pairs with the row's inline transitions
            double ya2[HXD_EXTRA], yb2[HXD_EXTRA], my2[HXD_EXTRA * 3];
            int ye2[HXD_EXTRA], mye[HXD_EXTRA * 3];
#pragma unroll
            for (int q = 0; q < HXD_EXTRA; ++q) {
              ya2[q] = yb2[q] = 0.; ye2[q] = HXD_EMIN;
#pragma unroll
              for (int k = 0; k < 3; ++k) { my2[q * 3 + k] = 0.; mye[q * 3 + k] = HXD_EMIN; }
              const bool hy = ydeg > HX_DAG_INLINE + q;
              const unsigned sly = ownB + col_part(ysx[q] + lane);
              if (hy && (ygo || mode == 2)) ye2[q] = ldgi<MULTI>(EX, sly >> 1);
              if (hy && ygo) { ya2[q] = ldg<MULTI>(LIN, vYa + sly); yb2[q] = ldg<MULTI>(LIN, vYb + sly); }
              if (hy && mode == 2) my2[q * 3] = ldg<MULTI>(LIN, sly);
              if (mode == 1) {
#pragma unroll
                for (int k = 0; k < 3; ++k)
                  if (hy && xdeg > k) {                         // pair (row transition k, column transition 3+q)
                    const unsigned sl = xr[k] + col_part(ysx[q] + xl[k]);
                    my2[q * 3 + k] = ldg<MULTI>(LIN, V_G4 * planeB + sl); mye[q * 3 + k] = ldgi<MULTI>(EX, sl >> 1);
                  }
              }
            }
#pragma unroll
            for (int q = 0; q < HXD_EXTRA; ++q) {
              const bool hy = ydeg > HX_DAG_INLINE + q;
              if (hy && ysx[q] == j - 1) {                      // the lane's own previous cell
                ya2[q] = ynull ? pv.v[V_IDM] : pv.v[V_G2]; yb2[q] = ynull ? pv.v[V_IMI] : pv.v[V_G3]; ye2[q] = pv.E;
                if (mode == 2) my2[q * 3] = pv.v[V_IMM];
              }
              if (hy && ygo) add2(ay, ya2[q], yb2[q], ye2[q], ywx[q]);
              if (hy && mode == 2) add1(am, my2[q * 3], ye2[q], ywx[q]);
              if (mode == 1) {
#pragma unroll
                for (int k = 0; k < 3; ++k)
                  if (hy && xdeg > k) add1(am, my2[q * 3 + k], mye[q * 3 + k], wxs[k] * ywx[q]);
              }
            }
          }
          if (xdeg > HX_DAG_INLINE || ydeg > HX_DAG_INLINE) {
            // ---- what is left: pairs of two transitions beyond the inline ones, transitions beyond the eighth ----
            if (mode == 1 && xdeg > HX_DAG_INLINE && ydeg > HX_DAG_INLINE)
              for (int a = HX_DAG_INLINE; a < xdeg; ++a) {
                const int srcx = xin_src[X.in_b + a];
                const double wxa = xin_w[X.in_b + a];
                const unsigned rb = (unsigned)(srcx >> 6) * ssB + ((unsigned)(srcx & 63) << 3);
                for (int b = HX_DAG_INLINE; b < ydeg; ++b) {
                  const unsigned sl = rb + col_part(yin_src[Y.in_b + b] + (srcx & 63));
                  add1(am, ldg<MULTI>(LIN, V_G4 * planeB + sl), ldgi<MULTI>(EX, sl >> 1), wxa * yin_w[Y.in_b + b]);
                }
              }
            for (int a = HX_DAG_INLINE + HXD_EXTRA; a < xdeg; ++a) {
              const int src = xin_src[X.in_b + a];
              const double wa = xin_w[X.in_b + a];
              const unsigned rb = (unsigned)(src >> 6) * ssB + ((unsigned)(src & 63) << 3);
              const unsigned sl = rb + col_part(jc + (src & 63));
              const bool adj = lane > 0 && src == i - 1;
              if (xgo) {
                if (adj) add2(ax, xnull ? up_imd : up_g0, xnull ? up_iiw : up_g1, up_E, wa);
                else add2(ax, ldg<MULTI>(LIN, vXa + sl), ldg<MULTI>(LIN, vXb + sl), ldgi<MULTI>(EX, sl >> 1), wa);
              }
              if (mode == 3) {
                if (adj) add1(am, up_imm, up_E, wa);
                else add1(am, ldg<MULTI>(LIN, sl), ldgi<MULTI>(EX, sl >> 1), wa);
              }
              if (mode == 1)
                for (int b = 0; b < (ydeg < HX_DAG_INLINE ? ydeg : HX_DAG_INLINE); ++b) {
                  const unsigned sp = rb + col_part((b == 0 ? Y.s0 : (b == 1 ? Y.s1 : Y.s2)) + (src & 63));
                  add1(am, ldg<MULTI>(LIN, V_G4 * planeB + sp), ldgi<MULTI>(EX, sp >> 1), wa * (b == 0 ? Y.w0 : (b == 1 ? Y.w1 : Y.w2)));
                }
            }
            for (int b = HX_DAG_INLINE + HXD_EXTRA; b < ydeg; ++b) {
              const int src = yin_src[Y.in_b + b];
              const double wb = yin_w[Y.in_b + b];
              const unsigned sl = ownB + col_part(src + lane);
              const bool adj = src == j - 1;
              if (ygo) {
                if (adj) add2(ay, ynull ? pv.v[V_IDM] : pv.v[V_G2], ynull ? pv.v[V_IMI] : pv.v[V_G3], pv.E, wb);
                else add2(ay, ldg<MULTI>(LIN, vYa + sl), ldg<MULTI>(LIN, vYb + sl), ldgi<MULTI>(EX, sl >> 1), wb);
              }
              if (mode == 2) {
                if (adj) add1(am, pv.v[V_IMM], pv.E, wb);
                else add1(am, ldg<MULTI>(LIN, sl), ldgi<MULTI>(EX, sl >> 1), wb);
              }
              if (mode == 1)
                for (int a = 0; a < (xdeg < HX_DAG_INLINE ? xdeg : HX_DAG_INLINE); ++a) {
                  const unsigned sp = (a == 0 ? xrB0 : (a == 1 ? xrB1 : xrB2)) + col_part(src + (a == 0 ? xl0 : (a == 1 ? xl1 : xl2)));
                  add1(am, ldg<MULTI>(LIN, V_G4 * planeB + sp), ldgi<MULTI>(EX, sp >> 1), (a == 0 ? X.w0 : (a == 1 ? X.w1 : X.w2)) * wb);
                }
            }
          }
          // ---- absorption / emission factors ----
          m_imd = ax.a; m_iiw = ax.b; e_imd = e_iiw = ax.E;
          if (!xnull && yok) {
            m_imd *= X.rs_m; e_imd = ax.E + X.rs_e;
            m_iiw *= X.ins_m; e_iiw = ax.E + X.ins_e;
          }
          m_idm = ay.a; m_imi = ay.b; e_idm = e_imi = ay.E;
          if (!ynull && xok) {
            m_idm *= Y.rs_m; e_idm = ay.E + Y.rs_e;
            m_imi *= Y.ins_m; e_imi = ay.E + Y.ins_e;
          }
          m_imm = am.s; e_imm = am.E;
          if (mode == 1) {
            double em; int ee;
            exp_split(elog, etab2, em, ee);
            m_imm *= em; e_imm = am.E + ee;
          }
          if (s == 0 && t == 0 && lane == 0) { m_imm = 1.; e_imm = 0; }    // cell (0,0): lpStart() = 0 (reference src/forward.cpp:73)
        }
        HXD_TR(2);     // source loads and accumulation
        // ---- the reference's format (five logarithms, stored at once) and the kernel's: a common exponent (that of the
        // largest state), the five outgoing sums - kept in registers until the next step has issued its loads ----
        stg<MULTI>(M, own_slot_m, log_scaled(m_imm, e_imm, ltab));                  // (nothing in this kernel reads these back)
        stg<MULTI>(M, planeB + own_slot_m, log_scaled(m_imd, e_imd, ltab));
        stg<MULTI>(M, 2 * planeB + own_slot_m, log_scaled(m_idm, e_idm, ltab));
        stg<MULTI>(M, 3 * planeB + own_slot_m, log_scaled(m_imi, e_imi, ltab));
        stg<MULTI>(M, 4 * planeB + own_slot_m, log_scaled(m_iiw, e_iiw, ltab));
        {
          const int b0 = m_imm > 0. ? e_imm + __builtin_amdgcn_frexp_exp(m_imm) : HXD_EMIN;
          const int b1 = m_imd > 0. ? e_imd + __builtin_amdgcn_frexp_exp(m_imd) : HXD_EMIN;
          const int b2 = m_idm > 0. ? e_idm + __builtin_amdgcn_frexp_exp(m_idm) : HXD_EMIN;
          const int b3 = m_imi > 0. ? e_imi + __builtin_amdgcn_frexp_exp(m_imi) : HXD_EMIN;
          const int b4 = m_iiw > 0. ? e_iiw + __builtin_amdgcn_frexp_exp(m_iiw) : HXD_EMIN;
          int E = b0 > b1 ? b0 : b1;
          E = E > b2 ? E : b2;
          E = E > b3 ? E : b3;
          E = E > b4 ? E : b4;
          const double imm = ldexp_fast(m_imm, (m_imm > 0. ? e_imm : HXD_EMIN) - E), imd = ldexp_fast(m_imd, (m_imd > 0. ? e_imd : HXD_EMIN) - E);
          const double idm = ldexp_fast(m_idm, (m_idm > 0. ? e_idm : HXD_EMIN) - E), imi = ldexp_fast(m_imi, (m_imi > 0. ? e_imi : HXD_EMIN) - E);
          const double iiw = ldexp_fast(m_iiw, (m_iiw > 0. ? e_iiw : HXD_EMIN) - E);
          pv.v[V_IMM] = imm; pv.v[V_IMD] = imd; pv.v[V_IDM] = idm; pv.v[V_IMI] = imi; pv.v[V_IIW] = iiw;
          pv.v[V_G0] = __builtin_fma(imi, P31, __builtin_fma(idm, P21, __builtin_fma(imd, P11, imm * P01)));
          pv.v[V_G1] = __builtin_fma(iiw, P44, __builtin_fma(imi, P34, imm * P04));
          pv.v[V_G2] = __builtin_fma(iiw, P42, __builtin_fma(idm, P22, __builtin_fma(imd, P12, imm * P02)));
          pv.v[V_G3] = __builtin_fma(imi, P33, imm * P03);
          pv.v[V_G4] = __builtin_fma(iiw, P40, __builtin_fma(imi, P30, __builtin_fma(idm, P20, __builtin_fma(imd, P10, imm * P00))));
          pv.E = E;
          pv.slot = own_slot;
          up_imm = wave_shr1(imm); up_imd = wave_shr1(imd); up_iiw = wave_shr1(iiw);
          up_g0 = wave_shr1(pv.v[V_G0]); up_g1 = wave_shr1(pv.v[V_G1]);
          up_E = wave_shr1_int(E);
        }
        HXD_TR(3);     // logarithms, alignment, sums, stores
        // ---- publish, every 8th step: drain, then all stores issued so far (the cells of steps <= t - 1) are in memory,
        // i.e. all 64 rows have completed the columns up to t - 1 - 63
        if ((t & 7) == 7) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          int done = t - 63;
          done = (dead || done > Cc) ? Cc : done;
          if (done > published) {
            published = done;
            publish(my_base + done);
          }
        }
        HXD_TR(4);     // publish
#ifdef HX_DAG_TRACE
        ++tr_steps;
#endif
      }
      // flush the last cell of the window
#pragma unroll
      for (int v = 0; v < V_PLANES; ++v) stg<MULTI>(LIN, v * planeB + pv.slot, pv.v[v]);
      stgi<MULTI>(EX, pv.slot >> 1, pv.E);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // columns between / after the windows hold no in-envelope cell of this strip
      const int upto = (w == 0 && whi[1] > wlo[1]) ? wlo[1] - 63 : Cc;
      const int done = (dead || upto > Cc) ? Cc : upto;
      if (done > published) {
        published = done;
        publish(my_base + done);
      }
    }
#ifdef HX_DAG_TRACE
    if (lane == 0 && tr_steps > 0)
      printf("trace job %d strip %d steps %d wait %lld setup %lld accumulate %lld finish %lld publish %lld loads1 %lld sums1 %lld\n", (int)blockIdx.x, s, tr_steps,
             tr_sum[0] / tr_steps, tr_sum[1] / tr_steps, tr_sum[2] / tr_steps, tr_sum[3] / tr_steps, tr_sum[4] / tr_steps, tr_sum[5] / tr_steps, tr_sum[6] / tr_steps);
    if (lane == 0 && tr_steps > 0) printf("probe job %d strip %d one load %lld old line %lld\n", (int)blockIdx.x, s, tr_sum[7] / tr_steps, tr_sum[6] / tr_steps);
#endif
    // (a strip without any window still has to release the strip below)
    if (published < Cc) {
      published = Cc;
      publish(my_base + Cc);
    }
    if (MULTI && s == n_strips - 1) {
      // the last strip finishes last (every strip follows the one above): all of the pair's cells are in memory.  What END
      // reads may lie in other workgroups' strips: drop this CU's L1 first (agent-scope acquire), as k_forward_dag_pipe does
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) *J.lp_end = dead ? __builtin_nan("") : forward_lp_end(J, ExactLse{exact_tab});
    }
  }
  if (MULTI) return;
  __syncthreads();
  if (threadIdx.x == 0) *J.lp_end = forward_lp_end(J, ExactLse{exact_tab});
}

}  // namespace

// Doubles of scratch (DevJob::agg) a job of this fill needs: ten mantissa planes, the exponent plane, the LinPacks.
int64_t dag_linear_scratch_doubles(int64_t plane, int nx, int ny, int tx, int ty) {
  return (int64_t)(V_PLANES + 1) * plane + 6 * (int64_t)(nx + ny) + (((int64_t)tx + ty + 3) & ~(int64_t)1);
}

// byte offsets inside a job's planes are 32-bit
bool dag_linear_fits(int64_t plane) { return (V_PLANES + 1) * plane * 8 < ((int64_t)1 << 32); }

// multi > 1: one or two pairs of many strips, each dealt to `multi` workgroups of `multi_waves` waves (MULTI instantiation);
// `counters`: 256 zeroed ints per pair
int launch_forward_dag_linear(const DevJob* d_jobs, int n_jobs, int max_rows, Tab8 tab8, Tab16 log_tab, int multi, int multi_waves,
                              int* counters, hipStream_t st) {
  int w = (max_rows + 63) / 64;
  w = w < 1 ? 1 : (w > HXD_MAX_WAVES ? HXD_MAX_WAVES : w);
  hipLaunchKernelGGL(k_lin_pack, dim3(n_jobs), dim3(256), 0, st, d_jobs);
  if (multi > 1) {
    if (multi_waves > HXD_MAX_WAVES) multi_waves = HXD_MAX_WAVES;
    HX_CHECK_LDS(k_forward_dag_linear<true>, 0, "k_forward_dag_linear<multi>");
    hipLaunchKernelGGL(k_forward_dag_linear<true>, dim3(n_jobs * multi), dim3(multi_waves * 64), 0, st, d_jobs, tab8.p, log_tab.p, multi, counters, multi_patience());
    return 0;
  }
  HX_CHECK_LDS(k_forward_dag_linear<false>, 0, "k_forward_dag_linear");
  hipLaunchKernelGGL(k_forward_dag_linear<false>, dim3(n_jobs), dim3(w * 64), 0, st, d_jobs, tab8.p, log_tab.p, 1, nullptr, 0);
  return 0;
}

void launch_dag_linear_clear(const DevJob* d_jobs, int n_jobs, hipStream_t st) {
  hipLaunchKernelGGL(k_lin_clear, dim3(n_jobs, 64), dim3(256), 0, st, d_jobs);
}

}  // namespace hx
