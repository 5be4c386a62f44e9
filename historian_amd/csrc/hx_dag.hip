// Strip-pipelined Forward / Backward fills for general profiles (state DAGs with null
// states, ready/wait flags, fan-in > 1 and an optional guide-alignment band):
//   reference src/forward.cpp:68-223 (ForwardMatrix::ForwardMatrix)
//   reference src/forward.cpp:975-1088 (BackwardMatrix::BackwardMatrix)
//
// Same decomposition as the chain kernels (hx_chain.hip): one workgroup per pair, its waves
// take 64-row strips round-robin, lane <-> row, step <-> anti-diagonal, every wave on its own
// clock.  What differs is where a cell's sources come from.  A profile state may have any
// number of in-transitions from any earlier state, so a source is not always "the lane above,
// one step ago": sources are read back from the matrix itself (L1/L2 hits: the workgroup's own
// recent stores), ordered by
//   * own strip:    vector-memory operations of a wave retire in issue order, so a load sees every
//                   store the wave issued before it;
//   * strips above: a wave publishes (monotonic LDS counter), after draining its stores, how many
//                   columns of its strip are complete; the wave below does not enter step t before
//                   columns 0..t of the strip above are published.  Strips further up completed
//                   those columns earlier still.
// All waves of a workgroup share one CU and one L1, so draining + the LDS counter is a
// workgroup-scope release/acquire; no cache maintenance is needed.
//
// k_fill_dag<1>        Backward: the recursion as the reference writes it, per-step drain.
// k_forward_dag_pipe   Forward, restructured for latency (see the comment above it): outgoing sums
//                      stored per cell, inline transition records, register forwarding of the
//                      previous step, stores issued behind the next step's loads.
//
// The emission term of an IMM cell (computeLogProbAbsorb, reference src/forward.h:112-124)
// does not depend on the DP values: it is evaluated for all cells up front by a fully parallel
// kernel into a sixth plane in the matrix layout (or taken from the class-pair table when the
// profiles have few distinct columns), so the dependent chain of a cell is the transition
// log-sum-exps only.
//
// With a band (GuideAlignmentEnvelope), a strip only visits the step windows in which it has
// in-envelope cells; everything else stays at the -inf the matrix was pre-filled with, which is
// what the reference's cell() returns for cells it never stored (src/forward.h:68-88).
//
// Compiled with -ffp-contract=off (see hx_lse.h).
#include <hip/hip_runtime.h>
#include "hx_device.h"
#include "hx_lse.h"
#include "hx_common.h"
#include "hx_policy.h"
#include "hx_kernels.h"

namespace hx {

namespace {

struct Side {
  const HX_GLOBAL uint8_t* flags;
  const HX_GLOBAL int32_t* in_off;
  const HX_GLOBAL int32_t* in_src;
  const HX_GLOBAL double* in_lp;
  const HX_GLOBAL int32_t* ao_off;
  const HX_GLOBAL int32_t* ao_dst;
  const HX_GLOBAL double* ao_lp;
  const HX_GLOBAL int32_t* no_off;
  const HX_GLOBAL int32_t* no_dst;
  const HX_GLOBAL double* no_lp;
  const HX_GLOBAL double* ins;
  const HX_GLOBAL double* rootsub;
  const HX_GLOBAL int32_t* env;
  const HX_GLOBAL int32_t* cls;
  int n, empty, n_cls;
};

__device__ __forceinline__ Side make_side(const DevProfile& P) {
  Side s;
  s.flags = as_global(P.flags);
  s.in_off = as_global(P.in_off); s.in_src = as_global(P.in_src); s.in_lp = as_global(P.in_lp);
  s.ao_off = as_global(P.ao_off); s.ao_dst = as_global(P.ao_dst); s.ao_lp = as_global(P.ao_lp);
  s.no_off = as_global(P.no_off); s.no_dst = as_global(P.no_dst); s.no_lp = as_global(P.no_lp);
  s.ins = as_global((const double*)P.ins); s.rootsub = as_global((const double*)P.rootsub);
  s.env = as_global(P.env); s.cls = as_global(P.cls);
  s.n = P.n; s.empty = P.empty; s.n_cls = P.n_cls;
  return s;
}

struct Mat {
  HX_GLOBAL double* M;                 // the matrix being filled
  const HX_GLOBAL double* etab;        // class-pair emission table, or null
  const HX_GLOBAL double* eplane;      // per-cell emission plane (Forward layout), or null
  int64_t plane, ss;
  int R, Cc, max_dist;
};

__device__ __forceinline__ C5 ld5(const HX_GLOBAL double* M, int64_t plane, int64_t slot) {
  return C5{M[slot], M[plane + slot], M[2 * plane + slot], M[3 * plane + slot], M[4 * plane + slot]};
}

__device__ __forceinline__ double emis_at(const Mat& m, const Side& x, const Side& y, int i, int j) {
  if (m.etab) {
    // a hand-edited profile may route an absorbing transition into a null state
    // (reference t/testnullforward.cpp:37-39); such a pair emits nothing
    const int cx = x.cls[i], cy = y.cls[j];
    return (cx < 0 || cy < 0) ? HX_NEG_INF : m.etab[(int64_t)cx * y.n_cls + cy];
  }
  return m.eplane[cell_slot(m.ss, i, j)];
}

__device__ __forceinline__ bool in_env(const Mat& m, uint8_t xf, uint8_t yf, int xe, int ye) {
  if ((xf | yf) & F_EDGE) return true;
  if (m.max_dist < 0) return true;
  int d = xe - ye;
  d = d < 0 ? -d : d;
  return d <= m.max_dist;
}

// one Forward cell (reference src/forward.cpp:78-203); the caller has checked the envelope
template <class LSE>
__device__ __forceinline__ C5 forward_cell_dag(const Mat& m, const Side& x, const Side& y, const double (*T)[6],
                                               const LSE& L, int i, int j, uint8_t xf, uint8_t yf) {
  const HX_GLOBAL double* M = m.M;
  const int64_t plane = m.plane, ss = m.ss;
  C5 r = c5_neg_inf();
  if (i == 0 && j == 0) r.imm = 0.0;
  const bool xnull = xf & F_NULL, ynull = yf & F_NULL;
  const bool yok = (yf & F_READY) || y.empty;   // yState.isReady() || yEmpty
  const bool xok = (xf & F_READY) || x.empty;
  const int xb = x.in_off[i], xe = x.in_off[i + 1];
  const int yb = y.in_off[j], ye = y.in_off[j + 1];

  if (!xnull) {
    if (yok) {
      for (int t = xb; t < xe; ++t) {
        const C5 s = ld5(M, plane, cell_slot(ss, x.in_src[t], j));
        const double lp = x.in_lp[t];
        double a = L(s.imm + T[0][1], s.imd + T[1][1]);
        double b = L(s.imm + T[0][4], s.imi + T[3][4]);
        a = L(a, s.idm + T[2][1]);
        b = L(b, s.iiw + T[4][4]);
        a = L(a, s.imi + T[3][1]);
        r.imd = L(r.imd, a + lp);
        r.iiw = L(r.iiw, b + lp);
      }
      r.imd += x.rootsub[i];
      r.iiw += x.ins[i];
    }
  } else if (yok) {
    for (int t = xb; t < xe; ++t) {
      const int64_t sl = cell_slot(ss, x.in_src[t], j);
      const double lp = x.in_lp[t];
      r.imd = L(r.imd, M[plane + sl] + lp);
      r.iiw = L(r.iiw, M[4 * plane + sl] + lp);
    }
  }

  if (!ynull) {
    if (xok) {
      for (int t = yb; t < ye; ++t) {
        const C5 s = ld5(M, plane, cell_slot(ss, i, y.in_src[t]));
        const double lp = y.in_lp[t];
        double a = L(s.imm + T[0][2], s.imd + T[1][2]);
        const double b = L(s.imm + T[0][3], s.imi + T[3][3]);
        a = L(a, s.idm + T[2][2]);
        a = L(a, s.iiw + T[4][2]);
        r.idm = L(r.idm, a + lp);
        r.imi = L(r.imi, b + lp);
      }
      r.idm += y.rootsub[j];
      r.imi += y.ins[j];
    }
  } else {
    for (int t = yb; t < ye; ++t) {
      const int64_t sl = cell_slot(ss, i, y.in_src[t]);
      const double lp = y.in_lp[t];
      r.idm = L(r.idm, M[2 * plane + sl] + lp);
      r.imi = L(r.imi, M[3 * plane + sl] + lp);
    }
  }

  if (!xnull && !ynull) {
    for (int tx = xb; tx < xe; ++tx) {
      const int sx = x.in_src[tx];
      const double lpx = x.in_lp[tx];
      for (int ty = yb; ty < ye; ++ty) {
        const C5 s = ld5(M, plane, cell_slot(ss, sx, y.in_src[ty]));
        double a = L(s.imm + T[0][0], s.imd + T[1][0]);
        a = L(a, s.idm + T[2][0]);
        a = L(a, s.imi + T[3][0]);
        a = L(a, s.iiw + T[4][0]);
        r.imm = L(r.imm, a + lpx + y.in_lp[ty]);
      }
    }
    r.imm += emis_at(m, x, y, i, j);
  } else if (ynull && (xf & F_EMIT_OR_START)) {
    for (int t = yb; t < ye; ++t) r.imm = L(r.imm, M[cell_slot(ss, i, y.in_src[t])] + y.in_lp[t]);
  } else if (yok) {
    for (int t = xb; t < xe; ++t) r.imm = L(r.imm, M[cell_slot(ss, x.in_src[t], j)] + x.in_lp[t]);
  }
  return r;
}

// one Backward cell (reference src/forward.cpp:996-1086); B is stored mirrored (hx_device.h)
template <class LSE>
__device__ __forceinline__ C5 backward_cell_dag(const Mat& m, const Side& x, const Side& y, const double (*T)[6],
                                                const LSE& L, int i, int j, uint8_t xf, uint8_t yf) {
  const HX_GLOBAL double* M = m.M;
  const int64_t plane = m.plane, ss = m.ss;
  const int R = m.R, Cc = m.Cc;
#define BS(a, b) bwd_slot(ss, R, Cc, (a), (b))
  C5 r = c5_neg_inf();
  // cells that feed END are initialised by assignment (src/forward.cpp:981-995)
  if ((xf & F_TO_END) && (yf & F_TO_END)) {
    const int xe = x.n - 1, ye = y.n - 1;
    for (int tx = x.in_off[xe]; tx < x.in_off[xe + 1]; ++tx)
      for (int ty = y.in_off[ye]; ty < y.in_off[ye + 1]; ++ty)
        if (x.in_src[tx] == i && y.in_src[ty] == j) {
          const double lp = x.in_lp[tx] + y.in_lp[ty];
          r = C5{lp + T[0][5], lp + T[1][5], lp + T[2][5], lp + T[3][5], lp + T[4][5]};
        }
  }
  const bool yok = (yf & F_READY) || y.empty;
  const bool xok = (xf & F_READY) || x.empty;
  const int xab = x.ao_off[i], xae = x.ao_off[i + 1];
  const int yab = y.ao_off[j], yae = y.ao_off[j + 1];
  const int xnb = x.no_off[i], xne = x.no_off[i + 1];
  const int ynb = y.no_off[j], yne = y.no_off[j + 1];

  for (int tx = xab; tx < xae; ++tx) {
    const int dx = x.ao_dst[tx];
    const double lpx = x.ao_lp[tx];
    for (int ty = yab; ty < yae; ++ty) {
      const int dy = y.ao_dst[ty];
      const double d = lpx + y.ao_lp[ty] + emis_at(m, x, y, dx, dy) + M[BS(dx, dy)];
      r.imm = L(r.imm, T[0][0] + d);
      r.imd = L(r.imd, T[1][0] + d);
      r.idm = L(r.idm, T[2][0] + d);
      r.imi = L(r.imi, T[3][0] + d);
      r.iiw = L(r.iiw, T[4][0] + d);
    }
  }
  if (yok)
    for (int tx = xab; tx < xae; ++tx) {
      const int dx = x.ao_dst[tx];
      const double lpx = x.ao_lp[tx];
      const int64_t sl = BS(dx, j);
      const double d1 = lpx + x.rootsub[dx] + M[plane + sl];
      const double d2 = lpx + x.ins[dx] + M[4 * plane + sl];
      r.imm = L(r.imm, T[0][1] + d1);
      r.imd = L(r.imd, T[1][1] + d1);
      r.idm = L(r.idm, T[2][1] + d1);
      r.imi = L(r.imi, T[3][1] + d1);
      r.imm = L(r.imm, T[0][4] + d2);
      r.imi = L(r.imi, T[3][4] + d2);
      r.iiw = L(r.iiw, T[4][4] + d2);
    }
  if (xok)
    for (int ty = yab; ty < yae; ++ty) {
      const int dy = y.ao_dst[ty];
      const double lpy = y.ao_lp[ty];
      const int64_t sl = BS(i, dy);
      const double d1 = lpy + y.rootsub[dy] + M[2 * plane + sl];
      const double d2 = lpy + y.ins[dy] + M[3 * plane + sl];
      r.imm = L(r.imm, T[0][2] + d1);
      r.imd = L(r.imd, T[1][2] + d1);
      r.idm = L(r.idm, T[2][2] + d1);
      r.iiw = L(r.iiw, T[4][2] + d1);
      r.imm = L(r.imm, T[0][3] + d2);
      r.imi = L(r.imi, T[3][3] + d2);
    }
  if (yok)
    for (int tx = xnb; tx < xne; ++tx) {
      const int dx = x.no_dst[tx];
      if (dx >= R) continue;   // END is not stored: xyCell(END,.) is the empty cell
      const double lpx = x.no_lp[tx];
      const int64_t sl = BS(dx, j);
      r.imd = L(r.imd, lpx + M[plane + sl]);
      r.iiw = L(r.iiw, lpx + M[4 * plane + sl]);
      r.imm = L(r.imm, lpx + M[sl]);
    }
  for (int ty = ynb; ty < yne; ++ty) {
    const int dy = y.no_dst[ty];
    if (dy >= Cc) continue;
    const double lpy = y.no_lp[ty];
    const int64_t sl = BS(i, dy);
    r.idm = L(r.idm, lpy + M[2 * plane + sl]);
    r.imi = L(r.imi, lpy + M[3 * plane + sl]);
    if (xf & F_EMIT_OR_START) r.imm = L(r.imm, lpy + M[sl]);
  }
#undef BS
  return r;
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward with state records (rounds 2-3).  The recursion above walks a state's out-transitions through the CSR arrays:
// offsets, then destinations and weights, then the destinations' constants, then the matrix - four loads deep, sixty-four
// loads per wave-step.  A 96-byte record per state holds the CSR ranges, the flags and envelope coordinate, the first TWO
// absorbing out-transitions in full (destination, weight, emission class, rootsub and ins of the destination: nine states
// in ten have one, all but two in a hundred at most two) and the first null out-transition; the workgroup builds the
// records of its two profiles into the pair's scratch planes (free after the Forward fill) before it starts.  A row's
// record stays in registers for the strip, a column's is six 16-byte loads; everything that hangs on the recorded
// transitions - up to four transition pairs, two cells each for the x- and y-absorbing moves of two transitions, three
// each for the null moves - is then fetched in ONE batch before the first look-up (round 3: a build with every out-degree
// clamped to one showed that the second and further transitions, taken one at a time with their own dependent loads, were
// 55-60 % of the fill's time).  Third and further transitions are walked as before.
// Same operations on the same operands in the same order as backward_cell_dag: bit-identical.
// ---------------------------------------------------------------------------------------------------------------------
typedef int i4v __attribute__((ext_vector_type(4)));
typedef double bw_d2 __attribute__((ext_vector_type(2)));
#define HX_BWD_REC_ABS 3               // absorbing out-transitions a record holds in full
struct BwdRec {
  int ab, ae, nb, ne;                  // absorbing / null out-transitions (CSR ranges)
  int d[3], c[3];                      // destinations of the first three absorbing transitions and their emission classes
  int flags, env, n[2];                // the state's flags and envelope coordinate; destinations of the first two null transitions
  double lp[3], rs[3], ins[3];         // the absorbing transitions' weights, rootsub and ins of their destinations
  double nlp[2];                       // the two null transitions' weights (a state that has null out-transitions usually has two)
};
// in memory: ten 16-byte quarters {ab, ae, nb, ne} {d0, d1, d2, c0} {c1, c2, flags, env} {n0, n1, -, -}
//   {lp0, lp1} {lp2, rs0} {rs1, rs2} {ins0, ins1} {ins2, nlp0} {nlp1, -}
#define HX_BWD_REC_BYTES 160
__device__ __forceinline__ void build_bwd_recs(const Side& s, HX_GLOBAL char* out, bool banded, int tid, int threads) {
  for (int i = tid; i < s.n; i += threads) {
    BwdRec r;
    r.ab = s.ao_off[i]; r.ae = s.ao_off[i + 1];
    r.nb = s.no_off[i]; r.ne = s.no_off[i + 1];
    r.flags = s.flags[i];
    r.env = banded ? s.env[i] : 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      r.n[k] = 0; r.nlp[k] = HX_NEG_INF;
      if (r.ne > r.nb + k) { r.n[k] = s.no_dst[r.nb + k]; r.nlp[k] = s.no_lp[r.nb + k]; }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      r.d[k] = 0; r.c[k] = -1; r.lp[k] = r.rs[k] = r.ins[k] = HX_NEG_INF;
      if (r.ae > r.ab + k) {
        r.d[k] = s.ao_dst[r.ab + k];
        r.lp[k] = s.ao_lp[r.ab + k];
        r.c[k] = s.cls[r.d[k]];
        r.rs[k] = s.rootsub[r.d[k]];
        r.ins[k] = s.ins[r.d[k]];
      }
    }
    HX_GLOBAL i4v* q = (HX_GLOBAL i4v*)(out + (size_t)i * HX_BWD_REC_BYTES);
    q[0] = i4v{r.ab, r.ae, r.nb, r.ne};
    q[1] = i4v{r.d[0], r.d[1], r.d[2], r.c[0]};
    q[2] = i4v{r.c[1], r.c[2], r.flags, r.env};
    q[3] = i4v{r.n[0], r.n[1], 0, 0};
    ((HX_GLOBAL bw_d2*)q)[4] = bw_d2{r.lp[0], r.lp[1]};
    ((HX_GLOBAL bw_d2*)q)[5] = bw_d2{r.lp[2], r.rs[0]};
    ((HX_GLOBAL bw_d2*)q)[6] = bw_d2{r.rs[1], r.rs[2]};
    ((HX_GLOBAL bw_d2*)q)[7] = bw_d2{r.ins[0], r.ins[1]};
    ((HX_GLOBAL bw_d2*)q)[8] = bw_d2{r.ins[2], r.nlp[0]};
    ((HX_GLOBAL bw_d2*)q)[9] = bw_d2{r.nlp[1], 0.};
  }
}
__device__ __forceinline__ BwdRec load_bwd_rec(const HX_GLOBAL char* base, int i) {
  const HX_GLOBAL i4v* q = (const HX_GLOBAL i4v*)(base + (size_t)i * HX_BWD_REC_BYTES);
  const HX_GLOBAL bw_d2* qd = (const HX_GLOBAL bw_d2*)q;
  const i4v a = q[0], b = q[1], c = q[2], d = q[3];
  const bw_d2 e = qd[4], f = qd[5], g = qd[6], h = qd[7], k = qd[8], l = qd[9];
  BwdRec r;
  r.ab = a.x; r.ae = a.y; r.nb = a.z; r.ne = a.w;
  r.d[0] = b.x; r.d[1] = b.y; r.d[2] = b.z; r.c[0] = b.w;
  r.c[1] = c.x; r.c[2] = c.y; r.flags = c.z; r.env = c.w;
  r.n[0] = d.x; r.n[1] = d.y;
  r.lp[0] = e.x; r.lp[1] = e.y; r.lp[2] = f.x; r.rs[0] = f.y; r.rs[1] = g.x; r.rs[2] = g.y;
  r.ins[0] = h.x; r.ins[1] = h.y; r.ins[2] = k.x; r.nlp[0] = k.y; r.nlp[1] = l.x;
  return r;
}

// COH: the matrix is being written by other workgroups too (k_backward_dag_multi): every load of a cell bypasses the L1
// (`sc1`, a relaxed agent-scope load) - the producing wave stored it `sc1` and published its progress behind its own
// s_waitcnt vmcnt(0), the form MI355X_MICROARCH.md lists as valid without an L2 write-back.
template <bool COH>
struct CellLoads {
  const HX_GLOBAL double* p;
  __device__ __forceinline__ double operator[](int64_t k) const {
    if (COH) return __hip_atomic_load((double*)(p + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return p[k];
  }
};
// emission of the pair of destination states (dx, dy) whose classes are known (emis_at without the two class loads)
__device__ __forceinline__ double emis_known(const Mat& m, int ncy, int cx, int cy, int dx, int dy) {
  if (m.etab) return (cx < 0 || cy < 0) ? HX_NEG_INF : m.etab[(int64_t)cx * ncy + cy];
  return m.eplane[cell_slot(m.ss, dx, dy)];
}
// KR: how many of the recorded absorbing transitions the cell takes from the records (the rest it walks): three where the
// registers allow (the fast table policy), two in the exact policy, whose look-ups hold more of them
template <class LSE, bool COH = false, int KR = 2>
__device__ __forceinline__ C5 backward_cell_rec(const Mat& m, const Side& x, const Side& y, const double (*T)[6],
                                                const LSE& L, int i, int j, const BwdRec& rx, const BwdRec& ry) {
  static_assert(KR >= 1 && KR <= HX_BWD_REC_ABS, "the records hold three absorbing transitions");
  const CellLoads<COH> M{m.M};
  const int64_t plane = m.plane, ss = m.ss;
  const int R = m.R, Cc = m.Cc;
  const uint8_t xf = (uint8_t)rx.flags, yf = (uint8_t)ry.flags;
#define BS(a, b) bwd_slot(ss, R, Cc, (a), (b))
  C5 r = c5_neg_inf();
  // cells that feed END are initialised by assignment (src/forward.cpp:981-995)
  if ((xf & F_TO_END) && (yf & F_TO_END)) {
    const int xe = x.n - 1, ye = y.n - 1;
    for (int tx = x.in_off[xe]; tx < x.in_off[xe + 1]; ++tx)
      for (int ty = y.in_off[ye]; ty < y.in_off[ye + 1]; ++ty)
        if (x.in_src[tx] == i && y.in_src[ty] == j) {
          const double lp = x.in_lp[tx] + y.in_lp[ty];
          r = C5{lp + T[0][5], lp + T[1][5], lp + T[2][5], lp + T[3][5], lp + T[4][5]};
        }
  }
  const bool yok = (yf & F_READY) || y.empty;
  const bool xok = (xf & F_READY) || x.empty;
  const int xab = rx.ab, xae = rx.ae, yab = ry.ab, yae = ry.ae;
  const int xnb = rx.nb, xne = rx.ne, ynb = ry.nb, yne = ry.ne;
  const int xn = xae - xab, yn = yae - yab;              // absorbing out-degrees

  // ---- the batch: everything that hangs on the recorded absorbing transitions ----
  double e[KR][KR], mm[KR][KR], x1[KR], x4[KR], y2[KR], y3[KR];
  double xn0[2], xn1[2], xn4[2], yn0[2], yn2[2], yn3[2];
#pragma unroll
  for (int a = 0; a < KR; ++a)
#pragma unroll
    for (int b = 0; b < KR; ++b) {
      e[a][b] = mm[a][b] = HX_NEG_INF;
      if (a < xn && b < yn) { e[a][b] = emis_known(m, y.n_cls, rx.c[a], ry.c[b], rx.d[a], ry.d[b]); mm[a][b] = M[BS(rx.d[a], ry.d[b])]; }
    }
#pragma unroll
  for (int a = 0; a < KR; ++a) {
    x1[a] = x4[a] = HX_NEG_INF;
    if (yok && a < xn) { const int64_t sl = BS(rx.d[a], j); x1[a] = M[plane + sl]; x4[a] = M[4 * plane + sl]; }
  }
#pragma unroll
  for (int b = 0; b < KR; ++b) {
    y2[b] = y3[b] = HX_NEG_INF;
    if (xok && b < yn) { const int64_t sl = BS(i, ry.d[b]); y2[b] = M[2 * plane + sl]; y3[b] = M[3 * plane + sl]; }
  }
  // ... and on the recorded null transitions: with two absorbing transitions in the same batch, with three behind the pairs'
  // look-ups (in registers those have freed; they are the last to be used)
  const auto null_loads = [&]() {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      xn0[k] = xn1[k] = xn4[k] = yn0[k] = yn2[k] = yn3[k] = HX_NEG_INF;
      if (yok && xnb + k < xne && rx.n[k] < R) { const int64_t sl = BS(rx.n[k], j); xn1[k] = M[plane + sl]; xn4[k] = M[4 * plane + sl]; xn0[k] = M[sl]; }
      if (ynb + k < yne && ry.n[k] < Cc) { const int64_t sl = BS(i, ry.n[k]); yn2[k] = M[2 * plane + sl]; yn3[k] = M[3 * plane + sl]; if (xf & F_EMIT_OR_START) yn0[k] = M[sl]; }
    }
  };
  if (KR < 3) null_loads();

  // ---- transition pairs, row transition by row transition (src/forward.cpp:1000-1016) ----
  const auto pair_term = [&](const double d) {
    r.imm = L(r.imm, T[0][0] + d);
    r.imd = L(r.imd, T[1][0] + d);
    r.idm = L(r.idm, T[2][0] + d);
    r.imi = L(r.imi, T[3][0] + d);
    r.iiw = L(r.iiw, T[4][0] + d);
  };
#pragma unroll
  for (int a = 0; a < KR; ++a)
    if (a < xn) {
#pragma unroll
      for (int b = 0; b < KR; ++b)
        if (b < yn) pair_term(rx.lp[a] + ry.lp[b] + e[a][b] + mm[a][b]);
      for (int ty = yab + KR; ty < yae; ++ty) {          // the column's further transitions
        const int dy = y.ao_dst[ty];
        pair_term(rx.lp[a] + y.ao_lp[ty] + emis_at(m, x, y, rx.d[a], dy) + M[BS(rx.d[a], dy)]);
      }
    }
  for (int tx = xab + KR; tx < xae; ++tx) {              // the row's further transitions
    const int dx = x.ao_dst[tx];
    const double lpx = x.ao_lp[tx];
    for (int ty = yab; ty < yae; ++ty) {
      const int dy = y.ao_dst[ty];                        // (from the CSR arrays, not the record: selecting a record field by a
      const double lpy = y.ao_lp[ty];                     //  run-time index would put the record into private memory at every step)
      pair_term(lpx + lpy + emis_at(m, x, y, dx, dy) + M[BS(dx, dy)]);
    }
  }
  if (KR >= 3) {
    __builtin_amdgcn_sched_barrier(0);
    null_loads();
  }
  // ---- x-absorbing moves ----
  if (yok) {
    const auto x_term = [&](const double d1, const double d2) {
      r.imm = L(r.imm, T[0][1] + d1);
      r.imd = L(r.imd, T[1][1] + d1);
      r.idm = L(r.idm, T[2][1] + d1);
      r.imi = L(r.imi, T[3][1] + d1);
      r.imm = L(r.imm, T[0][4] + d2);
      r.imi = L(r.imi, T[3][4] + d2);
      r.iiw = L(r.iiw, T[4][4] + d2);
    };
#pragma unroll
    for (int a = 0; a < KR; ++a)
      if (a < xn) x_term(rx.lp[a] + rx.rs[a] + x1[a], rx.lp[a] + rx.ins[a] + x4[a]);
    for (int tx = xab + KR; tx < xae; ++tx) {
      const int dx = x.ao_dst[tx];
      const double lpx = x.ao_lp[tx];
      const int64_t sl = BS(dx, j);
      x_term(lpx + x.rootsub[dx] + M[plane + sl], lpx + x.ins[dx] + M[4 * plane + sl]);
    }
  }
  // ---- y-absorbing moves ----
  if (xok) {
    const auto y_term = [&](const double d1, const double d2) {
      r.imm = L(r.imm, T[0][2] + d1);
      r.imd = L(r.imd, T[1][2] + d1);
      r.idm = L(r.idm, T[2][2] + d1);
      r.iiw = L(r.iiw, T[4][2] + d1);
      r.imm = L(r.imm, T[0][3] + d2);
      r.imi = L(r.imi, T[3][3] + d2);
    };
#pragma unroll
    for (int b = 0; b < KR; ++b)
      if (b < yn) y_term(ry.lp[b] + ry.rs[b] + y2[b], ry.lp[b] + ry.ins[b] + y3[b]);
    for (int ty = yab + KR; ty < yae; ++ty) {
      const int dy = y.ao_dst[ty];
      const double lpy = y.ao_lp[ty];
      const int64_t sl = BS(i, dy);
      y_term(lpy + y.rootsub[dy] + M[2 * plane + sl], lpy + y.ins[dy] + M[3 * plane + sl]);
    }
  }
  // ---- null moves ----
  if (yok) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (xnb + k < xne && rx.n[k] < R) {      // (END is not stored: xyCell(END,.) is the empty cell)
        r.imd = L(r.imd, rx.nlp[k] + xn1[k]);
        r.iiw = L(r.iiw, rx.nlp[k] + xn4[k]);
        r.imm = L(r.imm, rx.nlp[k] + xn0[k]);
      }
    for (int tx = xnb + 2; tx < xne; ++tx) {
      const int dx = x.no_dst[tx];
      if (dx >= R) continue;
      const double lpx = x.no_lp[tx];
      const int64_t sl = BS(dx, j);
      r.imd = L(r.imd, lpx + M[plane + sl]);
      r.iiw = L(r.iiw, lpx + M[4 * plane + sl]);
      r.imm = L(r.imm, lpx + M[sl]);
    }
  }
#pragma unroll
  for (int k = 0; k < 2; ++k)
    if (ynb + k < yne && ry.n[k] < Cc) {
      r.idm = L(r.idm, ry.nlp[k] + yn2[k]);
      r.imi = L(r.imi, ry.nlp[k] + yn3[k]);
      if (xf & F_EMIT_OR_START) r.imm = L(r.imm, ry.nlp[k] + yn0[k]);
    }
  for (int ty = ynb + 2; ty < yne; ++ty) {
    const int dy = y.no_dst[ty];
    if (dy >= Cc) continue;
    const double lpy = y.no_lp[ty];
    const int64_t sl = BS(i, dy);
    r.idm = L(r.idm, lpy + M[2 * plane + sl]);
    r.imi = L(r.imi, lpy + M[3 * plane + sl]);
    if (xf & F_EMIT_OR_START) r.imm = L(r.imm, lpy + M[sl]);
  }
#undef BS
  return r;
}

#define HX_DAG_MAX_WAVES 16
#define HX_DAG_REC_WAVES 8      // the state-record Backward kernels: 8 waves, so that a wave may use 256 registers (the batch of first-transition loads)

// DIR 0: Forward, DIR 1: Backward (swept in mirrored coordinates)
template <int DIR, class LSE, bool FAST, bool REC = false>
__global__ void __launch_bounds__((REC ? HX_DAG_REC_WAVES : HX_DAG_MAX_WAVES) * 64) k_fill_dag(const DevJob* __restrict__ jobs,
                                                                      const double* __restrict__ exact_tab,
                                                                      const double* __restrict__ fast_tab) {
  __shared__ volatile int prog[HX_DAG_MAX_WAVES];
  __shared__ __attribute__((aligned(16))) double ftab[FAST ? (HX_FAST_INTERVALS + 1) * 2 : 2];
  const int threads = blockDim.x, W = threads >> 6;
  if (FAST)
    for (int k = threadIdx.x; k < (HX_FAST_INTERVALS + 1) * 2; k += threads) ftab[k] = fast_tab[k];
  if (threadIdx.x < HX_DAG_MAX_WAVES) prog[threadIdx.x] = 0;
  __syncthreads();
  const LSE L = LSE::make(FAST ? (const double*)ftab : fast_tab)   /* exact mode: fast_tab is the pair table */;
  volatile HX_LDS int* progp = (volatile HX_LDS int*)prog;

  const DevJob& J = jobs[blockIdx.x];
  const Side x = make_side(J.x), y = make_side(J.y);
  Mat m;
  m.M = as_global(DIR ? J.bwd : J.fwd);
  m.etab = as_global((const double*)J.emis);
  m.eplane = as_global((const double*)J.emis_plane);
  m.plane = J.plane; m.ss = J.strip_stride;
  m.R = J.n_rows; m.Cc = J.n_cols; m.max_dist = J.max_dist;
  const int R = m.R, Cc = m.Cc;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_strips = (R + 63) >> 6;
  const int prev_wave = (wave + W - 1) % W;
  const bool banded = J.max_dist >= 0;
  const HX_GLOBAL int32_t* win = as_global(DIR ? J.bwd_windows : J.fwd_windows);
  // REC: the state records of both profiles, built here into the pair's scratch planes (the caller launches this
  // instantiation only for pairs that have them, and only after the Forward fill is done with them)
  HX_GLOBAL char* xrec = nullptr;
  HX_GLOBAL char* yrec = nullptr;
  if (REC) {
    xrec = (HX_GLOBAL char*)as_global(J.agg);
    yrec = xrec + (size_t)x.n * HX_BWD_REC_BYTES;
    build_bwd_recs(x, xrec, banded, (int)threadIdx.x, threads);
    build_bwd_recs(y, yrec, banded, (int)threadIdx.x, threads);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  for (int s = wave; s < n_strips; s += W) {
    const int im = (s << 6) + lane;              // row in sweep coordinates
    const bool rvalid = im < R;
    const int i = rvalid ? (DIR ? R - 1 - im : im) : 0;
    const uint8_t xf = x.flags[i];
    const int xenv = banded ? x.env[i] : 0;
    BwdRec rx;
    if (REC) rx = load_bwd_rec(xrec, i);
    const int above_base = ((s - 1) / W) * Cc;   // columns the wave above published in its earlier strips
    const int my_base = (s / W) * Cc;
    const int64_t store_base = (int64_t)s * m.ss + (lane << 1);
    int seen = 0;
    // step windows [w0,w1) and [w2,w3) that hold this strip's in-envelope cells (whole sweep when unbanded)
    int wlo[2] = {0, 0}, whi[2] = {Cc + 63, 0};
    if (banded) {
      wlo[0] = win[4 * s]; whi[0] = win[4 * s + 1];
      wlo[1] = win[4 * s + 2]; whi[1] = win[4 * s + 3];
    }
    int published = 0;
    for (int w = 0; w < 2; ++w) {
      for (int t = wlo[w]; t < whi[w]; ++t) {
        if (s > 0) {
          const int need = above_base + (t + 1 < Cc ? t + 1 : Cc);
          if (seen < need) {
            do {
              seen = __builtin_amdgcn_readfirstlane(progp[prev_wave]);
              if (seen < need) __builtin_amdgcn_s_sleep(1);
            } while (seen < need);
            asm volatile("" ::: "memory");
          }
        }
        const int jm = t - lane;
        if (rvalid && jm >= 0 && jm < Cc) {
          const int j = DIR ? Cc - 1 - jm : jm;
          BwdRec ry;
          if (REC) ry = load_bwd_rec(yrec, j);
          const uint8_t yf = REC ? (uint8_t)ry.flags : y.flags[j];
          if (in_env(m, xf, yf, xenv, banded ? (REC ? ry.env : y.env[j]) : 0)) {
            const C5 c = REC ? backward_cell_rec<LSE, false, 3>(m, x, y, J.T, L, i, j, rx, ry)
                       : DIR ? backward_cell_dag(m, x, y, J.T, L, i, j, xf, yf)
                             : forward_cell_dag(m, x, y, J.T, L, i, j, xf, yf);
            const int64_t sl = store_base + ((int64_t)(t >> 1) << 7) + (t & 1);
            m.M[sl] = c.imm;
            m.M[m.plane + sl] = c.imd;
            m.M[2 * m.plane + sl] = c.idm;
            m.M[3 * m.plane + sl] = c.imi;
            m.M[4 * m.plane + sl] = c.iiw;
          }
        }
        // The next step of this wave reads this step's cells back through memory: vector-memory operations of a wave
        // take effect in issue order, so no wait is needed for that.  The strip below learns of them through the counter:
        // every eighth step the wave drains its stores and publishes how many columns all of its rows have completed.
        if ((t & 7) == 7) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          int done = t - 62;
          done = done > Cc ? Cc : done;
          if (done > published) {
            published = done;
            if (lane == 0) progp[wave] = my_base + done;
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // columns between / after the windows hold no in-envelope cell of this strip: they count as complete
      // once everything before them is
      const int upto = (w == 0 && whi[1] > wlo[1]) ? wlo[1] - 63 : Cc;
      int done = upto > Cc ? Cc : upto;
      if (done > published) {
        published = done;
        if (lane == 0) progp[wave] = my_base + done;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (DIR == 0) *J.lp_end = forward_lp_end(J, ExactLse{exact_tab});
    else *J.lp_start = J.bwd[cell_slot(m.ss, R - 1, Cc - 1)];   // B(0,0).IMM in mirrored coordinates
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward for ONE pair (or two) on several CUs: the pair's strips are dealt to the 16 G waves of G workgroups instead of the
// 16 waves of one, so that a pair of thirty strips needs one pass instead of two and its waves share SIMDs less.  What the
// strips hand each other - the matrix itself - travels write-through: every cell is stored `sc1`, every cell is loaded `sc1`
// (CellLoads), a wave's progress counter lives in global memory (zeroed by the host before the launch), is stored `sc1`
// behind the wave's own s_waitcnt vmcnt(0) and polled with `sc1` loads.  A poll gives up after HX_MULTI_PATIENCE rounds: the
// pair's lpStart then reads NaN and the host reports an error - never a hang.  All workgroups of the launch must be resident
// at once (the caller launches at most sixteen).  State records as in k_fill_dag<.., REC>.
// ---------------------------------------------------------------------------------------------------------------------
template <class LSE, bool FAST>
__global__ void __launch_bounds__(HX_DAG_REC_WAVES * 64) k_backward_dag_multi(const DevJob* __restrict__ jobs,
                                                                                const double* __restrict__ exact_tab,
                                                                                const double* __restrict__ fast_tab, const int G, const int patience) {
  __shared__ __attribute__((aligned(16))) double ftab[FAST ? (HX_FAST_INTERVALS + 1) * 2 : 2];
  const int threads = blockDim.x, W = threads >> 6, WT = W * G;
  if (FAST)
    for (int k = threadIdx.x; k < (HX_FAST_INTERVALS + 1) * 2; k += threads) ftab[k] = fast_tab[k];
  __syncthreads();
  const LSE L = LSE::make(FAST ? (const double*)ftab : fast_tab);
  const int job = (int)blockIdx.x / G, grp = (int)blockIdx.x % G;
  const DevJob& J = jobs[job];
  const Side x = make_side(J.x), y = make_side(J.y);
  Mat m;
  m.M = as_global(J.bwd);
  m.etab = as_global((const double*)J.emis);
  m.eplane = as_global((const double*)J.emis_plane);
  m.plane = J.plane; m.ss = J.strip_stride;
  m.R = J.n_rows; m.Cc = J.n_cols; m.max_dist = J.max_dist;
  const int R = m.R, Cc = m.Cc;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gw = grp * W + wave;                   // this wave among the pair's WT waves
  const int n_strips = (R + 63) >> 6;
  const int prev_gw = (gw + WT - 1) % WT;
  const bool banded = J.max_dist >= 0;
  const HX_GLOBAL int32_t* win = as_global(J.bwd_windows);
  // records: every workgroup builds them (the same bytes at the same addresses), its own waves read them after its barrier
  HX_GLOBAL char* xrec = (HX_GLOBAL char*)as_global(J.agg);
  HX_GLOBAL char* yrec = xrec + (size_t)x.n * HX_BWD_REC_BYTES;
  build_bwd_recs(x, xrec, banded, (int)threadIdx.x, threads);
  build_bwd_recs(y, yrec, banded, (int)threadIdx.x, threads);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // progress counters of the pair's waves: the last 256 ints of its scratch planes
  HX_GLOBAL int* gprog = (HX_GLOBAL int*)as_global(reinterpret_cast<int*>(J.agg + 5 * J.plane) - 256);
  bool dead = false;
  HX_GLOBAL double* Mw = as_global(J.bwd);       // (global, not flat, instructions: the hand-off is only measured valid for those)
  const auto put = [&](int64_t k, double v) { __hip_atomic_store(Mw + k, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  const auto publish = [&](int value) {
    if (lane == 0) __hip_atomic_store(gprog + gw, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  for (int s = gw; s < n_strips; s += WT) {
    const int im = (s << 6) + lane;
    const bool rvalid = im < R;
    const int i = rvalid ? R - 1 - im : 0;
    const uint8_t xf = x.flags[i];
    const int xenv = banded ? x.env[i] : 0;
    const BwdRec rx = load_bwd_rec(xrec, i);
    const int above_base = ((s - 1) / WT) * Cc;
    const int my_base = (s / WT) * Cc;
    const int64_t store_base = (int64_t)s * m.ss + (lane << 1);
    int seen = 0;
    int wlo[2] = {0, 0}, whi[2] = {Cc + 63, 0};
    if (banded) {
      wlo[0] = win[4 * s]; whi[0] = win[4 * s + 1];
      wlo[1] = win[4 * s + 2]; whi[1] = win[4 * s + 3];
    }
    int published = 0;
    for (int w = 0; w < 2; ++w) {
      for (int t = wlo[w]; t < whi[w]; ++t) {
        if (s > 0 && !dead) {
          const int need = above_base + (t + 1 < Cc ? t + 1 : Cc);
          int polls = 0;
          while (seen < need) {
            seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(gprog + prev_gw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (seen < need) {
              if (++polls > patience) { dead = true; break; }
              __builtin_amdgcn_s_sleep(2);
            }
          }
          if (seen == HX_MULTI_POISON) dead = true;      // the wave above (or one above it) gave up
        }
        const int jm = t - lane;
        if (!dead && rvalid && jm >= 0 && jm < Cc) {
          const int j = Cc - 1 - jm;
          const BwdRec ry = load_bwd_rec(yrec, j);
          const uint8_t yf = (uint8_t)ry.flags;
          if (in_env(m, xf, yf, xenv, banded ? ry.env : 0)) {
            const C5 c = backward_cell_rec<LSE, true, 3>(m, x, y, J.T, L, i, j, rx, ry);
            const int64_t sl = store_base + ((int64_t)(t >> 1) << 7) + (t & 1);
            put(sl, c.imm); put(m.plane + sl, c.imd); put(2 * m.plane + sl, c.idm); put(3 * m.plane + sl, c.imi); put(4 * m.plane + sl, c.iiw);
          }
        }
        if ((t & 7) == 7) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          int done = t - 62;
          done = done > Cc ? Cc : done;
          if (dead && published != HX_MULTI_POISON) {   // (everyone below runs out as well, and knows why)
            published = HX_MULTI_POISON;
            publish(HX_MULTI_POISON);
          }
          if (!dead && done > published) {
            published = done;
            publish(my_base + done);
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int upto = (w == 0 && whi[1] > wlo[1]) ? wlo[1] - 63 : Cc;
      const int done = upto > Cc ? Cc : upto;
      if (dead && published != HX_MULTI_POISON) {
        published = HX_MULTI_POISON;
        publish(HX_MULTI_POISON);
      }
      if (!dead && done > published) {
        published = done;
        publish(my_base + done);
      }
    }
    // the wave of the last strip reports B(START, START).IMM (its own store, read back past the L1)
    if (s == n_strips - 1 && lane == 0) {
      const double v = __hip_atomic_load(Mw + cell_slot(m.ss, R - 1, Cc - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *J.lp_start = dead ? __builtin_nan("") : v;
    }
  }
}

// ---------------------------------------------------------------------------
// Forward, second formulation.  Every cell also stores its five "outgoing sums"
//   g0 = (+)_s cell[s] + T[s][IMD]   (s = IMM,IMD,IDM,IMI)     read by x-absorbing moves into IMD
//   g1 = (+)_s cell[s] + T[s][IIW]   (s = IMM,IMI,IIW)         ...                       into IIW
//   g2 = (+)_s cell[s] + T[s][IDM]   (s = IMM,IMD,IDM,IIW)     read by y-absorbing moves into IDM
//   g3 = (+)_s cell[s] + T[s][IMI]   (s = IMM,IMI)             ...                       into IMI
//   g4 = (+)_s cell[s] + T[s][IMM]   (all five)                read by xy-absorbing moves into IMM
// (left-nested in the reference's order, src/forward.cpp:103-115,139-150,171-180: the reference
// evaluates exactly these sums once per incoming transition, from the same source cell, so
// evaluating them once at the source is bit-identical).  A destination then needs two values per
// x- or y-transition and one per transition pair, and one log-sum-exp per value beyond the first.
// Null destination states read the raw cell planes instead (src/forward.cpp:118-128,153-163,183-200).
//
// The dependent chain of a step is: one round of loads (all issued together: the first three
// in-transitions of every state are inline in its 80-byte FwdPack; the row's pack lives in
// registers, the columns' packs in a per-wave LDS ring refilled 64 columns at a time) ->
// accumulate -> the 13 look-ups of the outgoing sums in 4 levels.  The two cells a step can need
// from the step before -- (i-1, j) of the previous lane and (i, j-1) of the lane itself -- are
// forwarded through registers (DPP wave_shr:1), so a cell's stores can be issued one step late,
// behind the next step's loads: loads then never queue behind stores younger than one step.
// ---------------------------------------------------------------------------
typedef double d2v __attribute__((ext_vector_type(2)));

struct PackRegs { double lp[HX_DAG_INLINE], rootsub, ins; int s[HX_DAG_INLINE], in_b, meta, env, cls; };

__device__ __forceinline__ PackRegs unpack(const d2v a, const d2v b, const d2v c, const d2v d, const d2v e) {
  PackRegs r;
  r.lp[0] = a.x; r.lp[1] = a.y; r.rootsub = b.x; r.ins = b.y;
  r.s[0] = __double2loint(c.x); r.s[1] = __double2hiint(c.x);
  r.in_b = __double2loint(c.y); r.meta = __double2hiint(c.y);
  r.env = __double2loint(d.x); r.cls = __double2hiint(d.x);
  r.s[2] = __double2loint(d.y);
  r.lp[2] = e.x;
  return r;
}

__device__ __forceinline__ PackRegs load_pack(const HX_GLOBAL FwdPack* p) {
  const HX_GLOBAL d2v* q = (const HX_GLOBAL d2v*)p;
  return unpack(q[0], q[1], q[2], q[3], q[4]);
}

// Slots (offsets of a cell inside a plane, in doubles) are 32-bit and unsigned: a plane of 2^32 doubles is 32 GiB, more than a
// pair's ten planes can have in 288 GB, and an address is then ONE 64-bit operation (plane pointer + zero-extended slot * 8)
// where 64-bit slots took four or five per load - most of what a step issued in front of its loads.
struct RowRef { unsigned base; int l; };
__device__ __forceinline__ RowRef row_ref(unsigned ss, int r) {
  return RowRef{(unsigned)(r >> 6) * ss + ((unsigned)(r & 63) << 1), r & 63};
}
__device__ __forceinline__ unsigned slot_at(const RowRef& r, int c) {
  const unsigned t = (unsigned)(c + r.l);
  return r.base + ((t >> 1) << 7) + (t & 1);
}

#define HX_DAGF_MAX_WAVES 8     // Forward pipeline: up to 8 waves (512 threads) so that a wave may use 256 VGPRs
#define HX_DAGF_MULTI_WAVES 4   // ... its several-workgroups launch: 4 waves, a wave may use 512

// Trace builds (-DHX_DAG_TRACE): per-strip cycle sums of the phases of a step, printed by lane 0 at the end of
// every strip (tools/dag_trace.sh).  The explicit waits a trace build adds perturb the schedule a little.
#ifdef HX_DAG_TRACE
#define HX_TR(k) do { const long long now_ = (long long)__builtin_readcyclecounter(); tr_sum[k] += now_ - tr_last; tr_last = now_; } while (0)
#define HX_TR_WAIT_LOADS() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define HX_TR(k) do { } while (0)
#define HX_TR_WAIT_LOADS() do { } while (0)
#endif

// what the next step may need from the cell a lane has just computed
// The matrix (and the scratch planes addressed relative to it) as the Forward pipeline reads and writes it.  COH: other
// workgroups fill other strips of the same pair (the MULTI launch of k_forward_dag_pipe): every access is `sc1` - stores
// write through, loads bypass the L1 - the hand-off of k_backward_dag_multi.
template <bool COH>
struct PairPlanes {
  HX_GLOBAL double* p;
  struct Ref {
    HX_GLOBAL double* q;
    __device__ __forceinline__ operator double() const {
      if (COH) return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return *q;
    }
    __device__ __forceinline__ void operator=(double v) const {
      if (COH) __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else *q = v;
    }
  };
  __device__ __forceinline__ Ref operator[](int64_t k) const { return Ref{p + k}; }
  __device__ __forceinline__ Ref operator[](unsigned k) const { return Ref{p + k}; }
  __device__ __forceinline__ PairPlanes at(int64_t off) const { return PairPlanes{p + off}; }      // another plane's view
};

struct Fwd10 { double imm, imd, idm, imi, iiw, g0, g1, g2, g3, g4; };

// MULTI: one pair's strips dealt to `groups` workgroups (one or two pairs of many strips: see k_backward_dag_multi for the
// hand-off; the progress counters - 256 ints per pair - are a region the host passes in, zeroed before the launch).
template <class LSE, bool FAST, bool MULTI = false>
__global__ void __launch_bounds__(MULTI ? HX_DAGF_MULTI_WAVES * 64 : HX_DAGF_MAX_WAVES * 64) k_forward_dag_pipe(const DevJob* __restrict__ jobs,
                                                                               const double* __restrict__ exact_tab,
                                                                               const double* __restrict__ fast_tab,
                                                                               const int groups = 1, int* const counters = nullptr,
                                                                               const int patience = 0) {
  __shared__ volatile int prog[HX_DAGF_MAX_WAVES];
  __shared__ __attribute__((aligned(16))) double ftab[FAST ? (HX_FAST_INTERVALS + 1) * 2 : 2];
  // per wave: the column constants (FwdPack) of the 128 columns around the wave's position, as four
  // arrays of 16-byte quarters so that 64 lanes reading 64 consecutive columns do not collide on banks
  __shared__ d2v ycols[HX_DAGF_MAX_WAVES][5][128];
  const int threads = blockDim.x, W = threads >> 6;
  if (FAST)
    for (int k = threadIdx.x; k < (HX_FAST_INTERVALS + 1) * 2; k += threads) ftab[k] = fast_tab[k];
  if (threadIdx.x < HX_DAGF_MAX_WAVES) prog[threadIdx.x] = 0;
  __syncthreads();
  const LSE L = LSE::make(FAST ? (const double*)ftab : fast_tab)   /* exact mode: fast_tab is the pair table */;
  volatile HX_LDS int* progp = (volatile HX_LDS int*)prog;

  const int G = MULTI ? groups : 1;
  const int job = MULTI ? (int)blockIdx.x / G : (int)blockIdx.x, grp = MULTI ? (int)blockIdx.x % G : 0;
  const DevJob& J = jobs[job];
  const double (*T)[6] = J.T;
  const int R = J.n_rows, Cc = J.n_cols;
  const int64_t plane = J.plane;
  const unsigned ss = (unsigned)J.strip_stride;
  const PairPlanes<MULTI> M{as_global(J.fwd)};
  HX_GLOBAL int* gprog = MULTI ? (HX_GLOBAL int*)as_global(counters + 256 * job) : nullptr;
  const int64_t aggoff = J.agg - J.fwd;          // the outgoing-sum planes, addressed relative to the matrix
  const HX_GLOBAL double* etab = as_global((const double*)J.emis);
  const HX_GLOBAL double* eplane = as_global((const double*)J.emis_plane);
  const HX_GLOBAL FwdPack* xpk = as_global((const FwdPack*)J.x.fpack);
  const HX_GLOBAL FwdPack* ypk = as_global((const FwdPack*)J.y.fpack);
  const HX_GLOBAL int32_t* xin_src = as_global(J.x.in_src);
  const HX_GLOBAL double* xin_lp = as_global(J.x.in_lp);
  const HX_GLOBAL int32_t* yin_src = as_global(J.y.in_src);
  const HX_GLOBAL double* yin_lp = as_global(J.y.in_lp);
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
  const double NI = HX_NEG_INF;
  // in-transitions beyond the inline ones are taken CHX (row) / CHY (column) at a time; EARLY: the first round's loads are
  // issued with the step's own (the several-workgroups launch has four waves per workgroup, i.e. 512 registers per lane)
  constexpr int CHX = MULTI ? 6 : 2, CHY = MULTI ? 4 : 2;
  constexpr bool EARLY = MULTI;
  constexpr bool BPAIRS = MULTI;     // (needs the registers of that launch too)
  // g / h: a further transition's pairs with the other side's inline transitions ([.][0] doubles as the IMM source when the
  // other state is null); gg: the pairs of two further transitions
  struct XRound { double va[CHX], vb[CHX], g[CHX][HX_DAG_INLINE], gg[CHX][CHY]; };
  struct YRound { double va[CHY], vb[CHY], h[CHY][HX_DAG_INLINE]; };
  HX_LDS d2v* ring = (HX_LDS d2v*)&ycols[wave][0][0];
  // columns c0 .. c0+63 (clamped into the profile) -> ring; one coalesced 4 KiB read per call
  auto stage = [&](const int c0) {
    int c = c0 + lane;
    c = c < 0 ? 0 : (c >= Cc ? Cc - 1 : c);
    const HX_GLOBAL d2v* q = (const HX_GLOBAL d2v*)(ypk + c);
    const d2v q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4];
    const int k = (c0 + lane) & 127;
    ring[k] = q0; ring[128 + k] = q1; ring[256 + k] = q2; ring[384 + k] = q3; ring[512 + k] = q4;
  };
  auto column = [&](const int j) {
    const int k = j & 127;
    return unpack(ring[k], ring[128 + k], ring[256 + k], ring[384 + k], ring[512 + k]);
  };

  for (int s = gw; s < n_strips; s += WT) {
    const int i = (s << 6) + lane;
    const bool rvalid = i < R;
    const PackRegs X = load_pack(xpk + (rvalid ? i : 0));
    const int xf = X.meta & 0xff, xdeg = X.meta >> 8;
    const bool xnull = xf & F_NULL, xok = (xf & F_READY) || xempty, xeos = xf & F_EMIT_OR_START;
    const RowRef own = RowRef{(unsigned)s * ss + ((unsigned)lane << 1), lane};
    constexpr int K = HX_DAG_INLINE;
    RowRef XR[K];
    bool adjx[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      XR[k] = row_ref(ss, X.s[k]);
      // a source in the row directly above, inside this strip, is the previous lane's cell of the previous
      // step: it is forwarded through registers (its store has not been issued yet, see below)
      adjx[k] = lane > 0 && xdeg > k && X.s[k] == i - 1;
    }
    const PairPlanes<MULTI> Xa = M.at(xnull ? plane : aggoff), Xb = M.at(xnull ? 4 * plane : aggoff + plane);
    const PairPlanes<MULTI> G4 = M.at(aggoff + 4 * plane);
    // the row's in-transitions K .. K + CHX - 1 (CSR entries), kept for the whole strip: see "transitions beyond the inline ones"
    int xs_first[CHX];
    double xl_first[CHX];
#pragma unroll
    for (int q = 0; q < CHX; ++q) {
      xs_first[q] = 0; xl_first[q] = 0.;
      if (xdeg > K + q) { xs_first[q] = xin_src[X.in_b + K + q]; xl_first[q] = xin_lp[X.in_b + K + q]; }
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
      // results of the previous step: `own` this lane's cell (i, j-1), `up` the previous lane's cell (i-1, j).
      // They are stored one step late (after the next step's loads have been issued): vector-memory
      // operations retire in issue order, so a load issued after a store waits for that store to be
      // acknowledged; issued before it, the load only queues behind stores that are a whole step old.
      Fwd10 own10 = Fwd10{NI, NI, NI, NI, NI, NI, NI, NI, NI, NI};
      double up_imm = NI, up_imd = NI, up_iiw = NI, up_g0 = NI, up_g1 = NI;
      // (the first step's store goes to the slot that step's own cell will overwrite one step later)
      unsigned pend_slot = own.base + ((unsigned)(wlo[w] >> 1) << 7) + (unsigned)(wlo[w] & 1);
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
        HX_TR(0);      // progress wait
        if (t > wlo[w] && ((t - wlo[w]) & 63) == 0) stage(t);
        const int j = t - lane;
        const PackRegs Y = column(j);
        const int yf = Y.meta & 0xff, ydeg = Y.meta >> 8;
        bool act = rvalid && j >= 0 && j < Cc && !dead;
        if (banded) {
          int dd = X.env - Y.env;
          dd = dd < 0 ? -dd : dd;
          act = act && (((xf | yf) & F_EDGE) || dd <= max_dist);
        }
        const bool ynull = yf & F_NULL, yok = (yf & F_READY) || yempty;
        const int mode = (!xnull && !ynull) ? 1 : ((ynull && xeos) ? 2 : (yok ? 3 : 0));
        const PairPlanes<MULTI> Ya = M.at(ynull ? 2 * plane : aggoff + 2 * plane), Yb = M.at(ynull ? 3 * plane : aggoff + 3 * plane);
        const bool xgo = act && yok, ygo = act && (ynull || xok);
        const int jc = j < 0 ? 0 : (j >= Cc ? Cc - 1 : j);       // a valid column for the addresses of idle lanes
        bool adjy[K];
        unsigned sXj[K], sOy[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
          adjy[k] = ydeg > k && Y.s[k] == j - 1;
          sXj[k] = slot_at(XR[k], jc);
          sOy[k] = slot_at(own, Y.s[k]);
        }
        // the column's in-transitions K .. K + CHY - 1 (CSR entries), in front of the step's loads
        int ys_first[CHY];
        double yl_first[CHY];
#pragma unroll
        for (int q = 0; q < CHY; ++q) {
          ys_first[q] = 0; yl_first[q] = 0.;
          if (act && ydeg > K + q) { ys_first[q] = yin_src[Y.in_b + K + q]; yl_first[q] = yin_lp[Y.in_b + K + q]; }
        }
        // ---- this step's loads: the first transitions for every lane (absent ones have source state 0 in
        // the pack, so the address is valid and the value is discarded below), further ones only where they
        // exist.  xa/xb feed IMD/IIW, ya/yb feed IDM/IMI, mv[] feeds IMM: transition pairs when both states
        // emit, else the y (mode 2) or x (mode 3) transitions.
        double xa[K], xb[K], ya[K], yb[K], mv[K * K];
#pragma unroll
        for (int k = 0; k < K; ++k) { xa[k] = NI; xb[k] = NI; ya[k] = NI; yb[k] = NI; }
#pragma unroll
        for (int k = 0; k < K * K; ++k) mv[k] = NI;
        xa[0] = Xa[sXj[0]]; xb[0] = Xb[sXj[0]];
        ya[0] = Ya[sOy[0]]; yb[0] = Yb[sOy[0]];
        mv[0] = (mode == 1 ? G4 : M)[mode == 1 ? slot_at(XR[0], Y.s[0]) : (mode == 2 ? sOy[0] : sXj[0])];
        double e;
        if (etab) e = etab[(int64_t)(X.cls < 0 ? 0 : X.cls) * Ky + (Y.cls < 0 ? 0 : Y.cls)];
        else e = eplane[slot_at(own, jc)];
#pragma unroll
        for (int k = 1; k < K; ++k) {
          if (xgo && xdeg > k) { xa[k] = Xa[sXj[k]]; xb[k] = Xb[sXj[k]]; }
          if (ygo && ydeg > k) { ya[k] = Ya[sOy[k]]; yb[k] = Yb[sOy[k]]; }
        }
        if (act && mode == 1) {
#pragma unroll
          for (int a = 0; a < K; ++a)
#pragma unroll
            for (int b = 0; b < K; ++b)
              if ((a | b) != 0 && xdeg > a && ydeg > b) mv[a * K + b] = G4[slot_at(XR[a], Y.s[b])];
        } else if (act && mode == 2) {
#pragma unroll
          for (int k = 1; k < K; ++k) if (ydeg > k) mv[k] = M[sOy[k]];
        } else if (act && mode == 3) {
#pragma unroll
          for (int k = 1; k < K; ++k) if (xdeg > k) mv[k] = M[sXj[k]];
        }
        if (X.cls < 0 || Y.cls < 0) e = NI;
        // ---- a round of further in-transitions (see "transitions beyond the inline ones" below): the loads of up to CHX row
        // transitions a0 .. / CHY column transitions b0 .. whose CSR entries are in registers, and later their look-ups
        const bool rpairs = mode == 1 && ydeg <= K;                 // the row's further transitions carry their pairs
        const bool cpairs = mode == 1 && xdeg == 1;                 // the column's do (pair with the row's only transition)
        // several row transitions AND further column transitions, all of them among the entries in registers: one round each
        // (only where there are registers for it: the several-workgroups launch; else such cells take the generic loop below)
        const bool bpairs = BPAIRS && mode == 1 && xdeg > 1 && ydeg > K && xdeg <= K + CHX && ydeg <= K + CHY;
        const double upA = xnull ? up_imd : up_g0, upB = xnull ? up_iiw : up_g1;
        const double ownA = ynull ? own10.idm : own10.g2, ownB = ynull ? own10.imi : own10.g3;
        XRound xv;
        YRound yv;
        const auto x_loads = [&](const int (&src)[CHX], const int a0) {
#pragma unroll
          for (int q = 0; q < CHX; ++q) {
            xv.va[q] = upA; xv.vb[q] = upB; xv.g[q][0] = up_imm;
            if (a0 + q < xdeg) {
              const bool adj = lane > 0 && src[q] == i - 1;         // the previous lane's cell of the previous step
              const RowRef rr = row_ref(ss, src[q]);
              const unsigned sl = slot_at(rr, j);
              if (xgo && !adj) { xv.va[q] = Xa[sl]; xv.vb[q] = Xb[sl]; }
              if (mode == 3 && !adj) xv.g[q][0] = M[sl];
              if (rpairs || bpairs) {                               // (a pair's source is >= 2 steps old: in memory)
#pragma unroll
                for (int b = 0; b < K; ++b)
                  if (b < ydeg) xv.g[q][b] = G4[slot_at(rr, Y.s[b])];
              }
              if (bpairs) {
#pragma unroll
                for (int r = 0; r < CHY; ++r)
                  if (K + r < ydeg) xv.gg[q][r] = G4[slot_at(rr, ys_first[r])];
              }
            }
          }
        };
        const auto y_loads = [&](const int (&src)[CHY], const int b0) {
#pragma unroll
          for (int q = 0; q < CHY; ++q) {
            yv.va[q] = ownA; yv.vb[q] = ownB; yv.h[q][0] = own10.imm;
            if (b0 + q < ydeg) {
              const bool adj = src[q] == j - 1;                     // the lane's own cell of the previous step
              const unsigned sl = slot_at(own, src[q]);
              if (ygo && !adj) { yv.va[q] = Ya[sl]; yv.vb[q] = Yb[sl]; }
              if (mode == 2 && !adj) yv.h[q][0] = M[sl];
              if (cpairs || bpairs) {
#pragma unroll
                for (int a = 0; a < K; ++a)
                  if (a < xdeg) yv.h[q][a] = G4[slot_at(XR[a], src[q])];
              }
            }
          }
        };
        if (EARLY && act) {
          // with registers to spare (the several-workgroups launch: four waves per workgroup) the first round's loads go out
          // with the step's own, so that a step with up to K + CHX row and K + CHY column transitions is one round trip
          if (xdeg > K) x_loads(xs_first, K);
          if (ydeg > K) y_loads(ys_first, K);
        }
        HX_TR(1);      // column record, addresses, load issue
        HX_TR_WAIT_LOADS();
        HX_TR(2);      // waiting for the loads
        // ---- values forwarded from the previous step, and -inf for what does not exist ----
#pragma unroll
        for (int k = 0; k < K; ++k) {
          if (adjx[k]) { xa[k] = upA; xb[k] = upB; }
          if (adjy[k]) { ya[k] = ownA; yb[k] = ownB; }
          if (mode == 2 && adjy[k]) mv[k] = own10.imm;
          if (mode == 3 && adjx[k]) mv[k] = up_imm;
        }
        if (!(xgo && xdeg > 0)) { xa[0] = NI; xb[0] = NI; }
        if (!(ygo && ydeg > 0)) { ya[0] = NI; yb[0] = NI; }
        {
          const bool h0 = mode == 1 ? (xdeg > 0 && ydeg > 0) : (mode == 2 ? ydeg > 0 : (mode == 3 && xdeg > 0));
          if (!h0) mv[0] = NI;
        }

        Fwd10 c = Fwd10{NI, NI, NI, NI, NI, NI, NI, NI, NI, NI};
        if (act) {
          // ---- the inline transitions (their cells were loaded with the step's other sources) ----
          double imd = xa[0] + X.lp[0], iiw = xb[0] + X.lp[0];      // x-absorbing (or x-null) moves
#pragma unroll
          for (int k = 1; k < K; ++k)
            if (xgo && xdeg > k) { imd = L(imd, xa[k] + X.lp[k]); iiw = L(iiw, xb[k] + X.lp[k]); }
          double idm = ya[0] + Y.lp[0], imi = yb[0] + Y.lp[0];      // y-absorbing (or y-null) moves
#pragma unroll
          for (int k = 1; k < K; ++k)
            if (ygo && ydeg > k) { idm = L(idm, ya[k] + Y.lp[k]); imi = L(imi, yb[k] + Y.lp[k]); }
          double imm = NI;
          // the transition pairs of IMM are summed row transition by row transition (reference src/forward.cpp:98-116): with at
          // most one row transition that is the column's order, so the column's further transitions add their pair below;
          // with at most K column transitions the row's further transitions add theirs; else the generic loop
          const bool pairs_inline = mode == 1 && (ydeg <= K || xdeg <= 1);
          if (pairs_inline) {
            imm = (mv[0] + X.lp[0]) + Y.lp[0];
#pragma unroll
            for (int a = 0; a < K; ++a)
#pragma unroll
              for (int b = 0; b < K; ++b)
                if ((a | b) != 0 && xdeg > a && ydeg > b) imm = L(imm, (mv[a * K + b] + X.lp[a]) + Y.lp[b]);
          } else if (mode == 2) {
            imm = mv[0] + Y.lp[0];
#pragma unroll
            for (int k = 1; k < K; ++k) if (ydeg > k) imm = L(imm, mv[k] + Y.lp[k]);
          } else if (mode == 3) {
            imm = mv[0] + X.lp[0];
#pragma unroll
            for (int k = 1; k < K; ++k) if (xdeg > k) imm = L(imm, mv[k] + X.lp[k]);
          }
          // ---- transitions beyond the inline ones.  Rare (half a percent of the states in gp120's profiles, two percent at
          // 64 leaves), but a row that has them has them at every step and sets the pace of its strip, and through the pipeline
          // of its pair; and one of the 64 columns of a step has them at every second or third step.  They are taken CHX / CHY
          // at a time: a round's CSR entries are in registers before it starts (the row's first for the whole strip, the
          // column's first fetched in front of the step's loads, a later round's together with the loads of the round before),
          // so a round is ONE batch of loads - the cells of its transitions and everything that hangs on them - and then the
          // look-ups, in the reference's order.
          if (ydeg > K) {
            int e_src[CHY];
            double e_lp[CHY];
#pragma unroll
            for (int q = 0; q < CHY; ++q) { e_src[q] = ys_first[q]; e_lp[q] = yl_first[q]; }
            for (int b = K; b < ydeg; b += CHY) {
              int n_src[CHY];
              double n_lp[CHY];
#pragma unroll
              for (int q = 0; q < CHY; ++q) {
                n_src[q] = 0; n_lp[q] = 0.;
                if (b + CHY + q < ydeg) { n_src[q] = yin_src[Y.in_b + b + CHY + q]; n_lp[q] = yin_lp[Y.in_b + b + CHY + q]; }
              }
              if (!EARLY || b > K) y_loads(e_src, b);
#pragma unroll
              for (int q = 0; q < CHY; ++q)
                if (b + q < ydeg) {
                  const double lp = e_lp[q];
                  if (ygo) { idm = L(idm, yv.va[q] + lp); imi = L(imi, yv.vb[q] + lp); }
                  if (mode == 2) imm = L(imm, yv.h[q][0] + lp);
                  if (cpairs) imm = L(imm, (yv.h[q][0] + X.lp[0]) + lp);
                }
#pragma unroll
              for (int q = 0; q < CHY; ++q) { e_src[q] = n_src[q]; e_lp[q] = n_lp[q]; }
            }
          }
          if (bpairs) {
            // the pairs of the row's inline transitions, row transition by row transition: with the column's inline transitions
            // (loaded with the step's sources), then with its further ones (the round above)
#pragma unroll
            for (int a = 0; a < K; ++a)
              if (a < xdeg) {
#pragma unroll
                for (int b = 0; b < K; ++b) imm = L(imm, (mv[a * K + b] + X.lp[a]) + Y.lp[b]);      // (K < ydeg)
#pragma unroll
                for (int r = 0; r < CHY; ++r)
                  if (K + r < ydeg) imm = L(imm, (yv.h[r][a] + X.lp[a]) + yl_first[r]);
              }
          }
          if (xdeg > K) {
            int e_src[CHX];
            double e_lp[CHX];
#pragma unroll
            for (int q = 0; q < CHX; ++q) { e_src[q] = xs_first[q]; e_lp[q] = xl_first[q]; }
            for (int a = K; a < xdeg; a += CHX) {
              int n_src[CHX];
              double n_lp[CHX];
#pragma unroll
              for (int q = 0; q < CHX; ++q) {
                n_src[q] = 0; n_lp[q] = 0.;
                if (a + CHX + q < xdeg) { n_src[q] = xin_src[X.in_b + a + CHX + q]; n_lp[q] = xin_lp[X.in_b + a + CHX + q]; }
              }
              if (!EARLY || a > K) x_loads(e_src, a);
#pragma unroll
              for (int q = 0; q < CHX; ++q)
                if (a + q < xdeg) {
                  const double lp = e_lp[q];
                  if (xgo) { imd = L(imd, xv.va[q] + lp); iiw = L(iiw, xv.vb[q] + lp); }
                  if (mode == 3) imm = L(imm, xv.g[q][0] + lp);
                  if (rpairs) {
#pragma unroll
                    for (int b = 0; b < K; ++b)
                      if (b < ydeg) imm = L(imm, (xv.g[q][b] + lp) + Y.lp[b]);
                  }
                }
#pragma unroll
              for (int q = 0; q < CHX; ++q) { e_src[q] = n_src[q]; e_lp[q] = n_lp[q]; }
            }
          }
          if (bpairs) {
            // ... and those of the row's further transitions
#pragma unroll
            for (int q = 0; q < CHX; ++q)
              if (K + q < xdeg) {
#pragma unroll
                for (int b = 0; b < K; ++b) imm = L(imm, (xv.g[q][b] + xl_first[q]) + Y.lp[b]);
#pragma unroll
                for (int r = 0; r < CHY; ++r)
                  if (K + r < ydeg) imm = L(imm, (xv.gg[q][r] + xl_first[q]) + yl_first[r]);
              }
          }
          if (mode == 1 && !pairs_inline && !bpairs) {
            // both states emit, the row has several in-transitions and the column more than K + CHY (or the row more than
            // K + CHX): all pairs in the reference's order, one at a time
            for (int a = 0; a < xdeg; ++a) {
              const RowRef rr = row_ref(ss, xin_src[X.in_b + a]);
              const double lpx = xin_lp[X.in_b + a];
              for (int b = 0; b < ydeg; ++b)
                imm = L(imm, (G4[slot_at(rr, yin_src[Y.in_b + b])] + lpx) + yin_lp[Y.in_b + b]);
            }
          }
          if (!xnull && yok) { imd += X.rootsub; iiw += X.ins; }
          if (!ynull && xok) { idm += Y.rootsub; imi += Y.ins; }
          if (mode == 1) imm += e;
          if (s == 0 && t == 0 && lane == 0) imm = 0.0;     // cell (0,0): lpStart() = 0 (reference src/forward.cpp:73)
          c.imm = imm; c.imd = imd; c.idm = idm; c.imi = imi; c.iiw = iiw;
        }
        HX_TR(3);      // accumulate (inline transitions and tails)
        // ---- the previous step's cell goes to memory now, behind this step's loads (unconditionally: an
        // idle lane writes -inf into padding or into an out-of-envelope cell, which holds -inf already) ----
        {
          M[pend_slot] = own10.imm;
          M[plane + pend_slot] = own10.imd;
          M[2 * plane + pend_slot] = own10.idm;
          M[3 * plane + pend_slot] = own10.imi;
          M[4 * plane + pend_slot] = own10.iiw;
          M[aggoff + pend_slot] = own10.g0;
          M[aggoff + plane + pend_slot] = own10.g1;
          M[aggoff + 2 * plane + pend_slot] = own10.g2;
          M[aggoff + 3 * plane + pend_slot] = own10.g3;
          M[aggoff + 4 * plane + pend_slot] = own10.g4;
        }
        if (act) {
          // outgoing sums, level by level (5 + 4 + 3 + 1 look-ups)
          typename LSE::Prep p0 = L.prep(c.imm + T[0][1], c.imd + T[1][1]);
          typename LSE::Prep p1 = L.prep(c.imm + T[0][4], c.imi + T[3][4]);
          typename LSE::Prep p2 = L.prep(c.imm + T[0][2], c.imd + T[1][2]);
          typename LSE::Prep p3 = L.prep(c.imm + T[0][3], c.imi + T[3][3]);
          typename LSE::Prep p4 = L.prep(c.imm + T[0][0], c.imd + T[1][0]);
          typename LSE::Piece c0 = L.fetch(p0), c1 = L.fetch(p1), c2 = L.fetch(p2), c3 = L.fetch(p3), c4 = L.fetch(p4);
          double g0 = L.finish(p0, c0), g1 = L.finish(p1, c1), g2 = L.finish(p2, c2);
          const double g3 = L.finish(p3, c3);
          double g4 = L.finish(p4, c4);
          p0 = L.prep(g0, c.idm + T[2][1]);
          p1 = L.prep(g1, c.iiw + T[4][4]);
          p2 = L.prep(g2, c.idm + T[2][2]);
          p4 = L.prep(g4, c.idm + T[2][0]);
          c0 = L.fetch(p0); c1 = L.fetch(p1); c2 = L.fetch(p2); c4 = L.fetch(p4);
          g0 = L.finish(p0, c0); g1 = L.finish(p1, c1); g2 = L.finish(p2, c2); g4 = L.finish(p4, c4);
          p0 = L.prep(g0, c.imi + T[3][1]);
          p2 = L.prep(g2, c.iiw + T[4][2]);
          p4 = L.prep(g4, c.imi + T[3][0]);
          c0 = L.fetch(p0); c2 = L.fetch(p2); c4 = L.fetch(p4);
          g0 = L.finish(p0, c0); g2 = L.finish(p2, c2); g4 = L.finish(p4, c4);
          g4 = L(g4, c.iiw + T[4][0]);
          c.g0 = g0; c.g1 = g1; c.g2 = g2; c.g3 = g3; c.g4 = g4;
        }
        HX_TR(4);      // stores issued, outgoing sums
        // ---- rotate: this step's cell becomes `own`, and the next lane's `up` ----
        own10 = c;
        pend_slot = own.base + ((unsigned)(t >> 1) << 7) + (unsigned)(t & 1);
        up_imm = wave_shr1(c.imm); up_imd = wave_shr1(c.imd); up_iiw = wave_shr1(c.iiw);
        up_g0 = wave_shr1(c.g0); up_g1 = wave_shr1(c.g1);
        // ---- publish, every 8th step: drain, then all stores issued so far (the cells of steps <= t-1)
        // are in memory, i.e. all 64 rows have completed the columns up to t - 1 - 63
        if ((t & 7) == 7) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          int done = t - 63;
          done = done > Cc ? Cc : done;
          if (dead) done = Cc;
          if (done > published) {
            published = done;
            publish(my_base + done);
          }
        }
        HX_TR(5);      // rotate, drain + publish
#ifdef HX_DAG_TRACE
        ++tr_steps;
#endif
      }
      // flush the last cell of the window
      {
        M[pend_slot] = own10.imm;
        M[plane + pend_slot] = own10.imd;
        M[2 * plane + pend_slot] = own10.idm;
        M[3 * plane + pend_slot] = own10.imi;
        M[4 * plane + pend_slot] = own10.iiw;
        M[aggoff + pend_slot] = own10.g0;
        M[aggoff + plane + pend_slot] = own10.g1;
        M[aggoff + 2 * plane + pend_slot] = own10.g2;
        M[aggoff + 3 * plane + pend_slot] = own10.g3;
        M[aggoff + 4 * plane + pend_slot] = own10.g4;
      }
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
      printf("trace job %d strip %d steps %d wait %lld issue %lld loads %lld accumulate %lld sums %lld rotate %lld\n", (int)blockIdx.x, s, tr_steps,
             tr_sum[0] / tr_steps, tr_sum[1] / tr_steps, tr_sum[2] / tr_steps, tr_sum[3] / tr_steps, tr_sum[4] / tr_steps, tr_sum[5] / tr_steps);
#endif
    // (a strip without any window still has to release the strip below)
    if (published < Cc) {
      published = Cc;
      publish(my_base + Cc);
    }
    if (MULTI && s == n_strips - 1) {
      // the last strip finishes last (every strip follows the one above): all of the pair's cells are in memory.  What END
      // reads may lie in other workgroups' strips: drop this CU's L1 first (agent-scope acquire = buffer_inv sc1).
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) *J.lp_end = dead ? __builtin_nan("") : forward_lp_end(J, ExactLse{exact_tab});
    }
  }
  if (MULTI) return;
  __syncthreads();
  if (threadIdx.x == 0) *J.lp_end = forward_lp_end(J, ExactLse{exact_tab});
}

// emission plane: one thread per matrix slot (stores coalesced); cells outside the envelope and
// cells of null states get 0 / -inf and are never added to a finite value
// FAST: the table policies other than the exact one (fast, truncating, scaled probabilities) evaluate the emission terms of
// profiles that are not leaves with the fast table too - the arithmetic their fills use for every other sum (the exact
// operator costs twice the instructions and a gather in the 800 KB table per term: 84 terms per cell with four mixture
// components).  Leaf profiles, whose best paths are held to the reference's in the traceback-identity sweeps, keep the exact
// operator in every policy.
template <bool FAST>
__global__ void __launch_bounds__(FAST ? 1024 : 256) k_emission_plane(const DevJob* __restrict__ jobs, const double* __restrict__ tab,
                                                                      const double* __restrict__ fast_tab) {
  __shared__ __attribute__((aligned(16))) double ftab[FAST ? (HX_FAST_INTERVALS + 1) * 2 : 2];
  if (FAST) {
    for (int k = threadIdx.x; k < (HX_FAST_INTERVALS + 1) * 2; k += blockDim.x) ftab[k] = fast_tab[k];
    __syncthreads();
  }
  const DevJob& J = jobs[blockIdx.y];
  if (!J.emis_plane) return;
  const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= J.plane) return;
  const int64_t strip = slot / J.strip_stride;
  const int64_t in = slot - strip * J.strip_stride;
  const int l = (int)((in & 127) >> 1);
  const int t = (int)((in >> 7) << 1) + (int)(in & 1);
  const int i = (int)(strip << 6) + l, j = t - l;
  double e = 0.0;
  if (i < J.n_rows && j >= 0 && j < J.n_cols) {
    const int cx = J.x.cls[i], cy = J.y.cls[j];
    if (cx < 0 || cy < 0) e = HX_NEG_INF;
    else if (in_envelope(J, i, j)) {
      const double* sx = J.x.subc + (size_t)cx * J.CA;
      const double* sy = J.y.subc + (size_t)cy * J.CA;
      if (FAST && !J.leaf_like) e = emission_rows(J, sx, sy, FastLse::make((const double*)ftab));
      else e = emission_rows(J, sx, sy, ExactLse{tab});
    }
  }
  J.emis_plane[slot] = e;
}

__global__ void k_fill_neg_inf(double* __restrict__ p, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) p[k] = HX_NEG_INF;
}

}  // namespace

// fast: the batch runs one of the policies with the fast table (see k_emission_plane)
void launch_emission_plane(const DevJob* d_jobs, int n_jobs, int64_t max_plane, Tab8 tab8, Tab16 fast_tab, bool fast, hipStream_t st) {
  const double* tab = tab8.p;
  if (max_plane <= 0) return;
  const int tpb = fast ? 1024 : 256;
  for (int j0 = 0; j0 < n_jobs; j0 += 32768) {       // (grid.y is limited to 65535)
    const int n = n_jobs - j0 < 32768 ? n_jobs - j0 : 32768;
    dim3 grid((unsigned)((max_plane + tpb - 1) / tpb), (unsigned)n);
    if (fast) hipLaunchKernelGGL(k_emission_plane<true>, grid, dim3(tpb), 0, st, d_jobs + j0, tab, fast_tab.p);
    else hipLaunchKernelGGL(k_emission_plane<false>, grid, dim3(tpb), 0, st, d_jobs + j0, tab, (const double*)nullptr);
  }
}

void launch_fill_neg_inf(double* p, int64_t n, hipStream_t st) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_fill_neg_inf, dim3(2048), dim3(256), 0, st, p, n);
}

static int dag_waves(int max_rows, int cap) {
  int w = (max_rows + 63) / 64;
  if (w < 1) w = 1;
  if (w > cap) w = cap;
  return w;
}

// multi > 1: one or two pairs of many strips, each dealt to `multi` workgroups of `multi_waves` waves (MULTI instantiation);
// `counters`: 256 zeroed ints per pair
int launch_forward_dag_pipe(const DevJob* d_jobs, int n_jobs, int max_rows, Tab8 tab8, Tab16 tab16,
                            bool fast, int multi, int multi_waves, int* counters, hipStream_t st) {
  const double* tab = tab8.p;
  const double* fast_tab = tab16.p;
  if (multi > 1) {
    const dim3 gm(n_jobs * multi), bm(multi_waves * 64);
    if (fast) hipLaunchKernelGGL((k_forward_dag_pipe<FastLse, true, true>), gm, bm, 0, st, d_jobs, tab, fast_tab, multi, counters, multi_patience());
    else hipLaunchKernelGGL((k_forward_dag_pipe<ExactLse3, false, true>), gm, bm, 0, st, d_jobs, tab, fast_tab, multi, counters, multi_patience());
    return 0;
  }
  const dim3 g(n_jobs), b(dag_waves(max_rows, HX_DAGF_MAX_WAVES) * 64);
  if (fast) {
    HX_CHECK_LDS((k_forward_dag_pipe<FastLse, true>), 0, "k_forward_dag_pipe<fast>");
    hipLaunchKernelGGL((k_forward_dag_pipe<FastLse, true>), g, b, 0, st, d_jobs, tab, fast_tab, 1, nullptr, 0);
  } else {
    HX_CHECK_LDS((k_forward_dag_pipe<ExactLse3, false>), 0, "k_forward_dag_pipe<exact>");
    hipLaunchKernelGGL((k_forward_dag_pipe<ExactLse3, false>), g, b, 0, st, d_jobs, tab, fast_tab, 1, nullptr, 0);
  }
  return 0;
}

// `records`: every pair of the launch has scratch planes (DevJob::agg) that the Forward fill no longer needs and that hold
// at least 20 doubles (HX_BWD_REC_BYTES) per state of its two profiles: the state-record formulation (backward_cell_rec)
int launch_backward_dag_pipe(const DevJob* d_jobs, int n_jobs, int max_rows, Tab8 tab8, Tab16 tab16,
                             bool fast, bool records, int multi, int multi_waves, hipStream_t st) {
  const double* tab = tab8.p;
  const double* fast_tab = tab16.p;
  const dim3 g(n_jobs), b(dag_waves(max_rows, HX_DAG_MAX_WAVES) * 64);
  // one or two pairs of more than sixteen strips: several workgroups per pair (k_backward_dag_multi); the caller has zeroed
  // the progress counters
  if (records && multi > 1) {
    const dim3 gm(n_jobs * multi), bm((multi_waves > 0 && multi_waves <= HX_DAG_REC_WAVES ? multi_waves : HX_DAG_REC_WAVES) * 64);
    if (fast) hipLaunchKernelGGL((k_backward_dag_multi<FastLse, true>), gm, bm, 0, st, d_jobs, tab, fast_tab, multi, multi_patience());
    else hipLaunchKernelGGL((k_backward_dag_multi<ExactLse3, false>), gm, bm, 0, st, d_jobs, tab, fast_tab, multi, multi_patience());
    return 0;
  }
  if (records) {
    const dim3 br(dag_waves(max_rows, HX_DAG_REC_WAVES) * 64);
    if (fast) hipLaunchKernelGGL((k_fill_dag<1, FastLse, true, true>), g, br, 0, st, d_jobs, tab, fast_tab);
    else hipLaunchKernelGGL((k_fill_dag<1, ExactLse3, false, true>), g, br, 0, st, d_jobs, tab, fast_tab);
    return 0;
  }
  if (fast) hipLaunchKernelGGL((k_fill_dag<1, FastLse, true>), g, b, 0, st, d_jobs, tab, fast_tab);
  else hipLaunchKernelGGL((k_fill_dag<1, ExactLse3, false>), g, b, 0, st, d_jobs, tab, fast_tab);
  return 0;
}

}  // namespace hx
