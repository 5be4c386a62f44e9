// The cell of the leaf-like pair DP in log space, shared by the strip pipeline (hx_chain.hip) and the banded
// rotating-row sweep (hx_band.hip): reference src/forward.cpp:98-199 (Forward) and :1018-1065 (Backward) specialised to
// in-degree / out-degree 1 profiles whose interior states all emit.
#pragma once
#include <hip/hip_runtime.h>
#include "hx_policy.h"

namespace hx {

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
__device__ __forceinline__ C5 leaf_cell(const double (*T)[6], const LSE& L, const XLeaf& X, const d4v& Y,
                                        double e, double pj, const C5& up, const C5& left, const C5& diag) {
  // the five left-nested n-ary sums of reference src/forward.cpp:103-115,139-150,171-180,
  // evaluated level by level: 5 + 4 + 3 + 1 look-ups, each level's fetches issued together
  typename LSE::Prep p0 = L.prep(up.imm + T[0][1], up.imd + T[1][1]);
  typename LSE::Prep p1 = L.prep(up.imm + T[0][4], up.imi + T[3][4]);
  typename LSE::Prep p2 = L.prep(left.imm + T[0][2], left.imd + T[1][2]);
  typename LSE::Prep p3 = L.prep(left.imm + T[0][3], left.imi + T[3][3]);
  typename LSE::Prep p4 = L.prep(diag.imm + T[0][0], diag.imd + T[1][0]);
  typename LSE::Piece c0 = L.fetch(p0), c1 = L.fetch(p1), c2 = L.fetch(p2), c3 = L.fetch(p3), c4 = L.fetch(p4);
  double a_imd = L.finish(p0, c0);
  double a_iiw = L.finish(p1, c1);
  double a_idm = L.finish(p2, c2);
  const double a_imi = L.finish(p3, c3);
  double a_imm = L.finish(p4, c4);

  p0 = L.prep(a_imd, up.idm + T[2][1]);
  p1 = L.prep(a_iiw, up.iiw + T[4][4]);
  p2 = L.prep(a_idm, left.idm + T[2][2]);
  p4 = L.prep(a_imm, diag.idm + T[2][0]);
  c0 = L.fetch(p0); c1 = L.fetch(p1); c2 = L.fetch(p2); c4 = L.fetch(p4);
  a_imd = L.finish(p0, c0);
  a_iiw = L.finish(p1, c1);
  a_idm = L.finish(p2, c2);
  a_imm = L.finish(p4, c4);

  p0 = L.prep(a_imd, up.imi + T[3][1]);
  p2 = L.prep(a_idm, left.iiw + T[4][2]);
  p4 = L.prep(a_imm, diag.imi + T[3][0]);
  c0 = L.fetch(p0); c2 = L.fetch(p2); c4 = L.fetch(p4);
  a_imd = L.finish(p0, c0);
  a_idm = L.finish(p2, c2);
  a_imm = L.finish(p4, c4);

  a_imm = L(a_imm, diag.iiw + T[4][0]);

  const double q1 = Y.w + pj;      // y state ready (or y empty), and the cell exists
  const double q2 = X.pen + pj;    // x state ready (or x empty), and the cell exists
  C5 r;
  r.imd = ((a_imd + X.lp) + X.rootsub) + q1;
  r.iiw = ((a_iiw + X.lp) + X.ins) + q1;
  r.idm = ((a_idm + Y.x) + Y.y) + q2;
  r.imi = ((a_imi + Y.x) + Y.z) + q2;
  r.imm = ((a_imm + X.lp + Y.x) + e) + pj;
  return r;
}

// Backward counterpart (reference src/forward.cpp:1018-1065 for in-degree/out-degree-1 leaf-like
// profiles), in mirrored coordinates: `up` = B(i+1,j), `left` = B(i,j+1), `diag` = B(i+1,j+1).
// X holds the constants of x state i+1 (lpTrans of i->i+1, rootsubx, insx, emission class) and the
// ready-penalty of state i; Y likewise for the y side.  The reference accumulates, in this order,
// the xy-absorbing term D, the x-absorbing terms d1x,d2x and the y-absorbing terms d1y,d2y into
// each state with log_accum_exp; the first accumulate into -inf is exact, the rest are left-nested.
template <class LSE>
__device__ __forceinline__ C5 leaf_cell_bwd(const double (*T)[6], const LSE& L, const XLeaf& X, const d4v& Y,
                                            double e, double pj, const C5& up, const C5& left, const C5& diag) {
  const double D = ((X.lp + Y.x) + e) + diag.imm;
  const double d1x = ((X.lp + X.rootsub) + up.imd) + Y.w;     // gated by the y state being ready
  const double d2x = ((X.lp + X.ins) + up.iiw) + Y.w;
  const double d1y = ((Y.x + Y.y) + left.idm) + X.pen;        // gated by the x state being ready
  const double d2y = ((Y.x + Y.z) + left.imi) + X.pen;
  typename LSE::Prep p0 = L.prep(T[0][0] + D, T[0][1] + d1x);
  typename LSE::Prep p1 = L.prep(T[1][0] + D, T[1][1] + d1x);
  typename LSE::Prep p2 = L.prep(T[2][0] + D, T[2][1] + d1x);
  typename LSE::Prep p3 = L.prep(T[3][0] + D, T[3][1] + d1x);
  typename LSE::Prep p4 = L.prep(T[4][0] + D, T[4][4] + d2x);
  typename LSE::Piece c0 = L.fetch(p0), c1 = L.fetch(p1), c2 = L.fetch(p2), c3 = L.fetch(p3), c4 = L.fetch(p4);
  double imm = L.finish(p0, c0), imd = L.finish(p1, c1), idm = L.finish(p2, c2), imi = L.finish(p3, c3), iiw = L.finish(p4, c4);
  p0 = L.prep(imm, T[0][4] + d2x);
  p1 = L.prep(imd, T[1][2] + d1y);
  p2 = L.prep(idm, T[2][2] + d1y);
  p3 = L.prep(imi, T[3][4] + d2x);
  p4 = L.prep(iiw, T[4][2] + d1y);
  c0 = L.fetch(p0); c1 = L.fetch(p1); c2 = L.fetch(p2); c3 = L.fetch(p3); c4 = L.fetch(p4);
  imm = L.finish(p0, c0); imd = L.finish(p1, c1); idm = L.finish(p2, c2); imi = L.finish(p3, c3); iiw = L.finish(p4, c4);
  p0 = L.prep(imm, T[0][2] + d1y);
  p3 = L.prep(imi, T[3][3] + d2y);
  c0 = L.fetch(p0); c3 = L.fetch(p3);
  imm = L.finish(p0, c0); imi = L.finish(p3, c3);
  imm = L(imm, T[0][3] + d2y);
  C5 r;
  r.imm = imm + pj; r.imd = imd + pj; r.idm = idm + pj; r.imi = imi + pj; r.iiw = iiw + pj;
  return r;
}

}  // namespace hx
