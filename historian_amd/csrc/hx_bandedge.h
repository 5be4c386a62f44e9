// Row 0 of a banded leaf pair beyond its band, under the scaled-probability policies (hx_band.hip, hx_band2.hip).
//
// Row 0 (x START) is inside the envelope throughout (reference src/forward.h:92-98), and beyond the band its cells are the
// chain IDM(0,j) = (IDM(0,j-1) + T[IDM][IDM]) + rootsuby[j] (IMI likewise with T[IMI][IMI] and insy): every other term of the
// reference's sums (src/forward.cpp:139-180) is -inf there.  The table policies evaluate it as that chain, bit for bit; the
// scaled-probability policies - whose contract is 1e-9, not bits - as a prefix sum over the wavefront's lanes, 64 columns at
// a time with the carry in a scalar: 2 000 dependent additions per pair become 32 short passes.  The sums are of the order
// of 1e4 and an addition rounds to 1e-12, so the two orders agree to ~1e-15 relative.
#pragma once
#include <hip/hip_runtime.h>
#include "hx_common.h"

namespace hx {

// One block of 64 columns, j = j0 + lane: lrs / lin = rootsuby[j] / insy[j] (anything for j >= n_cols), carry_* = the running
// sums up to column j0 - 1 (0 in front of the first block; updated).  k_idm / k_imi = the lane's cell values (-inf at column 0).
__device__ __forceinline__ void row0_chain_block(const int j0, const int lane, const int n_cols, const double lrs, const double lin,
                                                 const double T02, const double T03, const double T22, const double T33, const double pen0,
                                                 double& carry_idm, double& carry_imi, double& k_idm, double& k_imi) {
  const int j = j0 + lane;
  // column 0 holds no chain cell; column 1 starts the chain out of cell (0,0).IMM = 0 (src/forward.cpp:73)
  double a = (j == 0 || j >= n_cols) ? 0. : (j == 1 ? T02 : T22) + lrs;
  double b = (j == 0 || j >= n_cols) ? 0. : (j == 1 ? T03 : T33) + lin;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double ua = __shfl_up(a, d, 64), ub = __shfl_up(b, d, 64);
    if (lane >= d) { a += ua; b += ub; }
  }
  a += carry_idm; b += carry_imi;
  carry_idm = __shfl(a, 63, 64); carry_imi = __shfl(b, 63, 64);
  k_idm = j >= 1 ? a + pen0 : HX_NEG_INF;       // pen0: x START ready (or x empty) 0, else -inf
  k_imi = j >= 1 ? b + pen0 : HX_NEG_INF;
}

}  // namespace hx
