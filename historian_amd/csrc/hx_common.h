// Device helpers shared by the general (DAG) and the chain kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "hx_device.h"
#include "hx_lse.h"

namespace hx {

// log-sum-exp policy: the reference's table-interpolated operator, bit for bit
struct ExactLse {
  const double* __restrict__ tab;
  __device__ __forceinline__ double operator()(double a, double b) const { return lse(a, b, tab); }
  static __device__ __forceinline__ ExactLse make(const double* p) { return ExactLse{p}; }
};

// computeLogProbAbsorb (reference src/forward.h:112-124) from two leftMultiplied rows
template <class LSE>
__device__ __forceinline__ double emission_rows(const DevJob& J, const double* sx, const double* sy, const LSE& L) {
  double lip = HX_NEG_INF;
  for (int cpt = 0; cpt < J.C; ++cpt) {
    double inner = HX_NEG_INF;
    for (int a = 0; a < J.A; ++a) {
      const int k = cpt * J.A + a;
      inner = L(inner, J.log_root[k] + (sx[k] + sy[k]));
    }
    lip = L(lip, inner);
  }
  return lip;
}

__device__ __forceinline__ double emission(const DevJob& J, int i, int j, const double* __restrict__ tab) {
  if (J.emis) {
    // a hand-edited profile may route an absorbing transition into a null state
    // (reference t/testnullforward.cpp:37-39); such a pair emits nothing
    const int cx = J.x.cls[i], cy = J.y.cls[j];
    return (cx < 0 || cy < 0) ? HX_NEG_INF : J.emis[(size_t)cx * J.y.n_cls + cy];
  }
  return emission_rows(J, J.x.sub + (size_t)i * J.CA, J.y.sub + (size_t)j * J.CA, ExactLse{tab});
}

__device__ __forceinline__ bool in_envelope(const DevJob& J, int i, int j) {
  if ((J.x.flags[i] | J.y.flags[j]) & F_EDGE) return true;
  if (J.max_dist < 0) return true;
  int d = J.x.env[i] - J.y.env[j];
  d = d < 0 ? -d : d;
  return d <= J.max_dist;
}

// Slot of Forward cell (i, j) inside a state plane, or -1 when a band-compressed job (HX_BAND_COMPRESSED) does not store
// it: such a job keeps, per 64-row strip, only the step windows the fill sweeps (DevJob::fwd_windows / strip_base).
__device__ __forceinline__ int64_t stored_slot(const DevJob& J, int i, int j) {
  if (!J.strip_base) return cell_slot_blk(J.strip_stride, J.blk, i, j);
  const int s = i >> 6, l = i & (HX_STRIP - 1), t = j + l;
  const int32_t* w = J.fwd_windows + 4 * s;
  if (t >= w[0] && t < w[1]) return J.strip_base[2 * s] + ((int64_t)((t - w[0]) >> 1) << 7) + (l << 1) + (t & 1);
  if (t >= w[2] && t < w[3]) return J.strip_base[2 * s + 1] + ((int64_t)((t - w[2]) >> 1) << 7) + (l << 1) + (t & 1);
  return -1;
}

struct Cell5 { double v[5]; };

__device__ __forceinline__ Cell5 load_cell(const double* __restrict__ m, int64_t plane, int64_t slot) {
  Cell5 c;
#pragma unroll
  for (int s = 0; s < 5; ++s) c.v[s] = m[s * plane + slot];
  return c;
}

// transitions into EEE (reference src/forward.cpp:205-220)
template <class LSE>
__device__ inline double forward_lp_end(const DevJob& J, const LSE& L) {
  double lp_end = HX_NEG_INF;
  const int xe = J.x.n - 1, ye = J.y.n - 1;
  for (int tx = J.x.in_off[xe]; tx < J.x.in_off[xe + 1]; ++tx)
    for (int ty = J.y.in_off[ye]; ty < J.y.in_off[ye + 1]; ++ty) {
      const int64_t slot = stored_slot(J, J.x.in_src[tx], J.y.in_src[ty]);
      if (slot < 0) continue;                      // (band-compressed storage: not stored = -inf, contributes nothing)
      const Cell5 s = load_cell(J.fwd, J.plane, slot);
      double a = L(s.v[0] + J.T[0][5], s.v[1] + J.T[1][5]);
      a = L(a, s.v[2] + J.T[2][5]);
      a = L(a, s.v[3] + J.T[3][5]);
      a = L(a, s.v[4] + J.T[4][5]);
      lp_end = L(lp_end, a + J.x.in_lp[tx] + J.y.in_lp[ty]);
    }
  return lp_end;
}

}  // namespace hx
