// Device log-sum-exp, bit-identical to the reference's table-interpolated
// log_sum_exp (reference src/logsumexp.h:42-84).
//
// Build this translation unit with -ffp-contract=off: the reference is compiled for
// baseline x86-64 (no FMA), so `f0 + df * w` and `x - n * 1e-4` round twice.
#pragma once
#include <hip/hip_runtime.h>

namespace hx {

#define HX_NEG_INF (-__builtin_inf())

// Correctly rounded a / 1e-4 without the generic fp64 division sequence.
// y = RN(1/1e-4) is exactly 1e4; q0 = RN(a*y) is a faithful quotient, the residual
// r = a - 1e-4*q0 is exact in one FMA, and RN(q0 + r*y) is the correctly rounded
// quotient (Markstein's theorem; checked against IEEE division on every bin boundary and its
// neighbours, random arguments, remainders and bit patterns -- tests/test_host_numerics.py,
// tests/csrc/div1em4_check.c).  Holds while the residual does not underflow (|a| >= 1e-290):
// the arguments here are differences of log-probabilities and remainders of them, zero or
// many orders of magnitude above that.
__device__ __forceinline__ double div_by_1em4(double a) {
  const double q0 = a * 1e4;
  const double r = __builtin_fma(-1e-4, q0, a);
  return __builtin_fma(r, 1e4, q0);
}

// log(1 + exp(-x)) for x >= 0 by table lookup + linear interpolation;
// 0 for x >= 10, NaN or inf (reference src/logsumexp.h:42-64).
// Branch-free: for x >= 10, +inf or NaN the lookup is redirected to entry 0 (one
// broadcast address, no extra L2 traffic) and the result replaced by 0.
__device__ __forceinline__ double lse_unary(double x, const double* __restrict__ tab) {
  const bool in = x < 10.0;             // false for x >= 10, +inf and NaN
  const int n = in ? (int)div_by_1em4(x) : 0;
  // lookup[n], lookup[n+1]: one 16-byte access (8-byte aligned)
  const double f0 = tab[n];
  const double f1 = tab[n + 1];
  const double dx = x - ((double)n * 1e-4);
  const double df = f1 - f0;
  const double ret = f0 + df * div_by_1em4(dx);
  return in ? ret : 0.0;
}

// reference src/logsumexp.h:66-84.  a == b (incl. -inf,-inf) gives diff 0; here the
// -inf,-inf case yields diff NaN -> unary 0 -> -inf + 0: the same value.
__device__ __forceinline__ double lse(double a, double b, const double* __restrict__ tab) {
  const double mx = __builtin_fmax(a, b);
  const double diff = mx - __builtin_fmin(a, b);   // NaN for (-inf,-inf): handled as "no lookup"
  return mx + lse_unary(diff, tab);
}

}  // namespace hx
