// Log-sum-exp policies and address-space helpers shared by the strip-pipeline kernels
// (hx_chain.hip: linear-chain profiles, hx_dag.hip: general profiles).
#pragma once
#include <hip/hip_runtime.h>
#include "hx_device.h"
#include "hx_lse.h"

namespace hx {


// global-address-space views of pointers that were loaded from the job table (the compiler
// would otherwise have to use flat_* instructions, which tie up both memory counters)
#define HX_GLOBAL __attribute__((address_space(1)))
#define HX_LDS __attribute__((address_space(3)))
template <class T> __device__ __forceinline__ HX_GLOBAL T* as_global(T* p) { return (HX_GLOBAL T*)p; }
template <class T> __device__ __forceinline__ const HX_GLOBAL T* as_global(const T* p) { return (const HX_GLOBAL T*)p; }

typedef double d4v __attribute__((ext_vector_type(4)));

struct alignas(16) FastPiece { double c0; float c1, c2; };   // 16 bytes: one ds_read_b128 per log-sum-exp

// v_max_f64 / v_min_f64 without the canonicalisation moves the builtins add for sNaN inputs
// (v_min_f64 returns the non-NaN operand, which the clamping below relies on).
__device__ __forceinline__ double vmax(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double vmin(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// The same two instructions through the compiler's own operations, for operands that are results of arithmetic (never a
// signalling NaN: no canonicalising move is added).  Behind an inline-asm instruction the compiler pads every first use of the
// result with an s_nop - nine issue slots per step in the truncating sums of the scaled-probability fills.
__device__ __forceinline__ double fmax_plain(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ double fmin_plain(double a, double b) { return __builtin_fmin(a, b); }

// A log-sum-exp policy is used in three phases so that the independent table look-ups of a
// cell can be issued back to back:  prep (index arithmetic) -> fetch (the memory access) ->
// finish (interpolation + add).  operator() runs the three in sequence.
struct FastLse {
  const FastPiece* lds;   // [HX_FAST_INTERVALS + 1] quadratic pieces in t in [0,1); last piece all zero
  struct Prep { double mx, t; int k; };
  typedef FastPiece Piece;
  // Branch- and select-free.  d = |a - b| is scaled to table units and clamped to the
  // all-zero guard piece, which yields T = 0 for d >= 10, d = +inf and d = NaN (-inf - -inf):
  // exactly the reference's truncation.
  __device__ __forceinline__ Prep prep(double a, double b) const {
    Prep p;
    p.mx = vmax(a, b);
    const double d = a - b;
    const double s = vmin(__builtin_fabs(d) * (HX_FAST_INTERVALS / 10.0), (double)HX_FAST_INTERVALS);
    p.k = (int)s;
    p.t = __builtin_amdgcn_fract(s);
    return p;
  }
  __device__ __forceinline__ Piece fetch(const Prep& p) const {
    // exactly one 16-byte LDS access (ds_read_b128): c0 is the first double, the two fp32
    // coefficients are the halves of the second
    typedef double d2v __attribute__((ext_vector_type(2)));
    const d2v v = reinterpret_cast<const d2v*>(lds)[p.k];
    Piece c;
    c.c0 = v.x;
    c.c1 = __uint_as_float((unsigned)__double2loint(v.y));
    c.c2 = __uint_as_float((unsigned)__double2hiint(v.y));
    return c;
  }
  __device__ __forceinline__ double finish(const Prep& p, const Piece& c) const {
    return p.mx + __builtin_fma(__builtin_fma((double)c.c2, p.t, (double)c.c1), p.t, c.c0);
  }
  __device__ __forceinline__ double operator()(double a, double b) const {
    const Prep p = prep(a, b);
    return finish(p, fetch(p));
  }
  static __device__ __forceinline__ FastLse make(const double* p) { return FastLse{reinterpret_cast<const FastPiece*>(p)}; }
  static __device__ __forceinline__ FastLse make(const double* p, const double*, int) { return make(p); }
};

// The reference's table operator (hx_lse.h), phased the same way; bit-identical to lse().
struct ExactLse3 {
  // 16-byte aligned pairs {lookup[n], lookup[n+1] - lookup[n]} (built by hx_init from the host's table: the
  // difference is the same IEEE subtraction the reference performs at every look-up), so that a look-up is one
  // naturally aligned 16-byte gather.
  const double* __restrict__ tab;
  // Optionally the head of the table, lookup[0 .. n_lds], also sits in LDS.  The exact fill is bound by the
  // L1's miss handling (about 0.5 gather-lanes per clock and CU: PMC, tools/pmc_exact.sh), and the small
  // differences are by far the most frequent (15 % of all look-ups fall below 0.5, a third hit no entry at
  // all because the difference is >= 10): whatever LDS serves, the L1 does not see.
  const double* lds;
  int n_lds;
  struct Prep { double mx, x; int n; bool in; };
  struct Piece { double f0, df; };
  __device__ __forceinline__ Prep prep(double a, double b) const {
    Prep p;
    p.mx = vmax(a, b);
    p.x = p.mx - vmin(a, b);                 // NaN for (-inf,-inf): "no lookup"
    p.in = p.x < 10.0;
    p.n = p.in ? (int)div_by_1em4(p.x) : 0;
    return p;
  }
  __device__ __forceinline__ Piece fetch(const Prep& p) const {
    typedef double d2v __attribute__((ext_vector_type(2)));
    if (p.n < n_lds) {
      const HX_LDS double* l = (const HX_LDS double*)lds;
      const double f0 = l[p.n], f1 = l[p.n + 1];
      return Piece{f0, f1 - f0};
    }
    const d2v v = reinterpret_cast<const d2v*>(tab)[p.n];
    return Piece{v.x, v.y};
  }
  __device__ __forceinline__ double finish(const Prep& p, const Piece& c) const {
    const double dx = p.x - ((double)p.n * 1e-4);
    const double ret = c.f0 + c.df * div_by_1em4(dx);
    return p.mx + (p.in ? ret : 0.0);
  }
  __device__ __forceinline__ double operator()(double a, double b) const {
    const Prep p = prep(a, b);
    return finish(p, fetch(p));
  }
  static __device__ __forceinline__ ExactLse3 make(const double* p) { return ExactLse3{p, nullptr, 0}; }
  static __device__ __forceinline__ ExactLse3 make(const double* p, const double* lds_head, int n) { return ExactLse3{p, lds_head, n}; }
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

}  // namespace hx
