// Guide-alignment Viterbi: the reference's QuickAlignMatrix fill (src/quickalign.cpp:63-96) for a
// batch of independent sequence pairs -- the step before the Forward/Backward fills (SURVEY section
// 8f, row N1): a 3-state (match / insert / delete) max-plus pair DP with free end gaps,
//
//   mat(i,j) = max(mat(i-1,j-1)+m2m, del(i-1,j-1)+d2m, ins(i-1,j-1)+i2m, start+startGap(i,j)) + sub[x_i][y_j]
//   ins(i,j) = max(ins(i,j-1)+i2i, mat(i,j-1)+m2i)
//   del(i,j) = max(ins(i-1,j)+i2d, del(i-1,j)+d2d, mat(i-1,j)+m2d)
//   end      = max_{i,j} mat(i,j) + endGap(i,j)          (first maximum in (j,i) order wins)
//
// over the cells of a DiagonalEnvelope (a set of diagonals i-j; everything else reads as -inf).
// Only adds and maxima in fp64: bit-identical to the reference by construction (compiled with
// -ffp-contract=off so that gapOpen + n*gapExtend rounds twice, as on the host).
//
// Same strip pipeline as the Forward chain kernel (hx_chain.hip): one workgroup per pair, 64-row
// strips dealt round-robin to W waves, lane <-> row, step <-> anti-diagonal, every wave on its own
// clock; `left` is the lane's own previous cell, `up` and `diag` are the previous lane's cells of
// one and two steps ago (DPP wave_shr:1); the strip above's last row is block-loaded 64 columns at
// a time once that wave's LDS progress counter allows.  The matrix (3 planes, strip-skewed layout
// of hx_device.h) is written once, 24 B per cell, and kept for the traceback.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "hx_device.h"
#include "hx_lse.h"
#include "hx_policy.h"
#include "hx_kernels.h"

namespace hx {

namespace {

struct Q3 { double mat, ins, del; };
__device__ __forceinline__ Q3 q3_neg_inf() { return Q3{HX_NEG_INF, HX_NEG_INF, HX_NEG_INF}; }

__device__ __forceinline__ double shr1_keep0(double old, double v) {
  int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), 0x138, 0xf, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ Q3 shr1_keep0(const Q3& o, const Q3& c) {
  return Q3{shr1_keep0(o.mat, c.mat), shr1_keep0(o.ins, c.ins), shr1_keep0(o.del, c.del)};
}
__device__ __forceinline__ double lane_value(double v, int l) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
// std::max(a, b) = (a < b) ? b : a.  One v_max_f64 gives the same value for every pair of non-NaN
// operands except max(-0.0, +0.0) (sign of zero); the scores are sums of logarithms and never -0.0.
__device__ __forceinline__ double dmax(double a, double b) { return vmax(a, b); }

#define HX_QA_LAG 16
#define HX_QA_MAX_ALPH 31

// COLG: the per-column constants do not fit LDS (y longer than HX_QA_LDS_COLS): they live in a per-pair
// global scratch instead (L1/L2 resident; a slower path for very long sequences)
template <int W, bool FULL, bool COLG>
__global__ void __launch_bounds__(W * 64) k_quickalign(const DevQuick* __restrict__ jobs, const int max_cols) {
  __shared__ volatile int prog[W];
  __shared__ double sub[(HX_QA_MAX_ALPH + 1) * (HX_QA_MAX_ALPH + 1)];   // padded with a zero row / column for invalid tokens
  // per column (dynamic LDS, sized for the longest y of the batch): {startGap y part, endGap y part} and the
  // row offset of the column's token into `sub`
  extern __shared__ __attribute__((aligned(16))) double col_lds[];
  const DevQuick& J = jobs[blockIdx.x];
  const int A1 = J.alph + 1;
  typedef double d2v __attribute__((ext_vector_type(2)));
  d2v* colgap = COLG ? reinterpret_cast<d2v*>(J.col_scratch) : reinterpret_cast<d2v*>(col_lds);
  int* coltok = COLG ? reinterpret_cast<int*>(J.col_scratch + 2 * (size_t)J.ylen) : reinterpret_cast<int*>(col_lds + 2 * (size_t)max_cols);
  for (int k = threadIdx.x; k < A1 * A1; k += W * 64) {
    const int a = k / A1, b = k - a * A1;
    sub[k] = (a < J.alph && b < J.alph) ? J.submat[a * J.alph + b] : 0.0;
  }
  {
    const double gap_open = J.sc[8], gap_extend = J.sc[9], no_gap = J.sc[10];
    const int Cc = J.ylen;
    for (int c = threadIdx.x; c < Cc; c += W * 64) {
      const int j = c + 1;     // startGapScore / endGapScore, y part (src/quickalign.h:57-66; SeqIdx is unsigned)
      const double sgy = (j == 1) ? no_gap : gap_open + (double)(unsigned)(j - 2) * gap_extend;
      const double egy = (j == Cc) ? no_gap : gap_open + (double)(unsigned)(Cc - j - 2) * gap_extend;
      colgap[c] = d2v{sgy, egy};
      const int yt = J.ytok[c];
      coltok[c] = yt < 0 ? J.alph : yt;
    }
  }
  if (threadIdx.x < W) prog[threadIdx.x] = 0;
  __syncthreads();
  volatile HX_LDS int* progp = (volatile HX_LDS int*)prog;
  const int R = J.xlen, Cc = J.ylen;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t plane = J.plane, ss = J.strip_stride;
  const int blk = J.blk;
  HX_GLOBAL double* __restrict__ M = as_global(J.cells);
  const HX_GLOBAL uint8_t* in_env = as_global(J.in_env);
  const double m2m = J.sc[0], m2i = J.sc[1], m2d = J.sc[2], i2i = J.sc[3], i2m = J.sc[4], i2d = J.sc[5], d2d = J.sc[6],
               d2m = J.sc[7], gap_open = J.sc[8], gap_extend = J.sc[9], no_gap = J.sc[10];
  const int n_strips = (R + 63) >> 6;
  const int prev_wave = (wave + W - 1) % W;
  const int nsteps = Cc + 63;

  for (int s = wave; s < n_strips; s += W) {
    const int row0 = s << 6;
    const int r = row0 + lane;                    // row r <-> x position i = r + 1
    const bool rvalid = r < R;
    const int i = r + 1;
    int xt = rvalid ? J.xtok[r] : -1;
    if (xt < 0) xt = J.alph;
    const int sub_row = xt * A1;
    // startGapScore / endGapScore, x part (src/quickalign.h:57-66; SeqIdx is unsigned: xLen-i-2 wraps)
    const double sgx = (i == 1) ? no_gap : gap_open + (double)(unsigned)(i - 2) * gap_extend;
    const double egx = (i == R) ? no_gap : gap_open + (double)(unsigned)(R - i - 2) * gap_extend;
    double best = HX_NEG_INF;
    int best_j = 0;
    Q3 own = q3_neg_inf();                        // (r, c-1)
    Q3 u1 = q3_neg_inf(), u2 = q3_neg_inf();      // (r-1, c) and (r-1, c-1)
    Q3 bnd = q3_neg_inf();                        // 64 columns of the strip above's last row
    const bool has_above = s > 0;
    const int above_base = ((s - 1) / W) * Cc;
    const int my_base = (s / W) * Cc;
    const int64_t store_base = (int64_t)s * ss + (lane << 1);
    // one anti-diagonal step of the strip: the lane's new cell (r, t - lane), -inf when there is none
    auto step = [&](const int t) -> Q3 {
      if (has_above) {
        if ((t & 63) == 0 && t < Cc) {
          const int hi = (t + 64 < Cc) ? t + 64 : Cc;
          const int need = above_base + hi;
          while (progp[prev_wave] < need) __builtin_amdgcn_s_sleep(1);
          const int cc = t + lane;
          bnd = q3_neg_inf();
          if (cc < Cc) {
            const int64_t sl = cell_slot_blk(ss, blk, row0 - 1, cc);
            bnd.mat = __hip_atomic_load(M + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bnd.ins = __hip_atomic_load(M + plane + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bnd.del = __hip_atomic_load(M + 2 * plane + sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          asm volatile("" : "+v"(bnd.mat), "+v"(bnd.ins), "+v"(bnd.del));
        }
        const int sel = t & 63;
        Q3 a = Q3{lane_value(bnd.mat, sel), lane_value(bnd.ins, sel), lane_value(bnd.del, sel)};
        if (t >= Cc) a = q3_neg_inf();
        if (lane == 0) u1 = a;                    // (row0-1, t); u2 already holds (row0-1, t-1)
      }
      const int c = t - lane;                     // column c <-> y position j = c + 1
      const int j = c + 1;
      const int cl = c < 0 ? 0 : (c >= Cc ? Cc - 1 : c);
      int yt;
      d2v gy;
      if (COLG) {
        yt = as_global((const int*)coltok)[cl];
        gy = as_global((const d2v*)colgap)[cl];
      } else {
        yt = ((const HX_LDS int*)coltok)[cl];
        gy = ((const HX_LDS d2v*)colgap)[cl];
      }
      bool act = rvalid && c >= 0 && c < Cc;
      if (!FULL && in_env) act = act && in_env[(c < 0 || c >= Cc || !rvalid) ? 0 : (i - j + Cc)];   // (a job of a mixed batch may have the full envelope)
      Q3 nw = q3_neg_inf();                       // outside the envelope a cell reads as -inf (reference const getCell -> dummy)
      if (act) {
        double mat = dmax(dmax(u2.mat + m2m, u2.del + d2m), u2.ins + i2m);
        mat = dmax(mat, 0.0 + (sgx + gy.x));
        mat += sub[sub_row + yt];
        nw.mat = mat;
        nw.ins = dmax(own.ins + i2i, own.mat + m2i);
        nw.del = dmax(dmax(u1.ins + i2d, u1.del + d2d), u1.mat + m2d);
        const double ij_end = mat + (egx + gy.y);
        if (ij_end > best) { best = ij_end; best_j = j; }
      }
      own = nw;
      // rotate the row-above window: this step's `up` becomes the next step's `diag`; lane l+1's next
      // `up` is this lane's new cell (lane 0 is re-filled from the boundary block)
      u2 = u1;
      u1 = shr1_keep0(u1, nw);
      return nw;
    };
    // Two steps per iteration: in the strip-skewed layout the cells a row produces on two consecutive
    // anti-diagonals are adjacent, so a lane stores 16 contiguous bytes per plane every second step
    // (a wave: 1 KiB, fully coalesced).  Lanes without a cell store -inf into padding.
    const int64_t plane2 = plane >> 1;
    for (int t = 0; t < nsteps; t += 2) {
      const Q3 ca = step(t);
      Q3 cb = q3_neg_inf();
      if (t + 1 < nsteps) cb = step(t + 1);
      HX_GLOBAL d2v* M2 = (HX_GLOBAL d2v*)(M + store_base + (int64_t)(t >> 1) * blk);
      M2[0] = d2v{ca.mat, cb.mat};
      M2[plane2] = d2v{ca.ins, cb.ins};
      M2[2 * plane2] = d2v{ca.del, cb.del};
      // publish progress: the strip's last row has finished column (t + 1) - 63; a column counts once its
      // stores have left the wave (every iteration issues 5 vector-memory operations: everything stored
      // HX_QA_LAG steps ago is older than the wave's 40 youngest, see hx_chain.hip)
      const int fin = (t + 2 < nsteps ? t + 2 : nsteps) - 63;
      if (fin >= Cc) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) progp[wave] = my_base + Cc;
      } else {
        const int done = fin - HX_QA_LAG;
        if (done > 0 && ((done >> 6) != ((done - 2) >> 6))) {
          asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
          if (lane == 0) progp[wave] = my_base + done;
        }
      }
    }
    if (rvalid) {
      J.best_score[r] = best;
      J.best_j[r] = best_j;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // first maximum in (j, i) order, as the reference's column-major scan finds it (strict >)
    double end = HX_NEG_INF;
    int xe = 0, ye = 0;
    for (int r = 0; r < R; ++r) {
      const double b = J.best_score[r];
      const int bj = J.best_j[r];
      if (b > end || (b == end && b > HX_NEG_INF && (bj < ye || (bj == ye && r + 1 < xe)))) { end = b; xe = r + 1; ye = bj; }
    }
    J.result[0] = end;
    J.xy_end[0] = xe;
    J.xy_end[1] = ye;
  }
}

#define HX_QA_LDS_COLS 7000
template <int W>
int launch_w(const DevQuick* d_jobs, int n_jobs, int max_cols, bool full, hipStream_t st) {
  if (max_cols > HX_QA_LDS_COLS) {
    if (full) hipLaunchKernelGGL((k_quickalign<W, true, true>), dim3(n_jobs), dim3(W * 64), 16, st, d_jobs, max_cols);
    else hipLaunchKernelGGL((k_quickalign<W, false, true>), dim3(n_jobs), dim3(W * 64), 16, st, d_jobs, max_cols);
    return 0;
  }
  const size_t lds = (size_t)max_cols * (16 + 4) + 16;
  HX_CHECK_LDS((k_quickalign<W, true, false>), lds, "k_quickalign");
  if (full) hipLaunchKernelGGL((k_quickalign<W, true, false>), dim3(n_jobs), dim3(W * 64), lds, st, d_jobs, max_cols);
  else hipLaunchKernelGGL((k_quickalign<W, false, false>), dim3(n_jobs), dim3(W * 64), lds, st, d_jobs, max_cols);
  return 0;
}

}  // namespace

int launch_quickalign(const DevQuick* d_jobs, int n_jobs, int max_rows, int max_cols, bool all_full, hipStream_t st) {
  const char* v = getenv("HX_QA_WAVES");   // tuning hook
  const int forced = v ? atoi(v) : 0;
  if (forced == 8) return launch_w<8>(d_jobs, n_jobs, max_cols, all_full, st);
  if (forced == 4) return launch_w<4>(d_jobs, n_jobs, max_cols, all_full, st);
  if (forced == 2) return launch_w<2>(d_jobs, n_jobs, max_cols, all_full, st);
  if (forced == 1) return launch_w<1>(d_jobs, n_jobs, max_cols, all_full, st);
  // Waves per pair: enough to cover the rows when the batch is small (latency of one pair), fewer when the
  // batch alone fills the GPU -- a pair's strips start one after the other, so fewer waves per pair waste
  // less of the sweep on the ramp (measured on 512 pairs of 2000x2000: 16 waves 0.54, 8: 0.57, 4: 0.59 of
  // the HBM roofline; below 4 the per-column LDS tables limit the workgroups per CU).
  int cap = 16;
  if (n_jobs >= 384) cap = 4;
  else if (n_jobs >= 128) cap = 8;
  const int need = (max_rows + 63) / 64;
  const int w = need < cap ? need : cap;
  if (w <= 1) return launch_w<1>(d_jobs, n_jobs, max_cols, all_full, st);
  if (w <= 2) return launch_w<2>(d_jobs, n_jobs, max_cols, all_full, st);
  if (w <= 4) return launch_w<4>(d_jobs, n_jobs, max_cols, all_full, st);
  if (w <= 8) return launch_w<8>(d_jobs, n_jobs, max_cols, all_full, st);
  return launch_w<16>(d_jobs, n_jobs, max_cols, all_full, st);
}

}  // namespace hx
