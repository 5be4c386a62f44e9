// Device-side argmax traceback through a filled Forward matrix.
//
// Restates ForwardMatrix::bestTrace (reference src/forward.cpp:278-302) with its helpers
// sourceTransitionsWithoutEmitOrAbsorb (:326-398), lpCellEmitOrAbsorb (:404-440), sourceCells (:309-314)
// and bestCell (:245-255), so that a host that only needs the best path of every pair never copies the
// 40 B/cell matrices over PCIe.  Only additions and comparisons are involved: the path is the one the
// reference's code finds in the same matrix, ties included (bestCell keeps the first maximum in
// CellCoords order x, y, state).
//
// One wavefront per pair.  A step enumerates the source cells of the current cell — (in-transitions of
// the x state) x (in-transitions of the y state) x (pair-HMM source states) — one candidate per lane,
// 64 at a time, and reduces them with a wave-wide arg-max.  Steps are dependent (each needs the cell the
// previous one chose), so a pair runs at memory latency; the batch supplies the parallelism.
#include <hip/hip_runtime.h>
#include "hx_common.h"
#include "hx_kernels.h"

namespace hx {

namespace {

constexpr unsigned long long NO_KEY = ~0ull;

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
  const int lo = __shfl_xor((int)(unsigned)(v & 0xffffffffull), m, 64);
  const int hi = __shfl_xor((int)(unsigned)(v >> 32), m, 64);
  return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}

// a stored Forward cell, or -inf outside the storage (reference DPMatrix::cell, src/forward.h:74-88); the envelope test
// and the cell load are issued together
__device__ __forceinline__ double forward_cell(const DevJob& J, int i, int j, int s) {
  if (i < 0 || j < 0 || i >= J.n_rows || j >= J.n_cols) return HX_NEG_INF;
  const int64_t slot = stored_slot(J, i, j);
  const double v = J.fwd[(int64_t)s * J.plane + (slot < 0 ? 0 : slot)];
  bool ok = slot >= 0;
  if (J.max_dist >= 0) {
    const FwdPack* px = J.x.fpack + i;
    const FwdPack* py = J.y.fpack + j;
    const int fl = (px->meta | py->meta) & 0xFF;
    int d = px->env - py->env;
    d = d < 0 ? -d : d;
    ok = ok && ((fl & F_EDGE) || d <= J.max_dist);
  }
  return ok ? v : HX_NEG_INF;
}

}  // namespace

// paths: [n_jobs][cap][3] int32 (x, y, state), written from the END cell backwards; n_cells[job] = number of
// cells written, or -1 when lpEnd = -inf (the reference asserts), -2 when a cell had no source transitions
// (the reference's "traceback failure"), -3 when cap was too small.
// near_tie[job] (may be null): 1 when at some step of the walk the best source cell led a DIFFERENT source cell by no more than
// HX_TRACE_TIE_TOL relative to the value - there an arithmetic policy that is not the reference's bit for bit may have chosen
// the other route (two equally probable routes through a general profile); hx_batch_best_trace_ties.
#define HX_TRACE_TIE_TOL 1e-9
__global__ __launch_bounds__(64) void k_best_trace(const DevJob* __restrict__ jobs, int32_t* __restrict__ paths, int64_t cap,
                                                   int32_t* __restrict__ n_cells, const double* __restrict__ tab, int plane_valid,
                                                   int32_t* __restrict__ near_tie) {
  const DevJob& J = jobs[blockIdx.x];
  const int lane = threadIdx.x;
  int32_t* out = paths + (int64_t)blockIdx.x * cap * 3;
  const int Nx = J.x.n, Ny = J.y.n;
  if (!(*J.lp_end > HX_NEG_INF)) {
    if (lane == 0) { n_cells[blockIdx.x] = -1; if (near_tie) near_tie[blockIdx.x] = 0; }
    return;
  }
  int dx = Nx - 1, dy = Ny - 1, ds = 5;
  int64_t n = 0;
  int status = 0;
  int tie = 0;
  if (cap < 1) status = -3;
  else if (lane == 0) { out[0] = dx; out[1] = dy; out[2] = ds; }
  n = 1;
  while (status == 0 && (dx > 0 || dy > 0)) {
    // One 80-byte record per side holds what the step needs of the two states - flags, in-degree, the first three
    // in-transitions, rootsub / ins, emission class (FwdPack, written by k_scatter_prepared) - so that a step is two
    // dependent memory round trips (records, then source cells) instead of four.
    const FwdPack xp = J.x.fpack[dx], yp = J.y.fpack[dy];
    const int xf = xp.meta & 0xFF, yf = yp.meta & 0xFF;
    const bool x_null = xf & F_NULL, y_null = yf & F_NULL;
    const bool x_ready = (xf & F_READY) || J.x.empty, y_ready = (yf & F_READY) || J.y.empty;
    // which of the three factors a source cell may differ in (src/forward.cpp:326-398)
    bool move_x = false, move_y = false, hmm = false, any = false;
    double lp_abs = 0.;   // lpCellEmitOrAbsorb of the destination
    if (ds == 1 || ds == 4) {            // IMD, IIW
      move_x = true;
      if (x_null) any = y_ready && dx < Nx - 1;
      else { any = y_ready; hmm = true; lp_abs = ds == 1 ? xp.rootsub : xp.ins; }
    } else if (ds == 2 || ds == 3) {     // IDM, IMI
      move_y = true;
      if (y_null) any = dy < Ny - 1;
      else { any = x_ready; hmm = true; lp_abs = ds == 2 ? yp.rootsub : yp.ins; }
    } else if (ds == 0) {                // IMM
      if (y_null && (xf & F_EMIT_OR_START)) { move_y = true; any = dy < Ny - 1; }
      else if (x_null) { move_x = true; any = y_ready && dx < Nx - 1; }
      else if (!y_null) {
        move_x = move_y = hmm = any = true;
        if (J.emis) {
          const int cx = xp.cls, cy = yp.cls;
          lp_abs = (cx < 0 || cy < 0) ? HX_NEG_INF : J.emis[(size_t)cx * J.y.n_cls + cy];
        } else if (plane_valid)
          lp_abs = J.emis_plane[cell_slot(J.strip_stride, dx, dy)];
        else
          lp_abs = emission(J, dx, dy, tab);
      }
    } else {                             // EEE: only the end cell
      move_x = move_y = hmm = any = true;
    }
    const int nx = move_x ? (xp.meta >> 8) : 1;
    const int ny = move_y ? (yp.meta >> 8) : 1;
    const int ns = hmm ? 5 : 1;
    const int total = any ? nx * ny * ns : 0;
    if (total == 0) { status = -2; break; }
    double best = HX_NEG_INF, second = HX_NEG_INF;       // second: the best value among source cells other than the winner
    unsigned long long bkey = NO_KEY;
    for (int c = lane; c < total; c += 64) {
      const int si = c % ns, r = c / ns, yi = r % ny, xi = r / ny;
      int sx = dx, sy = dy;
      double xlp = 0., ylp = 0.;
      if (move_x) {
        if (xi < HX_DAG_INLINE) { sx = xi == 0 ? xp.s0 : xi == 1 ? xp.s1 : xp.s2; xlp = xi == 0 ? xp.lp0 : xi == 1 ? xp.lp1 : xp.lp2; }
        else { sx = J.x.in_src[xp.in_b + xi]; xlp = J.x.in_lp[xp.in_b + xi]; }
      }
      if (move_y) {
        if (yi < HX_DAG_INLINE) { sy = yi == 0 ? yp.s0 : yi == 1 ? yp.s1 : yp.s2; ylp = yi == 0 ? yp.lp0 : yi == 1 ? yp.lp1 : yp.lp2; }
        else { sy = J.y.in_src[yp.in_b + yi]; ylp = J.y.in_lp[yp.in_b + yi]; }
      }
      const int s = hmm ? si : ds;
      const double h = hmm ? J.T[si][ds] : 0.;
      // sourceTransitions then sourceCells: ((hmm + x) + y) + emit, then + cell (absent terms are +0.0, exact)
      const double v = (((h + xlp) + ylp) + lp_abs) + forward_cell(J, sx, sy, s);
      const unsigned long long key = ((unsigned long long)(unsigned)sx << 32) | ((unsigned long long)(unsigned)sy << 3) | (unsigned)s;
      if (v > best || (v == best && v > HX_NEG_INF && key < bkey)) {
        if (bkey != NO_KEY && bkey != key) second = fmax(second, best);
        best = v; bkey = key;
      } else if (key != bkey) second = fmax(second, v);
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
      const double ov = __shfl_xor(best, m, 64);
      const double os = __shfl_xor(second, m, 64);
      const unsigned long long ok = shfl_xor_u64(bkey, m);
      double ns2 = fmax(second, os);
      if (ov > best || (ov == best && ok < bkey)) {
        if (bkey != NO_KEY && bkey != ok) ns2 = fmax(ns2, best);
        best = ov; bkey = ok;
      } else if (ok != NO_KEY && ok != bkey) ns2 = fmax(ns2, ov);
      second = ns2;
    }
    if (second > HX_NEG_INF && best - second <= HX_TRACE_TIE_TOL * fmax(1.0, fabs(best))) tie = 1;
    if (bkey == NO_KEY) { dx = 0; dy = 0; ds = 5; }   // bestCell's default-constructed CellCoords (src/forward.h:32)
    else { dx = (int)(bkey >> 32); dy = (int)((bkey & 0xffffffffull) >> 3); ds = (int)(bkey & 7); }
    if (n >= cap) { status = -3; break; }
    if (lane == 0) { out[3 * n] = dx; out[3 * n + 1] = dy; out[3 * n + 2] = ds; }
    ++n;
  }
  if (lane == 0) {
    n_cells[blockIdx.x] = status < 0 ? status : (int32_t)n;
    if (near_tie) near_tie[blockIdx.x] = tie;
  }
}

// Expected indel events of a pair DP: BackwardMatrix::getCounts restricted to the IndelCounts members
// (reference src/forward.cpp:1183-1214 with transitionEigenCounts, :579-652), for profiles whose transitions carry no
// event counts of their own (leaf profiles).  A thread per matrix cell (i, j): for each of its five states, the source
// transitions exactly as the traceback enumerates them, each weighted with exp(F(src) + lp + B(dest) - lpEnd); six sums
// per thread, reduced over the workgroup, added to out[6] = {ins, del, insExt, delExt, insTime, delTime}.
// tm[6] = {l.t, r.t, l.insWait, l.delWait, r.insWait, r.delWait} (ProbModel members of the pair HMM's two branches).
__global__ __launch_bounds__(256) void k_indel_counts(const DevJob* __restrict__ jobs, const int job, const double* __restrict__ tm,
                                                      double* __restrict__ out, const double* __restrict__ tab, const int plane_valid) {
  const DevJob& J = jobs[job];
  const int Nx = J.x.n, Ny = J.y.n, R = J.n_rows, Cc = J.n_cols;
  const double lp_end = *J.lp_end;
  const double l_t = tm[0], r_t = tm[1], l_iw = tm[2], l_dw = tm[3], r_iw = tm[4], r_dw = tm[5];
  double c_ins = 0., c_del = 0., c_iext = 0., c_dext = 0., c_itime = 0., c_dtime = 0.;
  for (int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; cell < (int64_t)R * Cc && lp_end > HX_NEG_INF; cell += (int64_t)gridDim.x * blockDim.x) {
    const int dx = (int)(cell / Cc), dy = (int)(cell - (int64_t)dx * Cc);
    if (!in_envelope(J, dx, dy)) continue;
    const FwdPack xp = J.x.fpack[dx], yp = J.y.fpack[dy];
    const int xf = xp.meta & 0xFF, yf = yp.meta & 0xFF;
    const bool x_null = xf & F_NULL, y_null = yf & F_NULL;
    const bool x_ready = (xf & F_READY) || J.x.empty, y_ready = (yf & F_READY) || J.y.empty;
    const int64_t bslot = bwd_slot(J.strip_stride, R, Cc, dx, dy);
    for (int ds = 0; ds < 5; ++ds) {
      const double lp_dest = J.bwd[(int64_t)ds * J.plane + bslot];
      if (!(lp_dest > HX_NEG_INF)) continue;
      bool move_x = false, move_y = false, hmm = false, any = false;
      double lp_abs = 0.;
      if (ds == 1 || ds == 4) {
        move_x = true;
        if (x_null) any = y_ready && dx < Nx - 1;
        else { any = y_ready; hmm = true; lp_abs = ds == 1 ? xp.rootsub : xp.ins; }
      } else if (ds == 2 || ds == 3) {
        move_y = true;
        if (y_null) any = dy < Ny - 1;
        else { any = x_ready; hmm = true; lp_abs = ds == 2 ? yp.rootsub : yp.ins; }
      } else {
        if (y_null && (xf & F_EMIT_OR_START)) { move_y = true; any = dy < Ny - 1; }
        else if (x_null) { move_x = true; any = y_ready && dx < Nx - 1; }
        else if (!y_null) {
          move_x = move_y = hmm = any = true;
          if (J.emis) {
            const int cx = xp.cls, cy = yp.cls;
            lp_abs = (cx < 0 || cy < 0) ? HX_NEG_INF : J.emis[(size_t)cx * J.y.n_cls + cy];
          } else if (plane_valid)
            lp_abs = J.emis_plane[cell_slot(J.strip_stride, dx, dy)];
          else
            lp_abs = emission(J, dx, dy, tab);
        }
      }
      if (!any) continue;
      const int nx = move_x ? (xp.meta >> 8) : 1, ny = move_y ? (yp.meta >> 8) : 1, ns = hmm ? 5 : 1;
      for (int xi = 0; xi < nx; ++xi)
        for (int yi = 0; yi < ny; ++yi) {
          int sx = dx, sy = dy;
          double xlp = 0., ylp = 0.;
          if (move_x) {
            if (xi < HX_DAG_INLINE) { sx = xi == 0 ? xp.s0 : xi == 1 ? xp.s1 : xp.s2; xlp = xi == 0 ? xp.lp0 : xi == 1 ? xp.lp1 : xp.lp2; }
            else { sx = J.x.in_src[xp.in_b + xi]; xlp = J.x.in_lp[xp.in_b + xi]; }
          }
          if (move_y) {
            if (yi < HX_DAG_INLINE) { sy = yi == 0 ? yp.s0 : yi == 1 ? yp.s1 : yp.s2; ylp = yi == 0 ? yp.lp0 : yi == 1 ? yp.lp1 : yp.lp2; }
            else { sy = J.y.in_src[yp.in_b + yi]; ylp = J.y.in_lp[yp.in_b + yi]; }
          }
          for (int si = 0; si < ns; ++si) {
            const int s = hmm ? si : ds;
            const double h = hmm ? J.T[si][ds] : 0.;
            const double f = forward_cell(J, sx, sy, s);
            const double lw = (f + ((((h + xlp) + ylp) + lp_abs))) + lp_dest - lp_end;
            if (!(lw > HX_NEG_INF)) continue;
            const double w = exp(lw);
            // transitionEigenCounts, dest.state switch (src/forward.cpp:585-649)
            if (ds == 0) {
              if (!x_null && !y_null) {
                if (s == 0 || s == 1) { c_itime += w * l_t; c_dtime += w * l_t; }
                if (s == 0 || s == 2) { c_itime += w * r_t; c_dtime += w * r_t; }
              }
            } else if (ds == 1) {
              if (!x_null) {
                if (s == 0 || s == 1) { c_itime += w * l_t; c_dtime += w * l_t; }
                if (s == 1) c_dext += w;
                else { c_del += w; c_dtime += w * r_dw; }
              }
            } else if (ds == 4) {
              if (!x_null) {
                if (s == 4) c_iext += w;
                else { c_ins += w; c_itime += w * l_iw; }
              }
            } else if (ds == 2) {
              if (!y_null) {
                if (s == 0 || s == 2) { c_itime += w * r_t; c_dtime += w * r_t; }
                if (s == 2) c_dext += w;
                else { c_del += w; c_dtime += w * l_dw; }
              }
            } else {
              if (!y_null) {
                if (s == 3) c_iext += w;
                else { c_ins += w; c_itime += w * r_iw; }
              }
            }
          }
        }
    }
  }
  __shared__ double part[6][256];
  part[0][threadIdx.x] = c_ins; part[1][threadIdx.x] = c_del; part[2][threadIdx.x] = c_iext;
  part[3][threadIdx.x] = c_dext; part[4][threadIdx.x] = c_itime; part[5][threadIdx.x] = c_dtime;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if ((int)threadIdx.x < h)
#pragma unroll
      for (int k = 0; k < 6; ++k) part[k][threadIdx.x] += part[k][threadIdx.x + h];
    __syncthreads();
  }
  if (threadIdx.x < 6) atomicAdd(&out[threadIdx.x], part[threadIdx.x][0]);
}

// Sampled tracebacks of ONE pair: ForwardMatrix::sampleTrace (reference src/forward.cpp:257-276) with sampleCell (:225-243),
// n_walks walks one after the other.  The reference draws one uniform_real_distribution<double>(0, ptot) value per step from the
// generator it shares with everything else; the host hands over the canonical uniforms that generator would produce (two
// 32-bit draws each, std::generate_canonical) and discards what the walks used: step k of the concatenated walks takes
// uniforms[k].  A step is what sampleCell does, in its order: the source cells as the reference's std::map holds them (sorted
// by (x, y, state), a cell produced twice keeps the later value), lpmax, ptot = the sum of exp(lp - lpmax) in that order, the
// draw u * ptot, and the first cell at which the running remainder is used up.  One wavefront: candidates are evaluated a
// lane each, the two sums run over them in order (every lane the same).  What is NOT the reference's bit for bit is exp():
// this is the device library's, glibc's differs from it in the last place for some arguments, so a walk can leave the
// reference's only where a draw falls within an ulp of a boundary between two cells' shares (~1e-16 per step).
//
// paths: [n_walks][cap][3], a walk's cells from the END cell backwards; n_cells[w] = cells of walk w, or < 0: -1 lpEnd = -inf,
// -2 a cell without sources, -3 cap too small, -4 more than HX_SAMPLE_MAX_SOURCES source cells, -5 the remainder never used up
// (the reference aborts), -6 out of uniforms; walks behind a failed one are not run (n_cells 0).  draws[w] = uniforms used up
// to and including walk w.
#define HX_SAMPLE_MAX_SOURCES 1024
__global__ __launch_bounds__(64) void k_sample_traces(const DevJob* __restrict__ jobs, const int job, const int n_walks,
                                                      const double* __restrict__ uniforms, const int64_t n_uniforms,
                                                      int32_t* __restrict__ paths, const int64_t cap, int32_t* __restrict__ n_cells,
                                                      int64_t* __restrict__ draws, const double* __restrict__ tab, const int plane_valid) {
  __shared__ unsigned long long ckey[HX_SAMPLE_MAX_SOURCES], skey[HX_SAMPLE_MAX_SOURCES];
  __shared__ double cval[HX_SAMPLE_MAX_SOURCES], sval[HX_SAMPLE_MAX_SOURCES];
  __shared__ unsigned char alive[HX_SAMPLE_MAX_SOURCES];
  const DevJob& J = jobs[job];
  const int lane = threadIdx.x;
  const int Nx = J.x.n, Ny = J.y.n;
  int64_t used = 0;
  bool failed = false;
  for (int w = 0; w < n_walks; ++w) {
    int32_t* out = paths + (int64_t)w * cap * 3;
    if (failed) { if (lane == 0) { n_cells[w] = 0; draws[w] = used; } continue; }
    int status = 0;
    if (!(*J.lp_end > HX_NEG_INF)) status = -1;
    int dx = Nx - 1, dy = Ny - 1, ds = 5;
    int64_t n = 0;
    if (status == 0) {
      if (cap < 1) status = -3;
      else if (lane == 0) { out[0] = dx; out[1] = dy; out[2] = ds; }
      n = 1;
    }
    while (status == 0) {
      // ---- the source cells of (dx, dy, ds), as k_best_trace enumerates them ----
      const FwdPack xp = J.x.fpack[dx], yp = J.y.fpack[dy];
      const int xf = xp.meta & 0xFF, yf = yp.meta & 0xFF;
      const bool x_null = xf & F_NULL, y_null = yf & F_NULL;
      const bool x_ready = (xf & F_READY) || J.x.empty, y_ready = (yf & F_READY) || J.y.empty;
      bool move_x = false, move_y = false, hmm = false, any = false;
      double lp_abs = 0.;
      if (ds == 1 || ds == 4) {
        move_x = true;
        if (x_null) any = y_ready && dx < Nx - 1;
        else { any = y_ready; hmm = true; lp_abs = ds == 1 ? xp.rootsub : xp.ins; }
      } else if (ds == 2 || ds == 3) {
        move_y = true;
        if (y_null) any = dy < Ny - 1;
        else { any = x_ready; hmm = true; lp_abs = ds == 2 ? yp.rootsub : yp.ins; }
      } else if (ds == 0) {
        if (y_null && (xf & F_EMIT_OR_START)) { move_y = true; any = dy < Ny - 1; }
        else if (x_null) { move_x = true; any = y_ready && dx < Nx - 1; }
        else if (!y_null) {
          move_x = move_y = hmm = any = true;
          if (J.emis) {
            const int cx = xp.cls, cy = yp.cls;
            lp_abs = (cx < 0 || cy < 0) ? HX_NEG_INF : J.emis[(size_t)cx * J.y.n_cls + cy];
          } else if (plane_valid)
            lp_abs = J.emis_plane[cell_slot(J.strip_stride, dx, dy)];
          else
            lp_abs = emission(J, dx, dy, tab);
        }
      } else {
        move_x = move_y = hmm = any = true;
      }
      const int nx = move_x ? (xp.meta >> 8) : 1;
      const int ny = move_y ? (yp.meta >> 8) : 1;
      const int ns = hmm ? 5 : 1;
      const int total = any ? nx * ny * ns : 0;
      if (total == 0) { status = -2; break; }
      if (total > HX_SAMPLE_MAX_SOURCES) { status = -4; break; }
      __syncthreads();                                   // (the arrays of the step in front are done with)
      for (int c = lane; c < total; c += 64) {
        const int si = c % ns, r = c / ns, yi = r % ny, xi = r / ny;
        int sx = dx, sy = dy;
        double xlp = 0., ylp = 0.;
        if (move_x) {
          if (xi < HX_DAG_INLINE) { sx = xi == 0 ? xp.s0 : xi == 1 ? xp.s1 : xp.s2; xlp = xi == 0 ? xp.lp0 : xi == 1 ? xp.lp1 : xp.lp2; }
          else { sx = J.x.in_src[xp.in_b + xi]; xlp = J.x.in_lp[xp.in_b + xi]; }
        }
        if (move_y) {
          if (yi < HX_DAG_INLINE) { sy = yi == 0 ? yp.s0 : yi == 1 ? yp.s1 : yp.s2; ylp = yi == 0 ? yp.lp0 : yi == 1 ? yp.lp1 : yp.lp2; }
          else { sy = J.y.in_src[yp.in_b + yi]; ylp = J.y.in_lp[yp.in_b + yi]; }
        }
        const int st = hmm ? si : ds;
        const double h = hmm ? J.T[si][ds] : 0.;
        cval[c] = (((h + xlp) + ylp) + lp_abs) + forward_cell(J, sx, sy, st);
        ckey[c] = ((unsigned long long)(unsigned)sx << 32) | ((unsigned long long)(unsigned)sy << 3) | (unsigned)st;
      }
      __syncthreads();
      // ---- the std::map: a cell produced twice keeps its later value; cells in (x, y, state) order ----
      for (int c = lane; c < total; c += 64) {
        const unsigned long long k = ckey[c];
        bool a = true;
        for (int c2 = c + 1; c2 < total; ++c2) a = a && ckey[c2] != k;
        alive[c] = a ? 1 : 0;
      }
      __syncthreads();
      for (int c = lane; c < total; c += 64)
        if (alive[c]) {
          const unsigned long long k = ckey[c];
          int rank = 0;
          for (int c2 = 0; c2 < total; ++c2) rank += (alive[c2] && ckey[c2] < k) ? 1 : 0;
          skey[rank] = k; sval[rank] = cval[c];
        }
      int m = 0;
      for (int c = 0; c < total; ++c) m += alive[c];      // (every lane the same)
      __syncthreads();
      // ---- sampleCell ----
      double lpmax = HX_NEG_INF;
      for (int k = 0; k < m; ++k) lpmax = sval[k] > lpmax ? sval[k] : lpmax;
      for (int k = lane; k < m; k += 64) cval[k] = exp(sval[k] - lpmax);
      __syncthreads();
      double ptot = 0.;
      for (int k = 0; k < m; ++k) ptot += cval[k];
      if (used >= n_uniforms) { status = -6; break; }
      const double p0 = uniforms[used] * (ptot - 0.0) + 0.0;   // uniform_real_distribution<double>(0, ptot)
      ++used;
      double p = p0;
      int pick = -1;
      for (int k = 0; k < m; ++k)
        if ((p -= cval[k]) <= 0) { pick = k; break; }
      if (pick < 0) { status = -5; break; }
      const unsigned long long bkey = skey[pick];
      dx = (int)(bkey >> 32); dy = (int)((bkey & 0xffffffffull) >> 3); ds = (int)(bkey & 7);
      if (n >= cap) { status = -3; break; }
      if (lane == 0) { out[3 * n] = dx; out[3 * n + 1] = dy; out[3 * n + 2] = ds; }
      ++n;
      if (dx == 0 && dy == 0) break;
    }
    if (lane == 0) { n_cells[w] = status < 0 ? status : (int32_t)n; draws[w] = used; }
    failed = status < 0;
  }
}

void launch_sample_traces(const DevJob* d_jobs, int job, int n_walks, const double* d_uniforms, int64_t n_uniforms, int32_t* d_paths,
                          int64_t cap, int32_t* d_n_cells, int64_t* d_draws, Tab8 tab8, bool plane_valid, hipStream_t st) {
  hipLaunchKernelGGL(k_sample_traces, dim3(1), dim3(64), 0, st, d_jobs, job, n_walks, d_uniforms, n_uniforms, d_paths, cap, d_n_cells,
                     d_draws, tab8.p, plane_valid ? 1 : 0);
}

void launch_indel_counts(const DevJob* d_jobs, int job, const double* d_tm, double* d_out, int64_t cells, Tab8 tab8, bool plane_valid,
                         hipStream_t st) {
  int64_t blocks = (cells + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(k_indel_counts, dim3((unsigned)blocks), dim3(256), 0, st, d_jobs, job, d_tm, d_out, tab8.p, plane_valid ? 1 : 0);
}

// paths as walked (END cell first, [job][cap][3]) -> start-first, back to back at off[job]
__global__ void k_reverse_paths(const int32_t* __restrict__ paths, int64_t cap, const int32_t* __restrict__ n_cells,
                                const int64_t* __restrict__ off, int32_t* __restrict__ out) {
  const int job = blockIdx.x;
  const int n = n_cells[job];
  const int32_t* src = paths + (int64_t)job * cap * 3;
  int32_t* dst = out + off[job] * 3;
  for (int c = threadIdx.x; c < n; c += blockDim.x) {
    const int32_t* t = src + 3 * (int64_t)(n - 1 - c);
    dst[3 * (int64_t)c] = t[0]; dst[3 * (int64_t)c + 1] = t[1]; dst[3 * (int64_t)c + 2] = t[2];
  }
}

void launch_reverse_paths(const int32_t* d_paths, int64_t cap, const int32_t* d_n_cells, const int64_t* d_off, int32_t* d_out,
                          int n_jobs, hipStream_t st) {
  hipLaunchKernelGGL(k_reverse_paths, dim3(n_jobs), dim3(256), 0, st, d_paths, cap, d_n_cells, d_off, d_out);
}

void launch_best_trace(const DevJob* d_jobs, int n_jobs, int32_t* d_paths, int64_t cap, int32_t* d_n_cells, Tab8 tab8,
                       bool plane_valid, int32_t* d_near_tie, hipStream_t st) {
  const double* tab = tab8.p;
  hipLaunchKernelGGL(k_best_trace, dim3(n_jobs), dim3(64), 0, st, d_jobs, d_paths, cap, d_n_cells, tab, plane_valid ? 1 : 0, d_near_tie);
}

}  // namespace hx
