// HIP kernels (gfx950 / CDNA4) for the pair-HMM Forward/Backward hot path.
//
//   k_left_multiply   Profile::leftMultiply                (reference src/profile.cpp:78-91)
//   k_ins_rootsub     insx/rootsubx/insy/rootsuby          (src/forward.cpp:44-56)
//   k_emission_table  computeLogProbAbsorb per class pair  (src/forward.h:112-124)
//   k_forward_dag     ForwardMatrix fill, any profile      (src/forward.cpp:68-223)
//   k_backward_dag    BackwardMatrix fill, any profile     (src/forward.cpp:975-1088)
//   k_posterior_scan  cellsAbovePostProbThreshold          (src/forward.cpp:1302-1319)
//
// Compiled with -ffp-contract=off (see hx_lse.h).
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "hx_device.h"
#include "hx_lse.h"
#include "hx_common.h"
#include "hx_kernels.h"

namespace hx {

// ---------------------------------------------------------------------------
// profile preparation
// ---------------------------------------------------------------------------
// States with byte-identical lpAbsorb rows form an emission class (a leaf has at most A+1 of them);
// leftMultiply / insx / rootsubx depend on the row only, so they are evaluated once per class --
// the same operations on the same inputs as the per-state loops of the reference -- and scattered.
// one thread per (class, cpt, c): subc[k][cpt][c] = (+)_d logsub[cpt][c][d] + lpAbsorb[rep(k)][cpt][d]
__global__ void k_left_multiply(const DevJob* __restrict__ jobs, const double* __restrict__ tab) {
  const DevJob& J = jobs[blockIdx.y >> 1];
  const int side = blockIdx.y & 1;
  const DevProfile& P = side ? J.y : J.x;
  const double* logsub = side ? J.log_sub_r : J.log_sub_l;
  const int A = J.A, CA = J.CA;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= P.n_cls * CA) return;
  const int kc = idx / CA, k = idx - kc * CA;
  const int cpt = k / A, c = k - cpt * A;
  const double* row = P.lp_absorb + (size_t)P.cls_rep[kc] * CA + cpt * A;
  const double* ls = logsub + ((size_t)cpt * A + c) * A;
  double lp = HX_NEG_INF;
  for (int d = 0; d < A; ++d) lp = lse(lp, ls[d] + row[d], tab);
  P.subc[idx] = lp;
}

// one thread per class: insx / rootsubx (reference src/forward.cpp:44-56)
__global__ void k_ins_rootsub(const DevJob* __restrict__ jobs, const double* __restrict__ tab) {
  const DevJob& J = jobs[blockIdx.y >> 1];
  const int side = blockIdx.y & 1;
  const DevProfile& P = side ? J.y : J.x;
  const double* logins = side ? J.log_ins_r : J.log_ins_l;
  const double* logcw = side ? J.log_cptw_r : J.log_cptw_l;
  const int A = J.A, C = J.C, CA = J.CA;
  const int kc = blockIdx.x * blockDim.x + threadIdx.x;
  if (kc >= P.n_cls) return;
  double ins = HX_NEG_INF, rs = HX_NEG_INF;
  for (int cpt = 0; cpt < C; ++cpt) {
    const double* raw = P.lp_absorb + (size_t)P.cls_rep[kc] * CA + cpt * A;
    const double* sub = P.subc + (size_t)kc * CA + cpt * A;
    double lipi = HX_NEG_INF, lipr = HX_NEG_INF;
    for (int a = 0; a < A; ++a) lipi = lse(lipi, logins[cpt * A + a] + raw[a], tab);
    ins = lse(ins, logcw[cpt] + lipi, tab);
    for (int a = 0; a < A; ++a) lipr = lse(lipr, J.log_root[cpt * A + a] + sub[a], tab);
    rs = lse(rs, lipr, tab);
  }
  P.insc[kc] = ins;
  P.rootsubc[kc] = rs;
}

// one thread per state: scatter the class results, pack the chain kernels' per-state constants.  RECORDS: also the 80-byte
// state records (FwdPack) that the general-profile fills and the traceback / counting kernels read - the leaf-pair fills do
// not, and the records are two thirds of what this kernel writes, so a batch of leaf pairs only gets them when one of those
// kernels is about to run (hx_api.hip: ensure_state_records).  RECORDS == 2: nothing but the records.
template <int RECORDS>
__global__ void k_scatter_prepared(const DevJob* __restrict__ jobs) {
  const DevJob& J = jobs[blockIdx.y >> 1];
  const int side = blockIdx.y & 1;
  const DevProfile& P = side ? J.y : J.x;
  const int CA = J.CA;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  const int kc = P.cls[i];
  const bool emit = kc >= 0 && i >= 1 && i < P.n - 1;     // the reference fills insx/rootsubx for 1 <= i < size-1
  const double ins = emit ? P.insc[kc] : HX_NEG_INF;
  const double rs = emit ? P.rootsubc[kc] : HX_NEG_INF;
  // Per-state leftMultiply rows are only read on the device where there is no class-pair emission table (per-cell
  // emission terms, k_emission_plane); otherwise they are scattered when the host asks for them (k_scatter_sub).
  if (RECORDS != 2) {
    if (!J.emis)
      for (int k = 0; k < CA; ++k) P.sub[(size_t)i * CA + k] = kc >= 0 ? P.subc[(size_t)kc * CA + k] : HX_NEG_INF;
    P.ins[i] = ins;
    P.rootsub[i] = rs;
    const bool ok = (P.flags[i] & F_READY) || P.empty;
    double* pk = P.pack + 4 * (size_t)i;
    pk[0] = (i > 0 && P.in_off[i + 1] > P.in_off[i]) ? P.in_lp[P.in_off[i]] : 0.0;
    pk[1] = rs;
    pk[2] = ins;
    pk[3] = ok ? 0.0 : HX_NEG_INF;
  }
  if (RECORDS == 0) return;
  FwdPack f;
  const int b = P.in_off[i], deg = P.in_off[i + 1] - b;
  f.lp0 = deg > 0 ? P.in_lp[b] : 0.0;
  f.lp1 = deg > 1 ? P.in_lp[b + 1] : 0.0;
  f.rootsub = rs;
  f.ins = ins;
  f.s0 = deg > 0 ? P.in_src[b] : 0;
  f.s1 = deg > 1 ? P.in_src[b + 1] : 0;
  f.s2 = deg > 2 ? P.in_src[b + 2] : 0;
  f.lp2 = deg > 2 ? P.in_lp[b + 2] : 0.0;
  f.pad0_ = 0;
  f.pad1_ = 0.0;
  f.in_b = b;
  f.meta = (int)P.flags[i] | (deg << 8);
  f.env = P.env ? P.env[i] : 0;
  f.cls = kc;
  P.fpack[i] = f;
}

// subx / suby of the jobs that have a class-pair emission table, on demand (hx_batch_read_prepared)
__global__ void k_scatter_sub(const DevJob* __restrict__ jobs) {
  const DevJob& J = jobs[blockIdx.y >> 1];
  if (!J.emis) return;                       // (already scattered by k_scatter_prepared)
  const DevProfile& P = (blockIdx.y & 1) ? J.y : J.x;
  const int CA = J.CA;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  const int kc = P.cls[i];
  for (int k = 0; k < CA; ++k) P.sub[(size_t)i * CA + k] = kc >= 0 ? P.subc[(size_t)kc * CA + k] : HX_NEG_INF;
}

// one thread per (x class, y class)
__global__ void k_emission_table(const DevJob* __restrict__ jobs, const double* __restrict__ tab) {
  const DevJob& J = jobs[blockIdx.y];
  if (!J.emis) return;
  const int Kx = J.x.n_cls, Ky = J.y.n_cls;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Kx * Ky) return;
  const int kx = idx / Ky, ky = idx - kx * Ky;
  const double* sx = J.x.subc + (size_t)kx * J.CA;
  const double* sy = J.y.subc + (size_t)ky * J.CA;
  const double e = emission_rows(J, sx, sy, ExactLse{tab});
  J.emis[idx] = e;
  J.emis_pad[(size_t)kx * (Ky + 1) + ky] = e;
  if (ky == 0) J.emis_pad[(size_t)kx * (Ky + 1) + Ky] = 0.0;
  if (kx == 0) J.emis_pad[(size_t)Kx * (Ky + 1) + ky] = 0.0;
  if (idx == 0) J.emis_pad[(size_t)Kx * (Ky + 1) + Ky] = 0.0;
}

// ---------------------------------------------------------------------------
// Forward fill, general profiles (DAG, null states, ready/wait flags, envelope).
// One workgroup per pair; thread <-> row; one anti-diagonal per barrier.  Sources are
// re-read from the matrix itself (L1/L2); a 64-row strip's anti-diagonal segment is
// contiguous in the strip-skewed layout, so loads and stores are coalesced.
// ---------------------------------------------------------------------------
__device__ void forward_cell(const DevJob& J, int i, int j, const double* __restrict__ tab) {
  const int64_t plane = J.plane, ss = J.strip_stride;
  double* __restrict__ M = J.fwd;
  const int64_t slot = cell_slot(ss, i, j);
  double imm = HX_NEG_INF, imd = HX_NEG_INF, idm = HX_NEG_INF, imi = HX_NEG_INF, iiw = HX_NEG_INF;
  if (in_envelope(J, i, j)) {
    if (i == 0 && j == 0) imm = 0.0;
    const uint8_t xf = J.x.flags[i], yf = J.y.flags[j];
    const bool xnull = xf & F_NULL, ynull = yf & F_NULL;
    const bool yok = (yf & F_READY) || J.y.empty;   // yState.isReady() || yEmpty
    const bool xok = (xf & F_READY) || J.x.empty;
    const int xb = J.x.in_off[i], xe = J.x.in_off[i + 1];
    const int yb = J.y.in_off[j], ye = J.y.in_off[j + 1];
    const double (*T)[6] = J.T;

    if (!xnull) {
      if (yok) {
        for (int t = xb; t < xe; ++t) {
          const Cell5 s = load_cell(M, plane, cell_slot(ss, J.x.in_src[t], j));
          const double lp = J.x.in_lp[t];
          double a = lse(s.v[0] + T[0][1], s.v[1] + T[1][1], tab);
          a = lse(a, s.v[2] + T[2][1], tab);
          a = lse(a, s.v[3] + T[3][1], tab);
          imd = lse(imd, a + lp, tab);
          double b = lse(s.v[0] + T[0][4], s.v[3] + T[3][4], tab);
          b = lse(b, s.v[4] + T[4][4], tab);
          iiw = lse(iiw, b + lp, tab);
        }
        imd += J.x.rootsub[i];
        iiw += J.x.ins[i];
      }
    } else if (yok) {
      for (int t = xb; t < xe; ++t) {
        const int64_t sl = cell_slot(ss, J.x.in_src[t], j);
        const double lp = J.x.in_lp[t];
        imd = lse(imd, M[1 * plane + sl] + lp, tab);
        iiw = lse(iiw, M[4 * plane + sl] + lp, tab);
      }
    }

    if (!ynull) {
      if (xok) {
        for (int t = yb; t < ye; ++t) {
          const Cell5 s = load_cell(M, plane, cell_slot(ss, i, J.y.in_src[t]));
          const double lp = J.y.in_lp[t];
          double a = lse(s.v[0] + T[0][2], s.v[1] + T[1][2], tab);
          a = lse(a, s.v[2] + T[2][2], tab);
          a = lse(a, s.v[4] + T[4][2], tab);
          idm = lse(idm, a + lp, tab);
          const double b = lse(s.v[0] + T[0][3], s.v[3] + T[3][3], tab);
          imi = lse(imi, b + lp, tab);
        }
        idm += J.y.rootsub[j];
        imi += J.y.ins[j];
      }
    } else {
      for (int t = yb; t < ye; ++t) {
        const int64_t sl = cell_slot(ss, i, J.y.in_src[t]);
        const double lp = J.y.in_lp[t];
        idm = lse(idm, M[2 * plane + sl] + lp, tab);
        imi = lse(imi, M[3 * plane + sl] + lp, tab);
      }
    }

    if (!xnull && !ynull) {
      for (int tx = xb; tx < xe; ++tx) {
        const int sx = J.x.in_src[tx];
        const double lpx = J.x.in_lp[tx];
        for (int ty = yb; ty < ye; ++ty) {
          const Cell5 s = load_cell(M, plane, cell_slot(ss, sx, J.y.in_src[ty]));
          double a = lse(s.v[0] + T[0][0], s.v[1] + T[1][0], tab);
          a = lse(a, s.v[2] + T[2][0], tab);
          a = lse(a, s.v[3] + T[3][0], tab);
          a = lse(a, s.v[4] + T[4][0], tab);
          imm = lse(imm, a + lpx + J.y.in_lp[ty], tab);
        }
      }
      imm += emission(J, i, j, tab);
    } else if (ynull && (xf & F_EMIT_OR_START)) {
      for (int t = yb; t < ye; ++t)
        imm = lse(imm, M[cell_slot(ss, i, J.y.in_src[t])] + J.y.in_lp[t], tab);
    } else if (yok) {
      for (int t = xb; t < xe; ++t)
        imm = lse(imm, M[cell_slot(ss, J.x.in_src[t], j)] + J.x.in_lp[t], tab);
    }
  }
  M[slot] = imm;
  M[1 * plane + slot] = imd;
  M[2 * plane + slot] = idm;
  M[3 * plane + slot] = imi;
  M[4 * plane + slot] = iiw;
}

__global__ void __launch_bounds__(1024) k_forward_dag(const DevJob* __restrict__ jobs, const double* __restrict__ tab) {
  const DevJob& J = jobs[blockIdx.x];
  const int R = J.n_rows, Cc = J.n_cols;
  const int nd = R + Cc - 1;
  for (int d = 0; d < nd; ++d) {
    for (int i = threadIdx.x; i < R; i += blockDim.x) {
      const int j = d - i;
      if (j >= 0 && j < Cc) forward_cell(J, i, j, tab);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *J.lp_end = forward_lp_end(J, ExactLse{tab});
}

// ---------------------------------------------------------------------------
// Backward fill, general profiles (reference src/forward.cpp:975-1088)
// ---------------------------------------------------------------------------
__device__ void backward_cell(const DevJob& J, int i, int j, const double* __restrict__ tab) {
  const int64_t plane = J.plane, ss = J.strip_stride;
  double* __restrict__ M = J.bwd;
  const int R = J.n_rows, Cc = J.n_cols;
#define BS(a, b) bwd_slot(ss, R, Cc, (a), (b))
  const int64_t slot = BS(i, j);
  double imm = HX_NEG_INF, imd = HX_NEG_INF, idm = HX_NEG_INF, imi = HX_NEG_INF, iiw = HX_NEG_INF;
  if (in_envelope(J, i, j)) {
    const uint8_t xf = J.x.flags[i], yf = J.y.flags[j];
    const double (*T)[6] = J.T;
    // cells that feed END are initialised by assignment (src/forward.cpp:981-995)
    if ((xf & F_TO_END) && (yf & F_TO_END)) {
      const int xe = J.x.n - 1, ye = J.y.n - 1;
      for (int tx = J.x.in_off[xe]; tx < J.x.in_off[xe + 1]; ++tx)
        for (int ty = J.y.in_off[ye]; ty < J.y.in_off[ye + 1]; ++ty)
          if (J.x.in_src[tx] == i && J.y.in_src[ty] == j) {
            const double lp = J.x.in_lp[tx] + J.y.in_lp[ty];
            imm = lp + T[0][5]; imd = lp + T[1][5]; idm = lp + T[2][5]; imi = lp + T[3][5]; iiw = lp + T[4][5];
          }
    }
    const bool yok = (yf & F_READY) || J.y.empty;
    const bool xok = (xf & F_READY) || J.x.empty;
    const int xab = J.x.ao_off[i], xae = J.x.ao_off[i + 1];
    const int yab = J.y.ao_off[j], yae = J.y.ao_off[j + 1];
    const int xnb = J.x.no_off[i], xne = J.x.no_off[i + 1];
    const int ynb = J.y.no_off[j], yne = J.y.no_off[j + 1];

    for (int tx = xab; tx < xae; ++tx) {
      const int dx = J.x.ao_dst[tx];
      const double lpx = J.x.ao_lp[tx];
      for (int ty = yab; ty < yae; ++ty) {
        const int dy = J.y.ao_dst[ty];
        const double d = lpx + J.y.ao_lp[ty] + emission(J, dx, dy, tab) + M[BS(dx, dy)];
        imm = lse(imm, T[0][0] + d, tab);
        imd = lse(imd, T[1][0] + d, tab);
        idm = lse(idm, T[2][0] + d, tab);
        imi = lse(imi, T[3][0] + d, tab);
        iiw = lse(iiw, T[4][0] + d, tab);
      }
    }
    if (yok)
      for (int tx = xab; tx < xae; ++tx) {
        const int dx = J.x.ao_dst[tx];
        const double lpx = J.x.ao_lp[tx];
        const int64_t sl = BS(dx, j);
        const double d1 = lpx + J.x.rootsub[dx] + M[1 * plane + sl];
        const double d2 = lpx + J.x.ins[dx] + M[4 * plane + sl];
        imm = lse(imm, T[0][1] + d1, tab);
        imd = lse(imd, T[1][1] + d1, tab);
        idm = lse(idm, T[2][1] + d1, tab);
        imi = lse(imi, T[3][1] + d1, tab);
        imm = lse(imm, T[0][4] + d2, tab);
        imi = lse(imi, T[3][4] + d2, tab);
        iiw = lse(iiw, T[4][4] + d2, tab);
      }
    if (xok)
      for (int ty = yab; ty < yae; ++ty) {
        const int dy = J.y.ao_dst[ty];
        const double lpy = J.y.ao_lp[ty];
        const int64_t sl = BS(i, dy);
        const double d1 = lpy + J.y.rootsub[dy] + M[2 * plane + sl];
        const double d2 = lpy + J.y.ins[dy] + M[3 * plane + sl];
        imm = lse(imm, T[0][2] + d1, tab);
        imd = lse(imd, T[1][2] + d1, tab);
        idm = lse(idm, T[2][2] + d1, tab);
        iiw = lse(iiw, T[4][2] + d1, tab);
        imm = lse(imm, T[0][3] + d2, tab);
        imi = lse(imi, T[3][3] + d2, tab);
      }
    if (yok)
      for (int tx = xnb; tx < xne; ++tx) {
        const int dx = J.x.no_dst[tx];
        if (dx >= J.n_rows) continue;   // END column is not stored: xyCell(END,.) is the empty cell
        const double lpx = J.x.no_lp[tx];
        const int64_t sl = BS(dx, j);
        imd = lse(imd, lpx + M[1 * plane + sl], tab);
        iiw = lse(iiw, lpx + M[4 * plane + sl], tab);
        imm = lse(imm, lpx + M[sl], tab);
      }
    for (int ty = ynb; ty < yne; ++ty) {
      const int dy = J.y.no_dst[ty];
      if (dy >= J.n_cols) continue;
      const double lpy = J.y.no_lp[ty];
      const int64_t sl = BS(i, dy);
      idm = lse(idm, lpy + M[2 * plane + sl], tab);
      imi = lse(imi, lpy + M[3 * plane + sl], tab);
      if (xf & F_EMIT_OR_START) imm = lse(imm, lpy + M[sl], tab);
    }
  }
  M[slot] = imm;
  M[1 * plane + slot] = imd;
  M[2 * plane + slot] = idm;
  M[3 * plane + slot] = imi;
  M[4 * plane + slot] = iiw;
#undef BS
}

__global__ void __launch_bounds__(1024) k_backward_dag(const DevJob* __restrict__ jobs, const double* __restrict__ tab) {
  const DevJob& J = jobs[blockIdx.x];
  const int R = J.n_rows, Cc = J.n_cols;
  // mirrored sweep: thread <-> mirrored row, so that a wave's stores are contiguous in the mirrored layout
  for (int d = 0; d < R + Cc - 1; ++d) {
    for (int im = threadIdx.x; im < R; im += blockDim.x) {
      const int jm = d - im;
      if (jm >= 0 && jm < Cc) backward_cell(J, R - 1 - im, Cc - 1 - jm, tab);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *J.lp_start = J.bwd[bwd_slot(J.strip_stride, R, Cc, 0, 0)];
}

// ---------------------------------------------------------------------------
// posterior threshold scan with stream compaction
// ---------------------------------------------------------------------------
__global__ void k_posterior_scan(const DevJob* __restrict__ jobs, int job, double lpp_threshold,
                                 PostCell* __restrict__ out, unsigned long long cap,
                                 unsigned long long* __restrict__ counter) {
  const DevJob& J = jobs[job];
  const double fwd_end = *J.lp_end;
  const int64_t total = (int64_t)J.n_rows * J.n_cols;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(c / J.n_cols), j = (int)(c - (int64_t)i * J.n_cols);
    if (!in_envelope(J, i, j)) continue;
    const int64_t slot = cell_slot_blk(J.strip_stride, J.blk, i, j);
    const int64_t bslot = cell_slot_blk(J.strip_stride, J.blk, J.n_rows - 1 - i, J.n_cols - 1 - j);
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const double lpp = J.bwd[s * J.plane + bslot] + J.fwd[s * J.plane + slot] - fwd_end;
      if (lpp >= lpp_threshold) {
        const unsigned long long k = atomicAdd(counter, 1ull);
        if (k < cap) { out[k].xpos = i; out[k].ypos = j; out[k].state = s; out[k].pad = 0; out[k].lpp = lpp; }
      }
    }
  }
}

// gather of individual cells (traceback support); with a band, cells outside the envelope read as -inf
// whatever the matrix holds there (HX_SPARSE_ENVELOPE batches do not pre-fill)
__global__ void k_gather_cells(const DevJob* __restrict__ jobs, int job, const double* __restrict__ M, int mirrored,
                               const int* __restrict__ ij, int64_t n, double* __restrict__ out) {
  const DevJob& J = jobs[job];
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int i = ij[2 * k], j = ij[2 * k + 1];
  const bool ok = i >= 0 && j >= 0 && i < J.n_rows && j < J.n_cols && in_envelope(J, i, j);
  const int64_t slot = !ok ? 0 : (mirrored ? cell_slot_blk(J.strip_stride, J.blk, J.n_rows - 1 - i, J.n_cols - 1 - j) : stored_slot(J, i, j));
#pragma unroll
  for (int s = 0; s < 5; ++s) out[5 * k + s] = (ok && slot >= 0) ? M[s * J.plane + slot] : HX_NEG_INF;
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
namespace {
thread_local char g_launch_err[256] = "";
}
const char* launch_error() { return g_launch_err; }
int launch_fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_launch_err, sizeof(g_launch_err), fmt, ap);
  va_end(ap);
  return -1;
}

// The prep kernels put the job on grid.y (two blocks per job, x and y side); grid.y is limited to 65535, so large
// batches are launched in chunks of HX_PREP_CHUNK jobs.
#define HX_PREP_CHUNK 16384
void launch_state_records(const DevJob* d_jobs, int n_jobs, int max_states, hipStream_t st) {
  const int tpb = 256;
  for (int j0 = 0; j0 < n_jobs; j0 += HX_PREP_CHUNK) {
    const int n = n_jobs - j0 < HX_PREP_CHUNK ? n_jobs - j0 : HX_PREP_CHUNK;
    dim3 grid((unsigned)((max_states + tpb - 1) / tpb), (unsigned)(2 * n));
    hipLaunchKernelGGL(k_scatter_prepared<2>, grid, dim3(tpb), 0, st, d_jobs + j0);
  }
}

int launch_prep(const DevJob* d_jobs, int n_jobs, int max_states, int max_cls, int max_ca, int max_cls_pairs,
                Tab8 tab8, bool state_records, hipStream_t st) {
  const double* tab = tab8.p;
  const int tpb = 256;
  for (int j0 = 0; j0 < n_jobs; j0 += HX_PREP_CHUNK) {
    const int n = n_jobs - j0 < HX_PREP_CHUNK ? n_jobs - j0 : HX_PREP_CHUNK;
    const DevJob* jobs = d_jobs + j0;
    if (max_cls > 0) {
      dim3 grid((unsigned)((max_cls * max_ca + tpb - 1) / tpb), (unsigned)(2 * n));
      hipLaunchKernelGGL(k_left_multiply, grid, dim3(tpb), 0, st, jobs, tab);
      dim3 grid2((unsigned)((max_cls + tpb - 1) / tpb), (unsigned)(2 * n));
      hipLaunchKernelGGL(k_ins_rootsub, grid2, dim3(tpb), 0, st, jobs, tab);
    }
    {
      dim3 grid((unsigned)((max_states + tpb - 1) / tpb), (unsigned)(2 * n));
      if (state_records) hipLaunchKernelGGL(k_scatter_prepared<1>, grid, dim3(tpb), 0, st, jobs);
      else hipLaunchKernelGGL(k_scatter_prepared<0>, grid, dim3(tpb), 0, st, jobs);
    }
    if (max_cls_pairs > 0) {
      dim3 grid((unsigned)((max_cls_pairs + tpb - 1) / tpb), (unsigned)n);
      hipLaunchKernelGGL(k_emission_table, grid, dim3(tpb), 0, st, jobs, tab);
    }
  }
  return 0;
}

void launch_scatter_sub(const DevJob* d_jobs, int n_jobs, int max_states, hipStream_t st) {
  const int tpb = 256;
  for (int j0 = 0; j0 < n_jobs; j0 += HX_PREP_CHUNK) {
    const int n = n_jobs - j0 < HX_PREP_CHUNK ? n_jobs - j0 : HX_PREP_CHUNK;
    dim3 grid((unsigned)((max_states + tpb - 1) / tpb), (unsigned)(2 * n));
    hipLaunchKernelGGL(k_scatter_sub, grid, dim3(tpb), 0, st, d_jobs + j0);
  }
}

int launch_forward_dag(const DevJob* d_jobs, int n_jobs, int max_rows, Tab8 tab8, hipStream_t st) {
  const double* tab = tab8.p;
  int threads = ((max_rows + 63) / 64) * 64;
  if (threads > 1024) threads = 1024;
  if (threads < 64) threads = 64;
  hipLaunchKernelGGL(k_forward_dag, dim3(n_jobs), dim3(threads), 0, st, d_jobs, tab);
  return 0;
}

int launch_backward_dag(const DevJob* d_jobs, int n_jobs, int max_rows, Tab8 tab8, hipStream_t st) {
  const double* tab = tab8.p;
  int threads = ((max_rows + 63) / 64) * 64;
  if (threads > 1024) threads = 1024;
  if (threads < 64) threads = 64;
  hipLaunchKernelGGL(k_backward_dag, dim3(n_jobs), dim3(threads), 0, st, d_jobs, tab);
  return 0;
}

void launch_posterior_scan(const DevJob* d_jobs, int job, double lpp_threshold, PostCell* out,
                           unsigned long long cap, unsigned long long* counter, hipStream_t st) {
  hipLaunchKernelGGL(k_posterior_scan, dim3(1024), dim3(256), 0, st, d_jobs, job, lpp_threshold, out, cap, counter);
}

void launch_gather_cells(const DevJob* d_jobs, int job, const double* M, int mirrored, const int* ij, int64_t n,
                         double* out, hipStream_t st) {
  const int tpb = 256;
  hipLaunchKernelGGL(k_gather_cells, dim3((unsigned)((n + tpb - 1) / tpb)), dim3(tpb), 0, st, d_jobs, job, M, mirrored, ij, n, out);
}

}  // namespace hx
