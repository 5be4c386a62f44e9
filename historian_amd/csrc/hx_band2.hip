// Banded fills of leaf-like pairs on scaled probabilities, TWO PAIRS PER WAVEFRONT.
//
// Reference: ForwardMatrix::ForwardMatrix (src/forward.cpp:68-223) / BackwardMatrix::BackwardMatrix (:975-1088) inside a
// GuideAlignmentEnvelope (src/forward.h:92-98, src/alignpath.h:56-61) - the reference's default mode (band 20).
//
// The rotating-row sweep of hx_band.hip gives a pair one 64-lane wavefront, lane = row mod 64; an anti-diagonal of a
// band-20 envelope holds ~21 cells, so two thirds of every vector instruction of that sweep are idle lanes.  Here a
// wavefront sweeps TWO pairs: the even lanes are a ring of 32 rows (lane = 2 (row mod 32)) of one pair, the odd lanes of
// another.  Nothing else changes in the recursion: left is the lane's own previous cell, up and diagonal are the cells of
// one and two steps ago of the lane two below, handed over by two DPP wave_ror:1 moves per register.  A pair is admitted when rows i and i + 31 are never alive together (hx_api.hip:
// band2_admits); the row records, store bases and step counts are those of hx_band.hip (build_band_rows).
//
// One wavefront does everything for its two pairs - the one-dimensional envelope edges (row 0 beyond the band, the column
// feeding END) first, then the recursion, the five logarithms per cell, the stores - so there is no ring between a sweeping
// and a converting wave and no flow control; a workgroup is NW such wavefronts sharing the logarithm table.
// Per-pair constants that are wave-uniform in hx_band.hip (the 18 transition probabilities, plane bases, LDS block) are
// per-lane here, since the two halves belong to different pairs.
//
// Arithmetic: scaled probabilities (HX_LSE_LINEAR) or scaled probabilities with the reference's truncation (HX_LSE_TRUNC:
// trunc_sum drops the smaller term of every pairwise sum when it is at most e^-10 of the larger, as the reference's table
// does for differences >= 10, src/logsumexp.h:45; left-nested as src/logsumexp.h:86-100).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "hx_device.h"
#include "hx_lse.h"
#include "hx_common.h"
#include "hx_policy.h"
#include "hx_kernels.h"
#include "hx_bandedge.h"

namespace hx {

namespace {

typedef double d2v __attribute__((ext_vector_type(2)));
typedef int i2v __attribute__((ext_vector_type(2)));

#define HXB2_EMIN (-(1 << 28))
#define HXB2_LOG_ENTRIES 1536      // the logarithm table of hx_linear.hip (build_log_table): entry 0 and entries 1024..1535 are used
#define HXB2_HOLE_BEGIN 16         // bytes [16, 16384) of the table are never addressed: pair blocks live there
#define HXB2_HOLE_END 16384
#define HXB2_W 32                  // rows per ring
#define HXB2_INFLIGHT 30           // vector-memory operations that may be outstanding when a lane reads its next row's record ...
#define HXB2_LONG_ROW 16           // ... at the end of a row of at least this many owned steps: 5 stores in each of its first 7 step pairs

struct L5 { double imm, imd, idm, imi, iiw; int e; };
__device__ __forceinline__ L5 l5_zero() { return L5{0., 0., 0., 0., 0., HXB2_EMIN}; }

// value of the previous lane of the lane's ring.  The two rings are INTERLEAVED - lane = 2 (row mod 32) + (which pair) - so
// the previous row sits two lanes down, all the way round the wavefront: two DPP wave_ror:1 moves per dword, on the vector
// ALU of the wavefront's own SIMD.  (Halves as rings - lanes 0-31 / 32-63 - need a rotation inside 32 lanes, which only the
// LDS crossbar offers (ds_swizzle rotate mode, checked in tools/probes/swizzle_rotate.hip): eleven LDS instructions per
// step made the CU's one LDS pipeline the bound of the whole kernel - 3.5 ms for 2048 pairs whatever the arithmetic.)
__device__ __forceinline__ int ror1(int v) { return __builtin_amdgcn_mov_dpp(v, 0x13C /* wave_ror:1 */, 0xf, 0xf, false); }
__device__ __forceinline__ int rot(const int, const int v) { return ror1(ror1(v)); }
__device__ __forceinline__ double rot(const int addr, const double v) {
  return __hiloint2double(rot(addr, __double2hiint(v)), rot(addr, __double2loint(v)));
}
// a whole cell: the eleven first moves, then the eleven second ones - a move that reads the result of the move in front of it
// waits two issue slots (the compiler, left alone, pairs them up that way and pads every pair)
__device__ __forceinline__ double ror1(const double v) { return __hiloint2double(ror1(__double2hiint(v)), ror1(__double2loint(v))); }
__device__ __forceinline__ L5 rot(const int, const L5& c) {
  const L5 t{ror1(c.imm), ror1(c.imd), ror1(c.idm), ror1(c.imi), ror1(c.iiw), ror1(c.e)};
  __builtin_amdgcn_sched_barrier(0);
  const L5 r{ror1(t.imm), ror1(t.imd), ror1(t.idm), ror1(t.imi), ror1(t.iiw), ror1(t.e)};
  __builtin_amdgcn_sched_barrier(0);
  return r;
}

// (hx_linear.hip) one pairwise sum of the reference's log_sum_exp on probabilities
// (a dropped term keeps its low word: a number below 2^-1042 that no sum of mantissas scaled to the cell's exponent feels -
// one select instead of two)
__device__ __forceinline__ double trunc_sum(double a, double b) {
  const double hi = fmax_plain(a, b), lo = fmin_plain(a, b);
  const int keep = lo > hi * 4.5399929762484854e-05 ? __double2hiint(lo) : 0;
  return hi + __hiloint2double(keep, __double2loint(lo));
}
template <bool TRUNC> __device__ __forceinline__ double lin_acc(double m, double p, double acc) {
  if (TRUNC) return trunc_sum(acc, m * p);
  return __builtin_fma(m, p, acc);
}

// log(m * 2^e), m >= 0 (hx_linear.hip log_scaled: frexp, one 16-byte table entry, a cubic)
// The table sits at LDS address 0 (the kernel has no static LDS: the dynamic allocation starts there), so the entry's byte
// offset IS its address.
__device__ __forceinline__ double log_scaled(double m, int e) {
  const double f = __builtin_amdgcn_frexp_mant(m);
  const int k = __builtin_amdgcn_frexp_exp(m);
  const unsigned byte_off = ((unsigned)__double2hiint(f) >> 7) & 0x7FF0u;
  const d2v ce = *(const HX_LDS d2v*)(uintptr_t)byte_off;
  const double r = __builtin_fma(f, ce.x, -1.0);
  double p = __builtin_fma(r, 1.0 / 3.0, -0.5);
  p = __builtin_fma(p, r, 1.0);
  const double lf = __builtin_fma(p, r, ce.y);
  return __builtin_fma((double)(e + k), 0.693147180559945309417, lf);
}

__device__ __forceinline__ double read_lane(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// LDS plan (bytes, computed on the host): the logarithm table first (pair blocks 0.. sit in its hole while they fit), one
// block per pair: store bases of the strips, one byte per column {emission class : 7, not ready : 1}, the class constants
// of both sides, the class-pair emission table
struct Band2Plan { int sbase, ycol, yclass, xclass, elds, stride, in_hole, total; };

__device__ __forceinline__ int block_offset(const Band2Plan& p, const int pair) {
  return pair < p.in_hole ? HXB2_HOLE_BEGIN + pair * p.stride : 16 * HXB2_LOG_ENTRIES + (pair - p.in_hole) * p.stride;
}

// The envelope's one-dimensional edges of a pair (see hx_band.hip): row 0 beyond the band and the column feeding END
// (Forward), the first column and the last row (Backward).  One wavefront per pair, four pairs per workgroup, a launch of its
// own in front of the sweep: nothing here is read by the sweep, and the sweep's 176 registers per lane leave a CU no room for
// wavefronts that only write edges (as a fifth wavefront of the sweep's workgroups they kept a second workgroup off the CU; run
// by the sweeping wavefronts in front of their sweeps they were a third of a millisecond of dependent additions per pair in
// front of every sweep).
template <bool TRUNC, int DIR>
__global__ void __launch_bounds__(256)
k_band2_edges(const DevJob* __restrict__ jobs, const int n_jobs, const int write_edges) {
  const int lane = threadIdx.x & 63;
  const int job = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  if (job >= n_jobs) return;
  const DevJob J = jobs[job];                      // (a copy: the stores below cannot alias it)
  const int Re = J.n_rows, Ce = J.n_cols;
  const int nse = (Re + 63) >> 6;
  const int64_t plane = J.plane;
  const int blk = J.blk;
  HX_GLOBAL double* __restrict__ M = as_global(DIR ? J.bwd : J.fwd);
  const HX_GLOBAL i2v* xrecG = (const HX_GLOBAL i2v*)as_global(reinterpret_cast<const i2v*>(DIR ? J.band_rows_bwd : J.band_rows));
  if (DIR == 1) {
    const int64_t ssd = J.strip_stride;
    const int lo_last = reinterpret_cast<const int*>(reinterpret_cast<const i2v*>(J.band_rows_bwd) + (Re + 64))[nse];   // first column the sweep owns on the last row
    auto put_inf = [&](const int ip, const int jp) {
      const int64_t sl = cell_slot_blk(ssd, blk, ip, jp);
      M[sl] = HX_NEG_INF; M[plane + sl] = HX_NEG_INF; M[2 * plane + sl] = HX_NEG_INF; M[3 * plane + sl] = HX_NEG_INF;
      M[4 * plane + sl] = HX_NEG_INF;
    };
    if (write_edges) {
      for (int ip = 1 + lane; ip < Re; ip += 64) {
        const i2v rec = xrecG[ip];
        if ((rec.x & 0xFFFF) + ((rec.y >> 9) & 7) - ip > 0) put_inf(ip, 0);      // (the sweep owns the row from a later column on)
      }
      for (int jp = 1 + lane; jp < lo_last; jp += 64) put_inf(Re - 1, jp);
    }
    {
      // a first row whose band does not reach column 0: the END-feeding cell itself (src/forward.cpp:981-995)
      const i2v rec = xrecG[0];
      if (lane == 0 && (rec.x & 0xFFFF) + ((rec.y >> 9) & 7) > 0) {
        const double lpe = J.x.pack[4 * (size_t)Re] + J.y.pack[4 * (size_t)Ce];
        const int64_t sl = cell_slot_blk(ssd, blk, 0, 0);
        for (int st = 0; st < 5; ++st) M[st * plane + sl] = lpe + J.T[st][5];
      }
    }
  } else {
    // row 0 beyond what the sweep owns: the chain in log space
    const int own0 = (xrecG[0].x >> 16) & 0xFFFF;                    // row 0 is owned from step 0 to this step = column
    const i2v rec1 = xrecG[1];
    const bool row1_edge = ((rec1.x & 0xFFFF) + ((rec1.x >> 16) & 0xFFFF) - 1) < Ce - 1;   // row 1 does not own column Ny-2
    const double T02 = J.T[0][2], T03 = J.T[0][3], T22 = J.T[2][2], T33 = J.T[3][3];
    const double pen0 = J.x.pack[3];                                 // x START ready (or x empty): 0, else -inf
    double d_idm = HX_NEG_INF, d_imi = HX_NEG_INF;                   // cell (0, Ny-3): the diagonal source of (1, Ny-2)
    if (own0 < Ce - 1 || row1_edge) {
      // the chain as a prefix sum (hx_bandedge.h)
      // row 0's slots: strip 0's windows (band-compressed planes) or the dense layout, fetched once
      const bool packed = J.strip_base != nullptr;
      const int w0 = packed ? J.fwd_windows[0] : 0, w1 = packed ? J.fwd_windows[1] : 0, w2 = packed ? J.fwd_windows[2] : 0,
                w3 = packed ? J.fwd_windows[3] : 0;
      const int64_t b0 = packed ? J.strip_base[0] : 0, b1 = packed ? J.strip_base[1] : 0;
      auto row0_slot = [&](const int j) -> int64_t {
        if (!packed) return cell_slot_blk(J.strip_stride, J.blk, 0, j);
        if (j >= w0 && j < w1) return b0 + ((int64_t)((j - w0) >> 1) << 7) + (j & 1);
        if (j >= w2 && j < w3) return b1 + ((int64_t)((j - w2) >> 1) << 7) + (j & 1);
        return -1;
      };
      const HX_GLOBAL double* ypack = as_global(J.y.pack);
      // first even column from which on rows 1-3 own nothing (a row's last owned step: first + count of its record)
      int full_from = own0 + 1;
      for (int l = 1; l < 4 && l < Re; ++l) {
        const i2v r = xrecG[l];
        if ((r.x & 0xFFFF) != 0xFFFF) full_from = max(full_from, (r.x & 0xFFFF) + ((r.x >> 16) & 0xFFFF) + 1);
      }
      full_from = (full_from + 1) & ~1;
      double carry_idm = 0., carry_imi = 0.;                          // the sums up to the block in front
      for (int j0 = 0; j0 < Ce; j0 += 64) {
        const int j = j0 + lane;
        const size_t jl = 4 * (size_t)(j < Ce ? j : Ce - 1);
        const double lrs = ypack[jl + 1], lin = ypack[jl + 2];
        double kidm, kimi;
        row0_chain_block(j0, lane, Ce, lrs, lin, T02, T03, T22, T33, pen0, carry_idm, carry_imi, kidm, kimi);
        if (Ce - 2 >= j0 && Ce - 2 < j0 + 64) { d_idm = read_lane(kidm, Ce - 2 - j0); d_imi = read_lane(kimi, Ce - 2 - j0); }
        // Row 0's cells lie one to a kilobyte of a plane (lane 0 of strip 0's step-pair blocks): written alone, every pair of
        // them is a 16-byte piece of a 64-byte line of its own - five read-modify-writes per two columns.  The rest of such a
        // line are the cells of rows 1-3 on the same two steps, which beyond those rows' bands lie outside the envelope: they
        // hold -inf (pre-filled planes) or anything (sparse / band-compressed planes), so the line is written whole, -inf
        // around the two cells.  Two stores per plane and block of 64 columns: lane t writes piece t & 3 of line t >> 2 (+ 16).
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int line = (lane >> 2) + 16 * half, piece = lane & 3;
          const int je = j0 + 2 * line;                                 // the line's even column
          const double e_idm = __shfl(kidm, 2 * line, 64), o_idm = __shfl(kidm, 2 * line + 1, 64);
          const double e_imi = __shfl(kimi, 2 * line, 64), o_imi = __shfl(kimi, 2 * line + 1, 64);
          const int64_t sl = je + 1 < Ce ? row0_slot(je) : -1;
          if (je >= full_from && sl >= 0) {
            const d2v ninf = d2v{HX_NEG_INF, HX_NEG_INF};
            HX_GLOBAL d2v* L = (HX_GLOBAL d2v*)(M + sl) + piece;
            const int64_t plane2 = plane >> 1;
            L[0] = ninf; L[plane2] = ninf; L[4 * plane2] = ninf;
            L[2 * plane2] = piece == 0 ? d2v{e_idm, o_idm} : ninf;
            L[3 * plane2] = piece == 0 ? d2v{e_imi, o_imi} : ninf;
          }
        }
        // (the few columns next to the band, and an odd last one: cell by cell)
        if (j > own0 && j < Ce && ((j & ~1) < full_from || (j | 1) >= Ce)) {
          const int64_t sl = row0_slot(j);
          if (sl >= 0) {
            M[sl] = HX_NEG_INF; M[plane + sl] = HX_NEG_INF; M[2 * plane + sl] = kidm; M[3 * plane + sl] = kimi; M[4 * plane + sl] = HX_NEG_INF;
          }
        }
      }
    }
    // cell (1, Ny-2) when row 1's band does not reach it (see hx_band.hip): the nested sum over the diagonal cell
    // (0, Ny-3), in libm arithmetic (with the reference's truncation under HX_LSE_TRUNC), written as a log-probability
    if (row1_edge) {
      double dv[5] = {HX_NEG_INF, HX_NEG_INF, HX_NEG_INF, HX_NEG_INF, HX_NEG_INF};
      if (Ce - 2 == 0) dv[0] = 0.0; else { dv[2] = d_idm; dv[3] = d_imi; }
      const unsigned c = (unsigned)J.y.ecls[Ce - 1] & 0x7Fu;
      const unsigned eo = (unsigned)(rec1.y & 0xFF) * (unsigned)(J.y.n_cls + 1);
      double acc = HX_NEG_INF;
      for (int q = 0; q < 5; ++q) {
        const double t = dv[q] + J.T[q][0];
        const double hi = vmax(acc, t), lo = vmin(acc, t);
        const bool add = hi > HX_NEG_INF && lo > HX_NEG_INF && (!TRUNC || hi - lo < 10.0);
        acc = add ? hi + log1p(exp(lo - hi)) : hi;
      }
      const double imm = acc > HX_NEG_INF ? acc + J.emis_pad[eo + c] : HX_NEG_INF;
      const int64_t sl = stored_slot(J, 1, Ce - 1);
      if (lane == 0 && sl >= 0) {
        M[sl] = imm; M[plane + sl] = HX_NEG_INF; M[2 * plane + sl] = HX_NEG_INF; M[3 * plane + sl] = HX_NEG_INF; M[4 * plane + sl] = HX_NEG_INF;
      }
    }
    // the rest of column Ny-2 away from the band: -inf (only where the matrix was not pre-filled)
    if (write_edges)
      for (int i = 2 + lane; i < Re; i += 64) {
        const i2v rec = xrecG[i];
        const int last_col = (rec.x & 0xFFFF) + ((rec.x >> 16) & 0xFFFF) - i;   // column of the row's last owned step
        if (last_col < Ce - 1) {
          const int64_t sl = stored_slot(J, i, Ce - 1);
          if (sl >= 0) {
            M[sl] = HX_NEG_INF; M[plane + sl] = HX_NEG_INF; M[2 * plane + sl] = HX_NEG_INF; M[3 * plane + sl] = HX_NEG_INF; M[4 * plane + sl] = HX_NEG_INF;
          }
        }
      }
  }
}

// NW sweeping wavefronts (two pairs each) per workgroup.  DIR = 1: the Backward fill as the same sweep
// in mirrored coordinates (see hx_band.hip).
template <bool TRUNC, int NW, int DIR>
__global__ void __launch_bounds__(NW * 64, 2)
k_fill_band2(const DevJob* __restrict__ jobs, const double* __restrict__ exact_tab, const double* __restrict__ log_tab,
             const Band2Plan plan, const int n_jobs) {
  constexpr int THREADS = NW * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int half = lane & 1, r32 = lane >> 1;      // which of the wavefront's two pairs, and the lane's place in that pair's ring
  double* ptab = reinterpret_cast<double*>(lds);
  // (entries 1..1023 of the logarithm table are never addressed)
  for (int k = threadIdx.x; k < 2; k += THREADS) ptab[k] = log_tab[k];
  for (int k = 2048 + threadIdx.x; k < 2 * HXB2_LOG_ENTRIES; k += THREADS) ptab[k] = log_tab[k];
  const int first_job = (int)blockIdx.x * 2 * NW;

  // ---- the sweeping wavefronts stage their two pairs' blocks (32 lanes per pair) ----
  const int my_pair = 2 * wave + half;
  const int my_job = first_job + my_pair;
  const bool live = my_job < n_jobs;
  const DevJob* __restrict__ Jp = jobs + (my_job < n_jobs ? my_job : 0);
  unsigned char* blkp = lds + block_offset(plan, my_pair);
  HX_LDS int* sbaseL = (HX_LDS int*)(blkp + plan.sbase);
  HX_LDS unsigned char* ycolL = (HX_LDS unsigned char*)(blkp + plan.ycol);
  HX_LDS d2v* yclassL = (HX_LDS d2v*)(blkp + plan.yclass);
  HX_LDS d2v* xclassL = (HX_LDS d2v*)(blkp + plan.xclass);
  HX_LDS double* eldsL = (HX_LDS double*)(blkp + plan.elds);
  const int R = Jp->n_rows, Cc = Jp->n_cols;
  const int n_strips = (R + 63) >> 6;
  const int Ky1 = Jp->y.n_cls + 1, Kx1 = Jp->x.n_cls + 1;
  const i2v* rowsG = reinterpret_cast<const i2v*>(DIR ? Jp->band_rows_bwd : Jp->band_rows);
  if (live) {
    const int* sb = reinterpret_cast<const int*>(rowsG + (R + 64));
    for (int q = r32; q < n_strips; q += 32) sbaseL[q] = sb[q];
    for (int j = r32; j < Cc; j += 32) {
      // Backward: sweep column j is y state Cc-1-j; the class is that of the state an absorbing move leads to, the ready
      // bit that of the state itself
      const int jc = DIR ? Cc - 1 - j : j;
      ycolL[j] = (unsigned char)((unsigned)Jp->y.ecls[DIR ? jc + 1 : jc] | (Jp->y.pack[4 * (size_t)jc + 3] < 0.0 ? 0x80u : 0u));
    }
    for (int c = r32; c < Ky1; c += 32) {
      const bool real = c < Jp->y.n_cls;
      const int rep = real ? Jp->y.cls_rep[c] : 0;
      yclassL[c] = real ? d2v{exp(Jp->y.pack[4 * (size_t)rep + 1]), exp(Jp->y.pack[4 * (size_t)rep + 2])} : d2v{0., 0.};
    }
    for (int c = r32; c < Kx1; c += 32) {
      const bool real = c < Jp->x.n_cls;
      const int rep = real ? Jp->x.cls_rep[c] : 0;
      xclassL[c] = real ? d2v{exp(Jp->x.pack[4 * (size_t)rep + 1]), exp(Jp->x.pack[4 * (size_t)rep + 2])} : d2v{0., 0.};
    }
    for (int e = r32; e < Kx1 * Ky1; e += 32) eldsL[e] = exp(Jp->emis_pad[e]);
  }
  __syncthreads();

  {
    // =======================================================================================================
    // the sweep: lane = row mod 32 inside the lane's half
    // =======================================================================================================
    const int rot_addr = 0;
    const int64_t plane2 = Jp->plane >> 1;
    HX_GLOBAL double* __restrict__ M = as_global(DIR ? Jp->bwd : Jp->fwd);
    const HX_GLOBAL i2v* xrecG = (const HX_GLOBAL i2v*)as_global(rowsG);
    auto xrec_at = [&](const int i) -> i2v { return xrecG[i]; };
    // The record of the row a lane takes NEXT is fetched when the lane takes its current row and first read on that row's last
    // step.  Left to the compiler that read costs an `s_waitcnt vmcnt(0)` in every step - some lane of the wavefront is always
    // about to change rows - i.e. every step waited for the acknowledgement of the stores issued just before it and for the
    // fetch another lane had issued a few hundred instructions earlier (shader counters: the wavefronts waited 40-55 % of
    // their cycles, and the fill took the same time with 138 as with 239 vector instructions per step).  Vector-memory
    // operations retire in issue order, and between a lane's fetch and its first read of the result the lane owns cells in
    // every step pair of the row, so the five stores of each of those pairs were issued behind the fetch: for a row of at
    // least HXB2_LONG_ROW owned steps the fetch has retired once at most HXB2_INFLIGHT operations are outstanding.  The fetch
    // is therefore issued by hand into a register pair the compiler only ever sees READ (`landing`: defined once, in front
    // of the loop; nothing makes the compiler copy or move it), and the value is taken out of it behind a counted wait; a
    // wavefront in which a shorter row is ending (the band's corners) waits for everything.  tests/test_band2_isa.py checks the
    // compiled loop: one landing register pair, read only behind the waits.
    i2v landing;
    asm volatile("; landing registers of the row-record fetches: %0" : "=v"(landing));
    auto xrec_fetch = [&](const int i) {
      const HX_GLOBAL i2v* a = xrecG + i;
      asm volatile("global_load_dwordx2 %0, %1, off" : : "v"(landing), "v"(a) : "memory");
    };
    auto xrec_take = [&](const bool short_row) -> i2v {
      int x, y;
      if (__builtin_amdgcn_ballot_w64(short_row) != 0)
        asm volatile("s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(x), "=&v"(y) : "v"(landing.x), "v"(landing.y) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(%4)\n\tv_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(x), "=&v"(y) : "v"(landing.x), "v"(landing.y), "n"(HXB2_INFLIGHT) : "memory");
      return i2v{x, y};
    };
    // anti-diagonal steps: the longer of the wavefront's two pairs, in whole blocks of eight (extra steps own nothing)
    const int my_steps = live ? ((DIR ? Jp->band_steps_bwd : Jp->band_steps) + 7) & ~7 : 0;
    const int s0 = __builtin_amdgcn_readlane(my_steps, 0), s1 = __builtin_amdgcn_readlane(my_steps, 1);
    const int n_steps = s0 > s1 ? s0 : s1;

    // The lane's row, decoded: owned steps [os, oe], in-envelope steps [as, as + span], the row's store pointer (cell of step k:
    // rowp + (k >> 1) * 64 sixteen-byte units per state plane), its emission row and column bytes as LDS addresses (the byte of
    // step k is at ycur + k), its x-side constants.  And the raw record of the row the lane takes next (i + 32), fetched a
    // whole row ahead.
    int i = r32, os, oe, as;
    unsigned span;
    HX_GLOBAL d2v* rowp;
    unsigned erow;               // LDS address of the row's emission-table row
    unsigned ycur;               // LDS address of the column byte of step 0 (may lie before the array: clamped at use)
    double xc_rs, xc_in;         // exp(rootsubx), exp(insx)
    int x_wait;                  // x state not ready: 2^29 (an exponent shift)
    i2v nrec;
    d2v nxc = d2v{0., 0.};
    int nstore = 0;
    unsigned nerow = 0;
    const unsigned ycol_lo = (unsigned)(uintptr_t)ycolL, ycol_hi = ycol_lo + (unsigned)(Cc - 1);
    const unsigned elds_a = (unsigned)(uintptr_t)eldsL, yclass_a = (unsigned)(uintptr_t)yclassL;
    auto decode = [&](const i2v r, const d2v xc, const int sb, const unsigned er) {
      os = r.x & 0xFFFF; oe = os + ((r.x >> 16) & 0xFFFF);
      as = os + ((r.y >> 9) & 7);
      span = (unsigned)((oe - ((r.y >> 12) & 7)) - as);
      if ((r.x & 0xFFFF) == 0xFFFF || !live) { os = 0x7FFFFFF0; oe = 0x7FFFFFF1; as = os; span = 1; }     // sentinel: never owned
      rowp = (HX_GLOBAL d2v*)(M + sb);
      erow = er;
      x_wait = (r.y & 0x100) ? (1 << 29) : 0;
      xc_rs = xc.x; xc_in = xc.y;
      ycur = ycol_lo - (unsigned)i;
    };
    auto store_base = [&](const int row) -> int { return sbaseL[row < R ? row >> 6 : 0] + 2 * (row & 63); };
    auto emis_row = [&](const i2v r) -> unsigned { return elds_a + 8u * (unsigned)(r.y & 0xFF) * (unsigned)Ky1; };
    {
      const i2v r0 = xrec_at(r32 < R ? r32 : R);
      decode(r0, xclassL[r0.y & 0xFF], store_base(r32), emis_row(r0));
      nrec = i2v{0xFFFF, 0};
      xrec_fetch(r32 + HXB2_W < R ? r32 + HXB2_W : R);
    }
    // the 18 transition probabilities the recursion reads (dest 5 = EEE is only read by lpEnd): per lane - the halves
    // belong to different pairs
    double P[5][5];
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
      for (int d = 0; d < 5; ++d) {
        const bool used = d == 0 || (d == 1 && a != 4) || (d == 2 && a != 3) || (d == 3 && (a == 0 || a == 3)) ||
                          (d == 4 && (a == 0 || a == 3 || a == 4));
        P[a][d] = used ? exp(Jp->T[a][d]) : 0.;
      }
    // Backward: the cell feeding END
    double end_cell[5] = {0., 0., 0., 0., 0.};
    if (DIR == 1) {
      const double lpe = Jp->x.pack[4 * (size_t)R] + Jp->y.pack[4 * (size_t)Cc];
      for (int s = 0; s < 5; ++s) end_cell[s] = exp(lpe + Jp->T[s][5]);
    }

    // cell registers, ping-ponged: at an even step the lane's previous cell is in lb (the one before in la, which the new
    // cell overwrites), the previous lane's cells of one / two steps ago in lua / lub
    L5 la = l5_zero(), lb = l5_zero(), lua = l5_zero(), lub = l5_zero();

    // The y side of a step, fetched ahead (see hx_band.hip): the column's byte two steps ahead, the column's class constants
    // and the emission term one step ahead.  A lane that is about to change rows looks up the NEXT row's column.
    auto byte_at = [&](const unsigned addr) -> unsigned {       // (clamped to the pair's column bytes)
      const int a = (int)addr < (int)ycol_lo ? (int)ycol_lo : ((int)addr > (int)ycol_hi ? (int)ycol_hi : (int)addr);
      return *(const HX_LDS unsigned char*)(uintptr_t)(unsigned)a;
    };
    unsigned w_cur = byte_at(ycur), w_nxt = byte_at(ycur + 1);     // bytes of steps k, k + 1
    d2v rc_cur = yclassL[w_cur & 0x7Fu];                            // class constants of step k
    double em_cur = *(const HX_LDS double*)(uintptr_t)(erow + 8u * (w_cur & 0x7Fu));

    auto roll_even = [&](const int k) {
      if (k > oe) {
        i += HXB2_W;
        decode(nrec, nxc, nstore, nerow);                 // (nrec was taken on the last step of the row that ended)
        xrec_fetch(i + HXB2_W < R ? i + HXB2_W : R);
      }
    };
    auto roll_odd = [&](const int k) {
      if (k == oe) {
        nrec = xrec_take(oe - os < HXB2_LONG_ROW - 1);
        nxc = xclassL[nrec.y & 0xFF];
        nstore = store_base(i + HXB2_W);
        nerow = emis_row(nrec);
      }
    };
    struct YSide { unsigned w; d2v rc; double em; };
    auto y_side = [&](const int k) -> YSide {
      const YSide now{w_cur, rc_cur, em_cur};
      const unsigned c1 = w_nxt & 0x7Fu;
      rc_cur = *(const HX_LDS d2v*)(uintptr_t)(yclass_a + 16u * c1);
      em_cur = *(const HX_LDS double*)(uintptr_t)(((k + 1 > oe) ? nerow : erow) + 8u * c1);
      w_cur = w_nxt;
      w_nxt = byte_at(((oe <= k + 1) ? ycur - HXB2_W : ycur) + (unsigned)(k + 2));
      return now;
    };

    auto step = [&](const int k, const bool renorm, const L5& left, L5& out, L5& u1, L5& u2, const YSide ys) {
      const d2v rc = ys.rc;
      const int y_wait = (int)((ys.w & 0x80u) << 22);          // y state not ready: 2^29, else 0
      const double em = ys.em;
      int E = left.e > u1.e ? left.e : u1.e;
      E = E > u2.e ? E : u2.e;
      // (a cell outside the envelope - src/forward.h:92-98 - is shifted to zero as a whole: the exponent it is brought to lies 2^29 higher)
      const int Eo = E + ((unsigned)(k - as) <= span ? 0 : (1 << 29));
      const int du = (u1.e - y_wait) - Eo, dl = (left.e - x_wait) - Eo, dd = u2.e - Eo;
      if (DIR == 1) {
        // Backward (src/forward.cpp:1018-1065 for leaf-like profiles): the five destination terms brought to the cell's
        // exponent (a move that may not be made, a cell outside the envelope: shifted out of range), then the sums in the
        // reference's accumulation order
        const double tD = u2.imm * em;
        const double t1x = u1.imd * xc_rs, t2x = u1.iiw * xc_in;
        const double t1y = left.idm * rc.x, t2y = left.imi * rc.y;
        const double D = __builtin_ldexp(tD, dd);
        const double d1x = __builtin_ldexp(t1x, du), d2x = __builtin_ldexp(t2x, du);
        const double d1y = __builtin_ldexp(t1y, dl), d2y = __builtin_ldexp(t2y, dl);
        out.imm = lin_acc<TRUNC>(P[0][3], d2y, lin_acc<TRUNC>(P[0][2], d1y, lin_acc<TRUNC>(P[0][4], d2x, lin_acc<TRUNC>(P[0][1], d1x, P[0][0] * D))));
        out.imd = lin_acc<TRUNC>(P[1][2], d1y, lin_acc<TRUNC>(P[1][1], d1x, P[1][0] * D));
        out.idm = lin_acc<TRUNC>(P[2][2], d1y, lin_acc<TRUNC>(P[2][1], d1x, P[2][0] * D));
        out.imi = lin_acc<TRUNC>(P[3][3], d2y, lin_acc<TRUNC>(P[3][4], d2x, lin_acc<TRUNC>(P[3][1], d1x, P[3][0] * D)));
        out.iiw = lin_acc<TRUNC>(P[4][2], d1y, lin_acc<TRUNC>(P[4][4], d2x, P[4][0] * D));
      } else {
        // the five sums of src/forward.cpp:103-115,139-150,171-180 on probabilities, left-nested as the reference's
        double s_imd = u1.imm * P[0][1];
        double s_iiw = u1.imm * P[0][4];
        double s_idm = left.imm * P[0][2];
        double s_imi = left.imm * P[0][3];
        double s_imm = u2.imm * P[0][0];
        s_imd = lin_acc<TRUNC>(u1.imd, P[1][1], s_imd);
        s_iiw = lin_acc<TRUNC>(u1.imi, P[3][4], s_iiw);
        s_idm = lin_acc<TRUNC>(left.imd, P[1][2], s_idm);
        s_imi = lin_acc<TRUNC>(left.imi, P[3][3], s_imi);
        s_imm = lin_acc<TRUNC>(u2.imd, P[1][0], s_imm);
        s_imd = lin_acc<TRUNC>(u1.idm, P[2][1], s_imd);
        s_iiw = lin_acc<TRUNC>(u1.iiw, P[4][4], s_iiw);
        s_idm = lin_acc<TRUNC>(left.idm, P[2][2], s_idm);
        s_imm = lin_acc<TRUNC>(u2.idm, P[2][0], s_imm);
        s_imd = lin_acc<TRUNC>(u1.imi, P[3][1], s_imd);
        s_idm = lin_acc<TRUNC>(left.iiw, P[4][2], s_idm);
        s_imm = lin_acc<TRUNC>(u2.imi, P[3][0], s_imm);
        s_imm = lin_acc<TRUNC>(u2.iiw, P[4][0], s_imm);
        // the three source groups brought to the cell's exponent; a state that may not be entered (y or x state not ready:
        // src/forward.cpp:97,133) and a cell outside the envelope are shifted out of range: zero
        out.imd = __builtin_ldexp(s_imd * xc_rs, du);
        out.iiw = __builtin_ldexp(s_iiw * xc_in, du);
        out.idm = __builtin_ldexp(s_idm * rc.x, dl);
        out.imi = __builtin_ldexp(s_imi * rc.y, dl);
        out.imm = __builtin_ldexp(s_imm * em, dd);
      }
      out.e = E;
      // (compile-time: the first TWO steps of every block of eight.  A cell takes the largest exponent of its three sources,
      // and the sources of step k + 2 are cells of steps k and k + 1: renormalising one step only would leave the stale
      // exponent alive in the cells of the other parity for ever)
      if (renorm) {
        if (k == 0 && r32 == 0 && live) {
          if (DIR == 0) { out.imm = 1.0; out.e = 0; }   // cell (0,0): lpStart() = 0 (src/forward.cpp:73)
          else {
            // the cell feeding END is initialised by assignment (src/forward.cpp:981-995)
            out.imm = end_cell[0]; out.imd = end_cell[1]; out.idm = end_cell[2]; out.imi = end_cell[3]; out.iiw = end_cell[4];
            out.e = 0;
          }
        }
        const double mx = vmax(vmax(vmax(out.imm, out.imd), vmax(out.idm, out.imi)), out.iiw);
        const int kk = __builtin_amdgcn_frexp_exp(mx);
        out.imm = __builtin_ldexp(out.imm, -kk);
        out.imd = __builtin_ldexp(out.imd, -kk);
        out.idm = __builtin_ldexp(out.idm, -kk);
        out.imi = __builtin_ldexp(out.imi, -kk);
        out.iiw = __builtin_ldexp(out.iiw, -kk);
        out.e = mx > 0. ? out.e + kk : HXB2_EMIN;
      }
      u2 = rot(rot_addr, out);                       // the previous lane's new cell: next step's upper neighbour
    };

    // a pair of steps: in the strip-skewed layout the two cells are adjacent, 16 bytes per lane and state plane
    auto step_pair = [&](const int k, const bool renorm) {
      roll_even(k);
      const bool own = k >= os && k <= oe;           // (owned spans are whole step pairs)
      HX_GLOBAL d2v* M2 = rowp + ((k >> 1) << 6);     // (step-pair blocks of 128 doubles: hx_api.hip admits dense planes and compressed windows alike)
      step(k, renorm, lb, la, lua, lub, y_side(k));
      const double l0 = log_scaled(la.imm, la.e), l1 = log_scaled(la.imd, la.e), l2 = log_scaled(la.idm, la.e),
                   l3 = log_scaled(la.imi, la.e), l4 = log_scaled(la.iiw, la.e);
      roll_odd(k + 1);
      step(k + 1, renorm, la, lb, lub, lua, y_side(k + 1));
      const double h0 = log_scaled(lb.imm, lb.e), h1 = log_scaled(lb.imd, lb.e), h2 = log_scaled(lb.idm, lb.e),
                   h3 = log_scaled(lb.imi, lb.e), h4 = log_scaled(lb.iiw, lb.e);
      if (own) {
        // write-once data: non-temporal stores
        __builtin_nontemporal_store(d2v{l0, h0}, &M2[0]);
        __builtin_nontemporal_store(d2v{l1, h1}, &M2[plane2]);
        __builtin_nontemporal_store(d2v{l2, h2}, &M2[2 * plane2]);
        __builtin_nontemporal_store(d2v{l3, h3}, &M2[3 * plane2]);
        __builtin_nontemporal_store(d2v{l4, h4}, &M2[4 * plane2]);
      }
    };
    for (int k = 0; k < n_steps; k += 8) {
      step_pair(k, true);                            // (mantissas renormalised every eighth step)
      step_pair(k + 2, false);
      step_pair(k + 4, false);
      step_pair(k + 6, false);
    }
  }
  // (the drain tools/check_landing_regs.py looks for: every hand-issued fetch has retired)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// lpEnd / lpStart of the pairs, behind the sweep AND the edge kernel (which may run beside the sweep on another stream):
// lpEnd reads cell (Nx-2, Ny-2) - the sweep's last cell or, with a one-row band, an edge cell.
template <int DIR>
__global__ void __launch_bounds__(64)
k_band2_result(const DevJob* __restrict__ jobs, const double* __restrict__ exact_tab, const int n_jobs) {
  const int job = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (job >= n_jobs) return;
  const DevJob& J = jobs[job];
  if (DIR == 0) *J.lp_end = forward_lp_end(J, ExactLse{exact_tab});
  else *J.lp_start = J.bwd[cell_slot_blk(J.strip_stride, J.blk, J.n_rows - 1, J.n_cols - 1)];   // B(START, START).IMM: the sweep's last cell
}

Band2Plan plan_band2(int nw, int max_rows, int max_cols, int max_cls) {
  Band2Plan p;
  int a = 0;
  p.sbase = a; a += (4 * ((max_rows + 63) / 64 + 1) + 15) & ~15;
  p.ycol = a; a += (max_cols + 15) & ~15;
  p.yclass = a; a += 16 * (max_cls + 1);
  p.xclass = a; a += 16 * (max_cls + 1);
  p.elds = a; a += (8 * (max_cls + 1) * (max_cls + 1) + 15) & ~15;
  p.stride = a;
  p.in_hole = (HXB2_HOLE_END - HXB2_HOLE_BEGIN) / a;
  if (p.in_hole > 2 * nw) p.in_hole = 2 * nw;
  p.total = 16 * HXB2_LOG_ENTRIES + (2 * nw - p.in_hole) * a;
  return p;
}

template <bool TRUNC, int NW, int DIR>
int launch_b2(const DevJob* d_jobs, int n_jobs, const Band2Plan& p, const double* tab, const double* log_tab, int write_edges, hipStream_t st,
              hipStream_t edge_st) {
  if (p.total > HX_LDS_LIMIT) return launch_fail("k_fill_band2<%d> needs %d bytes of LDS (limit %d)", NW, p.total, HX_LDS_LIMIT);
  // the sweep first: it takes two workgroups' worth of every CU, the edge kernel's wavefronts fit beside them
  hipLaunchKernelGGL((k_fill_band2<TRUNC, NW, DIR>), dim3((n_jobs + 2 * NW - 1) / (2 * NW)), dim3(NW * 64), p.total, st, d_jobs, tab,
                     log_tab, p, n_jobs);
  hipLaunchKernelGGL((k_band2_edges<TRUNC, DIR>), dim3((n_jobs + 3) / 4), dim3(256), 0, edge_st, d_jobs, n_jobs, write_edges);
  return 0;
}

template <int DIR>
int launch_band2_dir(const DevJob* d_jobs, int n_jobs, bool trunc, int max_rows, int max_cols, int max_cls, Tab8 tab8, Tab16 log_tab,
                     bool write_edges, hipStream_t st, hipStream_t edge_st) {
  if (max_cls + 1 > 127) return launch_fail("%d emission classes exceed the two-pairs-per-wavefront sweep's column bytes", max_cls);
  const char* v = getenv("HX_BAND2_NW");           // tuning / test hook: sweeping wavefronts per workgroup (1, 2 or 4)
  int nw = v ? atoi(v) : 0;
  // (four: a workgroup is then one wavefront per SIMD of its CU and two workgroups fill the CU; workgroups of two wavefronts are
  // placed two to a SIMD pair - 2048 pairs 3.3 ms with four, 4.4 ms with two)
  if (nw != 1 && nw != 2 && nw != 4) nw = 4;
  const int we = write_edges ? 1 : 0;
#define HXB2_GO(NW_) do { const Band2Plan p = plan_band2(NW_, max_rows, max_cols, max_cls); \
    return trunc ? launch_b2<true, NW_, DIR>(d_jobs, n_jobs, p, tab8.p, log_tab.p, we, st, edge_st) \
                 : launch_b2<false, NW_, DIR>(d_jobs, n_jobs, p, tab8.p, log_tab.p, we, st, edge_st); } while (0)
  if (nw == 4) HXB2_GO(4);
  if (nw == 2) HXB2_GO(2);
  HXB2_GO(1);
#undef HXB2_GO
}

}  // namespace

bool band2_kernel_fits(int rows, int cols, int cls) {
  return cls + 1 <= 127 && plan_band2(1, rows, cols, cls).total <= HX_LDS_LIMIT;
}

void launch_band2_result(const DevJob* d_jobs, int n_jobs, int dir, Tab8 tab8, hipStream_t st) {
  if (dir == 0) hipLaunchKernelGGL(k_band2_result<0>, dim3((n_jobs + 63) / 64), dim3(64), 0, st, d_jobs, tab8.p, n_jobs);
  else hipLaunchKernelGGL(k_band2_result<1>, dim3((n_jobs + 63) / 64), dim3(64), 0, st, d_jobs, tab8.p, n_jobs);
}

int launch_forward_band2(const DevJob* d_jobs, int n_jobs, bool trunc, int max_rows, int max_cols, int max_cls, Tab8 tab8, Tab16 log_tab,
                         bool write_edges, hipStream_t st, hipStream_t edge_st) {
  return launch_band2_dir<0>(d_jobs, n_jobs, trunc, max_rows, max_cols, max_cls, tab8, log_tab, write_edges, st, edge_st);
}
int launch_backward_band2(const DevJob* d_jobs, int n_jobs, bool trunc, int max_rows, int max_cols, int max_cls, Tab8 tab8, Tab16 log_tab,
                          bool write_edges, hipStream_t st, hipStream_t edge_st) {
  return launch_band2_dir<1>(d_jobs, n_jobs, trunc, max_rows, max_cols, max_cls, tab8, log_tab, write_edges, st, edge_st);
}

}  // namespace hx
