// Column sum-product on a phylogenetic tree and the eigen-basis substitution-count accumulation of counts mode
// (SURVEY 8f row N3): the reference's SumProduct::initColumn / fillUp / fillDown (src/sumprod.cpp:58-198),
// accumulateRootCounts and accumulateEigenCounts (src/sumprod.cpp:264-271, 294-372), for a BATCH of alignment columns.
//
// In the reference one SumProduct object walks the columns one after the other (AlignColSumProduct, and once per
// posterior-weighted DP cell in BackwardMatrix::getCounts).  Columns and mixture components are independent, so here a
// (column, component) is a thread (k_sumprod_columns): every thread runs the tip-to-root and root-to-tip passes of its column - tiny tree recursions,
// A x A matrix-vector products per branch and mixture component - with its messages E, F, G in a global scratch laid out
// in blocks of 64 columns ([block][.][64 lanes], two vector entries per lane and access), then turns the messages of every branch
// into the eigen basis (D_k, U_l).  The reference adds weight x D_k J_kl U_l to the count matrix column by column; J_kl
// (eigenSubCount of the branch) does not depend on the column, so the sum over columns is an outer-product sum
// J_kl x sum_col D_k(col) U_l(col): a second kernel (k_outer_counts) reduces the columns of one (component, branch) per
// workgroup, A x A accumulators in registers, the bases staged through LDS, and multiplies by J once at the end.  When
// every eigenvector is real (reversible models: all the reference's presets) the imaginary halves are neither written nor
// read.  Root counts are written per column and summed by rows (k_row_sums).
//
// Arithmetic follows the reference: linear-space messages with per-node log scale factors, rescaling below 1e-30, the
// column likelihood combined over components with the table log_sum_exp.  Sums over columns are taken in a different
// order from the reference's: results agree to rounding (tests/test_gpu_sumprod.py: 1e-10), not bit for bit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/historian_hip.h"
#include "hx_lse.h"
#include "hx_policy.h"
#include "hx_kernels.h"

namespace hx {

namespace {

#define HX_SP_RESCALE 1e-30        // SUMPROD_RESCALE_THRESHOLD (src/sumprod.cpp:8)
#define HX_SP_TILE 32              // columns per LDS tile of k_outer_counts
#ifndef HX_SP_WAVES_PER_SIMD
#define HX_SP_WAVES_PER_SIMD 2    // the column kernel's register budget: 512 / this.  Four waves per SIMD (128 registers, 49 spilled doubles)
                                  // run at the same rate: the kernel is bound by the traffic it moves, not by the latency it hides
#endif
#define HX_SP_PAIRS 16             // (k,l) pairs per thread of k_outer_counts: A*A <= 256 * 16, A <= 64

struct SpModel {
  int A, C, N, real_basis;
  const int* parent;               // [N] post-order: children before parents, root last; -1 for the root
  const int* child;                // [N][2] children or -1 (binary trees, as the reconstruction builds them)
  const double* ins_prob;          // [C][A]
  const double* log_cpt_weight;    // [C]
  const double* branch_sub;        // [C][N][A][A] exp(R t_r)
  const double* evec_re;           // [C][A][A] right eigenvectors, and their inverse
  const double* evec_im;
  const double* einv_re;
  const double* einv_im;
  const double* esc_re;            // [C][N][A][A] eigenSubCount(t_r)
  const double* esc_im;
};

// scratch of one chunk of columns, in blocks of 64 columns: messages E, G [block][cpt][node][a][64]; scale factors and max U
// [block][cpt][node][64]; F at the column's root and the root-count terms [block][cpt][a][64]; bases
// [block][cpt][node][Ur, Dr, (Ui, Di)][a][64]
struct SpScratch { double* E; double* G; double* logE; double* logF; double* logG; double* maxU; double* Froot; double* rootc; double* basis; };

// Model matrices are read through the constant address space: their addresses are uniform over the wavefront and
// nothing writes them, so the compiler fetches them with scalar loads and feeds the FMAs from SGPRs.
typedef const __attribute__((address_space(4))) double* CMat;
__device__ __forceinline__ CMat cmat(const double* p) { return (CMat)(unsigned long long)p; }

// Matrix-vector products with a matrix that a wave keeps in LDS (row-major, A x A, 16-byte aligned rows: A even).  Every
// lane reads the same entries (broadcast reads, two entries per ds_read_b128); half a row is fetched half a row ahead of
// its use into registers of its own, so that no read is waited for - written the plain way the compiler, short of
// registers, reused one register pair for every read and waited for each (one multiply-add per ~50 cycles).  Products
// with the transposed matrix (y = x M) use the same routine on a transposed LDS copy: one vector in registers, results
// handed out as they are complete - the kernel stays within 128 registers, four waves per SIMD.
typedef double sp_d2 __attribute__((ext_vector_type(2)));
template <int H>                                    // H entries (H even)
struct LdsPiece {
  sp_d2 v[H / 2];
  __device__ __forceinline__ void fetch(const HX_LDS double* p) {
#pragma unroll
    for (int q = 0; q < H / 2; ++q) v[q] = reinterpret_cast<const HX_LDS sp_d2*>(p)[q];
  }
  __device__ __forceinline__ double at(const int b) const { return (b & 1) ? v[b >> 1].y : v[b >> 1].x; }
};
// y[a] = sum_b M[a][b] x[b], handed out two rows at a time (A even), the way the scratch stores them; two partial sums per
// row and half (the additions do not wait for one another)
template <int A, class Emit>
__device__ __forceinline__ void lds_mat_vec(const HX_LDS double* M, const double (&x)[A], const Emit& emit) {
  constexpr int H0 = ((A / 2) + 1) & ~1, H1 = A - H0;      // a row in two pieces of even length (20 = 10 + 10, 4 = 2 + 2)
  static_assert(H1 >= 0 && (H1 & 1) == 0, "even alphabet sizes");
  LdsPiece<H0> lo;
  LdsPiece<(H1 > 0 ? H1 : 2)> hi;
  lo.fetch(M);
  double ya = 0.;
#pragma unroll
  for (int a = 0; a < A; ++a) {
    if (H1 > 0) hi.fetch(M + a * A + H0);
    double p0 = 0., p1 = 0.;
#pragma unroll
    for (int b = 0; b < H0; b += 2) {
      p0 = __builtin_fma(lo.at(b), x[b], p0);
      p1 = __builtin_fma(lo.at(b + 1), x[b + 1], p1);
    }
    if (a + 1 < A) lo.fetch(M + (a + 1) * A);
    if (H1 > 0) {
#pragma unroll
      for (int b = 0; b < H1; b += 2) {
        p0 = __builtin_fma(hi.at(b), x[H0 + b], p0);
        p1 = __builtin_fma(hi.at(b + 1), x[H0 + b + 1], p1);
      }
    }
    const double y = p0 + p1;
    if (a & 1) emit(a >> 1, ya, y);
    else ya = y;
  }
}

// TA: the alphabet size as a compile-time constant (message vectors live in registers, loops unrolled), or 0 for any
// alphabet of up to 64 symbols (vectors in private memory).
// LM: a wave keeps the matrices it multiplies with in LDS - exp(R t) of the branch at hand (re-staged per node), the real
// parts of the eigenvectors and of their inverse (per component) - and reads their entries as LDS broadcasts; without it
// they come through the scalar cache, which the 3 KB per matrix-vector product overrun (2.9 TFLOP/s, one FMA per ~100 cycles).
template <int TA, bool LM>
__global__ void __launch_bounds__(512, HX_SP_WAVES_PER_SIMD) k_sumprod_columns(const SpModel m, const signed char* __restrict__ tok, const double* __restrict__ weight,
                                                         const long long n_cols, const SpScratch s, const double* __restrict__ lse_tab,
                                                         double* __restrict__ col_log_like, double* __restrict__ root_post) {
  constexpr int AX = TA ? TA : 64;
  const int A = TA ? TA : m.A, C = m.C, N = m.N, AA = A * A, AP = (A + 1) & ~1;
  const int parts = m.real_basis ? 2 : 4;
  // Scratch layout: blocks of 64 columns, [block][.][64] - everything a wavefront touches in a pass lies in one
  // contiguous stretch (a few hundred KB) instead of one 512-byte piece per row of a [.][all columns] array, rows
  // n_cols * 8 bytes apart (two thousand pages per wavefront).  `cb`: this wave's block, `cl` = the lane.
  // Inside a block a message vector is stored two entries per lane and row, [a / 2][64][2]: a lane moves 16 bytes per
  // memory instruction (8-byte accesses run at 0.5-0.7 of the 16-byte rate).  AP = A rounded up to even.
#define ROWP(P, idx) ((P) + (idx) * (long long)(AP * 64) + cl * 2)
#define AT(P, cpt, r, a) ROWP(P, (cb * C + (cpt)) * N + (r))[((a) >> 1) * 128 + ((a) & 1)]
#define LG(P, cpt, r) P[((cb * C + (cpt)) * N + (r)) * 64 + cl]
#define BS(cpt, r, part, l) ROWP(s.basis, ((cb * C + (cpt)) * N + (r)) * parts + (part))[((l) >> 1) * 128 + ((l) & 1)]
#define FR(cpt, a) s.Froot[((cb * C + (cpt)) * A + (a)) * 64 + cl]
  // a wavefront is 64 columns of one mixture component (the component is uniform over the wave, so the model's matrices
  // still come through scalar loads); the waves of a workgroup share the columns and split the components.  They meet
  // twice: for the column likelihood (through LDS) and for the root posterior (through the scratch).
  extern __shared__ double sh_ll[];                 // [C][64], then per wave [3][A * A]: exp(R t), evecInv, evec (real parts)
  const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63, Wb = (int)blockDim.x >> 6;
  HX_LDS double* Lsub = (HX_LDS double*)(sh_ll + 64 * C + (LM ? 3 * AA * wave : 0));
  HX_LDS double* Linv = Lsub + AA;
  HX_LDS double* Lvec = Linv + AA;
  auto wave_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
  for (long long base = (long long)blockIdx.x * 64; base < n_cols; base += (long long)gridDim.x * 64) {
    const bool busy = base + lane < n_cols;
    const long long col = busy ? base + lane : 0;
    const long long cb = base >> 6;
    const int cl = lane;
    // a whole row of the scratch, 16 bytes per access when the alphabet size is a compile-time even number
    auto put_row = [&](double* rp, auto&& gen) {
      if constexpr (TA != 0 && TA % 2 == 0) {
#pragma unroll
        for (int q = 0; q < TA / 2; ++q) *reinterpret_cast<sp_d2*>(rp + q * 128) = sp_d2{gen(2 * q), gen(2 * q + 1)};
      } else {
        for (int a = 0; a < A; ++a) rp[(a >> 1) * 128 + (a & 1)] = gen(a);
      }
    };
    auto get_row = [&](const double* rp, double (&v)[AX]) {
      if constexpr (TA != 0 && TA % 2 == 0) {
#pragma unroll
        for (int q = 0; q < TA / 2; ++q) {
          const sp_d2 t2 = *reinterpret_cast<const sp_d2*>(rp + q * 128);
          v[2 * q] = t2.x; v[2 * q + 1] = t2.y;
        }
      } else {
        for (int a = 0; a < A; ++a) v[a] = rp[(a >> 1) * 128 + (a & 1)];
      }
    };
#define EROW(P, cpt, r) ROWP(P, (cb * C + (cpt)) * N + (r))
#define BROW(cpt, r, part) ROWP(s.basis, ((cb * C + (cpt)) * N + (r)) * parts + (part))
    const signed char* t = tok + col * N;       // -2 gap, -1 wildcard, else the residue's token
    const double w = weight ? weight[col] : 1.;
    int root = -1;
    for (int r = 0; r < N; ++r)
      if (t[r] != -2 && (m.parent[r] < 0 || t[m.parent[r]] == -2)) root = r;     // (one root per column: the caller's contract)
    double cll = HX_NEG_INF;
    // ---- tip-to-root (src/sumprod.cpp:99-161); U of every branch in the eigen basis on the way (src/sumprod.cpp:318-340) ----
    for (int cpt = wave; cpt < C; cpt += Wb) {
      double cpt_ll = 0.;
      CMat ins = cmat(m.ins_prob + cpt * A);
      if (LM) {
        wave_sync();
        for (int k = lane; k < AA; k += 64) {
          Linv[k] = m.einv_re[(long long)cpt * AA + k];
          // (the root-to-tip pass multiplies with the transposed matrices: staged transposed where the row routine is used)
          if constexpr (TA != 0 && TA % 2 == 0) Lvec[(k % A) * A + k / A] = m.evec_re[(long long)cpt * AA + k];
          else Lvec[k] = m.evec_re[(long long)cpt * AA + k];
        }
      }
      for (int r = 0; r < N; ++r) {
        if (LM && m.parent[r] >= 0) {
          wave_sync();                               // (the reads of the previous node's matrix are done)
          for (int k = lane; k < AA; k += 64) Lsub[k] = m.branch_sub[((long long)cpt * N + r) * AA + k];
          wave_sync();
        }
        if (!busy) continue;
        const int c0 = m.child[2 * r], c1 = m.child[2 * r + 1];
        double lf = (c0 >= 0 ? LG(s.logE, cpt, c0) : 0.) + (c1 >= 0 ? LG(s.logE, cpt, c1) : 0.);
        const int tk = t[r];
        if (tk == -2) {
          // a gap: its message to the parent is all ones, and no counts on the branch above it
          put_row(EROW(s.E, cpt, r), [](int) { return 1.; });
          LG(s.logE, cpt, r) = 0.;
          LG(s.logF, cpt, r) = lf;
          put_row(BROW(cpt, r, 0), [](int) { return 0.; });
          if (parts == 4)
            for (int l = 0; l < A; ++l) BS(cpt, r, 2, l) = 0.;
          continue;
        }
        CMat sub_s = cmat(m.branch_sub + ((long long)cpt * N + r) * AA);
        CMat ir_s = cmat(m.einv_re + (long long)cpt * AA);
        CMat ii = cmat(m.einv_im + (long long)cpt * AA);
        auto sub = [&](const int k) -> double { return LM ? Lsub[k] : sub_s[k]; };
        auto ir = [&](const int k) -> double { return LM ? Linv[k] : ir_s[k]; };
        if (tk >= 0) {
          // a residue: F is one-hot, E and U are columns of the matrices
          double f = (c0 >= 0 ? AT(s.E, cpt, c0, tk) : 1.) * (c1 >= 0 ? AT(s.E, cpt, c1, tk) : 1.);
          if (f < HX_SP_RESCALE) { lf += log(f); f = 1.; }
          LG(s.logF, cpt, r) = lf;
          if (r == root) {
#pragma unroll
            for (int a = 0; a < A; ++a) FR(cpt, a) = a == tk ? f : 0.;
            cpt_ll += lf + log(f * m.ins_prob[cpt * A + tk]);
          } else {
            LG(s.logE, cpt, r) = lf;
            LG(s.maxU, cpt, r) = f;
            const double* subg = m.branch_sub + ((long long)cpt * N + r) * AA;
            put_row(EROW(s.E, cpt, r), [&](const int a) { return (LM ? Lsub[a * A + tk] : subg[a * A + tk]) * f; });
            put_row(BROW(cpt, r, 0), [&](const int l) { return (LM ? Linv[l * A + tk] : m.einv_re[(long long)cpt * AA + l * A + tk]) * (f / f); });
            if (parts == 4)
              for (int l = 0; l < A; ++l) BS(cpt, r, 2, l) = m.einv_im[(long long)cpt * AA + l * A + tk] * (f / f);
          }
          continue;
        }
        // a wildcard: the full vector
        double f[AX];
        double fmax = 0.;
        {
          double e1[AX];
#pragma unroll
          for (int a = 0; a < A; ++a) f[a] = e1[a] = 1.;
          if (c0 >= 0) get_row(EROW(s.E, cpt, c0), f);
          if (c1 >= 0) get_row(EROW(s.E, cpt, c1), e1);
#pragma unroll
          for (int a = 0; a < A; ++a) {
            f[a] *= e1[a];
            fmax = f[a] > fmax ? f[a] : fmax;
          }
        }
        if (fmax < HX_SP_RESCALE) {
#pragma unroll
          for (int a = 0; a < A; ++a) f[a] /= fmax;
          lf += log(fmax);
          fmax = 1.;
        }
        LG(s.logF, cpt, r) = lf;
        if (r == root) {
          double ip = 0.;
#pragma unroll
          for (int a = 0; a < A; ++a) {
            FR(cpt, a) = f[a];
            ip += f[a] * ins[a];
          }
          cpt_ll += lf + log(ip);
          continue;
        }
        LG(s.logE, cpt, r) = lf;
        LG(s.maxU, cpt, r) = fmax;
        if constexpr (LM && TA != 0 && TA % 2 == 0) {
          {
            double* rp = EROW(s.E, cpt, r);
            lds_mat_vec<AX>(Lsub, f, [&](const int q, const double ea, const double eb) { *reinterpret_cast<sp_d2*>(rp + q * 128) = sp_d2{ea, eb}; });
          }
          const double inv_fmax = 1. / fmax;
#pragma unroll
          for (int a = 0; a < A; ++a) f[a] *= inv_fmax;
          {
            double* rp = BROW(cpt, r, 0);
            lds_mat_vec<AX>(Linv, f, [&](const int q, const double ua, const double ub) { *reinterpret_cast<sp_d2*>(rp + q * 128) = sp_d2{ua, ub}; });
          }
        } else {
#pragma unroll
          for (int a = 0; a < A; ++a) {
            double e = 0.;
#pragma unroll
            for (int b = 0; b < A; ++b) e += sub(a * A + b) * f[b];
            AT(s.E, cpt, r, a) = e;
          }
#pragma unroll
          for (int a = 0; a < A; ++a) f[a] /= fmax;
#pragma unroll
          for (int l = 0; l < A; ++l) {
            double ur = 0.;
#pragma unroll
            for (int b = 0; b < A; ++b) ur += ir(l * A + b) * f[b];
            BS(cpt, r, 0, l) = ur;
          }
        }
        if (parts == 4)
          for (int l = 0; l < A; ++l) {
            double ui = 0.;
#pragma unroll
            for (int b = 0; b < A; ++b) ui += ii[l * A + b] * f[b];
            BS(cpt, r, 2, l) = ui;
          }
      }
      sh_ll[cpt * 64 + lane] = cpt_ll;
    }
    __syncthreads();
    for (int cpt = 0; cpt < C; ++cpt) cll = lse(cll, m.log_cpt_weight[cpt] + sh_ll[cpt * 64 + lane], lse_tab);
    if (wave == 0 && busy) col_log_like[col] = cll;
    // ---- root-to-tip (src/sumprod.cpp:163-198); D of every branch in the eigen basis on the way (src/sumprod.cpp:341-360) ----
    for (int cpt = wave; cpt < C; cpt += Wb) {
      CMat ins = cmat(m.ins_prob + cpt * A);
      if (LM && C > Wb) {                            // (with a wave per component the eigenvectors are still there)
        wave_sync();
        for (int k = lane; k < AA; k += 64) {
          if constexpr (TA != 0 && TA % 2 == 0) Lvec[(k % A) * A + k / A] = m.evec_re[(long long)cpt * AA + k];
          else Lvec[k] = m.evec_re[(long long)cpt * AA + k];
        }
      }
      for (int r = N - 1; r >= 0; --r) {
        if (LM && m.parent[r] >= 0) {
          wave_sync();
          for (int k = lane; k < AA; k += 64) {
            if constexpr (TA != 0 && TA % 2 == 0) Lsub[(k % A) * A + k / A] = m.branch_sub[((long long)cpt * N + r) * AA + k];
            else Lsub[k] = m.branch_sub[((long long)cpt * N + r) * AA + k];
          }
          wave_sync();
        }
        if (!busy) continue;
        if (t[r] == -2 || r == root) {
          if (r == root) {
            put_row(EROW(s.G, cpt, r), [&](const int a) { return ins[a]; });
            LG(s.logG, cpt, r) = 0.;
            put_row(BROW(cpt, r, 0), [](int) { return 0.; });         // no branch above the root: U was not written on the way up
            if (parts == 4)
              for (int l = 0; l < A; ++l) BS(cpt, r, 2, l) = 0.;
          }
          put_row(BROW(cpt, r, 1), [](int) { return 0.; });
          if (parts == 4)
            for (int l = 0; l < A; ++l) BS(cpt, r, 3, l) = 0.;
          continue;
        }
        const int p = m.parent[r];
        const int sib = m.child[2 * p] == r ? m.child[2 * p + 1] : m.child[2 * p];
        const double le_sib = sib >= 0 ? LG(s.logE, cpt, sib) : 0.;
        const double lg_p = LG(s.logG, cpt, p);
        LG(s.logG, cpt, r) = lg_p + le_sib;
        CMat sub_s = cmat(m.branch_sub + ((long long)cpt * N + r) * AA);
        CMat vr_s = cmat(m.evec_re + (long long)cpt * AA);
        CMat vi = cmat(m.evec_im + (long long)cpt * AA);
        auto sub = [&](const int k) -> double { return LM ? Lsub[k] : sub_s[k]; };
        auto vr = [&](const int k) -> double { return LM ? Lvec[k] : vr_s[k]; };
        // what flows down the branch: the parent's outside message times the sibling's subtree
        double d[AX];
        double max_d = 0.;
        {
          double es[AX];
#pragma unroll
          for (int a = 0; a < A; ++a) es[a] = 1.;
          get_row(EROW(s.G, cpt, p), d);
          if (sib >= 0) get_row(EROW(s.E, cpt, sib), es);
#pragma unroll
          for (int a = 0; a < A; ++a) {
            d[a] *= es[a];
            max_d = d[a] > max_d ? d[a] : max_d;
          }
        }
        if constexpr (LM && TA != 0 && TA % 2 == 0) {
          double* rp = EROW(s.G, cpt, r);
          lds_mat_vec<AX>(Lsub, d, [&](const int q, const double ga, const double gb) { *reinterpret_cast<sp_d2*>(rp + q * 128) = sp_d2{ga, gb}; });
        } else {
#pragma unroll
          for (int b = 0; b < A; ++b) {
            double g = 0.;
#pragma unroll
            for (int a = 0; a < A; ++a) g += d[a] * sub(a * A + b);
            AT(s.G, cpt, r, b) = g;
          }
        }
        const double norm = exp(cll - m.log_cpt_weight[cpt] - LG(s.logF, cpt, r) - lg_p - le_sib) / (LG(s.maxU, cpt, r) * max_d);
        const double scale = w / norm;
        if constexpr (LM && TA != 0 && TA % 2 == 0) {
          const double inv_max = 1. / max_d;
#pragma unroll
          for (int a = 0; a < A; ++a) d[a] *= inv_max;
          double* rp = BROW(cpt, r, 1);
          lds_mat_vec<AX>(Lvec, d, [&](const int q, const double da, const double db) { *reinterpret_cast<sp_d2*>(rp + q * 128) = sp_d2{da * scale, db * scale}; });
        } else {
#pragma unroll
          for (int a = 0; a < A; ++a) d[a] /= max_d;
#pragma unroll
          for (int k = 0; k < A; ++k) {
            double dr = 0.;
#pragma unroll
            for (int a = 0; a < A; ++a) dr += vr(a * A + k) * d[a];
            BS(cpt, r, 1, k) = dr * scale;
          }
        }
        if (parts == 4)
          for (int k = 0; k < A; ++k) {
            double di = 0.;
#pragma unroll
            for (int a = 0; a < A; ++a) di += vi[a * A + k] * d[a];
            BS(cpt, r, 3, k) = di * scale;
          }
      }
    }
    __syncthreads();            // (the other components' F, G at the root, written to the scratch by the neighbouring threads)
    // ---- posterior of the root's residue (src/sumprod.cpp:208-217) ----
    if (root_post && busy && wave == 0)
      for (int a = 0; a < A; ++a) {
        double lp = HX_NEG_INF;
        if (root >= 0)
          for (int cpt = 0; cpt < C; ++cpt)
            lp = lse(lp, m.log_cpt_weight[cpt] + LG(s.logF, cpt, root) + log(FR(cpt, a)) + LG(s.logG, cpt, root) +
                             log(AT(s.G, cpt, root, a)) - cll, lse_tab);
        root_post[col * A + a] = lp < 0. ? lp : 0.;
      }
    // ---- this column's terms of the root counts (src/sumprod.cpp:264-271) ----
    for (int cpt = wave; cpt < C && busy; cpt += Wb) {
      const double norm = root >= 0 ? exp(m.log_cpt_weight[cpt] + LG(s.logF, cpt, root) - cll) : 0.;
      for (int a = 0; a < A; ++a)
        s.rootc[((cb * C + cpt) * A + a) * 64 + cl] = root >= 0 ? w * m.ins_prob[cpt * A + a] * FR(cpt, a) * norm : 0.;
    }
    __syncthreads();
  }
#undef AT
#undef ROWP
#undef EROW
#undef BROW
#undef LG
#undef BS
#undef FR
}

// eigenCounts[cpt][k][l] += J[cpt][node][k][l] * sum over this chunk's columns of D_k(col) U_l(col)   (src/sumprod.cpp:361-370)
// grid (C * N branches, column slices); 256 threads, thread t owns the pairs (k,l) = t, t + 256, ...
template <bool REAL>
__global__ void __launch_bounds__(256) k_outer_counts(const SpModel m, const double* __restrict__ basis, const long long n_cols,
                                                      double* __restrict__ eig_re, double* __restrict__ eig_im) {
  extern __shared__ double tile[];                 // [parts * A][HX_SP_TILE + 1]
  constexpr int PARTS = REAL ? 2 : 4, TS = HX_SP_TILE + 1;
  const int A = m.A, AA = A * A, N = m.N;
  const int branch = blockIdx.x, cpt = branch / N, r = branch - cpt * N;
  if (m.parent[r] < 0) return;
  const int AP = (A + 1) & ~1;
  const long long blk_doubles = (long long)m.C * N * PARTS * AP * 64;     // one 64-column block of the scratch (see k_sumprod_columns)
  const double* bs = basis + (long long)branch * PARTS * AP * 64;
  double acc_re[HX_SP_PAIRS], acc_im[HX_SP_PAIRS];
#pragma unroll
  for (int q = 0; q < HX_SP_PAIRS; ++q) acc_re[q] = acc_im[q] = 0.;
  const long long n_tiles = (n_cols + HX_SP_TILE - 1) / HX_SP_TILE;
  for (long long tl = blockIdx.y; tl < n_tiles; tl += gridDim.y) {
    const long long c0 = tl * HX_SP_TILE;
    for (int e = threadIdx.x; e < PARTS * A * HX_SP_TILE; e += 256) {
      const int row = e / HX_SP_TILE, c = e - row * HX_SP_TILE;
      const long long cc = c0 + c;
      const int part = row / A, l = row - part * A;      // (rows of a part are stored in pairs: [l / 2][64][2])
      tile[row * TS + c] = cc < n_cols ? bs[(cc >> 6) * blk_doubles + (long long)part * AP * 64 + ((l >> 1) * 64 + (cc & 63)) * 2 + (l & 1)] : 0.;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < HX_SP_PAIRS; ++q) {
      const int pr = q * 256 + threadIdx.x;
      if (q * 256 >= AA) break;
      if (pr < AA) {
        const int k = pr / A, l = pr - k * A;
        const double* u_re = tile + l * TS;
        const double* d_re = tile + (A + k) * TS;
        double sr = acc_re[q], si = acc_im[q];
        if (REAL) {
#pragma unroll 8
          for (int c = 0; c < HX_SP_TILE; ++c) sr += d_re[c] * u_re[c];
        } else {
          const double* u_im = tile + (2 * A + l) * TS;
          const double* d_im = tile + (3 * A + k) * TS;
#pragma unroll 4
          for (int c = 0; c < HX_SP_TILE; ++c) {
            sr += d_re[c] * u_re[c] - d_im[c] * u_im[c];
            si += d_re[c] * u_im[c] + d_im[c] * u_re[c];
          }
        }
        acc_re[q] = sr;
        acc_im[q] = si;
      }
    }
    __syncthreads();
  }
  const double* jr = m.esc_re + (long long)branch * AA;
  const double* ji = m.esc_im + (long long)branch * AA;
#pragma unroll
  for (int q = 0; q < HX_SP_PAIRS; ++q) {
    const int pr = q * 256 + threadIdx.x;
    if (pr < AA) {
      atomicAdd(&eig_re[(long long)cpt * AA + pr], acc_re[q] * jr[pr] - acc_im[q] * ji[pr]);
      atomicAdd(&eig_im[(long long)cpt * AA + pr], acc_re[q] * ji[pr] + acc_im[q] * jr[pr]);
    }
  }
}

// The same sum for real bases on the matrix cores: sum_col D_k(col) U_l(col) is the product D (A x columns) times U^T
// (columns x A).  A workgroup takes one (component, branch) and a slice of the 64-column blocks; a block's 2 A rows of 64
// columns are one contiguous piece of the scratch (see k_sumprod_columns), copied to LDS with rows padded to 65 entries; the
// four waves share the 16 x 16 tiles of the A x A result, sixteen v_mfma_f64_16x16x4_f64 per tile and block (k = the block's
// columns, four at a time).  Operand maps of that instruction: lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15]; result
// register q of lane l is C[(l >> 4) + 4 q][l & 15].  Rows past A are zero in LDS.
typedef double sp_d4 __attribute__((ext_vector_type(4)));
#define HX_SP_MAX_TILES 4          // tiles per wave: A <= 64 -> (A / 16)^2 <= 16 tiles over four waves
template <int PER>                                  // 16-byte entries of a block per thread: >= ceil(A / 2) * 2 * 64 / 256
__global__ void __launch_bounds__(256) k_outer_counts_mfma(const SpModel m, const double* __restrict__ basis, const long long n_cols,
                                                           double* __restrict__ eig_re) {
  extern __shared__ double tile[];                 // [2][Mp][65]: U rows, then D rows
  const int A = m.A, AA = A * A, N = m.N;
  const int tm = (A + 15) >> 4, Mp = tm << 4, n_tiles = tm * tm;
  const int branch = blockIdx.x, cpt = branch / N, r = branch - cpt * N;
  if (m.parent[r] < 0) return;
  const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
  const int AP = (A + 1) & ~1;
  const long long blk_doubles = (long long)m.C * N * 2 * AP * 64;
  const double* bs = basis + (long long)branch * 2 * AP * 64;
  for (int e = threadIdx.x; e < 2 * Mp * 65; e += 256) tile[e] = 0.;
  sp_d4 acc[HX_SP_MAX_TILES];
#pragma unroll
  for (int q = 0; q < HX_SP_MAX_TILES; ++q) acc[q] = sp_d4{0., 0., 0., 0.};
  const long long n_blocks = (n_cols + 63) >> 6;
  // a block's entries travel through registers, two blocks ahead: the blocks after this one are in flight while it is multiplied
  sp_d2 bufA[PER], bufB[PER];
  const int total = AP * 64;                        // a block's U and D rows: AP / 2 row pairs of 64 lanes each, 16 bytes per entry
  const auto fetch = [&](sp_d2 (&buf)[PER], const long long b) {
    if (b >= n_blocks) return;
    const sp_d2* src = reinterpret_cast<const sp_d2*>(bs + b * blk_doubles);
    const long long c0 = b << 6;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int e = q * 256 + (int)threadIdx.x;
      if (q * 256 < total) {
        sp_d2 v = e < total ? src[e] : sp_d2{0., 0.};
        if (c0 + (e & 63) >= n_cols) v = sp_d2{0., 0.};
        buf[q] = v;
      }
    }
  };
  const auto multiply = [&](const sp_d2 (&buf)[PER]) {
    __syncthreads();                               // (the previous block's products are done; the first time: the zero fill)
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int e = q * 256 + (int)threadIdx.x;
      if (q * 256 < total && e < total) {
        // entry e: row pair e / 64 (pairs 0 .. AP/2 - 1: U, then D), column e % 64; its two values are rows 2 p and 2 p + 1
        const int pr = e >> 6, c = e & 63;
        const int part = pr >= AP / 2, l0 = 2 * (pr - part * (AP / 2));
        const int trow = part * Mp + l0;
        tile[trow * 65 + c] = buf[q].x;
        if (l0 + 1 < A) tile[(trow + 1) * 65 + c] = buf[q].y;
      }
    }
    __syncthreads();
  };
  const auto products = [&]() {
#pragma unroll
    for (int q = 0; q < HX_SP_MAX_TILES; ++q) {
      const int t = wave + 4 * q;
      if (t >= n_tiles) break;
      const int tk = t / tm, tl = t - tk * tm;
      const double* drow = tile + (Mp + 16 * tk + (lane & 15)) * 65 + (lane >> 4);
      const double* urow = tile + (16 * tl + (lane & 15)) * 65 + (lane >> 4);
      sp_d4 c = acc[q];
#pragma unroll
      for (int s4 = 0; s4 < 16; ++s4) c = __builtin_amdgcn_mfma_f64_16x16x4f64(drow[4 * s4], urow[4 * s4], c, 0, 0, 0);
      acc[q] = c;
    }
  };
  const long long step = gridDim.y;
  fetch(bufA, blockIdx.y);
  fetch(bufB, blockIdx.y + step);
  for (long long b = blockIdx.y; b < n_blocks; b += 2 * step) {
    multiply(bufA);
    fetch(bufA, b + 2 * step);
    products();
    if (b + step >= n_blocks) break;
    multiply(bufB);
    fetch(bufB, b + 3 * step);
    products();
  }
  const double* jr = m.esc_re + (long long)branch * AA;
#pragma unroll
  for (int q = 0; q < HX_SP_MAX_TILES; ++q) {
    const int t = wave + 4 * q;
    if (t >= n_tiles) break;
    const int tk = t / tm, tl = t - tk * tm;
    const int l = 16 * tl + (lane & 15);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int k = 16 * tk + (lane >> 4) + 4 * g;
      if (k < A && l < A) atomicAdd(&eig_re[(long long)cpt * AA + k * A + l], acc[q][g] * jr[k * A + l]);
    }
  }
}

// out[row] += sum of the row's n_cols entries; one workgroup per row
__global__ void __launch_bounds__(256) k_row_sums(const double* __restrict__ rows, const long long n_cols, double* __restrict__ out) {
  __shared__ double part[256];
  // rows are laid out in blocks of 64 columns: [block][row][64]
  const double* row = rows + (long long)blockIdx.x * 64;
  const long long blk_doubles = (long long)gridDim.x * 64;
  double sum = 0.;
  // grid (rows, slices of the columns): a slice's partial sum is added with one atomic
  for (long long c = (long long)blockIdx.y * 256 + threadIdx.x; c < n_cols; c += (long long)gridDim.y * 256)
    sum += row[(c >> 6) * blk_doubles + (c & 63)];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if ((int)threadIdx.x < h) part[threadIdx.x] += part[threadIdx.x + h];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(&out[blockIdx.x], part[0]);
}

thread_local float g_sp_ms = 0.f;

struct Buf {
  void* p = nullptr;
  ~Buf() { if (p) (void)hipFree(p); }
};

}  // namespace

const double* device_lse_table(int device);    // hx_api.hip: the 8-byte log_sum_exp table of an initialised device, or null
int api_fail(int code, const char* what);       // hx_api.hip: sets hx_last_error()

}  // namespace hx

using namespace hx;

extern "C" {

// See include/historian_hip.h.  One call = upload, the kernels over chunks of columns that fit the scratch budget, download.
int hx_sumprod_columns(const hx_sumprod_model* hm, const int8_t* tokens, const double* weight, int64_t n_cols, double* col_log_like,
                       double* root_counts, double* eigen_re, double* eigen_im, double* root_post, void* stream) {
  if (!hm || !tokens || n_cols <= 0 || !col_log_like || !root_counts || !eigen_re || !eigen_im)
    return api_fail(HX_ERR_INVALID_ARG, "hx_sumprod_columns: null argument or no columns");
  const int A = hm->alph_size, C = hm->components, N = hm->n_nodes, AA = A * A;
  if (A <= 0 || C <= 0 || N <= 0 || !hm->parent || !hm->ins_prob || !hm->log_cpt_weight || !hm->branch_sub || !hm->evec_re || !hm->evec_im ||
      !hm->evec_inv_re || !hm->evec_inv_im || !hm->esc_re || !hm->esc_im)
    return api_fail(HX_ERR_INVALID_ARG, "hx_sumprod_columns: incomplete model");
  if (AA > 256 * HX_SP_PAIRS) return api_fail(HX_ERR_INVALID_ARG, "hx_sumprod_columns: alphabets of more than 64 symbols are not supported");
  if (C > 128) return api_fail(HX_ERR_INVALID_ARG, "hx_sumprod_columns: more than 128 mixture components are not supported");
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return api_fail(HX_ERR_NO_DEVICE, "no HIP device");
  const double* lse_tab = device_lse_table(device);
  if (!lse_tab) return api_fail(HX_ERR_NOT_INITIALIZED, "hx_init has not been called for the current device");
  // children from parents; binary, children before parents
  std::vector<int> child(2 * (size_t)N, -1);
  for (int r = 0; r < N; ++r) {
    const int p = hm->parent[r];
    if (p < 0) continue;
    if (p <= r || p >= N) return api_fail(HX_ERR_NOT_TOPOSORTED, "hx_sumprod_columns: a node precedes its child");
    if (child[2 * p] < 0) child[2 * p] = r;
    else if (child[2 * p + 1] < 0) child[2 * p + 1] = r;
    else return api_fail(HX_ERR_INVALID_ARG, "hx_sumprod_columns: a node has more than two children");
  }
  // tokens index the alphabet on the device: refuse anything else here, not with a fault there
  for (int64_t k = 0; k < n_cols * N; ++k)
    if (tokens[k] < -2 || tokens[k] >= A) return api_fail(HX_ERR_RANGE, "hx_sumprod_columns: a token outside -2 .. alphabet size - 1");
  bool real_basis = true;
  for (size_t k = 0; k < (size_t)C * AA && real_basis; ++k) real_basis = hm->evec_im[k] == 0. && hm->evec_inv_im[k] == 0.;
  for (size_t k = 0; k < (size_t)C * N * AA && real_basis; ++k) real_basis = hm->esc_im[k] == 0.;
  const int parts = real_basis ? 2 : 4;
  hipStream_t st = static_cast<hipStream_t>(stream);
  Buf b_int, b_dbl, b_tok, b_w, b_scr, b_out;
  const size_t dbls = (size_t)C * A + C + (size_t)C * N * AA + 4 * (size_t)C * AA + 2 * (size_t)C * N * AA;
  if (hipMalloc(&b_int.p, 3 * (size_t)N * sizeof(int)) != hipSuccess || hipMalloc(&b_dbl.p, dbls * sizeof(double)) != hipSuccess ||
      hipMalloc(&b_tok.p, (size_t)n_cols * N) != hipSuccess)
    return api_fail(HX_ERR_OUT_OF_MEMORY, "hx_sumprod_columns: device allocation failed");
  int* d_int = static_cast<int*>(b_int.p);
#define UP(dst, src, n) if (hipMemcpy(dst, src, (n), hipMemcpyHostToDevice) != hipSuccess) return api_fail(HX_ERR_HIP, "hx_sumprod_columns: copy failed")
  UP(d_int, hm->parent, N * sizeof(int));
  UP(d_int + N, child.data(), 2 * N * sizeof(int));
  SpModel m;
  m.A = A; m.C = C; m.N = N; m.real_basis = real_basis; m.parent = d_int; m.child = d_int + N;
  double* q = static_cast<double*>(b_dbl.p);
  bool copied = true;
  auto put = [&](const double* src, size_t n) -> const double* {
    double* at = q;
    q += n;
    copied = copied && hipMemcpy(at, src, n * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
    return at;
  };
  m.ins_prob = put(hm->ins_prob, (size_t)C * A);
  m.log_cpt_weight = put(hm->log_cpt_weight, C);
  m.branch_sub = put(hm->branch_sub, (size_t)C * N * AA);
  m.evec_re = put(hm->evec_re, (size_t)C * AA);
  m.evec_im = put(hm->evec_im, (size_t)C * AA);
  m.einv_re = put(hm->evec_inv_re, (size_t)C * AA);
  m.einv_im = put(hm->evec_inv_im, (size_t)C * AA);
  m.esc_re = put(hm->esc_re, (size_t)C * N * AA);
  m.esc_im = put(hm->esc_im, (size_t)C * N * AA);
  if (!copied) return api_fail(HX_ERR_HIP, "hx_sumprod_columns: copy failed");
  UP(b_tok.p, tokens, (size_t)n_cols * N);
  if (weight) {
    if (hipMalloc(&b_w.p, n_cols * sizeof(double)) != hipSuccess) return api_fail(HX_ERR_OUT_OF_MEMORY, "hx_sumprod_columns: device allocation failed");
    UP(b_w.p, weight, n_cols * sizeof(double));
  }
#undef UP
  // columns per chunk: the scratch of a chunk stays within the budget (HX_SUMPROD_SCRATCH_MB, default 16 GiB)
  const int AP = (A + 1) & ~1;                     // message vectors are stored in pairs of entries
  const size_t per_col = (2 + (size_t)parts) * C * N * AP + 4 * (size_t)C * N + 2 * (size_t)C * A;
  size_t budget = (size_t)16 << 30;
  if (const char* e = getenv("HX_SUMPROD_SCRATCH_MB")) budget = (size_t)atoll(e) << 20;
  long long chunk = (long long)(budget / (per_col * sizeof(double)));
  chunk &= ~63LL;                                   // whole blocks of 64 columns
  if (chunk < 64) chunk = 64;
  if (chunk > n_cols) chunk = n_cols;
  const size_t n_out = (size_t)n_cols * (1 + (root_post ? A : 0)) + (size_t)C * A + 2 * (size_t)C * AA;
  if (hipMalloc(&b_scr.p, per_col * (((size_t)chunk + 63) & ~(size_t)63) * sizeof(double)) != hipSuccess || hipMalloc(&b_out.p, n_out * sizeof(double)) != hipSuccess)
    return api_fail(HX_ERR_OUT_OF_MEMORY, "hx_sumprod_columns: device allocation failed");
  double* out = static_cast<double*>(b_out.p);
  double* d_cll = out;
  double* d_post = root_post ? out + n_cols : nullptr;
  double* d_root = out + (size_t)n_cols * (1 + (root_post ? A : 0));
  double* d_re = d_root + (size_t)C * A;
  double* d_im = d_re + (size_t)C * AA;
  if (hipMemsetAsync(d_root, 0, ((size_t)C * A + 2 * (size_t)C * AA) * sizeof(double), st) != hipSuccess)
    return api_fail(HX_ERR_HIP, "hx_sumprod_columns: HIP call failed");
  const size_t lds = sizeof(double) * (size_t)parts * A * (HX_SP_TILE + 1);
  if (lds > HX_LDS_LIMIT) return api_fail(HX_ERR_INVALID_ARG, "hx_sumprod_columns: basis tile exceeds the LDS of a CU");
  struct Events {                                   // destroyed on every way out
    hipEvent_t a = nullptr, b = nullptr;
    ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
  } ev;
  if (hipEventCreate(&ev.a) != hipSuccess || hipEventCreate(&ev.b) != hipSuccess) return api_fail(HX_ERR_HIP, "hx_sumprod_columns: HIP call failed");
  const hipEvent_t e0 = ev.a, e1 = ev.b;
  (void)hipEventRecord(e0, st);
  for (long long first = 0; first < n_cols; first += chunk) {
    const long long nc = n_cols - first < chunk ? n_cols - first : chunk;
    double* scr = static_cast<double*>(b_scr.p);
    SpScratch s;
    const size_t nc64 = ((size_t)nc + 63) & ~(size_t)63;         // the scratch holds whole blocks of 64 columns
    const size_t msg = (size_t)C * N * AP * nc64, lg = (size_t)C * N * nc64;
    s.E = scr; s.G = scr + msg;
    s.logE = scr + 2 * msg; s.logF = s.logE + lg; s.logG = s.logF + lg; s.maxU = s.logG + lg;
    s.Froot = s.maxU + lg;
    s.rootc = s.Froot + (size_t)C * A * nc64;
    s.basis = s.rootc + (size_t)C * A * nc64;
    const int tpb = 64 * (C < 8 ? C : 8);            // a wave per mixture component, up to eight
    long long blocks = (nc + 63) / 64;
    if (blocks > 65535) blocks = 65535;
    const size_t ll_lds = sizeof(double) * 64 * (size_t)C;
    const signed char* d_tok = static_cast<const signed char*>(b_tok.p) + first * N;
    const double* d_w = b_w.p ? static_cast<const double*>(b_w.p) + first : nullptr;
    double* d_p = d_post ? d_post + first * A : nullptr;
    const size_t mat_lds = sizeof(double) * 3 * (size_t)AA * (tpb / 64);
    const bool lm = ll_lds + mat_lds <= 96 * 1024;
    const size_t col_lds = ll_lds + (lm ? mat_lds : 0);
#define HX_SP_GO(TA_, LM_) hipLaunchKernelGGL((k_sumprod_columns<TA_, LM_>), dim3((unsigned)blocks), dim3(tpb), col_lds, st, m, d_tok, d_w, nc, s, lse_tab, d_cll + first, d_p)
    if (A == 4) { if (lm) HX_SP_GO(4, true); else HX_SP_GO(4, false); }
    else if (A == 20) { if (lm) HX_SP_GO(20, true); else HX_SP_GO(20, false); }
    else { if (lm) HX_SP_GO(0, true); else HX_SP_GO(0, false); }
#undef HX_SP_GO
    const long long tiles = (nc + HX_SP_TILE - 1) / HX_SP_TILE;
    long long slices = 4096 / ((long long)C * N) + 1;
    if (slices > tiles) slices = tiles;
    if (real_basis && sizeof(double) * 2 * (((A + 15) / 16) * 16) * 65 <= 64 * 1024 && !getenv("HX_SUMPROD_NO_MFMA")) {
      const int mp = ((A + 15) / 16) * 16;
      const long long blocks64 = (nc + 63) / 64;
      long long sl = 8192 / ((long long)C * N) + 1;          // (measured: 4.46 / 4.35 / 4.23 ms per 100 000 columns at 8 / 34 / 68 slices)
      if (const char* e = getenv("HX_SUMPROD_SLICES")) sl = atoll(e);
      if (sl < 1) sl = 1;
      if (sl > blocks64) sl = blocks64;
      const size_t tile_lds = sizeof(double) * 2 * mp * 65;
      const dim3 og((unsigned)(C * N), (unsigned)sl);
      if (AP <= 4) hipLaunchKernelGGL(k_outer_counts_mfma<1>, og, dim3(256), tile_lds, st, m, s.basis, nc, d_re);
      else if (AP <= 20) hipLaunchKernelGGL(k_outer_counts_mfma<5>, og, dim3(256), tile_lds, st, m, s.basis, nc, d_re);
      else hipLaunchKernelGGL(k_outer_counts_mfma<16>, og, dim3(256), tile_lds, st, m, s.basis, nc, d_re);
    } else if (real_basis)
      hipLaunchKernelGGL(k_outer_counts<true>, dim3((unsigned)(C * N), (unsigned)slices), dim3(256), lds, st, m, s.basis, nc, d_re, d_im);
    else
      hipLaunchKernelGGL(k_outer_counts<false>, dim3((unsigned)(C * N), (unsigned)slices), dim3(256), lds, st, m, s.basis, nc, d_re, d_im);
    {
      long long rs = (nc + 4095) / 4096;                      // at least 4 096 columns per slice
      if (rs > 64) rs = 64;
      hipLaunchKernelGGL(k_row_sums, dim3((unsigned)(C * A), (unsigned)rs), dim3(256), 0, st, s.rootc, nc, d_root);
    }
  }
  (void)hipEventRecord(e1, st);
  int rc = HX_OK;
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = HX_ERR_HIP;
  if (rc == HX_OK) (void)hipEventElapsedTime(&g_sp_ms, e0, e1);
  if (rc != HX_OK) return api_fail(rc, "hx_sumprod_columns: kernel launch or execution failed");
#define DOWN(dst, src, n) if (hipMemcpy(dst, src, (n) * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return api_fail(HX_ERR_HIP, "hx_sumprod_columns: copy failed")
  DOWN(col_log_like, d_cll, (size_t)n_cols);
  if (root_post) DOWN(root_post, d_post, (size_t)n_cols * A);
  DOWN(root_counts, d_root, (size_t)C * A);
  DOWN(eigen_re, d_re, (size_t)C * AA);
  DOWN(eigen_im, d_im, (size_t)C * AA);
#undef DOWN
  return HX_OK;
}

int hx_sumprod_last_kernel_ms(float* ms) {
  if (!ms) return HX_ERR_INVALID_ARG;
  *ms = g_sp_ms;
  return HX_OK;
}

}  // extern "C"
