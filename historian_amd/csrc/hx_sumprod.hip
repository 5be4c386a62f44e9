// Column sum-product on a phylogenetic tree and the eigen-basis substitution-count accumulation of counts mode
// (SURVEY 8f row N3): the reference's SumProduct::initColumn / fillUp / fillDown (src/sumprod.cpp:58-198),
// accumulateRootCounts and accumulateEigenCounts (src/sumprod.cpp:264-271, 294-372), for a BATCH of alignment columns.
//
// In the reference one SumProduct object walks the columns one after the other (AlignColSumProduct, and once per
// posterior-weighted DP cell in BackwardMatrix::getCounts).  Columns and mixture components are independent, so here a
// (column, component) is a thread (k_sumprod_columns): every thread runs the tip-to-root and root-to-tip passes of its column - tiny tree recursions,
// A x A matrix-vector products per branch and mixture component - with its messages E, F, G in a global scratch laid out
// [.][column] so that the threads of a wavefront touch consecutive addresses, then turns the messages of every branch
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

// scratch of one chunk of columns: messages E, G [cpt][node][a][column]; scale factors and max U [cpt][node][column];
// F at the column's root and the root-count terms [cpt][a][column]; bases [cpt][node][Ur, Dr, (Ui, Di)][a][column]
struct SpScratch { double* E; double* G; double* logE; double* logF; double* logG; double* maxU; double* Froot; double* rootc; double* basis; };

// Model matrices are read through the constant address space: their addresses are uniform over the wavefront and
// nothing writes them, so the compiler fetches them with scalar loads and feeds the FMAs from SGPRs.
typedef const __attribute__((address_space(4))) double* CMat;
__device__ __forceinline__ CMat cmat(const double* p) { return (CMat)(unsigned long long)p; }

// TA: the alphabet size as a compile-time constant (message vectors live in registers, loops unrolled), or 0 for any
// alphabet of up to 64 symbols (vectors in private memory).
// LM: a wave keeps the matrices it multiplies with in LDS - exp(R t) of the branch at hand (re-staged per node), the real
// parts of the eigenvectors and of their inverse (per component) - and reads their entries as LDS broadcasts; without it
// they come through the scalar cache, which the 3 KB per matrix-vector product overrun (2.9 TFLOP/s, one FMA per ~100 cycles).
template <int TA, bool LM>
__global__ void __launch_bounds__(512) k_sumprod_columns(const SpModel m, const signed char* __restrict__ tok, const double* __restrict__ weight,
                                                         const long long n_cols, const SpScratch s, const double* __restrict__ lse_tab,
                                                         double* __restrict__ col_log_like, double* __restrict__ root_post) {
  constexpr int AX = TA ? TA : 64;
  const int A = TA ? TA : m.A, C = m.C, N = m.N, AA = A * A;
  const long long stride = n_cols;
  const int parts = m.real_basis ? 2 : 4;
#define AT(P, cpt, r, a) P[(((long long)(cpt) * N + (r)) * A + (a)) * stride + col]
#define LG(P, cpt, r) P[((long long)(cpt) * N + (r)) * stride + col]
#define BS(cpt, r, part, l) s.basis[((((long long)(cpt) * N + (r)) * parts + (part)) * A + (l)) * stride + col]
#define FR(cpt, a) s.Froot[((long long)(cpt) * A + (a)) * stride + col]
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
        for (int k = lane; k < AA; k += 64) { Linv[k] = m.einv_re[(long long)cpt * AA + k]; Lvec[k] = m.evec_re[(long long)cpt * AA + k]; }
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
#pragma unroll
          for (int a = 0; a < A; ++a) AT(s.E, cpt, r, a) = 1.;
          LG(s.logE, cpt, r) = 0.;
          LG(s.logF, cpt, r) = lf;
#pragma unroll
          for (int l = 0; l < A; ++l) BS(cpt, r, 0, l) = 0.;
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
#pragma unroll
            for (int a = 0; a < A; ++a) AT(s.E, cpt, r, a) = (LM ? Lsub[a * A + tk] : subg[a * A + tk]) * f;
#pragma unroll
            for (int l = 0; l < A; ++l) BS(cpt, r, 0, l) = (LM ? Linv[l * A + tk] : m.einv_re[(long long)cpt * AA + l * A + tk]) * (f / f);
            if (parts == 4)
              for (int l = 0; l < A; ++l) BS(cpt, r, 2, l) = m.einv_im[(long long)cpt * AA + l * A + tk] * (f / f);
          }
          continue;
        }
        // a wildcard: the full vector
        double f[AX];
        double fmax = 0.;
#pragma unroll
        for (int a = 0; a < A; ++a) {
          f[a] = (c0 >= 0 ? AT(s.E, cpt, c0, a) : 1.) * (c1 >= 0 ? AT(s.E, cpt, c1, a) : 1.);
          fmax = f[a] > fmax ? f[a] : fmax;
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
        for (int k = lane; k < AA; k += 64) Lvec[k] = m.evec_re[(long long)cpt * AA + k];
      }
      for (int r = N - 1; r >= 0; --r) {
        if (LM && m.parent[r] >= 0) {
          wave_sync();
          for (int k = lane; k < AA; k += 64) Lsub[k] = m.branch_sub[((long long)cpt * N + r) * AA + k];
          wave_sync();
        }
        if (!busy) continue;
        if (t[r] == -2 || r == root) {
          if (r == root) {
#pragma unroll
            for (int a = 0; a < A; ++a) AT(s.G, cpt, r, a) = ins[a];
            LG(s.logG, cpt, r) = 0.;
            for (int l = 0; l < A; ++l) BS(cpt, r, 0, l) = 0.;        // no branch above the root: U was not written on the way up
            if (parts == 4)
              for (int l = 0; l < A; ++l) BS(cpt, r, 2, l) = 0.;
          }
#pragma unroll
          for (int l = 0; l < A; ++l) BS(cpt, r, 1, l) = 0.;
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
#pragma unroll
        for (int a = 0; a < A; ++a) {
          d[a] = AT(s.G, cpt, p, a) * (sib >= 0 ? AT(s.E, cpt, sib, a) : 1.);
          max_d = d[a] > max_d ? d[a] : max_d;
        }
#pragma unroll
        for (int b = 0; b < A; ++b) {
          double g = 0.;
#pragma unroll
          for (int a = 0; a < A; ++a) g += d[a] * sub(a * A + b);
          AT(s.G, cpt, r, b) = g;
        }
        const double norm = exp(cll - m.log_cpt_weight[cpt] - LG(s.logF, cpt, r) - lg_p - le_sib) / (LG(s.maxU, cpt, r) * max_d);
        const double scale = w / norm;
#pragma unroll
        for (int a = 0; a < A; ++a) d[a] /= max_d;
#pragma unroll
        for (int k = 0; k < A; ++k) {
          double dr = 0.;
#pragma unroll
          for (int a = 0; a < A; ++a) dr += vr(a * A + k) * d[a];
          BS(cpt, r, 1, k) = dr * scale;
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
        s.rootc[((long long)cpt * A + a) * stride + col] = root >= 0 ? w * m.ins_prob[cpt * A + a] * FR(cpt, a) * norm : 0.;
    }
    __syncthreads();
  }
#undef AT
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
  const double* bs = basis + (long long)branch * PARTS * A * n_cols;
  double acc_re[HX_SP_PAIRS], acc_im[HX_SP_PAIRS];
#pragma unroll
  for (int q = 0; q < HX_SP_PAIRS; ++q) acc_re[q] = acc_im[q] = 0.;
  const long long n_tiles = (n_cols + HX_SP_TILE - 1) / HX_SP_TILE;
  for (long long tl = blockIdx.y; tl < n_tiles; tl += gridDim.y) {
    const long long c0 = tl * HX_SP_TILE;
    for (int e = threadIdx.x; e < PARTS * A * HX_SP_TILE; e += 256) {
      const int row = e / HX_SP_TILE, c = e - row * HX_SP_TILE;
      tile[row * TS + c] = c0 + c < n_cols ? bs[(long long)row * n_cols + c0 + c] : 0.;
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

// out[row] += sum of the row's n_cols entries; one workgroup per row
__global__ void __launch_bounds__(256) k_row_sums(const double* __restrict__ rows, const long long n_cols, double* __restrict__ out) {
  __shared__ double part[256];
  const double* row = rows + (long long)blockIdx.x * n_cols;
  double sum = 0.;
  for (long long c = threadIdx.x; c < n_cols; c += 256) sum += row[c];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if ((int)threadIdx.x < h) part[threadIdx.x] += part[threadIdx.x + h];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] += part[0];
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
  const size_t per_col = (2 + (size_t)parts) * C * N * A + 4 * (size_t)C * N + 2 * (size_t)C * A;
  size_t budget = (size_t)16 << 30;
  if (const char* e = getenv("HX_SUMPROD_SCRATCH_MB")) budget = (size_t)atoll(e) << 20;
  long long chunk = (long long)(budget / (per_col * sizeof(double)));
  if (chunk < 1) chunk = 1;
  if (chunk > n_cols) chunk = n_cols;
  const size_t n_out = (size_t)n_cols * (1 + (root_post ? A : 0)) + (size_t)C * A + 2 * (size_t)C * AA;
  if (hipMalloc(&b_scr.p, per_col * chunk * sizeof(double)) != hipSuccess || hipMalloc(&b_out.p, n_out * sizeof(double)) != hipSuccess)
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
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return api_fail(HX_ERR_HIP, "hx_sumprod_columns: HIP call failed");
  (void)hipEventRecord(e0, st);
  for (long long first = 0; first < n_cols; first += chunk) {
    const long long nc = n_cols - first < chunk ? n_cols - first : chunk;
    double* scr = static_cast<double*>(b_scr.p);
    SpScratch s;
    const size_t msg = (size_t)C * N * A * nc, lg = (size_t)C * N * nc;
    s.E = scr; s.G = scr + msg;
    s.logE = scr + 2 * msg; s.logF = s.logE + lg; s.logG = s.logF + lg; s.maxU = s.logG + lg;
    s.Froot = s.maxU + lg;
    s.rootc = s.Froot + (size_t)C * A * nc;
    s.basis = s.rootc + (size_t)C * A * nc;
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
    if (real_basis)
      hipLaunchKernelGGL(k_outer_counts<true>, dim3((unsigned)(C * N), (unsigned)slices), dim3(256), lds, st, m, s.basis, nc, d_re, d_im);
    else
      hipLaunchKernelGGL(k_outer_counts<false>, dim3((unsigned)(C * N), (unsigned)slices), dim3(256), lds, st, m, s.basis, nc, d_re, d_im);
    hipLaunchKernelGGL(k_row_sums, dim3((unsigned)(C * A)), dim3(256), 0, st, s.rootc, nc, d_root);
  }
  (void)hipEventRecord(e1, st);
  int rc = HX_OK;
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = HX_ERR_HIP;
  if (rc == HX_OK) (void)hipEventElapsedTime(&g_sp_ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
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
