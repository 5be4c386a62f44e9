// Column sum-product on a phylogenetic tree and the eigen-basis substitution-count accumulation of counts mode
// (SURVEY 8f row N3): the reference's SumProduct::initColumn / fillUp / fillDown (src/sumprod.cpp:58-198),
// accumulateRootCounts and accumulateEigenCounts (src/sumprod.cpp:264-271, 294-372), for a BATCH of alignment columns.
//
// In the reference one SumProduct object walks the columns one after the other (AlignColSumProduct, and once per
// posterior-weighted DP cell in BackwardMatrix::getCounts).  Columns are independent, so here a column is a thread:
// every thread runs the tip-to-root and root-to-tip passes of its column - tiny tree recursions, A x A matrix-vector
// products per branch and mixture component - with its messages E, F, G in a global scratch laid out [.][column] so that
// the threads of a wavefront touch consecutive addresses, then turns the branch messages into the eigen basis and adds
// weight x D_k J_kl U_l to the count matrices.  The count matrices are accumulated per workgroup in LDS (fp64 LDS atomics)
// and added to the global result once per workgroup.
//
// Arithmetic follows the reference: linear-space messages with per-node log scale factors, rescaling below 1e-30, the
// column likelihood combined over components with the table log_sum_exp.  Sums over columns are atomic, so their order
// differs from the reference's: results agree to rounding (tests/test_gpu_sumprod.py: 1e-10 relative), not bit for bit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../include/historian_hip.h"
#include "hx_lse.h"
#include "hx_kernels.h"

namespace hx {

namespace {

#define HX_SP_RESCALE 1e-30        // SUMPROD_RESCALE_THRESHOLD (src/sumprod.cpp:8)

struct SpModel {
  int A, C, N;
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

// scratch of one batch: [kind][cpt][node][a][column]
struct SpScratch { double* E; double* F; double* G; double* logE; double* logF; double* logG; double* basis; };

__global__ void __launch_bounds__(128) k_sumprod_columns(const SpModel m, const signed char* __restrict__ tok, const double* __restrict__ weight,
                                                         const long long n_cols, const SpScratch s, const double* __restrict__ lse_tab,
                                                         double* __restrict__ col_log_like, double* __restrict__ root_post,
                                                         double* __restrict__ root_counts, double* __restrict__ eig_re, double* __restrict__ eig_im) {
  extern __shared__ double acc[];   // [C][A] root counts, [C][A][A] re, [C][A][A] im
  const int A = m.A, C = m.C, N = m.N, AA = A * A;
  double* acc_root = acc;
  double* acc_re = acc + C * A;
  double* acc_im = acc_re + C * AA;
  for (int k = threadIdx.x; k < C * A + 2 * C * AA; k += blockDim.x) acc[k] = 0.;
  __syncthreads();
  const long long stride = n_cols;
#define AT(P, cpt, r, a) P[(((long long)(cpt) * N + (r)) * A + (a)) * stride + col]
#define LG(P, cpt, r) P[((long long)(cpt) * N + (r)) * stride + col]
  for (long long col = (long long)blockIdx.x * blockDim.x + threadIdx.x; col < n_cols; col += (long long)gridDim.x * blockDim.x) {
    const signed char* t = tok + col * N;       // -2 gap, -1 wildcard, else the residue's token
    const double w = weight ? weight[col] : 1.;
    int root = -1;
    for (int r = 0; r < N; ++r)
      if (t[r] != -2 && (m.parent[r] < 0 || t[m.parent[r]] == -2)) root = r;     // (one root per column: the caller's contract)
    double cll = HX_NEG_INF;
    // ---- tip-to-root (src/sumprod.cpp:99-161) ----
    for (int cpt = 0; cpt < C; ++cpt) {
      double cpt_ll = 0.;
      for (int r = 0; r < N; ++r) {
        const int c0 = m.child[2 * r], c1 = m.child[2 * r + 1];
        double lf = (c0 >= 0 ? LG(s.logE, cpt, c0) : 0.) + (c1 >= 0 ? LG(s.logE, cpt, c1) : 0.);
        if (t[r] == -2) {
          // a gap: its message to the parent is all ones
          for (int a = 0; a < A; ++a) AT(s.E, cpt, r, a) = 1.;
          LG(s.logE, cpt, r) = 0.;
          LG(s.logF, cpt, r) = lf;
          continue;
        }
        if (t[r] == -1) {
          double fmax = 0.;
          for (int a = 0; a < A; ++a) {
            const double f = (c0 >= 0 ? AT(s.E, cpt, c0, a) : 1.) * (c1 >= 0 ? AT(s.E, cpt, c1, a) : 1.);
            AT(s.F, cpt, r, a) = f;
            fmax = f > fmax ? f : fmax;
          }
          if (fmax < HX_SP_RESCALE) {
            for (int a = 0; a < A; ++a) AT(s.F, cpt, r, a) /= fmax;
            lf += log(fmax);
          }
        } else {
          const int tk = t[r];
          double f = (c0 >= 0 ? AT(s.E, cpt, c0, tk) : 1.) * (c1 >= 0 ? AT(s.E, cpt, c1, tk) : 1.);
          if (f < HX_SP_RESCALE) { lf += log(f); f = 1.; }
          for (int a = 0; a < A; ++a) AT(s.F, cpt, r, a) = a == tk ? f : 0.;
        }
        LG(s.logF, cpt, r) = lf;
        if (r == root) {
          double ip = 0.;
          for (int a = 0; a < A; ++a) ip += AT(s.F, cpt, r, a) * m.ins_prob[cpt * A + a];
          cpt_ll += lf + log(ip);
        } else {
          LG(s.logE, cpt, r) = lf;
          const double* sub = m.branch_sub + ((long long)cpt * N + r) * AA;
          for (int a = 0; a < A; ++a) {
            double e = 0.;
            for (int b = 0; b < A; ++b) e += sub[a * A + b] * AT(s.F, cpt, r, b);
            AT(s.E, cpt, r, a) = e;
          }
        }
      }
      cll = lse(cll, m.log_cpt_weight[cpt] + cpt_ll, lse_tab);
    }
    col_log_like[col] = cll;
    // ---- root-to-tip (src/sumprod.cpp:163-198) ----
    for (int cpt = 0; cpt < C; ++cpt)
      for (int r = N - 1; r >= 0; --r) {
        if (t[r] == -2) continue;
        if (r == root) {
          for (int a = 0; a < A; ++a) AT(s.G, cpt, r, a) = m.ins_prob[cpt * A + a];
          LG(s.logG, cpt, r) = 0.;
          continue;
        }
        const int p = m.parent[r];
        const int sib = m.child[2 * p] == r ? m.child[2 * p + 1] : m.child[2 * p];
        LG(s.logG, cpt, r) = LG(s.logG, cpt, p) + (sib >= 0 ? LG(s.logE, cpt, sib) : 0.);
        const double* sub = m.branch_sub + ((long long)cpt * N + r) * AA;
        const bool sib_in = sib >= 0 && t[sib] != -2;
        for (int b = 0; b < A; ++b) {
          double g = 0.;
          for (int a = 0; a < A; ++a) {
            double pw = AT(s.G, cpt, p, a) * sub[a * A + b];
            if (sib_in) pw *= AT(s.E, cpt, sib, a);
            g += pw;
          }
          AT(s.G, cpt, r, b) = g;
        }
      }
    if (root < 0) continue;
    // ---- posterior of the root's residue (src/sumprod.cpp:208-217) ----
    if (root_post)
      for (int a = 0; a < A; ++a) {
        double lp = HX_NEG_INF;
        for (int cpt = 0; cpt < C; ++cpt)
          lp = lse(lp, m.log_cpt_weight[cpt] + LG(s.logF, cpt, root) + log(AT(s.F, cpt, root, a)) + LG(s.logG, cpt, root) +
                           log(AT(s.G, cpt, root, a)) - cll, lse_tab);
        root_post[col * A + a] = lp < 0. ? lp : 0.;
      }
    // ---- root counts (src/sumprod.cpp:264-271) ----
    for (int cpt = 0; cpt < C; ++cpt) {
      const double norm = exp(m.log_cpt_weight[cpt] + LG(s.logF, cpt, root) - cll);
      for (int a = 0; a < A; ++a) atomicAdd(&acc_root[cpt * A + a], w * m.ins_prob[cpt * A + a] * AT(s.F, cpt, root, a) * norm);
    }
    // ---- eigen-basis substitution counts of every branch above an ungapped node (src/sumprod.cpp:294-372) ----
    for (int r = 0; r < N; ++r) {
      if (t[r] == -2 || r == root) continue;
      const int p = m.parent[r];
      const int sib = m.child[2 * p] == r ? m.child[2 * p + 1] : m.child[2 * p];
      for (int cpt = 0; cpt < C; ++cpt) {
        double max_u = 0., max_d = 0.;
        for (int a = 0; a < A; ++a) {
          const double u = AT(s.F, cpt, r, a), d = AT(s.G, cpt, p, a) * (sib >= 0 ? AT(s.E, cpt, sib, a) : 1.);
          max_u = u > max_u ? u : max_u;
          max_d = d > max_d ? d : max_d;
        }
        const double norm = exp(cll - m.log_cpt_weight[cpt] - LG(s.logF, cpt, r) - LG(s.logG, cpt, p) - (sib >= 0 ? LG(s.logE, cpt, sib) : 0.)) /
                            (max_u * max_d);
        const double scale = w / norm;
        const double* vr = m.evec_re + (long long)cpt * AA;
        const double* vi = m.evec_im + (long long)cpt * AA;
        const double* ir = m.einv_re + (long long)cpt * AA;
        const double* ii = m.einv_im + (long long)cpt * AA;
        // Ubasis[l] = sum_b evecInv[l][b] U[b];  Dbasis[k] = sum_a D[a] evec[a][k]   (scratch: [4][A] per column)
        for (int l = 0; l < A; ++l) {
          double ur = 0., ui = 0., dr = 0., di = 0.;
          for (int b = 0; b < A; ++b) {
            const double u = AT(s.F, cpt, r, b) / max_u;
            ur += ir[l * A + b] * u; ui += ii[l * A + b] * u;
            const double d = AT(s.G, cpt, p, b) * (sib >= 0 ? AT(s.E, cpt, sib, b) : 1.) / max_d;
            dr += vr[b * A + l] * d; di += vi[b * A + l] * d;
          }
          s.basis[(0 * A + l) * stride + col] = ur; s.basis[(1 * A + l) * stride + col] = ui;
          s.basis[(2 * A + l) * stride + col] = dr; s.basis[(3 * A + l) * stride + col] = di;
        }
        const double* jr = m.esc_re + ((long long)cpt * N + r) * AA;
        const double* ji = m.esc_im + ((long long)cpt * N + r) * AA;
        for (int k = 0; k < A; ++k) {
          const double dr = s.basis[(2 * A + k) * stride + col], di = s.basis[(3 * A + k) * stride + col];
          for (int l = 0; l < A; ++l) {
            const double ur = s.basis[(0 * A + l) * stride + col], ui = s.basis[(1 * A + l) * stride + col];
            // D_k * (J_kl * U_l)
            const double xr = jr[k * A + l] * ur - ji[k * A + l] * ui, xi = jr[k * A + l] * ui + ji[k * A + l] * ur;
            atomicAdd(&acc_re[(cpt * A + k) * A + l], (dr * xr - di * xi) * scale);
            atomicAdd(&acc_im[(cpt * A + k) * A + l], (dr * xi + di * xr) * scale);
          }
        }
      }
    }
  }
#undef AT
#undef LG
  __syncthreads();
  for (int k = threadIdx.x; k < C * A; k += blockDim.x) atomicAdd(&root_counts[k], acc_root[k]);
  for (int k = threadIdx.x; k < C * AA; k += blockDim.x) {
    atomicAdd(&eig_re[k], acc_re[k]);
    atomicAdd(&eig_im[k], acc_im[k]);
  }
}

thread_local float g_sp_ms = 0.f;

}  // namespace

const double* device_lse_table(int device);    // hx_api.hip: the 8-byte log_sum_exp table of an initialised device, or null

}  // namespace hx

using namespace hx;

extern "C" {

// See include/historian_hip.h.  One call = upload, one launch over all columns, download.
int hx_sumprod_columns(const hx_sumprod_model* hm, const int8_t* tokens, const double* weight, int64_t n_cols, double* col_log_like,
                       double* root_counts, double* eigen_re, double* eigen_im, double* root_post, void* stream) {
  if (!hm || !tokens || n_cols <= 0 || !col_log_like || !root_counts || !eigen_re || !eigen_im) return HX_ERR_INVALID_ARG;
  const int A = hm->alph_size, C = hm->components, N = hm->n_nodes, AA = A * A;
  if (A <= 0 || C <= 0 || N <= 0 || !hm->parent || !hm->ins_prob || !hm->log_cpt_weight || !hm->branch_sub || !hm->evec_re || !hm->evec_im ||
      !hm->evec_inv_re || !hm->evec_inv_im || !hm->esc_re || !hm->esc_im)
    return HX_ERR_INVALID_ARG;
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return HX_ERR_NO_DEVICE;
  const double* lse_tab = device_lse_table(device);
  if (!lse_tab) return HX_ERR_NOT_INITIALIZED;
  const size_t lds = sizeof(double) * ((size_t)C * A + 2 * (size_t)C * AA);
  if (lds > HX_LDS_LIMIT) return HX_ERR_INVALID_ARG;
  // children from parents; binary, post-order
  std::vector<int> child(2 * (size_t)N, -1);
  for (int r = 0; r < N; ++r) {
    const int p = hm->parent[r];
    if (p < 0) continue;
    if (p <= r || p >= N) return HX_ERR_NOT_TOPOSORTED;
    if (child[2 * p] < 0) child[2 * p] = r;
    else if (child[2 * p + 1] < 0) child[2 * p + 1] = r;
    else return HX_ERR_INVALID_ARG;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  // one arena for the model, one for the scratch
  const size_t n_model = (size_t)N + 2 * N + 0;
  (void)n_model;
  struct Buf { void* p = nullptr; ~Buf() { if (p) (void)hipFree(p); } };
  Buf b_int, b_dbl, b_tok, b_w, b_scr, b_out;
  const size_t ints = 3 * (size_t)N;
  const size_t dbls = (size_t)C * A + C + (size_t)C * N * AA + 4 * (size_t)C * AA + 2 * (size_t)C * N * AA;
  if (hipMalloc(&b_int.p, ints * sizeof(int)) != hipSuccess || hipMalloc(&b_dbl.p, dbls * sizeof(double)) != hipSuccess ||
      hipMalloc(&b_tok.p, (size_t)n_cols * N) != hipSuccess)
    return HX_ERR_OUT_OF_MEMORY;
  int* d_int = static_cast<int*>(b_int.p);
  double* d_dbl = static_cast<double*>(b_dbl.p);
#define UP(dst, src, n) if (hipMemcpy(dst, src, (n), hipMemcpyHostToDevice) != hipSuccess) return HX_ERR_HIP
  UP(d_int, hm->parent, N * sizeof(int));
  UP(d_int + N, child.data(), 2 * N * sizeof(int));
  SpModel m;
  m.A = A; m.C = C; m.N = N; m.parent = d_int; m.child = d_int + N;
  double* q = d_dbl;
  auto put = [&](const double* src, size_t n) -> const double* { double* at = q; q += n; return hipMemcpy(at, src, n * sizeof(double), hipMemcpyHostToDevice) == hipSuccess ? at : nullptr; };
  m.ins_prob = put(hm->ins_prob, (size_t)C * A);
  m.log_cpt_weight = put(hm->log_cpt_weight, C);
  m.branch_sub = put(hm->branch_sub, (size_t)C * N * AA);
  m.evec_re = put(hm->evec_re, (size_t)C * AA); m.evec_im = put(hm->evec_im, (size_t)C * AA);
  m.einv_re = put(hm->evec_inv_re, (size_t)C * AA); m.einv_im = put(hm->evec_inv_im, (size_t)C * AA);
  m.esc_re = put(hm->esc_re, (size_t)C * N * AA); m.esc_im = put(hm->esc_im, (size_t)C * N * AA);
  if (!m.ins_prob || !m.log_cpt_weight || !m.branch_sub || !m.evec_re || !m.evec_im || !m.einv_re || !m.einv_im || !m.esc_re || !m.esc_im) return HX_ERR_HIP;
  UP(b_tok.p, tokens, (size_t)n_cols * N);
  if (weight) {
    if (hipMalloc(&b_w.p, n_cols * sizeof(double)) != hipSuccess) return HX_ERR_OUT_OF_MEMORY;
    UP(b_w.p, weight, n_cols * sizeof(double));
  }
#undef UP
  const size_t per_col = 3 * (size_t)C * N * A + 3 * (size_t)C * N + 4 * (size_t)A;
  const size_t n_out = (size_t)n_cols * (1 + (root_post ? A : 0)) + (size_t)C * A + 2 * (size_t)C * AA;
  if (hipMalloc(&b_scr.p, per_col * n_cols * sizeof(double)) != hipSuccess || hipMalloc(&b_out.p, n_out * sizeof(double)) != hipSuccess)
    return HX_ERR_OUT_OF_MEMORY;
  double* scr = static_cast<double*>(b_scr.p);
  SpScratch s;
  const size_t msg = (size_t)C * N * A * n_cols, lg = (size_t)C * N * n_cols;
  s.E = scr; s.F = scr + msg; s.G = scr + 2 * msg; s.logE = scr + 3 * msg; s.logF = s.logE + lg; s.logG = s.logF + lg; s.basis = s.logG + lg;
  double* out = static_cast<double*>(b_out.p);
  double* d_cll = out;
  double* d_post = root_post ? out + n_cols : nullptr;
  double* d_root = out + (size_t)n_cols * (1 + (root_post ? A : 0));
  double* d_re = d_root + (size_t)C * A;
  double* d_im = d_re + (size_t)C * AA;
  if (hipMemsetAsync(d_root, 0, ((size_t)C * A + 2 * (size_t)C * AA) * sizeof(double), st) != hipSuccess) return HX_ERR_HIP;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return HX_ERR_HIP;
  const int tpb = 128;
  long long blocks = (n_cols + tpb - 1) / tpb;
  if (blocks > 4096) blocks = 4096;
  (void)hipEventRecord(e0, st);
  hipLaunchKernelGGL(k_sumprod_columns, dim3((unsigned)blocks), dim3(tpb), lds, st, m, static_cast<const signed char*>(b_tok.p),
                     static_cast<const double*>(b_w.p), (long long)n_cols, s, lse_tab, d_cll, d_post, d_root, d_re, d_im);
  (void)hipEventRecord(e1, st);
  int rc = HX_OK;
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = HX_ERR_HIP;
  if (rc == HX_OK) (void)hipEventElapsedTime(&g_sp_ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (rc != HX_OK) return rc;
#define DOWN(dst, src, n) if (hipMemcpy(dst, src, (n) * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return HX_ERR_HIP
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
