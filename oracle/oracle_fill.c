/*
 * oracle_fill.c -- plain-C CPU restatement of the reference's Forward / Backward
 * fills, for parity checks at sizes the pure-Python oracle cannot reach and as the
 * `cpu_baseline` ("port") leg of bench.py.
 *
 * TEST INFRASTRUCTURE ONLY: nothing under historian_amd/ may link or call this.
 * Parity status: pinned transitively -- tests/test_oracle_c.py checks every cell of
 * this file bit-for-bit against oracle/historian_oracle.py, which reproduces the
 * reference's golden files byte-for-byte (tests/test_oracle_golden.py).
 *
 * Follows, line by line:
 *   log_sum_exp            reference src/logsumexp.h:42-84
 *   leftMultiply           reference src/profile.cpp:78-91
 *   insx/rootsubx          reference src/forward.cpp:44-56
 *   edges                  reference src/forward.cpp:58-65
 *   Forward fill + lpEnd   reference src/forward.cpp:68-223
 *   Backward fill          reference src/forward.cpp:975-1088
 * Input is the POD job image of include/historian_hip.h (the interface definition
 * only; no product code is used).  Output cells are dense row-major AoS:
 * cells[(i*n_cols + j)*5 + state], -inf outside the envelope.
 *
 * Build with -ffp-contract=off and no -march flags (the reference is built for
 * baseline x86-64: every multiply and add rounds separately).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/historian_hip.h"

#define NEG_INF (-INFINITY)

static const double* g_tab;

void orc_set_table(const double* tab) { g_tab = tab; }

/* reference src/logsumexp.cpp:8-16 (+1 guard entry, see historian_oracle.py) */
void orc_build_table(double* tab) {
  int n;
  for (n = 0; n < HX_LSE_TABLE_ENTRIES; ++n) tab[n] = log(1. + exp(-(n * .0001)));
}

static inline double lse_unary(double x) {
  if (x >= 10 || isnan(x) || isinf(x)) return 0;
  if (x < 0) return -x;
  {
    const int n = (int)(x / .0001);
    const double f0 = g_tab[n];
    const double dx = x - (n * .0001);
    const double f1 = g_tab[n + 1];
    const double df = f1 - f0;
    return f0 + df * (dx / .0001);
  }
}

static inline double lse(double a, double b) {
  double max, diff;
  if (a == b) { max = a; diff = 0; }
  else if (a < b) { max = b; diff = b - a; }
  else { max = a; diff = a - b; }
  return max + lse_unary(diff);
}

double orc_log_sum_exp(double a, double b) { return lse(a, b); }

/* Test switch for the Forward recursion only (not the profile preparation, emission terms or lpEnd, which keep
 * the reference's operator): log(exp(a) + exp(b)) evaluated with libm and without the reference's table or its
 * truncation of differences >= 10.  The scaled-linear HIP kernel (hx_linear.hip) computes the recursion on
 * probabilities, i.e. in this arithmetic; the tests bound its distance to this as well as to the reference's.
 * on == 2: the same with the reference's truncation kept (the term log1p(exp(-diff)) is 0 once diff >= 10,
 * src/logsumexp.h:45): the arithmetic of the truncating scaled-probability policy (HX_LSE_TRUNC), which differs from
 * the reference's own only by the interpolation error of the reference's table (< 3e-10 per operation). */
static int g_true_math;
void orc_set_true_math(int on) { g_true_math = on; }
static inline double cell_lse(double a, double b) {
  double max, diff;
  if (!g_true_math) return lse(a, b);
  if (a == b) { max = a; diff = 0; }
  else if (a < b) { max = b; diff = b - a; }
  else { max = a; diff = a - b; }
  if (isnan(diff) || isinf(diff)) return max;
  if (g_true_math == 2 && diff >= 10.0) return max;
  return max + log1p(exp(-diff));
}

typedef struct {
  int n;
  int empty;
  const hx_profile* p;
  double* sub;      /* [n][CA] */
  double* ins;      /* [n] */
  double* rootsub;  /* [n] */
  unsigned char* edge, *ready, *emit_or_start;
} side_t;

static int side_init(side_t* s, const hx_profile* p, const hx_hmm* h, int is_y) {
  const int N = p->n_states, A = h->alph_size, C = h->components, CA = A * C;
  const double* logsub = is_y ? h->log_sub_r : h->log_sub_l;
  const double* logins = is_y ? h->log_ins_r : h->log_ins_l;
  const double* logcw = is_y ? h->log_cptw_r : h->log_cptw_l;
  int i, cpt, c, d, k;
  s->n = N;
  s->p = p;
  s->sub = (double*)malloc(sizeof(double) * (size_t)N * CA);
  s->ins = (double*)malloc(sizeof(double) * N);
  s->rootsub = (double*)malloc(sizeof(double) * N);
  s->edge = (unsigned char*)calloc(N, 1);
  s->ready = (unsigned char*)calloc(N, 1);
  s->emit_or_start = (unsigned char*)calloc(N, 1);
  if (!s->sub || !s->ins || !s->rootsub || !s->edge || !s->ready || !s->emit_or_start) return -1;
  s->empty = 1;
  for (i = 0; i < N; ++i) {
    if (!p->is_null[i]) s->empty = 0;
    s->ready[i] = p->nout_off[i + 1] == p->nout_off[i];
    s->emit_or_start[i] = !p->is_null[i] || p->in_off[i + 1] == p->in_off[i];
  }
  /* leftMultiply */
  for (i = 0; i < N; ++i)
    for (cpt = 0; cpt < C; ++cpt)
      for (c = 0; c < A; ++c) {
        double lp = NEG_INF;
        if (!p->is_null[i])
          for (d = 0; d < A; ++d)
            lp = lse(lp, logsub[(cpt * A + c) * A + d] + p->lp_absorb[(size_t)i * CA + cpt * A + d]);
        s->sub[(size_t)i * CA + cpt * A + c] = lp;
      }
  for (i = 0; i < N; ++i) {
    s->ins[i] = NEG_INF;
    s->rootsub[i] = NEG_INF;
    if (i >= 1 && i < N - 1 && !p->is_null[i])
      for (cpt = 0; cpt < C; ++cpt) {
        double lip = NEG_INF;
        for (c = 0; c < A; ++c) lip = lse(lip, logins[cpt * A + c] + p->lp_absorb[(size_t)i * CA + cpt * A + c]);
        s->ins[i] = lse(s->ins[i], logcw[cpt] + lip);
        lip = NEG_INF;
        for (c = 0; c < A; ++c) lip = lse(lip, h->log_root[cpt * A + c] + s->sub[(size_t)i * CA + cpt * A + c]);
        s->rootsub[i] = lse(s->rootsub[i], lip);
      }
  }
  if (!is_y) {
    s->edge[0] = 1;
    for (i = 0; i < N; ++i)
      if (s->edge[i])
        for (k = p->nout_off[i]; k < p->nout_off[i + 1]; ++k) s->edge[p->trans_dst[p->nout_idx[k]]] = 1;
  } else {
    for (k = p->in_off[N - 1]; k < p->in_off[N]; ++k) s->edge[p->trans_src[p->in_idx[k]]] = 1;
  }
  return 0;
}

static void side_free(side_t* s) {
  free(s->sub); free(s->ins); free(s->rootsub); free(s->edge); free(s->ready); free(s->emit_or_start);
}

static inline int in_envelope(const side_t* X, const side_t* Y, int max_dist, int i, int j) {
  int d;
  if (X->edge[i] || Y->edge[j]) return 1;
  if (max_dist < 0) return 1;
  d = X->p->env_pos[i] - Y->p->env_pos[j];
  return abs(d) <= max_dist;
}

static double emission(const hx_hmm* h, const side_t* X, const side_t* Y, int i, int j) {
  const int A = h->alph_size, C = h->components, CA = A * C;
  double lip = NEG_INF;
  int cpt, a;
  for (cpt = 0; cpt < C; ++cpt) {
    double inner = NEG_INF;
    for (a = 0; a < A; ++a) {
      const int k = cpt * A + a;
      inner = lse(inner, h->log_root[k] + (X->sub[(size_t)i * CA + k] + Y->sub[(size_t)j * CA + k]));
    }
    lip = lse(lip, inner);
  }
  return lip;
}

/* Cell storage: a dense array, or - ORC_MAP_STORAGE, see oracle_fill_map.cpp - the reference's own structure, a
 * std::map per row (src/forward.h:22,68), for the CPU baseline that has the reference's cost structure. */
#ifdef ORC_MAP_STORAGE
#define CELL(i, j) orc_map_cell((i), (j))
#else
#define CELL(i, j) (cells + ((size_t)(i) * Cc + (j)) * 5)
#endif
#define TSRC(P, k) ((P)->trans_src[(P)->in_idx[k]])
#define TLP(P, k) ((P)->trans_lp[(P)->in_idx[k]])

/* optional outputs may be NULL */
int orc_forward(const hx_pair_job* job, double* cells, double* lp_end_out,
                double* subx_out, double* suby_out, double* insx_out, double* rootsubx_out,
                double* insy_out, double* rootsuby_out) {
  const hx_profile *x = job->x, *y = job->y;
  const hx_hmm* h = job->hmm;
  const double (*T)[6] = h->lp_trans;
  const int R = x->n_states - 1, Cc = y->n_states - 1, CA = h->alph_size * h->components;
  side_t X, Y;
  int i, j, k, kx, ky;
  double lp_end;
  if (!g_tab) return -2;
  if (side_init(&X, x, h, 0) || side_init(&Y, y, h, 1)) return -1;
  for (i = 0; i < R; ++i)
    for (j = 0; j < Cc; ++j) {
      double* c = CELL(i, j);
      c[0] = c[1] = c[2] = c[3] = c[4] = NEG_INF;
    }
#define lse(a, b) cell_lse(a, b)
  CELL(0, 0)[0] = 0;
  for (i = 0; i < R; ++i) {
    const int xnull = x->is_null[i];
    for (j = 0; j < Cc; ++j) {
      const int ynull = y->is_null[j];
      double* dest;
      double imm, imd, idm, imi, iiw;
      if (!in_envelope(&X, &Y, job->max_distance, i, j)) continue;
      dest = CELL(i, j);
      imm = dest[0]; imd = dest[1]; idm = dest[2]; imi = dest[3]; iiw = dest[4];
      if (!xnull) {
        if (Y.ready[j] || Y.empty) {
          for (k = x->in_off[i]; k < x->in_off[i + 1]; ++k) {
            const double* s = CELL(TSRC(x, k), j);
            imd = lse(imd, lse(lse(lse(s[0] + T[0][1], s[1] + T[1][1]), s[2] + T[2][1]), s[3] + T[3][1]) + TLP(x, k));
            iiw = lse(iiw, lse(lse(s[0] + T[0][4], s[3] + T[3][4]), s[4] + T[4][4]) + TLP(x, k));
          }
          imd += X.rootsub[i];
          iiw += X.ins[i];
        }
      } else {
        if (Y.ready[j] || Y.empty)
          for (k = x->in_off[i]; k < x->in_off[i + 1]; ++k) {
            const double* s = CELL(TSRC(x, k), j);
            imd = lse(imd, s[1] + TLP(x, k));
            iiw = lse(iiw, s[4] + TLP(x, k));
          }
      }
      if (!ynull) {
        if (X.ready[i] || X.empty) {
          for (k = y->in_off[j]; k < y->in_off[j + 1]; ++k) {
            const double* s = CELL(i, TSRC(y, k));
            idm = lse(idm, lse(lse(lse(s[0] + T[0][2], s[1] + T[1][2]), s[2] + T[2][2]), s[4] + T[4][2]) + TLP(y, k));
            imi = lse(imi, lse(s[0] + T[0][3], s[3] + T[3][3]) + TLP(y, k));
          }
          idm += Y.rootsub[j];
          imi += Y.ins[j];
        }
      } else {
        for (k = y->in_off[j]; k < y->in_off[j + 1]; ++k) {
          const double* s = CELL(i, TSRC(y, k));
          idm = lse(idm, s[2] + TLP(y, k));
          imi = lse(imi, s[3] + TLP(y, k));
        }
      }
      if (!xnull && !ynull) {
        for (kx = x->in_off[i]; kx < x->in_off[i + 1]; ++kx)
          for (ky = y->in_off[j]; ky < y->in_off[j + 1]; ++ky) {
            const double* s = CELL(TSRC(x, kx), TSRC(y, ky));
            imm = lse(imm, lse(lse(lse(lse(s[0] + T[0][0], s[1] + T[1][0]), s[2] + T[2][0]), s[3] + T[3][0]), s[4] + T[4][0])
                               + TLP(x, kx) + TLP(y, ky));
          }
        imm += emission(h, &X, &Y, i, j);
      } else if (ynull && X.emit_or_start[i]) {
        for (k = y->in_off[j]; k < y->in_off[j + 1]; ++k) imm = lse(imm, CELL(i, TSRC(y, k))[0] + TLP(y, k));
      } else {
        if (Y.ready[j] || Y.empty)
          for (k = x->in_off[i]; k < x->in_off[i + 1]; ++k) imm = lse(imm, CELL(TSRC(x, k), j)[0] + TLP(x, k));
      }
      dest[0] = imm; dest[1] = imd; dest[2] = idm; dest[3] = imi; dest[4] = iiw;
    }
  }
#undef lse
  lp_end = NEG_INF;
  for (kx = x->in_off[R]; kx < x->in_off[R + 1]; ++kx)
    for (ky = y->in_off[Cc]; ky < y->in_off[Cc + 1]; ++ky) {
      const double* s = CELL(TSRC(x, kx), TSRC(y, ky));
      lp_end = lse(lp_end, lse(lse(lse(lse(s[0] + T[0][5], s[1] + T[1][5]), s[2] + T[2][5]), s[3] + T[3][5]), s[4] + T[4][5])
                               + TLP(x, kx) + TLP(y, ky));
    }
  *lp_end_out = lp_end;
  if (subx_out) memcpy(subx_out, X.sub, sizeof(double) * (size_t)X.n * CA);
  if (suby_out) memcpy(suby_out, Y.sub, sizeof(double) * (size_t)Y.n * CA);
  if (insx_out) memcpy(insx_out, X.ins, sizeof(double) * X.n);
  if (rootsubx_out) memcpy(rootsubx_out, X.rootsub, sizeof(double) * X.n);
  if (insy_out) memcpy(insy_out, Y.ins, sizeof(double) * Y.n);
  if (rootsuby_out) memcpy(rootsuby_out, Y.rootsub, sizeof(double) * Y.n);
  side_free(&X);
  side_free(&Y);
  return 0;
}

#define ODST(P, off, idx, k) ((P)->trans_dst[(P)->idx[k]])
#define OLP(P, idx, k) ((P)->trans_lp[(P)->idx[k]])

int orc_backward(const hx_pair_job* job, double* cells, double* lp_start_out) {
  const hx_profile *x = job->x, *y = job->y;
  const hx_hmm* h = job->hmm;
  const double (*T)[6] = h->lp_trans;
  const int R = x->n_states - 1, Cc = y->n_states - 1;
  static const double empty[5] = {NEG_INF, NEG_INF, NEG_INF, NEG_INF, NEG_INF};
  side_t X, Y;
  int i, j, k, kx, ky, s;
  if (!g_tab) return -2;
  if (side_init(&X, x, h, 0) || side_init(&Y, y, h, 1)) return -1;
  for (i = 0; i < R; ++i)
    for (j = 0; j < Cc; ++j) {
      double* c = CELL(i, j);
      c[0] = c[1] = c[2] = c[3] = c[4] = NEG_INF;
    }
  for (kx = x->in_off[R]; kx < x->in_off[R + 1]; ++kx)
    for (ky = y->in_off[Cc]; ky < y->in_off[Cc + 1]; ++ky)
      if (in_envelope(&X, &Y, job->max_distance, TSRC(x, kx), TSRC(y, ky))) {
        double* c = CELL(TSRC(x, kx), TSRC(y, ky));
        for (s = 0; s < 5; ++s) c[s] = TLP(x, kx) + TLP(y, ky) + T[s][5];
      }
#define lse(a, b) cell_lse(a, b)
  for (i = R - 1; i >= 0; --i)
    for (j = Cc - 1; j >= 0; --j) {
      double* src;
      double imm, imd, idm, imi, iiw;
      if (!in_envelope(&X, &Y, job->max_distance, i, j)) continue;
      src = CELL(i, j);
      imm = src[0]; imd = src[1]; idm = src[2]; imi = src[3]; iiw = src[4];
      for (kx = x->aout_off[i]; kx < x->aout_off[i + 1]; ++kx) {
        const int dx = ODST(x, aout_off, aout_idx, kx);
        for (ky = y->aout_off[j]; ky < y->aout_off[j + 1]; ++ky) {
          const int dy = ODST(y, aout_off, aout_idx, ky);
          const double d = OLP(x, aout_idx, kx) + OLP(y, aout_idx, ky) + emission(h, &X, &Y, dx, dy) + CELL(dx, dy)[0];
          imm = lse(imm, T[0][0] + d);
          imd = lse(imd, T[1][0] + d);
          idm = lse(idm, T[2][0] + d);
          imi = lse(imi, T[3][0] + d);
          iiw = lse(iiw, T[4][0] + d);
        }
      }
      if (Y.ready[j] || Y.empty)
        for (kx = x->aout_off[i]; kx < x->aout_off[i + 1]; ++kx) {
          const int dx = ODST(x, aout_off, aout_idx, kx);
          const double* dc = CELL(dx, j);
          const double d1 = OLP(x, aout_idx, kx) + X.rootsub[dx] + dc[1];
          const double d2 = OLP(x, aout_idx, kx) + X.ins[dx] + dc[4];
          imm = lse(imm, T[0][1] + d1);
          imd = lse(imd, T[1][1] + d1);
          idm = lse(idm, T[2][1] + d1);
          imi = lse(imi, T[3][1] + d1);
          imm = lse(imm, T[0][4] + d2);
          imi = lse(imi, T[3][4] + d2);
          iiw = lse(iiw, T[4][4] + d2);
        }
      if (X.ready[i] || X.empty)
        for (ky = y->aout_off[j]; ky < y->aout_off[j + 1]; ++ky) {
          const int dy = ODST(y, aout_off, aout_idx, ky);
          const double* dc = CELL(i, dy);
          const double d1 = OLP(y, aout_idx, ky) + Y.rootsub[dy] + dc[2];
          const double d2 = OLP(y, aout_idx, ky) + Y.ins[dy] + dc[3];
          imm = lse(imm, T[0][2] + d1);
          imd = lse(imd, T[1][2] + d1);
          idm = lse(idm, T[2][2] + d1);
          iiw = lse(iiw, T[4][2] + d1);
          imm = lse(imm, T[0][3] + d2);
          imi = lse(imi, T[3][3] + d2);
        }
      if (Y.ready[j] || Y.empty)
        for (kx = x->nout_off[i]; kx < x->nout_off[i + 1]; ++kx) {
          const int dx = ODST(x, nout_off, nout_idx, kx);
          const double* dc = dx < R ? CELL(dx, j) : empty;
          imd = lse(imd, OLP(x, nout_idx, kx) + dc[1]);
          iiw = lse(iiw, OLP(x, nout_idx, kx) + dc[4]);
          imm = lse(imm, OLP(x, nout_idx, kx) + dc[0]);
        }
      for (ky = y->nout_off[j]; ky < y->nout_off[j + 1]; ++ky) {
        const int dy = ODST(y, nout_off, nout_idx, ky);
        const double* dc = dy < Cc ? CELL(i, dy) : empty;
        idm = lse(idm, OLP(y, nout_idx, ky) + dc[2]);
        imi = lse(imi, OLP(y, nout_idx, ky) + dc[3]);
        if (X.emit_or_start[i]) imm = lse(imm, OLP(y, nout_idx, ky) + dc[0]);
      }
      src[0] = imm; src[1] = imd; src[2] = idm; src[3] = imi; src[4] = iiw;
    }
#undef lse
  *lp_start_out = CELL(0, 0)[0];
  side_free(&X);
  side_free(&Y);
  return 0;
}

/* The reference's storage is a per-row std::map (src/forward.h:22,68); this variant
 * times the same recursion over a dense array, i.e. it is the optimistic stand-in for
 * the reference's cost.  Returns the number of lattice cells visited. */
int64_t orc_cells(const hx_pair_job* job) {
  return (int64_t)(job->x->n_states - 1) * (job->y->n_states - 1);
}
