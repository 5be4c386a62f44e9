/* TEST INFRASTRUCTURE ONLY -- plain-C restatement of the reference's QuickAlignMatrix fill
 * (reference src/quickalign.cpp:63-96, src/quickalign.h:43-66), dense storage.
 * Cells outside the envelope hold -inf, as the reference's const getCell() returns for them.
 * Checked against oracle/quickalign_oracle.py (which is pinned by data/testquickalign.out.fa). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

typedef struct {
  double m2m, m2i, m2d, i2i, i2m, i2d, d2d, d2m, gap_open, gap_extend, no_gap;
} qa_scores;

static double gap(const qa_scores* s, uint32_t n) { return s->gap_open + (double)n * s->gap_extend; }

static double maxd(double a, double b) { return a < b ? b : a; }   /* std::max(a,b) */

/* cells: [(xlen+1)*(ylen+1)][3] = mat, ins, del, row-major in (i,j); in_env: bitmap over diagonals d = i-j,
 * indexed d + ylen (size xlen+ylen+1), or NULL for the full envelope.  Returns the Viterbi score. */
double qa_fill(const int32_t* xtok, int32_t xlen, const int32_t* ytok, int32_t ylen, int32_t alph,
               const double* submat, const qa_scores* s, const uint8_t* in_env, double* cells,
               int32_t* x_end, int32_t* y_end) {
  const double NI = -INFINITY;
  const size_t W = (size_t)ylen + 1;
  for (size_t k = 0; k < (size_t)(xlen + 1) * W * 3; ++k) cells[k] = NI;
#define C(i, j, k) cells[((size_t)(i) * W + (size_t)(j)) * 3 + (k)]
  double end = NI;
  *x_end = *y_end = 0;
  for (int32_t j = 1; j <= ylen; ++j)
    for (int32_t i = 1; i <= xlen; ++i) {
      if (in_env && !in_env[i - j + ylen]) continue;
      double mat = maxd(maxd(C(i - 1, j - 1, 0) + s->m2m, C(i - 1, j - 1, 2) + s->d2m), C(i - 1, j - 1, 1) + s->i2m);
      const double sg = (i == 1 ? s->no_gap : gap(s, (uint32_t)(i - 2))) + (j == 1 ? s->no_gap : gap(s, (uint32_t)(j - 2)));
      mat = maxd(mat, 0.0 + sg);
      const int32_t xt = xtok[i - 1], yt = ytok[j - 1];
      mat += (xt < 0 || yt < 0) ? 0.0 : submat[(size_t)xt * alph + yt];
      const double ins = maxd(C(i, j - 1, 1) + s->i2i, C(i, j - 1, 0) + s->m2i);
      const double del = maxd(maxd(C(i - 1, j, 1) + s->i2d, C(i - 1, j, 2) + s->d2d), C(i - 1, j, 0) + s->m2d);
      C(i, j, 0) = mat; C(i, j, 1) = ins; C(i, j, 2) = del;
      const double eg = (i == xlen ? s->no_gap : gap(s, (uint32_t)(xlen - i - 2))) +
                        (j == ylen ? s->no_gap : gap(s, (uint32_t)(ylen - j - 2)));
      const double ij_end = mat + eg;
      if (ij_end > end) { *x_end = i; *y_end = j; end = ij_end; }
    }
#undef C
  return end;
}
