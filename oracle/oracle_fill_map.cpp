// Test infrastructure only (CPU baseline, SURVEY section 8d variant ii): the oracle's fills compiled over the reference's
// own cell storage - one std::map<index, XYCell> per row, cells created by operator[] on first touch
// (reference src/forward.h:22,68) - instead of a dense array.  Same recursion, same arithmetic, same results; only the
// cost structure changes (tree inserts and look-ups per cell, one allocation per cell), which is what the reference pays.
// Built as its own shared library (same symbol names as liboracle_fill.so).
#include <stddef.h>
#include <array>
#include <map>
#include <vector>

static std::vector<std::map<int, std::array<double, 5> > >* orc_rows;
static inline double* orc_map_cell(int i, int j) { return (*orc_rows)[(size_t)i][j].data(); }

#define ORC_MAP_STORAGE
extern "C" {
#include "oracle_fill.c"

// Forward fill of one job over map storage; returns lpEnd (the cells are dropped)
int orc_forward_map(const hx_pair_job* job, double* lp_end_out) {
  std::vector<std::map<int, std::array<double, 5> > > rows((size_t)job->x->n_states);
  orc_rows = &rows;
  const int rc = orc_forward(job, (double*)0, lp_end_out, 0, 0, 0, 0, 0, 0);
  orc_rows = 0;
  return rc;
}
}
