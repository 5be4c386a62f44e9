// Kernel launchers (implemented in hx_kernels.hip / hx_chain.hip).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include "hx_device.h"

namespace hx {

struct PostCell { int32_t xpos, ypos, state, pad; double lpp; };   // == hx_cell

// Launch guards.  A fill launcher checks on the host that the shape it was handed matches what its kernel and grid
// assume (LDS plan within the CU's 160 KB, grid dimensions within the limits) BEFORE launching, and returns 0 or -1
// with launch_error() describing the refusal: an unsupported shape becomes an error code of the C ABI, never a
// faulting launch (a device fault would take the caller's process down inside the next synchronisation).
#define HX_LDS_LIMIT (160 * 1024)

// The device tables come in two element sizes behind the same `const double*`: the reference's log_sum_exp table as 8-byte
// entries (profile prep, emission tables, lpEnd), and 16-byte entries - {f0, df} pairs of the exact policy, FastPiece
// records of the fast policy, {c, -log c} of the scaled-probability logarithm.  Indexing the 8-byte table as 16-byte
// entries reads up to 800 KB past its end, so the launchers take them as distinct types.
struct Tab8 { const double* p; };
struct Tab16 { const double* p; };
const char* launch_error();
int launch_fail(const char* fmt, ...);
// static + dynamic LDS of a kernel against the workgroup limit
#define HX_CHECK_LDS(kernel, dyn_bytes, what)                                                                         \
  do {                                                                                                                \
    hipFuncAttributes attr_;                                                                                          \
    if (hipFuncGetAttributes(&attr_, reinterpret_cast<const void*>(&kernel)) == hipSuccess &&                        \
        attr_.sharedSizeBytes + (size_t)(dyn_bytes) > (size_t)HX_LDS_LIMIT)                                           \
      return launch_fail("%s needs %zu bytes of LDS (limit %d)", what, attr_.sharedSizeBytes + (size_t)(dyn_bytes),  \
                         HX_LDS_LIMIT);                                                                               \
  } while (0)

int launch_prep(const DevJob* d_jobs, int n_jobs, int max_states, int max_cls, int max_ca, int max_cls_pairs,
                Tab8 tab, bool state_records, hipStream_t st);
// the 80-byte state records alone (for a batch whose launch_prep left them out)
void launch_state_records(const DevJob* d_jobs, int n_jobs, int max_states, hipStream_t st);
void launch_scatter_sub(const DevJob* d_jobs, int n_jobs, int max_states, hipStream_t st);
int launch_forward_dag(const DevJob* d_jobs, int n_jobs, int max_rows, Tab8 tab, hipStream_t st);
int launch_forward_chain(const DevJob* d_jobs, int n_jobs, int max_rows, Tab8 tab, Tab16 fast_tab,
                          bool fast, int leaf, bool banded, int yl_cols, int yl_emis, int multi, int* counters, hipStream_t st);
int launch_backward_chain(const DevJob* d_jobs, int n_jobs, int max_rows, Tab8 tab, Tab16 fast_tab,
                           bool fast, int leaf, bool banded, int yl_cols, int yl_emis, int multi, int* counters, hipStream_t st);
int chain_multi_groups(int n_jobs, int max_rows, int max_pairs);   // workgroups per pair of a small batch of unbanded leaf pairs (1: the ordinary launch)
void launch_emission_plane(const DevJob* d_jobs, int n_jobs, int64_t max_plane, Tab8 tab, Tab16 fast_tab, bool fast, hipStream_t st);
void launch_fill_neg_inf(double* p, int64_t n, hipStream_t st);
// scaled-probability Forward fill of general profiles (hx_daglin.hip); scratch per job in DevJob::agg
int64_t dag_linear_scratch_doubles(int64_t plane, int nx, int ny, int tx, int ty);
bool dag_linear_fits(int64_t plane);
int launch_forward_dag_linear(const DevJob* d_jobs, int n_jobs, int max_rows, Tab8 tab, Tab16 log_tab, int multi, int multi_waves,
                              int* counters, hipStream_t st);
void launch_dag_linear_clear(const DevJob* d_jobs, int n_jobs, hipStream_t st);
int launch_forward_dag_pipe(const DevJob* d_jobs, int n_jobs, int max_rows, Tab8 tab, Tab16 fast_tab,
                            bool fast, int multi, int multi_waves, int* counters, hipStream_t st);
int launch_backward_dag_pipe(const DevJob* d_jobs, int n_jobs, int max_rows, Tab8 tab, Tab16 fast_tab,
                             bool fast, bool records, int multi, int multi_waves, hipStream_t st);
int launch_backward_dag(const DevJob* d_jobs, int n_jobs, int max_rows, Tab8 tab, hipStream_t st);
void launch_posterior_scan(const DevJob* d_jobs, int job, double lpp_threshold, PostCell* out,
                           unsigned long long cap, unsigned long long* counter, hipStream_t st);
void launch_gather_cells(const DevJob* d_jobs, int job, const double* M, int mirrored, const int* ij, int64_t n,
                         double* out, hipStream_t st);

// scaled-linear Forward fill of leaf-like pairs (hx_linear.hip); log_tab from build_log_table
int log_table_doubles();
void build_log_table(double* out /* [log_table_doubles()] */);
int launch_forward_leaf_linear(const DevJob* d_jobs, int n_jobs, int max_rows, bool banded, Tab8 tab, Tab16 log_tab,
                                int yl_cols, int yl_emis, int yl_cls, int multi, int* counters, bool trunc, hipStream_t st);

int launch_backward_leaf_linear(const DevJob* d_jobs, int n_jobs, int max_rows, bool banded, Tab8 tab, Tab16 log_tab,
                                 int yl_cols, int yl_emis, int yl_cls, int multi, int* counters, bool trunc, hipStream_t st);

// banded leaf-like pairs, rotating-row sweep (hx_band.hip); pol: 0 = scaled probabilities, 1 = fast, 2 = exact, 3 = scaled probabilities with the reference's truncation
bool band_kernel_fits(int pol, int rows, int cols, int cls);
int launch_forward_band(const DevJob* d_jobs, int n_jobs, int pol, int max_rows, int max_cols, int max_cls, Tab8 tab,
                        Tab16 pol_tab, bool write_edges, hipStream_t st);
// the Backward sweep of the same kernel (mirrored coordinates; pol 1 fast, 2 exact; DevJob::band_rows_bwd)
int launch_backward_band(const DevJob* d_jobs, int n_jobs, int pol, int max_rows, int max_cols, int max_cls, Tab8 tab, Tab16 pol_tab,
                         bool write_edges, hipStream_t st);

// banded leaf-like pairs on scaled probabilities, two pairs per wavefront (hx_band2.hip): pairs whose rows i and i + 31 are
// never alive together (DevJob::band_w32)
bool band2_kernel_fits(int rows, int cols, int cls);
int launch_forward_band2(const DevJob* d_jobs, int n_jobs, bool trunc, int max_rows, int max_cols, int max_cls, Tab8 tab, Tab16 log_tab,
                         bool write_edges, hipStream_t st, hipStream_t edge_st);
// lpEnd (dir 0) / lpStart (dir 1) of the pairs: behind the sweep and the edge kernel of launch_*_band2
void launch_band2_result(const DevJob* d_jobs, int n_jobs, int dir, Tab8 tab, hipStream_t st);
int launch_backward_band2(const DevJob* d_jobs, int n_jobs, bool trunc, int max_rows, int max_cols, int max_cls, Tab8 tab, Tab16 log_tab,
                          bool write_edges, hipStream_t st, hipStream_t edge_st);

// Several-workgroups-per-pair launches of the general-profile fills (hx_dag.hip, hx_daglin.hip): polls of another workgroup's
// progress before a wave gives up (HX_MULTI_PATIENCE overrides; 0 makes every unsatisfied wait give up - the tests' way of
// forcing the error path), and the progress value a wave that gave up publishes: every wave that waits on it - directly or
// through the waves in between - gives up as well, down to the wave of the last strip, which reports NaN.
#define HX_MULTI_POISON 0x7FFFFFFF
inline int multi_patience() {
  const char* e = getenv("HX_MULTI_PATIENCE");
  return e ? atoi(e) : (1 << 22);
}

void launch_indel_counts(const DevJob* d_jobs, int job, const double* d_tm, double* d_out, int64_t cells, Tab8 tab, bool plane_valid,
                         hipStream_t st);
void launch_best_trace(const DevJob* d_jobs, int n_jobs, int32_t* d_paths, int64_t cap, int32_t* d_n_cells, Tab8 tab,
                       bool plane_valid, int32_t* d_near_tie, hipStream_t st);

void launch_sample_traces(const DevJob* d_jobs, int job, int n_walks, const double* d_uniforms, int64_t n_uniforms, int32_t* d_paths,
                          int64_t cap, int32_t* d_n_cells, int64_t* d_draws, Tab8 tab, bool plane_valid, hipStream_t st);

void launch_reverse_paths(const int32_t* d_paths, int64_t cap, const int32_t* d_n_cells, const int64_t* d_off, int32_t* d_out,
                          int n_jobs, hipStream_t st);

int launch_quickalign(const DevQuick* d_jobs, int n_jobs, int max_rows, int max_cols, bool all_full, hipStream_t st);

}  // namespace hx
