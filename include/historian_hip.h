/*
 * historian_hip.h -- C ABI of the MI355X (gfx950) pair-HMM Forward/Backward engine.
 *
 * This is the drop-in boundary for the hot path of `historian reconstruct`: what
 * the constructors of the reference's DPMatrix / ForwardMatrix / BackwardMatrix
 * (reference src/forward.h:11-227, src/forward.cpp:11-223, 975-1097) compute.
 * The reference has no FFI; its boundary is that C++ class interface, which
 * historian_amd/csrc/host/ mirrors on top of the entry points below
 * (see INTEGRATION.md for the binding a maintainer would add).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, caller-owned input buffers (copied to
 *     the device inside hx_batch_create; the caller may free them afterwards).
 *   - every function returns HX_OK (0) or a negative hx_status; nothing aborts or
 *     throws across the ABI.  A batch is grouped by kernel class internally (leaf pairs, chain profiles, general
 *     profiles; banded or not) and every class is launched with the kernel that suits it; a shape no kernel
 *     supports (an LDS plan over the CU's 160 KB ...) is refused with HX_ERR_INVALID_ARG before anything is launched.  A zero-likelihood fill is NOT an error: lp_end is
 *     -inf and the caller widens the band (reference src/recon.cpp:956-975).
 *   - all log-probabilities are IEEE fp64; -inf means probability zero.
 *   - "stream" arguments are a hipStream_t passed as void* (NULL = default stream).
 */
#ifndef HISTORIAN_HIP_H
#define HISTORIAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum hx_status {
  HX_OK = 0,
  HX_ERR_INVALID_ARG = -1,   /* null pointer, negative size, malformed CSR ...        */
  HX_ERR_NOT_INITIALIZED = -2,
  HX_ERR_NO_DEVICE = -3,     /* no HIP device / HIP runtime error at init             */
  HX_ERR_HIP = -4,           /* a HIP call failed; see hx_last_error()                */
  HX_ERR_NOT_TOPOSORTED = -5,/* a transition has src >= dest (reference profile.cpp:122-125) */
  HX_ERR_OUT_OF_MEMORY = -6,
  HX_ERR_STATE = -7,         /* call order violated (e.g. backward before forward)    */
  HX_ERR_RANGE = -8          /* index out of range                                    */
} hx_status;

/* Pair-HMM state indices, as reference src/pairhmm.h:14-18. */
enum { HX_IMM = 0, HX_IMD = 1, HX_IDM = 2, HX_IMI = 3, HX_IIW = 4, HX_STATES = 5, HX_EEE = 5 };

/* Number of doubles in the log_sum_exp lookup table the caller must supply:
 * LOG_SUM_EXP_LOOKUP_ENTRIES (100001, reference src/logsumexp.h:22-25) plus one
 * guard entry, because (int)(x/1e-4) can reach 100000 and the reference then reads
 * lookup[n+1] (src/logsumexp.h:53-57).  The table must be built by HOST libm
 * (log(1+exp(-n*1e-4)), reference src/logsumexp.cpp:8-16): device libm differs. */
#define HX_LSE_TABLE_ENTRIES 100002

/* ---- fill modes (hx_batch_create flags) ----------------------------------
 * Without a policy bit a batch runs the DEFAULT policy, HX_LSE_TRUNC: the fastest arithmetic whose best paths are the
 * reference's (0 of 2000 random pairs differ, profiles/r03/trace_identity_sweep_seed7.json) and whose log-likelihoods agree
 * with the reference's to ~1e-11 relative.  HX_LSE_EXACT asks for the reference's table arithmetic bit for bit (2-3x slower)
 * and overrides the other policy bits. */
#define HX_LSE_EXACT 128u /* table + linear interpolation + d>=10 truncation: every cell
                            bit-identical to the reference recursion                  */
#define HX_LSE_FAST  1u  /* same truncation, higher-order LDS-resident table; cells
                            differ from HX_LSE_EXACT by <= ~1e-9 per op               */
#define HX_LSE_LINEAR 17u /* HX_LSE_FAST, and the Forward fill of leaf-profile batches (the headline workload,
                            banded or not, and their Backward fill) runs on scaled PROBABILITIES instead of log-sum-exps: fp64 multiply-adds
                            with one integer exponent per cell, log-probabilities produced at the store
                            (hx_linear.hip).  Exact up to fp64 rounding, so it does NOT reproduce the
                            reference's truncation of terms below e^-10: lpEnd agrees with the reference to
                            ~3e-6 relative (north_star tolerance 1e-4), cells to ~1e-5 per alignment column.
                            Batches that kernel does not cover run as HX_LSE_FAST.                          */
#define HX_LSE_TRUNC 81u  /* HX_LSE_LINEAR with the reference's truncation: the same scaled-probability fills, but every pairwise
                            sum of the reference's left-nested log_sum_exp (src/logsumexp.h:66-100) drops its smaller term when
                            that is at most e^-10 of the larger - what the reference's table does for differences >= 10
                            (src/logsumexp.h:45) - instead of adding it.  No table, no logarithm until the store.  Cells agree
                            with HX_LSE_EXACT to the interpolation error of the reference's table (~3e-10 per operation), as
                            HX_LSE_FAST does; best paths are the reference's.  General profiles run as HX_LSE_FAST.           */
#define HX_KEEP_BACKWARD 2u /* pre-allocate the Backward matrices at hx_batch_create        */
#define HX_FORCE_GENERIC 4u /* always use the general (DAG) kernels, even for chain profiles */
#define HX_SPARSE_ENVELOPE 8u /* banded jobs: do not pre-fill the matrices with -inf.  Cells outside the
                                 envelope are then undefined in hx_batch_read_matrix (test the envelope, as the
                                 reference's sparse cell storage makes its callers do, src/forward.h:68-98);
                                 hx_batch_read_cells and hx_batch_posterior_scan still treat them as -inf.
                                 Honoured by the chain (leaf) pipelines; general-profile batches always pre-fill. */

#define HX_BAND_COMPRESSED 32u /* banded jobs: keep only the cells the fill sweeps - per 64-row strip the (at most two)
                                 step windows that hold its in-envelope cells - instead of dense planes: a 2x2000 pair
                                 with band 20 takes ~15 MB instead of 169 MB, so thousands of pairs fit one batch (the
                                 reference's sparse cell map, src/forward.h:22,68, is the same idea).  hx_layout::compressed
                                 is set and hx_batch_strip_windows describes the planes; hx_batch_read_cells,
                                 hx_batch_best_trace and lpEnd work as usual.  As with HX_SPARSE_ENVELOPE, a stored cell that
                                 lies outside the envelope is undefined in hx_batch_read_matrix (test the envelope).  Implemented by the Forward fills of chain
                                 (leaf) profiles, in every arithmetic policy; no general profiles, no Backward fill.   */

/* POD image of a reference Profile (src/profile.h:13-76) restricted to what the
 * fills read.  Transitions are listed once; the three per-state lists hold
 * transition indices in the reference's vector order (ProfileState::in, absorbOut,
 * nullOut), CSR-encoded.  That order is the accumulation order of the fills. */
typedef struct hx_profile {
  int32_t n_states;          /* N = state.size() (START ... END)                      */
  int32_t n_trans;           /* T = trans.size()                                      */
  const int32_t* trans_src;  /* [T] ProfileTransition::src                            */
  const int32_t* trans_dst;  /* [T] ProfileTransition::dest                           */
  const double*  trans_lp;   /* [T] ProfileTransition::lpTrans                        */
  const int32_t* in_off;     /* [N+1]                                                 */
  const int32_t* in_idx;     /* [in_off[N]] transition indices, ProfileState::in      */
  const int32_t* aout_off;   /* [N+1]                                                 */
  const int32_t* aout_idx;   /* ProfileState::absorbOut                               */
  const int32_t* nout_off;   /* [N+1]                                                 */
  const int32_t* nout_idx;   /* ProfileState::nullOut                                 */
  const uint8_t* is_null;    /* [N] ProfileState::isNull() (lpAbsorb.empty())         */
  const double*  lp_absorb;  /* [N][C][A] ProfileState::lpAbsorb; ignored rows for null states */
  const int32_t* env_pos;    /* [N] cumulativeMatches[rowPosToCol[closestLeafPos[i]]]
                                (reference src/forward.cpp:36-42, alignpath.h:56-61);
                                may be NULL when max_distance < 0                     */
} hx_profile;

/* POD image of a reference PairHMM (src/pairhmm.h, src/pairhmm.cpp:5-44) plus the
 * per-branch LogProbModel / log substitution matrices the DPMatrix constructor
 * reads (src/forward.cpp:20-21,44-56; src/profile.cpp:78-91). */
typedef struct hx_hmm {
  int32_t alph_size;         /* A */
  int32_t components;        /* C */
  double  lp_trans[5][6];    /* [src][dest], dest 5 = EEE; -inf where PairHMM::lpTrans has no case */
  const double* log_root;    /* [C][A] PairHMM::logRoot (log cptWeight folded in)      */
  const double* log_sub_l;   /* [C][A][A] log(l.subMat[cpt](c,d)), host libm log       */
  const double* log_sub_r;   /* [C][A][A]                                              */
  const double* log_ins_l;   /* [C][A] logl.logInsProb                                 */
  const double* log_ins_r;   /* [C][A]                                                 */
  const double* log_cptw_l;  /* [C] logl.logCptWeight                                  */
  const double* log_cptw_r;  /* [C]                                                    */
} hx_hmm;

typedef struct hx_pair_job {
  const hx_profile* x;       /* left child profile                                     */
  const hx_profile* y;       /* right child profile                                    */
  const hx_hmm* hmm;
  int32_t max_distance;      /* GuideAlignmentEnvelope::maxDistance; < 0 = no band     */
} hx_pair_job;

/* Where cell (i,j), 0 <= i < n_rows = Nx-1, 0 <= j < n_cols = Ny-1, lives in the
 * buffers returned by hx_batch_read_matrix:
 *     l = i % strip_rows,  t = j + l
 *     slot(i,j) = (i / strip_rows) * strip_stride + (t / 2) * block_stride + l * 2 + t % 2
 *     value(i,j,state) = buf[state * plane_stride + slot(i,j)]          (buf: matrix_doubles doubles)
 * (64-row strips, anti-diagonal-major inside a strip, two anti-diagonals interleaved: a
 * wavefront stepping along anti-diagonals writes 16 contiguous bytes per lane and state
 * every second step).  Two instances: separate state planes (block_stride = 2 * strip_rows,
 * plane_stride = n_strips * strip_stride), and - the scaled-probability fills of unbanded leaf
 * batches - the five states of a step pair adjacent (block_stride = 10 * strip_rows,
 * plane_stride = 2 * strip_rows): read the fields, do not assume either.  Cells outside the envelope hold -inf, as
 * DPMatrix::cell() returns for them (reference src/forward.h:79-84). */
typedef struct hx_layout {
  int32_t n_rows, n_cols;
  int32_t strip_rows;        /* 64 */
  int32_t n_strips;
  int64_t strip_stride;      /* doubles per strip per plane                            */
  int64_t plane_stride;      /* doubles per state plane = n_strips * strip_stride      */
  int32_t mirrored;          /* 1 for the Backward matrix: apply the formula to
                                (n_rows-1-i, n_cols-1-j) -- its fill sweeps from the far corner */
  int32_t compressed;        /* 1: HX_BAND_COMPRESSED job.  strip_stride is unused; with the windows {lo0,hi0,lo1,hi1}
                                and offsets {base0,base1} of strip i / strip_rows from hx_batch_strip_windows,
                                  slot(i,j) = base_w + ((t - lo_w) / 2) * (2 * strip_rows) + l * 2 + t % 2
                                for the window w with lo_w <= t < hi_w; a cell in neither window is not stored
                                (it is outside the envelope and reads as -inf).                                  */
  int64_t block_stride;      /* doubles per block of two anti-diagonals of a strip                    */
  int64_t matrix_doubles;    /* doubles hx_batch_read_matrix writes (all five states)                 */
} hx_layout;

typedef struct hx_cell {
  int32_t xpos, ypos, state;
  int32_t pad_;
  double  log_post_prob;
} hx_cell;

typedef struct hx_batch hx_batch;   /* opaque: inputs + matrices of n independent pair DPs, device resident */

/* -- lifetime ---------------------------------------------------------------- */
/* Select the device, upload the host-built lookup table.  Replaces the reference's
 * static LogSumExpLookupTable (src/logsumexp.cpp:6-16). */
int hx_init(int device_ordinal, const double* lse_table, size_t n_entries);
/* Several devices per process (independent pair DPs farmed over the GPUs of a node): call hx_init once per
 * ordinal; the last one initialised is the default device of hx_batch_create / hx_quick_batch_create.  A batch lives on
 * one device for its whole life; batches on different devices may be driven from one host thread (asynchronous launches,
 * one stream per device) or from one thread per device.  hx_shutdown releases every device's tables. */
int hx_device_count(void);
int hx_shutdown(void);
const char* hx_last_error(void);
int hx_version(void);

/* -- batch of independent pair DPs -------------------------------------------- */
/* Validates the jobs, plans the layouts, allocates device memory and copies the
 * inputs.  Replaces the member initialisers of DPMatrix::DPMatrix
 * (src/forward.cpp:11-35). */
int hx_batch_create(const hx_pair_job* jobs, int32_t n_jobs, uint32_t flags, hx_batch** out);
int hx_batch_create_on(int device_ordinal, const hx_pair_job* jobs, int32_t n_jobs, uint32_t flags, hx_batch** out);
int hx_batch_device(const hx_batch* b);
int hx_batch_destroy(hx_batch* b);

/* Asynchronous on `stream`: profile prep (leftMultiply, insx/rootsubx: reference
 * src/profile.cpp:78-91, src/forward.cpp:44-56) + Forward fill + lpEnd
 * (src/forward.cpp:68-223) for every job of the batch. */
int hx_batch_forward(hx_batch* b, void* stream);
/* Asynchronous: Backward fill (src/forward.cpp:975-1088).  Needs a previous
 * hx_batch_forward (the prepared vectors are shared, unlike the reference which
 * recomputes them at src/forward.cpp:976).  The Backward matrices are allocated on
 * the first call unless HX_KEEP_BACKWARD pre-allocated them (that first call then
 * synchronises). */
int hx_batch_backward(hx_batch* b, void* stream);
int hx_batch_sync(hx_batch* b);

/* Results.  All of these synchronise with the batch's last stream. */
int hx_batch_lp_end(hx_batch* b, double* out /* [n_jobs] ForwardMatrix::lpEnd */);
int hx_batch_lp_start(hx_batch* b, double* out /* [n_jobs] BackwardMatrix::lpStart() */);
int hx_batch_layout(const hx_batch* b, int32_t job, int32_t which, hx_layout* out);
/* which: 0 = Forward, 1 = Backward.  out holds hx_layout::matrix_doubles doubles. */
int hx_batch_read_matrix(hx_batch* b, int32_t job, int32_t which, double* out);
/* The same copy, started on the device's copy stream (one per device, shared by its batches) and not waited for: `out` must be page-locked (hx_host_alloc)
 * and must not be read before hx_batch_wait_read(b, job, which) has returned.  Lets a caller that walks the jobs of a
 * batch one after the other on the host (tracebacks in node order, reference src/recon.cpp:1006-1011) have the next
 * matrices in flight while it works on the current one. */
int hx_batch_read_matrix_async(hx_batch* b, int32_t job, int32_t which, double* out);
int hx_batch_wait_read(hx_batch* b, int32_t job, int32_t which);
/* Gather n cells (ij[2k], ij[2k+1]) -> out[5k..5k+4] without copying the matrix. */
int hx_batch_read_cells(hx_batch* b, int32_t job, int32_t which, const int32_t* ij, int64_t n, double* out);
/* Prepared per-state vectors of DPMatrix (src/forward.h:24-25,54): any pointer may be NULL.
 * subx/suby: [N][C][A] leftMultiply results; insx..rootsuby: [N]. */
int hx_batch_read_prepared(hx_batch* b, int32_t job, double* subx, double* suby,
                           double* insx, double* rootsubx, double* insy, double* rootsuby);
/* BackwardMatrix::cellsAbovePostProbThreshold (src/forward.cpp:1302-1319): all
 * (cell,state) with fwd+back-lpEnd >= log(min_post_prob), unordered.  *n_out gets the
 * number found; at most cap are written. */
int hx_batch_posterior_scan(hx_batch* b, int32_t job, double min_post_prob,
                            hx_cell* out, int64_t cap, int64_t* n_out);

/* Device-side ForwardMatrix::bestTrace() (src/forward.cpp:278-302, with sourceCells :309-398 and
 * bestCell :245-255) for EVERY job of the batch: the arg-max path from the END cell back to the start
 * cell, found in the device-resident Forward matrix, so that a caller who needs only best paths never
 * copies 40 B/cell over PCIe.  cells is [n_jobs][cap]; job k's path is written in the reference's Path
 * order (start cell first, (Nx-1,Ny-1,EEE) last) and n_cells[k] gets its length (<= Nx+Ny), or -1 when
 * the job's lpEnd is -inf (the reference asserts), or -2 when a cell had no source transitions (the
 * reference's "traceback failure").  Returns HX_ERR_RANGE when cap was too small.  Additions and
 * comparisons only: the path is bit-identical to the reference's traceback through the same matrix. */
typedef struct hx_trace_cell {
  int32_t xpos, ypos, state;   /* state: 0..4 = IMM,IMD,IDM,IMI,IIW, 5 = EEE */
} hx_trace_cell;
int hx_batch_best_trace(hx_batch* b, hx_trace_cell* cells, int64_t cap, int32_t* n_cells);

/* Near ties met by the walks of the last hx_batch_best_trace: near_tie[k] = 1 when, at some step of job k's path, the best
 * source cell led a different source cell by no more than 1e-9 relative to the value (or tied with it) - a place where
 * bestCell's choice (src/forward.cpp:245-255) hangs on the last bits of the fill's arithmetic.  Two routes of equal
 * probability through a general (internal-node) profile are such places; a policy other than HX_LSE_EXACT may then part from
 * the reference's path (3 of 640 random internal-node pairs, none of 10 000 leaf pairs).  A caller who wants
 * the reference's path there fills the flagged jobs again under HX_LSE_EXACT and takes that trace - what the C++ mirror does
 * in ForwardMatrix::bestTrace when HX_TIE_REFILL=1 (near ties are common on long sequences - half the nodes of a 32-leaf tree -
 * so it is not the default). */
int hx_batch_best_trace_ties(hx_batch* b, int32_t* near_tie);

/* Sampled tracebacks of job `job` in the device-resident Forward matrix: ForwardMatrix::sampleTrace (reference
 * src/forward.cpp:257-276) with DPMatrix::sampleCell (:225-243), n_walks walks one after the other, so that a host which
 * only needs the sampled paths (profile building) copies no matrix.  The reference draws one
 * uniform_real_distribution<double>(0, ptot) value per step from the generator it shares with every other node of the tree;
 * the caller passes the canonical uniforms in [0, 1) that generator produces from its current state
 * (std::generate_canonical<double, 53>: two draws of the 32-bit engine each) and afterwards discards what the walks used:
 * step k of the concatenated walks takes uniforms[k].  cells: [n_walks][cap], a walk's cells start cell first as the
 * reference's Path; n_cells[w] = its length, or < 0 for the walk that failed (-1 lpEnd = -inf, -2 a cell without source
 * cells - both "traceback failure" in the reference -, -3 cap too small, -4 more than 1024 source cells in one step, -5 the
 * reference's "sampleCell fail", -6 out of uniforms) and 0 for the walks behind it, which are not run; draws_used[w] =
 * uniforms consumed up to and including walk w.  Additions and comparisons are the reference's in the reference's order;
 * exp() is the device library's, which differs from glibc's in the last place for some arguments: a walk can leave the
 * reference's only when a draw falls within an ulp of the boundary between two source cells' shares.
 * One wavefront walks them all (the draws order the steps): ~2 microseconds per step. */
int hx_batch_sample_traces(hx_batch* b, int32_t job, int32_t n_walks, const double* uniforms, int64_t n_uniforms,
                           hx_trace_cell* cells, int64_t cap, int32_t* n_cells, int64_t* draws_used);

/* HX_BAND_COMPRESSED jobs: the step windows [n_strips][4] and their plane offsets [n_strips][2] (see hx_layout). */
int hx_batch_strip_windows(const hx_batch* b, int32_t job, int32_t* windows, int64_t* bases);

/* Total in-envelope-or-not lattice cells of the batch, sum (Nx-1)(Ny-1). */
int64_t hx_batch_total_cells(const hx_batch* b);

/* Diagnostics: which fill kernel takes pair `job`.  *forward_class = the kernel class of DESIGN.md section 5 (0 leaf pairs in
 * LDS, 1 the same banded, 2 the banded rotating-row sweep, 3/4 other leaf pairs, 5/6 in-degree-1 profiles, 7/8 general
 * profiles, 9 the barrier-per-diagonal kernels); *backward_sweep = 1 when hx_batch_backward also runs the rotating-row
 * sweep for it (class 2, dense planes), 0 when it runs the class's strip pipeline.  Either may be NULL. */
int hx_batch_job_kernel(const hx_batch* b, int32_t job, int32_t* forward_class, int32_t* backward_sweep);

/* Diagnostics: the number of pairs whose banded fill runs two pairs per wavefront (hx_band2.hip: scaled-probability policies,
 * every banded leaf pair of the batch admitted, more than 1024 such pairs or HX_BAND2=1), 0 when none does; negative: an error. */
int hx_batch_shared_wavefront_pairs(const hx_batch* b);

/* Diagnostics: how often the batch's fills were launched again with one workgroup per pair because, in a launch that deals a
 * pair's strips to several workgroups (small batches of large pairs), waves gave up waiting for one another - possible only
 * when not all of that launch's workgroups were resident.  hx_batch_lp_end / hx_batch_lp_start do this by themselves, once;
 * HX_ERR_HIP is returned only if the repeated fill fails too.  Negative: an error. */
int hx_batch_relaunches(const hx_batch* b);

/* Duration in milliseconds of the most recent hx_batch_forward / hx_batch_backward
 * fill kernel (HIP events recorded around that kernel on its stream). */
int hx_batch_last_kernel_ms(hx_batch* b, int32_t which, float* ms);

/* Expected indel events of pair `job` from its Forward and Backward matrices: BackwardMatrix::getCounts restricted to
 * the IndelCounts members (reference src/forward.cpp:1183-1214, transitionEigenCounts :579-652) - every transition between
 * two cells weighted with its posterior probability exp(F(src) + lp + B(dest) - lpEnd).  For profiles whose transitions
 * carry no event counts of their own (leaf profiles; the x.getTrans(..)->counts terms of the reference are then zero).
 * branch_times[6] = {l.t, r.t, l.insWait, l.delWait, r.insWait, r.delWait} (ProbModel members of the pair HMM's branches,
 * src/model.cpp:374-391); out[6] = {ins, del, insExt, delExt, insTime, delTime}.  Needs both fills; synchronises. */
int hx_batch_indel_counts(hx_batch* b, int32_t job, const double* branch_times, double* out);

/* -- guide-alignment Viterbi (reference src/quickalign.cpp, src/diagenv.cpp) -------------------
 * The pairwise DP that builds the guide alignment the Forward fills are banded around: a batch of
 * independent QuickAlignMatrix fills.  Results are bit-identical to the reference (adds and maxima). */
typedef struct hx_quick_job {
  const int32_t* x_tok;      /* [x_len] FastSeq::unvalidatedTokens: -1 for characters outside the alphabet */
  const int32_t* y_tok;      /* [y_len]                                                      */
  int32_t x_len, y_len;
  int32_t alph_size;         /* A <= 31                                                      */
  int32_t n_diagonals;       /* 0 with diagonals == NULL: full envelope (DiagonalEnvelope::initFull) */
  const double* submat;      /* [A][A] QuickAlignMatrix::submat (log odds, src/quickalign.cpp:26-31) */
  const int32_t* diagonals;  /* DiagonalEnvelope::diagonals: the d = i - j the DP visits, or NULL */
  double scores[11];         /* m2m m2i m2d i2i i2m i2d d2d d2m gapOpen gapExtend noGap (src/quickalign.cpp:33-54) */
} hx_quick_job;

typedef struct hx_quick_batch hx_quick_batch;

int hx_quick_batch_create(const hx_quick_job* jobs, int32_t n_jobs, hx_quick_batch** out);
int hx_quick_batch_create_on(int device_ordinal, const hx_quick_job* jobs, int32_t n_jobs, hx_quick_batch** out);
int hx_quick_batch_destroy(hx_quick_batch* b);
/* The fills (QuickAlignMatrix constructor, src/quickalign.cpp:63-99); asynchronous on `stream`. */
int hx_quick_batch_run(hx_quick_batch* b, void* stream);
/* QuickAlignMatrix::result, xEnd, yEnd of every pair (synchronises). */
int hx_quick_batch_results(hx_quick_batch* b, double* score, int32_t* x_end, int32_t* y_end);
/* Layout of pair `job`'s matrix: row = i - 1, column = j - 1 (1 <= i <= xLen, 1 <= j <= yLen), three
 * planes mat, ins, del (hx_layout formula; mirrored = 0).  Cells outside the envelope hold -inf. */
int hx_quick_batch_layout(const hx_quick_batch* b, int32_t job, hx_layout* out);
int hx_quick_batch_read_matrix(hx_quick_batch* b, int32_t job, double* out /* [3 * plane_stride] */);
int64_t hx_quick_batch_total_cells(const hx_quick_batch* b);     /* sum xLen * yLen */
int hx_quick_batch_last_kernel_ms(hx_quick_batch* b, float* ms);

/* -- counts mode: column sum-product and substitution counts (reference src/sumprod.cpp) ----------
 * A batch of alignment columns on one tree: SumProduct::initColumn / fillUp / fillDown for every column
 * (src/sumprod.cpp:58-198), the column log-likelihoods (colLogLike), the root residue posteriors
 * (logNodePostProb at the column's root, src/sumprod.cpp:208-217) and the weighted sums over columns of
 * accumulateRootCounts (src/sumprod.cpp:264-271) and accumulateEigenCounts (src/sumprod.cpp:294-372) - what
 * AlignColSumProduct-driven EigenCounts::accumulateSubstitutionCounts (src/sumprod.cpp:430-459) adds up
 * before getSubCounts turns the eigen-basis matrix into wait times and substitution counts on the host.
 * The eigen decomposition itself (GSL in the reference) stays with the caller: it hands over the eigenvectors,
 * their inverse, exp(R t) and EigenModel::eigenSubCount(t) of every branch. */
typedef struct hx_sumprod_model {
  int32_t alph_size;             /* A                                                                    */
  int32_t components;            /* C mixture components                                                 */
  int32_t n_nodes;               /* N tree nodes, numbered children-before-parents (Tree's node order), root last */
  const int32_t* parent;         /* [N] parent node or -1 (Tree::parentNode); at most two children per node */
  const double* ins_prob;        /* [C][A] RateModel::insProb                                            */
  const double* log_cpt_weight;  /* [C] log RateModel::cptWeight                                         */
  const double* branch_sub;      /* [C][N][A][A] EigenModel::getSubProbMatrix(branchLength(n)) (unused at the root) */
  const double* evec_re;         /* [C][A][A] EigenModel::evec, split into real and imaginary parts      */
  const double* evec_im;
  const double* evec_inv_re;     /* [C][A][A] EigenModel::evecInv                                        */
  const double* evec_inv_im;
  const double* esc_re;          /* [C][N][A][A] EigenModel::eigenSubCount(branchLength(n)) (src/model.cpp) */
  const double* esc_im;
} hx_sumprod_model;

/* tokens: [n_cols][N] one byte per node and column: the residue's token, -1 for a wildcard, -2 for a gap
 * (a column's ungapped nodes form one subtree: Alignment::isGap / SumProduct::initColumn's contract).
 * weight: [n_cols] multiplier of each column's counts, or NULL for 1.
 * Outputs, host memory: col_log_like [n_cols]; root_counts [C][A]; eigen_re, eigen_im [C][A][A]
 * (EigenCounts::rootCount, eigenCount); root_post [n_cols][A] log posteriors, or NULL.
 * Runs on the calling thread's current HIP device, which hx_init must have been called for; synchronous. */
int hx_sumprod_columns(const hx_sumprod_model* model, const int8_t* tokens, const double* weight, int64_t n_cols,
                       double* col_log_like, double* root_counts, double* eigen_re, double* eigen_im, double* root_post,
                       void* stream);
/* Duration of the most recent hx_sumprod_columns kernel on this thread (HIP events on its stream). */
int hx_sumprod_last_kernel_ms(float* ms);

/* Page-locked host memory for the destination of hx_batch_read_matrix: a device-to-host copy into
 * pageable memory runs at a fraction of the link rate (measured 4.7 GB/s for a 55 MB matrix).
 * Plain memory to the caller; release with hx_host_free. */
int hx_host_alloc(size_t bytes, void** out);
int hx_host_free(void* p);

/* ---- next row N4: per-branch pair DPs -----------------------------------------------------------
 * The three-state (Match / Insert / Delete) alignment of a parent sequence profile x with a child sequence profile y
 * across one tree branch inside a GuideAlignmentEnvelope: Refiner::BranchMatrix (reference src/refiner.cpp:10-60, Viterbi;
 * `historian reconstruct -refine` runs it on every branch) and Sampler::BranchMatrix (src/sampler.cpp:1034-1084, the same
 * lattice with log_sum_exp).  A batch is a set of independent branches (all branches of a tree in a refinement sweep).
 * Cells and lpEnd are bit-identical to the reference recursion for the same inputs in both forms. */
typedef struct hx_branch_job {
  int32_t x_len, y_len;      /* positions of the parent / child profile; the matrix has (x_len + 1) x (y_len + 1) cells   */
  int32_t components;        /* mixture components C                                                                     */
  int32_t alphabet;          /* A                                                                                        */
  const double* x_pwm;       /* [x_len][C][A] log weights of the parent profile (PosWeightMatrix xSeq)                   */
  const double* y_sub;       /* [y_len][C][A] the child profile left-multiplied by the branch's log substitution matrix
                                (BranchMatrixBase::ySub = TreeAlignFuncs::preMultiply, src/sampler.cpp:452-463)          */
  const double* y_emit;      /* [y_len] insertion scores (BranchMatrixBase::yEmit = calcInsProbs, src/sampler.cpp:465-476) */
  double trans[3][4];        /* log ProbModel::transProb(src, dest): src Match, Insert, Delete; dest ... and End (3)      */
  const int32_t* x_env;      /* [x_len + 1] envelope coordinate of every position: cumulativeMatches[row1PosToCol[xEnvPos[i]]]
                                of the GuideAlignmentEnvelope (src/alignpath.h:56-61); NULL with max_distance < 0          */
  const int32_t* y_env;      /* [y_len + 1]                                                                              */
  int32_t max_distance;      /* GuideAlignmentEnvelope::maxDistance; < 0: no band (every cell in the envelope)           */
} hx_branch_job;
typedef struct hx_branch_batch hx_branch_batch;

int hx_branch_batch_create(const hx_branch_job* jobs, int32_t n_jobs, hx_branch_batch** out);   /* on the current device */
int hx_branch_batch_destroy(hx_branch_batch* b);
/* viterbi != 0: the max-plus fill of Refiner::BranchMatrix; 0: the log_sum_exp fill of Sampler::BranchMatrix */
int hx_branch_batch_run(hx_branch_batch* b, int32_t viterbi, void* stream);
int hx_branch_batch_results(hx_branch_batch* b, double* lp_end /* [n_jobs] */);
/* dense [x_len + 1][y_len + 1][3] (Match, Insert, Delete); -inf outside the envelope, as SparseDPMatrix::cell() returns */
int hx_branch_batch_read_matrix(hx_branch_batch* b, int32_t job, double* out);
int64_t hx_branch_batch_total_cells(const hx_branch_batch* b);
int hx_branch_batch_last_kernel_ms(hx_branch_batch* b, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* HISTORIAN_HIP_H */
