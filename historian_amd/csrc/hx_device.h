// Device-side data model shared by the kernels and the C-ABI host code.
// Names follow the reference's domain: profiles, states, transitions, envelope, cells.
#pragma once
#include <stdint.h>

namespace hx {

// per-state flag byte
enum : uint8_t {
  F_NULL = 1,          // ProfileState::isNull()        (reference src/profile.h:32)
  F_READY = 2,         // ProfileState::isReady()       (src/profile.h:36)
  F_EMIT_OR_START = 4, // ProfileState::isEmitOrStart() (src/profile.h:35)
  F_EDGE = 8,          // xNearStart[i] for x, yNearEnd[j] for y (src/forward.cpp:58-65)
  F_TO_END = 16        // state is the source of a transition into END
};

// Per-state constants of the general-profile Forward pipeline (hx_dag.hip), one 80-byte record:
// the first three in-transitions (reference order) are inline, further ones are read from the CSR.
#define HX_DAG_INLINE 3
struct alignas(16) FwdPack {
  double lp0, lp1;            // lpTrans of in-transitions 0 and 1 (0 when absent)
  double rootsub, ins;        // rootsubx / insx of the state (-inf for null states)
  int32_t s0, s1;             // source states of in-transitions 0 and 1 (0 when absent)
  int32_t in_b;               // CSR position of in-transition 0
  int32_t meta;               // flags | in-degree << 8
  int32_t env;                // envelope coordinate (0 without a band)
  int32_t cls;                // emission class, -1 for null states
  int32_t s2;                 // source state of in-transition 2
  int32_t pad0_;
  double lp2;                 // lpTrans of in-transition 2
  double pad1_;
};

struct DevProfile {
  int32_t n;                  // number of states
  int32_t empty;              // Profile::isEmpty()
  const uint8_t* flags;       // [n]
  const int32_t* in_off;      // [n+1] CSR over ProfileState::in (reference order)
  const int32_t* in_src;
  const double*  in_lp;
  const int32_t* ao_off;      // absorbOut
  const int32_t* ao_dst;
  const double*  ao_lp;
  const int32_t* no_off;      // nullOut
  const int32_t* no_dst;
  const double*  no_lp;
  const double*  lp_absorb;   // [n][C*A] raw lpAbsorb
  double* sub;                // [n][C*A] leftMultiply result (subx / suby)
  double* ins;                // [n] insx / insy
  double* rootsub;            // [n] rootsubx / rootsuby
  const int32_t* env;         // [n] envelope coordinate, or nullptr
  const int32_t* cls;         // [n] emission class of the state (-1 for null states)
  const int32_t* cls_rep;     // [n_cls] a representative state of each class
  int32_t n_cls;
  int32_t pad_;
  // chain kernels: per-state constants packed for one 32-byte fetch
  //   pack[i] = { lpTrans of the in-transition (0 for state 0), rootsub, ins, ok-penalty }
  // ok-penalty is 0 when (isReady() || profile empty) else -inf (reference forward.cpp:97,133)
  double* pack;               // [n][4]
  // profile prep runs once per emission class and is scattered to the states
  double* subc;               // [n_cls][C*A] leftMultiply of the class representative
  double* insc;               // [n_cls]
  double* rootsubc;           // [n_cls]
  const int32_t* ecls;        // [n] emission class, n_cls for null states (-> zero row of emis_pad)
  FwdPack* fpack;             // [n]
};

struct DevJob {
  DevProfile x, y;
  double T[5][6];             // PairHMM::lpTrans(src,dest), dest 5 = EEE
  const double* log_root;     // [C][A]
  const double* log_sub_l;    // [C][A][A]
  const double* log_sub_r;
  const double* log_ins_l;    // [C][A]
  const double* log_ins_r;
  const double* log_cptw_l;   // [C]
  const double* log_cptw_r;
  int32_t A, C, CA;
  int32_t max_dist;           // < 0: no band
  int32_t n_rows, n_cols;     // Nx-1, Ny-1
  int32_t n_strips;
  int32_t chain;              // both profiles are linear chains (state i's only in-transition comes from i-1)
  int64_t strip_stride;       // doubles per 64-row strip per plane = (n_cols + 63) * 64
  int64_t plane;              // doubles per state plane
  double* fwd;                // [5][plane]
  double* bwd;                // [5][plane] or nullptr
  double* emis;               // [x.n_cls][y.n_cls] class-pair emission table, or nullptr (per-cell emission)
  double* emis_pad;           // [x.n_cls+1][y.n_cls+1]: emis plus a zero row/column for null states
  int32_t leaf_like;          // chain profiles whose interior states all emit (leaves)
  int32_t pad2_;
  double* lp_end;             // -> one double
  double* lp_start;           // -> one double
  // general-profile strip pipeline (hx_dag.hip)
  double* emis_plane;         // [plane] per-cell emission term in the Forward layout, or nullptr (class table used)
  double* agg;                // [5][plane] per-cell outgoing sums of the Forward fill (hx_dag.hip), or nullptr
  const int32_t* fwd_windows; // [n_strips][4] step windows {lo0,hi0,lo1,hi1} holding the strip's in-envelope cells
  const int32_t* bwd_windows; // same for the mirrored (Backward) sweep; both nullptr when there is no band
  const int64_t* strip_base;  // [n_strips][2] band-compressed storage (HX_BAND_COMPRESSED): offset of each Forward window in a
                              // state plane; fwd_windows then holds the windows exactly as swept.  nullptr: dense planes
  const uint32_t* yword;      // [n_cols + 328] per-column words of the banded scaled-probability fill (hx_linear.hip), or nullptr
  int32_t blk;                // doubles per step-pair block of a strip: 128 (state planes apart), or 640 = the five states of a
                              // step pair adjacent (interleaved layout; plane is then 128 and strip_stride five times as large)
  int32_t band_w32;           // hx_band.hip row records whose rows i and i + 31 are never alive together: the pair may share a wavefront (hx_band2.hip)
  int64_t matrix_doubles;     // doubles of one whole matrix (all five states)
  const uint32_t* yword_bwd;  // the same for the Backward sweep (mirrored column order, class of the state an absorbing move leads to)
  // banded rotating-row sweep (hx_band.hip): one 16-byte record per row + 64 sentinel records, and the number of
  // anti-diagonal steps of the sweep; nullptr when the pair does not run on that kernel
  const void* band_rows;
  int32_t band_steps;
  int32_t band_steps_bwd;     // the same for the Backward sweep (mirrored rows and columns), 0 when band_rows_bwd is null
  const void* band_rows_bwd;
};

// One pair of the guide-alignment Viterbi batch (hx_quick.hip)
struct DevQuick {
  const int32_t* xtok;        // [xlen] tokens, -1 = outside the alphabet
  const int32_t* ytok;        // [ylen]
  int32_t xlen, ylen, alph, n_strips;
  const double* submat;       // [alph][alph] log odds
  double sc[11];              // m2m m2i m2d i2i i2m i2d d2d d2m gapOpen gapExtend noGap
  const uint8_t* in_env;      // [xlen+ylen+1] indexed (i - j) + ylen, or nullptr = full envelope
  double* cells;              // [3][plane]: mat, ins, del in the strip-skewed layout, row = i-1, column = j-1
  int64_t plane, strip_stride;    // state stride and strip stride of the (interleaved, see cell_slot_blk) layout
  int32_t blk, pad_;              // doubles per step-pair block: 3 * 128
  double* col_scratch;        // [2*ylen doubles + ylen ints] per-column constants when they do not fit LDS, else nullptr
  double* best_score;         // [xlen] per-row best mat + endGapScore ...
  int32_t* best_j;            // ... and the first column that attains it
  double* result;             // -> Viterbi score
  int32_t* xy_end;            // -> xEnd, yEnd
};

#define HX_STRIP 64
#define HX_FAST_INTERVALS 4096   // quadratic pieces of the fast log-sum-exp table over [0,10)

// Strip-skewed layout.  Rows are grouped in strips of 64; inside a strip, cell (i,j) with
// l = i % 64 and skewed column t = j + l lives at  (t / 2) * 128 + l * 2 + t % 2.
// A wavefront stepping along anti-diagonals (lane <-> row) therefore writes, every two
// steps, 16 contiguous bytes per lane = 1 KiB contiguous per wave and state plane.
__host__ __device__ inline int64_t cell_slot(int64_t strip_stride, int i, int j) {
  const int l = i & (HX_STRIP - 1);
  const int t = j + l;
  return (int64_t)(i >> 6) * strip_stride + ((int64_t)(t >> 1) << 7) + (l << 1) + (t & 1);
}
// General form: `blk` doubles per step-pair block.  value(i, j, state) = M[state * plane + cell_slot_blk(...)], with
// (plane, blk) = (doubles per state plane, 128) for separate state planes, or (128, 640) when the five states of a step
// pair are adjacent - a wavefront then writes 5 KiB contiguous per iteration (scaled-probability fills, hx_linear.hip).
__host__ __device__ inline int64_t cell_slot_blk(int64_t strip_stride, int blk, int i, int j) {
  const int l = i & (HX_STRIP - 1);
  const int t = j + l;
  return (int64_t)(i >> 6) * strip_stride + (int64_t)(t >> 1) * blk + (l << 1) + (t & 1);
}
// The Backward matrix uses the same layout in mirrored coordinates (i -> R-1-i, j -> C-1-j): its
// fill sweeps from the bottom-right corner, and this keeps the sweep's stores coalesced.
__host__ __device__ inline int64_t bwd_slot(int64_t strip_stride, int n_rows, int n_cols, int i, int j) {
  return cell_slot(strip_stride, n_rows - 1 - i, n_cols - 1 - j);
}
__host__ __device__ inline int64_t strip_stride_for(int n_cols) {
  return ((int64_t)((n_cols + HX_STRIP - 1) >> 1) + 1) * (2 * HX_STRIP);
}

}  // namespace hx
