"""ctypes wrapper of oracle/oracle_fill.c (TEST INFRASTRUCTURE ONLY).

Takes the same POD job image as the C ABI (struct definitions imported from
historian_amd.capi, which only mirrors include/historian_hip.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

from historian_amd import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_fill.so")
_lib = None
_tab = None
_f64p = C.POINTER(C.c_double)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load():
    global _lib, _tab
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        lib = C.CDLL(_SO)
        lib.orc_forward.argtypes = [C.POINTER(capi.HxPairJob), _f64p, _f64p] + [_f64p] * 6
        lib.orc_backward.argtypes = [C.POINTER(capi.HxPairJob), _f64p, _f64p]
        lib.orc_build_table.argtypes = [_f64p]
        lib.orc_set_table.argtypes = [_f64p]
        lib.orc_log_sum_exp.argtypes = [C.c_double, C.c_double]
        lib.orc_log_sum_exp.restype = C.c_double
        _tab = np.empty(capi.HX_LSE_TABLE_ENTRIES)
        lib.orc_build_table(_tab.ctypes.data_as(_f64p))
        lib.orc_set_table(_tab.ctypes.data_as(_f64p))
        _lib = lib
    return _lib


def table():
    load()
    return _tab


def _ptr(a):
    return a.ctypes.data_as(_f64p)


def forward(x, y, hmm, max_distance=-1, true_math=False):
    """Returns dict(cells [R][Cc][5], lp_end, subx, suby, insx, rootsubx, insy, rootsuby).
    true_math: the cell recursion sums probabilities with libm's log1p/exp instead of the reference's table
    operator (a second yardstick for the scaled-linear HIP kernel; everything else stays the reference's);
    true_math=2: the same with the reference's truncation of differences >= 10 kept (yardstick of HX_LSE_TRUNC)."""
    lib = load()
    lib.orc_set_true_math(int(true_math))
    jobs = capi.make_jobs([(x, y, hmm, max_distance)])
    R, Cc, ca = x.n_states - 1, y.n_states - 1, hmm.alph_size * hmm.components
    cells = np.empty((R, Cc, 5))
    lp_end = C.c_double()
    subx, suby = np.empty((x.n_states, ca)), np.empty((y.n_states, ca))
    insx, rsx, insy, rsy = (np.empty(x.n_states), np.empty(x.n_states), np.empty(y.n_states), np.empty(y.n_states))
    rc = lib.orc_forward(jobs, _ptr(cells), C.byref(lp_end), _ptr(subx), _ptr(suby), _ptr(insx), _ptr(rsx),
                         _ptr(insy), _ptr(rsy))
    lib.orc_set_true_math(0)
    assert rc == 0, rc
    return dict(cells=cells, lp_end=lp_end.value, subx=subx, suby=suby, insx=insx, rootsubx=rsx, insy=insy,
                rootsuby=rsy)


def backward(x, y, hmm, max_distance=-1, true_math=False):
    lib = load()
    lib.orc_set_true_math(int(true_math))
    jobs = capi.make_jobs([(x, y, hmm, max_distance)])
    cells = np.empty((x.n_states - 1, y.n_states - 1, 5))
    lp_start = C.c_double()
    rc = lib.orc_backward(jobs, _ptr(cells), C.byref(lp_start))
    lib.orc_set_true_math(0)
    assert rc == 0, rc
    return dict(cells=cells, lp_start=lp_start.value)


# ---- guide-alignment Viterbi: oracle_quickalign.c ----------------------------
class QAScores(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("m2m", "m2i", "m2d", "i2i", "i2m", "i2d", "d2d", "d2m", "gap_open",
                                          "gap_extend", "no_gap")]


def qa_scores(sc):
    """oracle.quickalign_oracle.QuickAlignScores -> QAScores"""
    return QAScores(sc.m2m, sc.m2i, sc.m2d, sc.i2i, sc.i2m, sc.i2d, sc.d2d, sc.d2m, sc.gap_open, sc.gap_extend, sc.no_gap)


def quickalign(xtok, ytok, alph_size, submat, scores, diagonals=None):
    """Dense QuickAlignMatrix fill.  Returns dict(cells [(xlen+1),(ylen+1),3], score, x_end, y_end);
    diagonals: iterable of d = i - j in the envelope, or None for the full envelope."""
    lib = load()
    i32p = C.POINTER(C.c_int32)
    lib.qa_fill.argtypes = [i32p, C.c_int32, i32p, C.c_int32, C.c_int32, _f64p, C.POINTER(QAScores),
                            C.POINTER(C.c_uint8), _f64p, i32p, i32p]
    lib.qa_fill.restype = C.c_double
    xt = np.ascontiguousarray(xtok, dtype=np.int32)
    yt = np.ascontiguousarray(ytok, dtype=np.int32)
    sm = np.ascontiguousarray(submat, dtype=np.float64)
    xlen, ylen = len(xt), len(yt)
    env = None
    if diagonals is not None:
        env = np.zeros(xlen + ylen + 1, dtype=np.uint8)
        for d in diagonals:
            env[d + ylen] = 1
    cells = np.empty((xlen + 1, ylen + 1, 3))
    xe, ye = C.c_int32(0), C.c_int32(0)
    sc = scores if isinstance(scores, QAScores) else qa_scores(scores)
    score = lib.qa_fill(xt.ctypes.data_as(i32p), xlen, yt.ctypes.data_as(i32p), ylen, alph_size, _ptr(sm), C.byref(sc),
                        env.ctypes.data_as(C.POINTER(C.c_uint8)) if env is not None else None, _ptr(cells),
                        C.byref(xe), C.byref(ye))
    return dict(cells=cells, score=score, x_end=xe.value, y_end=ye.value)


_map_lib = None


def forward_map(x, y, hmm, max_distance=-1):
    """lpEnd of the same Forward fill over the reference's cell storage (a std::map per row, oracle_fill_map.cpp):
    the CPU baseline with the reference's cost structure.  The cells are not returned."""
    global _map_lib
    load()
    if _map_lib is None:
        lib = C.CDLL(os.path.join(_HERE, "_build", "liboracle_fill_map.so"))
        lib.orc_set_table.argtypes = [_f64p]
        lib.orc_forward_map.argtypes = [C.POINTER(capi.HxPairJob), C.POINTER(C.c_double)]
        lib.orc_set_table(_tab.ctypes.data_as(_f64p))
        _map_lib = lib
    jobs = capi.make_jobs([(x, y, hmm, max_distance)])
    lp_end = C.c_double()
    rc = _map_lib.orc_forward_map(jobs, C.byref(lp_end))
    assert rc == 0, rc
    return lp_end.value
