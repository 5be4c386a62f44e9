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


def forward(x, y, hmm, max_distance=-1):
    """Returns dict(cells [R][Cc][5], lp_end, subx, suby, insx, rootsubx, insy, rootsuby)."""
    lib = load()
    jobs = capi.make_jobs([(x, y, hmm, max_distance)])
    R, Cc, ca = x.n_states - 1, y.n_states - 1, hmm.alph_size * hmm.components
    cells = np.empty((R, Cc, 5))
    lp_end = C.c_double()
    subx, suby = np.empty((x.n_states, ca)), np.empty((y.n_states, ca))
    insx, rsx, insy, rsy = (np.empty(x.n_states), np.empty(x.n_states), np.empty(y.n_states), np.empty(y.n_states))
    rc = lib.orc_forward(jobs, _ptr(cells), C.byref(lp_end), _ptr(subx), _ptr(suby), _ptr(insx), _ptr(rsx),
                         _ptr(insy), _ptr(rsy))
    assert rc == 0, rc
    return dict(cells=cells, lp_end=lp_end.value, subx=subx, suby=suby, insx=insx, rootsubx=rsx, insy=insy,
                rootsuby=rsy)


def backward(x, y, hmm, max_distance=-1):
    lib = load()
    jobs = capi.make_jobs([(x, y, hmm, max_distance)])
    cells = np.empty((x.n_states - 1, y.n_states - 1, 5))
    lp_start = C.c_double()
    rc = lib.orc_backward(jobs, _ptr(cells), C.byref(lp_start))
    assert rc == 0, rc
    return dict(cells=cells, lp_start=lp_start.value)
