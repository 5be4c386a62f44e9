"""CPU-side checks of the drop-in boundary: the shared library loads and exports every
symbol include/historian_hip.h declares, and argument errors come back as codes."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from historian_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "historian_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hx_[a-z_]+)\s*\(", text)))


def test_header_declares_what_the_binding_expects():
    assert declared_symbols() == sorted(capi.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_version_and_error_codes_without_a_device():
    lib = capi.load()
    assert lib.hx_version() == 1
    # wrong table size is rejected before any HIP call
    t = np.zeros(10)
    assert lib.hx_init(0, t.ctypes.data_as(C.POINTER(C.c_double)), 10) == -1
    assert b"lse_table" in lib.hx_last_error()
    # batch creation before hx_init
    h = C.c_void_p()
    assert lib.hx_batch_create(None, 0, 0, C.byref(h)) in (-1, -2)


def test_struct_sizes_match_the_header():
    assert C.sizeof(capi.HxCell) == 24
    assert C.sizeof(capi.HxLayout) == 56
    assert C.sizeof(capi.HxHmm) == 8 + 5 * 6 * 8 + 7 * 8
    assert C.sizeof(capi.HxPairJob) == 32


def test_slot_formula_is_a_bijection():
    ss = (((77 + 63) >> 1) + 1) * 128
    l = capi.HxLayout(n_rows=130, n_cols=77, strip_rows=64, n_strips=3, strip_stride=ss, plane_stride=3 * ss, block_stride=128)
    ii, jj = np.meshgrid(np.arange(130), np.arange(77), indexing="ij")
    s = capi.slot_index(l, ii, jj).ravel()
    assert len(np.unique(s)) == s.size and s.min() >= 0 and s.max() < l.plane_stride
