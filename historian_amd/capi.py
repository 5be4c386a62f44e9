"""ctypes binding of the C ABI in include/historian_hip.h.

Plumbing only: it marshals numpy arrays into the POD structs and calls the shared
library.  There is no CPU fallback -- if the HIP library is missing or the device
is absent every entry point raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HX_LIB_PATH") or os.path.join(_HERE, "lib", "libhistorian_hip.so")

HX_LSE_TABLE_ENTRIES = 100002
HX_LSE_EXACT, HX_LSE_FAST, HX_KEEP_BACKWARD, HX_FORCE_GENERIC, HX_SPARSE_ENVELOPE = 128, 1, 2, 4, 8
HX_LSE_DEFAULT = 256    # the library's default policy (flags without any policy bit: HX_LSE_TRUNC)
HX_BAND_COMPRESSED = 32   # banded jobs keep only the swept step windows (include/historian_hip.h)
HX_LSE_TRUNC = 81       # HX_LSE_LINEAR with the reference's truncation of terms at most e^-10 of their sum (include/historian_hip.h)
HX_LSE_LINEAR = 17      # HX_LSE_FAST + scaled-probability Forward fill where it applies (include/historian_hip.h)
IMM, IMD, IDM, IMI, IIW, EEE = 0, 1, 2, 3, 4, 5

_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)


class HxProfile(C.Structure):
    _fields_ = [("n_states", C.c_int32), ("n_trans", C.c_int32),
                ("trans_src", _i32p), ("trans_dst", _i32p), ("trans_lp", _f64p),
                ("in_off", _i32p), ("in_idx", _i32p),
                ("aout_off", _i32p), ("aout_idx", _i32p),
                ("nout_off", _i32p), ("nout_idx", _i32p),
                ("is_null", _u8p), ("lp_absorb", _f64p), ("env_pos", _i32p)]


class HxHmm(C.Structure):
    _fields_ = [("alph_size", C.c_int32), ("components", C.c_int32),
                ("lp_trans", (C.c_double * 6) * 5),
                ("log_root", _f64p), ("log_sub_l", _f64p), ("log_sub_r", _f64p),
                ("log_ins_l", _f64p), ("log_ins_r", _f64p),
                ("log_cptw_l", _f64p), ("log_cptw_r", _f64p)]


class HxPairJob(C.Structure):
    _fields_ = [("x", C.POINTER(HxProfile)), ("y", C.POINTER(HxProfile)),
                ("hmm", C.POINTER(HxHmm)), ("max_distance", C.c_int32)]


class HxLayout(C.Structure):
    _fields_ = [("n_rows", C.c_int32), ("n_cols", C.c_int32), ("strip_rows", C.c_int32),
                ("n_strips", C.c_int32), ("strip_stride", C.c_int64), ("plane_stride", C.c_int64),
                ("mirrored", C.c_int32), ("compressed", C.c_int32),
                ("block_stride", C.c_int64), ("matrix_doubles", C.c_int64)]


class HxQuickJob(C.Structure):
    _fields_ = [("x_tok", C.POINTER(C.c_int32)), ("y_tok", C.POINTER(C.c_int32)), ("x_len", C.c_int32),
                ("y_len", C.c_int32), ("alph_size", C.c_int32), ("n_diagonals", C.c_int32),
                ("submat", C.POINTER(C.c_double)), ("diagonals", C.POINTER(C.c_int32)), ("scores", C.c_double * 11)]


class HxBranchJob(C.Structure):
    _fields_ = [("x_len", C.c_int32), ("y_len", C.c_int32), ("components", C.c_int32), ("alphabet", C.c_int32),
                ("x_pwm", C.POINTER(C.c_double)), ("y_sub", C.POINTER(C.c_double)), ("y_emit", C.POINTER(C.c_double)),
                ("trans", (C.c_double * 4) * 3), ("x_env", C.POINTER(C.c_int32)), ("y_env", C.POINTER(C.c_int32)),
                ("max_distance", C.c_int32)]


class HxSumprodModel(C.Structure):
    _fields_ = [("alph_size", C.c_int32), ("components", C.c_int32), ("n_nodes", C.c_int32), ("parent", _i32p),
                ("ins_prob", _f64p), ("log_cpt_weight", _f64p), ("branch_sub", _f64p),
                ("evec_re", _f64p), ("evec_im", _f64p), ("evec_inv_re", _f64p), ("evec_inv_im", _f64p),
                ("esc_re", _f64p), ("esc_im", _f64p)]


class HxCell(C.Structure):
    _fields_ = [("xpos", C.c_int32), ("ypos", C.c_int32), ("state", C.c_int32), ("pad_", C.c_int32),
                ("log_post_prob", C.c_double)]


EXPORTS = ["hx_init", "hx_shutdown", "hx_last_error", "hx_version", "hx_batch_create", "hx_batch_destroy",
           "hx_batch_forward", "hx_batch_backward", "hx_batch_sync", "hx_batch_lp_end", "hx_batch_lp_start",
           "hx_batch_layout", "hx_batch_read_matrix", "hx_batch_read_cells", "hx_batch_read_prepared",
           "hx_batch_posterior_scan", "hx_batch_best_trace", "hx_batch_best_trace_ties", "hx_batch_sample_traces", "hx_device_count", "hx_batch_create_on", "hx_batch_device",
           "hx_quick_batch_create_on", "hx_batch_strip_windows", "hx_batch_total_cells", "hx_batch_job_kernel", "hx_batch_last_kernel_ms", "hx_host_alloc",
           "hx_host_free", "hx_quick_batch_create", "hx_quick_batch_destroy", "hx_quick_batch_run",
           "hx_quick_batch_results", "hx_quick_batch_layout", "hx_quick_batch_read_matrix",
           "hx_quick_batch_total_cells", "hx_quick_batch_last_kernel_ms", "hx_sumprod_columns", "hx_sumprod_last_kernel_ms",
           "hx_batch_read_matrix_async", "hx_batch_wait_read", "hx_batch_indel_counts", "hx_batch_shared_wavefront_pairs", "hx_batch_relaunches",
           "hx_branch_batch_create", "hx_branch_batch_destroy", "hx_branch_batch_run", "hx_branch_batch_results",
           "hx_branch_batch_read_matrix", "hx_branch_batch_total_cells", "hx_branch_batch_last_kernel_ms"]


class HxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("historian_hip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load():
    """Load the HIP library; fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback for the product path)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.hx_init.argtypes = [C.c_int, _f64p, C.c_size_t]
    lib.hx_last_error.restype = C.c_char_p
    lib.hx_batch_create.argtypes = [C.POINTER(HxPairJob), C.c_int32, C.c_uint32, C.POINTER(vp)]
    lib.hx_batch_create_on.argtypes = [C.c_int, C.POINTER(HxPairJob), C.c_int32, C.c_uint32, C.POINTER(vp)]
    lib.hx_batch_device.argtypes = [vp]
    lib.hx_batch_destroy.argtypes = [vp]
    lib.hx_batch_forward.argtypes = [vp, vp]
    lib.hx_batch_backward.argtypes = [vp, vp]
    lib.hx_batch_sync.argtypes = [vp]
    lib.hx_batch_lp_end.argtypes = [vp, _f64p]
    lib.hx_batch_lp_start.argtypes = [vp, _f64p]
    lib.hx_batch_layout.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(HxLayout)]
    lib.hx_batch_read_matrix.argtypes = [vp, C.c_int32, C.c_int32, _f64p]
    lib.hx_batch_read_cells.argtypes = [vp, C.c_int32, C.c_int32, _i32p, C.c_int64, _f64p]
    lib.hx_batch_read_prepared.argtypes = [vp, C.c_int32] + [_f64p] * 6
    lib.hx_batch_posterior_scan.argtypes = [vp, C.c_int32, C.c_double, C.POINTER(HxCell), C.c_int64,
                                            C.POINTER(C.c_int64)]
    lib.hx_batch_best_trace.argtypes = [vp, vp, C.c_int64, _i32p]
    lib.hx_batch_best_trace_ties.argtypes = [vp, _i32p]
    lib.hx_batch_sample_traces.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.c_int64, vp, C.c_int64, _i32p, C.POINTER(C.c_int64)]
    lib.hx_batch_strip_windows.argtypes = [vp, C.c_int32, _i32p, C.POINTER(C.c_int64)]
    lib.hx_batch_job_kernel.argtypes = [vp, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.hx_batch_total_cells.argtypes = [vp]
    lib.hx_batch_shared_wavefront_pairs.argtypes = [vp]
    lib.hx_batch_relaunches.argtypes = [vp]
    lib.hx_batch_total_cells.restype = C.c_int64
    lib.hx_batch_last_kernel_ms.argtypes = [vp, C.c_int32, C.POINTER(C.c_float)]
    lib.hx_quick_batch_create.argtypes = [C.POINTER(HxQuickJob), C.c_int32, C.POINTER(vp)]
    lib.hx_branch_batch_create.argtypes = [C.POINTER(HxBranchJob), C.c_int32, C.POINTER(vp)]
    lib.hx_branch_batch_destroy.argtypes = [vp]
    lib.hx_branch_batch_run.argtypes = [vp, C.c_int32, vp]
    lib.hx_branch_batch_results.argtypes = [vp, _f64p]
    lib.hx_branch_batch_read_matrix.argtypes = [vp, C.c_int32, _f64p]
    lib.hx_branch_batch_total_cells.argtypes = [vp]
    lib.hx_branch_batch_total_cells.restype = C.c_int64
    lib.hx_branch_batch_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.hx_quick_batch_destroy.argtypes = [vp]
    lib.hx_quick_batch_run.argtypes = [vp, vp]
    lib.hx_quick_batch_results.argtypes = [vp, _f64p, _i32p, _i32p]
    lib.hx_quick_batch_layout.argtypes = [vp, C.c_int32, C.POINTER(HxLayout)]
    lib.hx_quick_batch_read_matrix.argtypes = [vp, C.c_int32, _f64p]
    lib.hx_quick_batch_total_cells.argtypes = [vp]
    lib.hx_quick_batch_total_cells.restype = C.c_int64
    lib.hx_quick_batch_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.hx_sumprod_columns.argtypes = [C.POINTER(HxSumprodModel), C.POINTER(C.c_int8), _f64p, C.c_int64, _f64p, _f64p, _f64p, _f64p,
                                       _f64p, vp]
    lib.hx_sumprod_last_kernel_ms.argtypes = [C.POINTER(C.c_float)]
    lib.hx_batch_read_matrix_async.argtypes = [vp, C.c_int32, C.c_int32, vp]
    lib.hx_batch_wait_read.argtypes = [vp, C.c_int32, C.c_int32]
    lib.hx_batch_indel_counts.argtypes = [vp, C.c_int32, _f64p, _f64p]
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise HxError(rc, load().hx_last_error().decode())


def _p(a, typ):
    return a.ctypes.data_as(typ) if a is not None and a.size else C.cast(None, typ)


class ProfileImage:
    """Owns the numpy arrays behind one hx_profile."""

    def __init__(self, trans_src, trans_dst, trans_lp, in_lists, aout_lists, nout_lists, is_null, lp_absorb,
                 env_pos=None):
        n = len(is_null)
        self.n_states = n
        self.trans_src = np.ascontiguousarray(trans_src, dtype=np.int32)
        self.trans_dst = np.ascontiguousarray(trans_dst, dtype=np.int32)
        self.trans_lp = np.ascontiguousarray(trans_lp, dtype=np.float64)
        self.in_off, self.in_idx = self._csr(in_lists)
        self.aout_off, self.aout_idx = self._csr(aout_lists)
        self.nout_off, self.nout_idx = self._csr(nout_lists)
        self.is_null = np.ascontiguousarray(is_null, dtype=np.uint8)
        self.lp_absorb = np.ascontiguousarray(lp_absorb, dtype=np.float64)   # [N][C][A]
        self.env_pos = None if env_pos is None else np.ascontiguousarray(env_pos, dtype=np.int32)
        s = HxProfile()
        s.n_states, s.n_trans = n, len(self.trans_src)
        s.trans_src, s.trans_dst, s.trans_lp = _p(self.trans_src, _i32p), _p(self.trans_dst, _i32p), _p(self.trans_lp, _f64p)
        s.in_off, s.in_idx = _p(self.in_off, _i32p), _p(self.in_idx, _i32p)
        s.aout_off, s.aout_idx = _p(self.aout_off, _i32p), _p(self.aout_idx, _i32p)
        s.nout_off, s.nout_idx = _p(self.nout_off, _i32p), _p(self.nout_idx, _i32p)
        s.is_null, s.lp_absorb = _p(self.is_null, _u8p), _p(self.lp_absorb, _f64p)
        s.env_pos = _p(self.env_pos, _i32p) if self.env_pos is not None else C.cast(None, _i32p)
        self.struct = s

    @staticmethod
    def _csr(lists):
        off = np.zeros(len(lists) + 1, dtype=np.int32)
        for i, l in enumerate(lists):
            off[i + 1] = off[i] + len(l)
        idx = np.fromiter((t for l in lists for t in l), dtype=np.int32, count=int(off[-1]))
        return off, idx

    @staticmethod
    def chain(is_null, lp_absorb, env_pos=None):
        """Linear-chain profile (a leaf: reference src/profile.cpp:23-76): state k -> k+1
        with lpTrans 0; the last transition (into END) is a null transition."""
        n = len(is_null)
        t = np.arange(n - 1, dtype=np.int32)
        in_lists = [[]] + [[k] for k in range(n - 1)]
        aout = [[k] if (k < n - 1 and not is_null[k + 1]) else [] for k in range(n)]
        nout = [[k] if (k < n - 1 and is_null[k + 1]) else [] for k in range(n)]
        return ProfileImage(t, t + 1, np.zeros(n - 1), in_lists, aout, nout, is_null, lp_absorb, env_pos)


class HmmImage:
    def __init__(self, lp_trans, log_root, log_sub_l, log_sub_r, log_ins_l, log_ins_r, log_cptw_l, log_cptw_r):
        self.log_root = np.ascontiguousarray(log_root, dtype=np.float64)
        c, a = self.log_root.shape
        self.log_sub_l = np.ascontiguousarray(log_sub_l, dtype=np.float64).reshape(c, a, a)
        self.log_sub_r = np.ascontiguousarray(log_sub_r, dtype=np.float64).reshape(c, a, a)
        self.log_ins_l = np.ascontiguousarray(log_ins_l, dtype=np.float64).reshape(c, a)
        self.log_ins_r = np.ascontiguousarray(log_ins_r, dtype=np.float64).reshape(c, a)
        self.log_cptw_l = np.ascontiguousarray(log_cptw_l, dtype=np.float64).reshape(c)
        self.log_cptw_r = np.ascontiguousarray(log_cptw_r, dtype=np.float64).reshape(c)
        self.lp_trans = np.ascontiguousarray(lp_trans, dtype=np.float64).reshape(5, 6)
        s = HxHmm()
        s.alph_size, s.components = a, c
        for i in range(5):
            for j in range(6):
                s.lp_trans[i][j] = self.lp_trans[i, j]
        s.log_root = _p(self.log_root, _f64p)
        s.log_sub_l, s.log_sub_r = _p(self.log_sub_l, _f64p), _p(self.log_sub_r, _f64p)
        s.log_ins_l, s.log_ins_r = _p(self.log_ins_l, _f64p), _p(self.log_ins_r, _f64p)
        s.log_cptw_l, s.log_cptw_r = _p(self.log_cptw_l, _f64p), _p(self.log_cptw_r, _f64p)
        self.struct = s
        self.alph_size, self.components = a, c


def make_jobs(triples):
    """triples: list of (ProfileImage x, ProfileImage y, HmmImage hmm, max_distance)."""
    arr = (HxPairJob * len(triples))()
    for k, (x, y, h, md) in enumerate(triples):
        arr[k].x = C.pointer(x.struct)
        arr[k].y = C.pointer(y.struct)
        arr[k].hmm = C.pointer(h.struct)
        arr[k].max_distance = md
    return arr


def init(device=0, table=None):
    from .hostmodel import lse_table
    t = np.ascontiguousarray(lse_table() if table is None else table, dtype=np.float64)
    _check(load().hx_init(device, _p(t, _f64p), t.size))


def shutdown():
    _check(load().hx_shutdown())


def slot_index(layout, i, j):
    """hx_layout cell addressing (include/historian_hip.h)."""
    sr = layout.strip_rows
    i = np.asarray(i, dtype=np.int64)
    j = np.asarray(j, dtype=np.int64)
    if layout.mirrored:
        i, j = layout.n_rows - 1 - i, layout.n_cols - 1 - j
    l = i % sr
    t = j + l
    return (i // sr) * layout.strip_stride + (t // 2) * (layout.block_stride or 2 * sr) + l * 2 + t % 2


class Batch:
    """n independent pair DPs resident on the device."""

    def __init__(self, triples, flags=0, device=None):
        """flags: HX_* bits.  This binding is what the parity tests drive, so WITHOUT a policy bit it asks for the bit-exact
        policy (HX_LSE_EXACT) - unlike the C ABI itself, whose default is the fastest policy with the reference's best
        paths (HX_LSE_TRUNC); pass HX_LSE_DEFAULT to get the library's default."""
        if flags & HX_LSE_DEFAULT:
            flags &= ~HX_LSE_DEFAULT
        elif not flags & (HX_LSE_TRUNC | HX_LSE_EXACT):
            flags |= HX_LSE_EXACT
        self._keep = triples
        self.n = len(triples)
        self._jobs = make_jobs(triples)
        self._h = C.c_void_p()
        if device is None:
            _check(load().hx_batch_create(self._jobs, self.n, flags, C.byref(self._h)))
        else:
            _check(load().hx_batch_create_on(device, self._jobs, self.n, flags, C.byref(self._h)))

    def close(self):
        if self._h:
            load().hx_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forward(self, stream=None):
        _check(load().hx_batch_forward(self._h, C.c_void_p(stream or 0)))

    def backward(self, stream=None):
        _check(load().hx_batch_backward(self._h, C.c_void_p(stream or 0)))

    def sync(self):
        _check(load().hx_batch_sync(self._h))

    def lp_end(self):
        out = np.empty(self.n)
        _check(load().hx_batch_lp_end(self._h, _p(out, _f64p)))
        return out

    def lp_start(self):
        out = np.empty(self.n)
        _check(load().hx_batch_lp_start(self._h, _p(out, _f64p)))
        return out

    def layout(self, job, which=0):
        l = HxLayout()
        _check(load().hx_batch_layout(self._h, job, which, C.byref(l)))
        return l

    def total_cells(self):
        return int(load().hx_batch_total_cells(self._h))

    def job_kernel(self, job):
        """(kernel class of the Forward fill, whether Backward runs the banded rotating-row sweep) of pair `job`."""
        c, s = C.c_int32(0), C.c_int32(0)
        _check(load().hx_batch_job_kernel(self._h, job, C.byref(c), C.byref(s)))
        return c.value, bool(s.value)

    def shared_wavefront_pairs(self):
        """pairs whose banded fill runs two pairs per wavefront (hx_band2.hip)"""
        n = load().hx_batch_shared_wavefront_pairs(self._h)
        if n < 0:
            _check(n)
        return n

    def relaunches(self):
        """fills repeated with one workgroup per pair after waves of a several-workgroups launch gave up"""
        return int(load().hx_batch_relaunches(self._h))

    def kernel_ms(self, which=0):
        ms = C.c_float()
        _check(load().hx_batch_last_kernel_ms(self._h, which, C.byref(ms)))
        return ms.value

    def read_matrix(self, job, which=0):
        """Dense [n_rows][n_cols][5] copy of a matrix (un-skewed on the host)."""
        l = self.layout(job, which)
        buf = np.empty(l.matrix_doubles)
        _check(load().hx_batch_read_matrix(self._h, job, which, _p(buf, _f64p)))
        if l.compressed:
            return self._dense_from_compressed(job, l, buf)
        ii, jj = np.meshgrid(np.arange(l.n_rows), np.arange(l.n_cols), indexing="ij")
        slot = slot_index(l, ii, jj)
        return np.stack([buf[s * l.plane_stride + slot] for s in range(5)], axis=-1)

    def indel_counts(self, job, branch_times):
        """hx_batch_indel_counts: branch_times = (l.t, r.t, l.insWait, l.delWait, r.insWait, r.delWait)
        -> dict ins, del, insExt, delExt, insTime, delTime"""
        tm = np.ascontiguousarray(branch_times, dtype=np.float64)
        out = np.zeros(6)
        _check(load().hx_batch_indel_counts(self._h, job, _p(tm, _f64p), _p(out, _f64p)))
        return dict(zip(("ins", "del", "insExt", "delExt", "insTime", "delTime"), out.tolist()))

    def strip_windows(self, job):
        l = self.layout(job)
        win = np.zeros((l.n_strips, 4), dtype=np.int32)
        bases = np.zeros((l.n_strips, 2), dtype=np.int64)
        _check(load().hx_batch_strip_windows(self._h, job, _p(win, _i32p), bases.ctypes.data_as(C.POINTER(C.c_int64))))
        return win, bases

    def _dense_from_compressed(self, job, l, buf):
        """HX_BAND_COMPRESSED: cells that are not stored read as -inf (hx_layout.compressed)."""
        win, bases = self.strip_windows(job)
        planes = buf.reshape(5, l.plane_stride)
        out = np.full((l.n_rows, l.n_cols, 5), -np.inf)
        sr = l.strip_rows
        ii, jj = np.meshgrid(np.arange(l.n_rows), np.arange(l.n_cols), indexing="ij")
        strip, lane = ii // sr, ii % sr
        t = jj + lane
        for w in range(2):
            lo, hi, base = win[strip, 2 * w], win[strip, 2 * w + 1], bases[strip, w]
            inside = (t >= lo) & (t < hi)
            slot = base + ((t - lo) // 2) * (2 * sr) + lane * 2 + t % 2
            for s in range(5):
                out[..., s][inside] = planes[s][slot[inside]]
        return out

    def read_cells(self, job, ij, which=0):
        ij = np.ascontiguousarray(ij, dtype=np.int32).reshape(-1, 2)
        out = np.empty((len(ij), 5))
        _check(load().hx_batch_read_cells(self._h, job, which, _p(ij, _i32p), len(ij), _p(out, _f64p)))
        return out

    def read_prepared(self, job):
        x, y, h, _ = self._keep[job]
        ca = h.alph_size * h.components
        subx, suby = np.empty((x.n_states, ca)), np.empty((y.n_states, ca))
        insx, rsx = np.empty(x.n_states), np.empty(x.n_states)
        insy, rsy = np.empty(y.n_states), np.empty(y.n_states)
        _check(load().hx_batch_read_prepared(self._h, job, _p(subx, _f64p), _p(suby, _f64p), _p(insx, _f64p),
                                             _p(rsx, _f64p), _p(insy, _f64p), _p(rsy, _f64p)))
        return dict(subx=subx, suby=suby, insx=insx, rootsubx=rsx, insy=insy, rootsuby=rsy)

    def posterior_scan(self, job, min_post_prob, cap=1 << 20):
        out = (HxCell * cap)()
        n = C.c_int64()
        _check(load().hx_batch_posterior_scan(self._h, job, min_post_prob, out, cap, C.byref(n)))
        k = min(n.value, cap)
        return n.value, [(out[i].xpos, out[i].ypos, out[i].state, out[i].log_post_prob) for i in range(k)]

    def best_trace(self, cap=None, raw=False):
        """ForwardMatrix::bestTrace() of every job, found on the device: a list (one entry per job) of
        [(xpos, ypos, state), ...] from the start cell to the END cell; None where lpEnd is -inf.
        raw=True returns the arrays (cells [n, cap, 3], n_cells [n]) instead."""
        if cap is None:
            cap = max(self.layout(k).n_rows + self.layout(k).n_cols for k in range(self.n)) + 4
        cells = np.zeros((self.n, cap, 3), dtype=np.int32)
        n_cells = np.zeros(self.n, dtype=np.int32)
        _check(load().hx_batch_best_trace(self._h, cells.ctypes.data_as(C.c_void_p), cap, _p(n_cells, _i32p)))
        if raw:
            return cells, n_cells
        out = []
        for k in range(self.n):
            if n_cells[k] == -1:
                out.append(None)
            elif n_cells[k] < 0:
                raise HxError(int(n_cells[k]), "traceback failure in job %d" % k)
            else:
                out.append([tuple(int(v) for v in c) for c in cells[k, :n_cells[k]]])
        return out


    def best_trace_ties(self):
        """Near-tie flags of the last best_trace() (hx_batch_best_trace_ties): one int per job."""
        flags = np.zeros(self.n, dtype=np.int32)
        _check(load().hx_batch_best_trace_ties(self._h, _p(flags, _i32p)))
        return flags

    def sample_traces(self, job, n_walks, uniforms, cap=None):
        """ForwardMatrix::sampleTrace, n_walks times in sequence, on the device (hx_batch_sample_traces): uniforms = the
        canonical draws of the caller's generator, in order.  -> (paths, draws): paths[w] = [(xpos, ypos, state), ...] from
        the start cell to the END cell, draws[w] = uniforms used up to and including walk w.  Raises HxError on a failed walk."""
        if cap is None:
            cap = self.layout(job).n_rows + self.layout(job).n_cols + 4
        u = np.ascontiguousarray(uniforms, dtype=np.float64)
        cells = np.zeros((n_walks, cap, 3), dtype=np.int32)
        n_cells = np.zeros(n_walks, dtype=np.int32)
        draws = np.zeros(n_walks, dtype=np.int64)
        _check(load().hx_batch_sample_traces(self._h, job, n_walks, u.ctypes.data_as(C.POINTER(C.c_double)), len(u),
                                             cells.ctypes.data_as(C.c_void_p), cap, _p(n_cells, _i32p),
                                             draws.ctypes.data_as(C.POINTER(C.c_int64))))
        out = []
        for w in range(n_walks):
            if n_cells[w] <= 0:
                raise HxError(int(n_cells[w]), "sampled walk %d of job %d failed (code %d)" % (w, job, n_cells[w]))
            out.append([tuple(int(v) for v in c) for c in cells[w, :n_cells[w]]])
        return out, [int(d) for d in draws]


class BranchBatch:
    """n independent per-branch pair DPs (Refiner::BranchMatrix / Sampler::BranchMatrix) resident on the device.
    jobs: list of (x_pwm [x_len][C][A], y_sub [y_len][C][A], y_emit [y_len], trans [3][4], x_env or None, y_env or None,
    max_distance)."""

    def __init__(self, jobs):
        self.n = len(jobs)
        self._keep = []
        arr = (HxBranchJob * self.n)()
        for k, (xp, ys, ye, trans, xenv, yenv, md) in enumerate(jobs):
            xp = np.ascontiguousarray(xp, dtype=np.float64)
            ys = np.ascontiguousarray(ys, dtype=np.float64)
            ye = np.ascontiguousarray(ye, dtype=np.float64)
            xe = None if xenv is None else np.ascontiguousarray(xenv, dtype=np.int32)
            yv = None if yenv is None else np.ascontiguousarray(yenv, dtype=np.int32)
            self._keep.append((xp, ys, ye, xe, yv))
            j = arr[k]
            j.x_len, j.y_len = xp.shape[0] if xp.ndim == 3 else 0, ys.shape[0] if ys.ndim == 3 else 0
            ref = xp if xp.ndim == 3 and xp.shape[0] else ys
            j.components, j.alphabet = (ref.shape[1], ref.shape[2]) if ref.ndim == 3 else (1, 1)
            j.x_pwm, j.y_sub, j.y_emit = _p(xp, _f64p), _p(ys, _f64p), _p(ye, _f64p)
            for s in range(3):
                for d in range(4):
                    j.trans[s][d] = trans[s][d]
            j.x_env = C.cast(None, _i32p) if xe is None else xe.ctypes.data_as(_i32p)
            j.y_env = C.cast(None, _i32p) if yv is None else yv.ctypes.data_as(_i32p)
            j.max_distance = md
        self._jobs = arr
        self.shapes = [(j.x_len + 1, j.y_len + 1) for j in arr]
        self._h = C.c_void_p()
        _check(load().hx_branch_batch_create(arr, self.n, C.byref(self._h)))

    def close(self):
        if self._h:
            load().hx_branch_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, viterbi=True, stream=None):
        _check(load().hx_branch_batch_run(self._h, 1 if viterbi else 0, C.c_void_p(stream or 0)))

    def lp_end(self):
        out = np.empty(self.n)
        _check(load().hx_branch_batch_results(self._h, _p(out, _f64p)))
        return out

    def read_matrix(self, job):
        out = np.empty(self.shapes[job] + (3,))
        _check(load().hx_branch_batch_read_matrix(self._h, job, _p(out, _f64p)))
        return out

    def total_cells(self):
        return int(load().hx_branch_batch_total_cells(self._h))

    def kernel_ms(self):
        ms = C.c_float()
        _check(load().hx_branch_batch_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value


class QuickBatch:
    """n independent guide-alignment Viterbi fills (QuickAlignMatrix) resident on the device.
    pairs: list of (x_tok, y_tok, alph_size, submat [A,A], scores [11], diagonals or None)."""

    SCORE_NAMES = ("m2m", "m2i", "m2d", "i2i", "i2m", "i2d", "d2d", "d2m", "gap_open", "gap_extend", "no_gap")

    def __init__(self, pairs):
        self.n = len(pairs)
        self._keep = []
        jobs = (HxQuickJob * self.n)()
        for k, (xt, yt, a, submat, scores, diags) in enumerate(pairs):
            xt = np.ascontiguousarray(xt, dtype=np.int32)
            yt = np.ascontiguousarray(yt, dtype=np.int32)
            sm = np.ascontiguousarray(submat, dtype=np.float64)
            dg = None if diags is None else np.ascontiguousarray(diags, dtype=np.int32)
            self._keep.append((xt, yt, sm, dg))
            j = jobs[k]
            j.x_tok, j.y_tok = _p(xt, _i32p), _p(yt, _i32p)
            j.x_len, j.y_len, j.alph_size = len(xt), len(yt), a
            j.submat = _p(sm, _f64p)
            j.diagonals = C.cast(None, _i32p) if dg is None else dg.ctypes.data_as(_i32p)
            j.n_diagonals = 0 if dg is None else len(dg)
            for s, v in enumerate(scores):
                j.scores[s] = v
        self._jobs = jobs
        self._h = C.c_void_p()
        _check(load().hx_quick_batch_create(jobs, self.n, C.byref(self._h)))

    def close(self):
        if self._h:
            load().hx_quick_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, stream=None):
        _check(load().hx_quick_batch_run(self._h, C.c_void_p(stream or 0)))

    def results(self):
        score = np.empty(self.n)
        xe, ye = np.empty(self.n, dtype=np.int32), np.empty(self.n, dtype=np.int32)
        _check(load().hx_quick_batch_results(self._h, _p(score, _f64p), _p(xe, _i32p), _p(ye, _i32p)))
        return score, xe, ye

    def layout(self, job):
        lay = HxLayout()
        _check(load().hx_quick_batch_layout(self._h, job, C.byref(lay)))
        return lay

    def read_matrix(self, job):
        """[(xlen+1), (ylen+1), 3] array indexed by the reference's (i, j): row / column 0 are -inf."""
        lay = self.layout(job)
        buf = np.empty(lay.matrix_doubles)
        _check(load().hx_quick_batch_read_matrix(self._h, job, _p(buf, _f64p)))
        i, j = np.meshgrid(np.arange(lay.n_rows), np.arange(lay.n_cols), indexing="ij")
        slot = slot_index(lay, i, j)
        out = np.full((lay.n_rows + 1, lay.n_cols + 1, 3), -np.inf)
        for s in range(3):
            out[1:, 1:, s] = buf[s * lay.plane_stride + slot]
        return out

    def total_cells(self):
        return int(load().hx_quick_batch_total_cells(self._h))

    def kernel_ms(self):
        ms = C.c_float()
        _check(load().hx_quick_batch_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value


def sumprod_columns(parent, ins_prob, log_cpt_weight, branch_sub, evec, evec_inv, esc, tokens, weight=None, want_root_post=False):
    """hx_sumprod_columns: parent [N]; ins_prob [C][A]; log_cpt_weight [C]; branch_sub [C][N][A][A]; evec, evec_inv [C][A][A]
    complex; esc [C][N][A][A] complex; tokens [n_cols][N] int8 (-1 wildcard, -2 gap); weight [n_cols] or None.
    Returns col_log_like [n_cols], root_counts [C][A], eigen_counts [C][A][A] complex, root_post [n_cols][A] or None."""
    ins_prob = np.ascontiguousarray(ins_prob, dtype=np.float64)
    c, a = ins_prob.shape
    parent = np.ascontiguousarray(parent, dtype=np.int32)
    n = parent.size
    tokens = np.ascontiguousarray(tokens, dtype=np.int8)
    if tokens.ndim != 2 or tokens.shape[1] != n:
        raise ValueError("tokens must be [n_cols][n_nodes]")
    n_cols = tokens.shape[0]
    parts = {}

    def real(name, arr, shape):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        if arr.shape != shape:
            raise ValueError("%s must have shape %s, not %s" % (name, shape, arr.shape))
        parts[name] = arr
        return _p(arr, _f64p)

    def split(name, arr, shape):
        arr = np.asarray(arr, dtype=np.complex128)
        return real(name + "_re", arr.real, shape), real(name + "_im", arr.imag, shape)
    m = HxSumprodModel()
    m.alph_size, m.components, m.n_nodes = a, c, n
    m.parent = _p(parent, _i32p)
    m.ins_prob = _p(ins_prob, _f64p)
    m.log_cpt_weight = real("log_cpt_weight", log_cpt_weight, (c,))
    m.branch_sub = real("branch_sub", branch_sub, (c, n, a, a))
    m.evec_re, m.evec_im = split("evec", evec, (c, a, a))
    m.evec_inv_re, m.evec_inv_im = split("evec_inv", evec_inv, (c, a, a))
    m.esc_re, m.esc_im = split("esc", esc, (c, n, a, a))
    w = None if weight is None else np.ascontiguousarray(weight, dtype=np.float64)
    if w is not None and w.shape != (n_cols,):
        raise ValueError("weight must be [n_cols]")
    cll = np.empty(n_cols)
    root = np.empty((c, a))
    ere, eim = np.empty((c, a, a)), np.empty((c, a, a))
    post = np.empty((n_cols, a)) if want_root_post else None
    _check(load().hx_sumprod_columns(C.byref(m), _p(tokens, C.POINTER(C.c_int8)), _p(w, _f64p) if w is not None else C.cast(None, _f64p),
                                     n_cols, _p(cll, _f64p), _p(root, _f64p), _p(ere, _f64p), _p(eim, _f64p),
                                     _p(post, _f64p) if post is not None else C.cast(None, _f64p), None))
    return cll, root, ere + 1j * eim, post


def sumprod_kernel_ms():
    ms = C.c_float()
    _check(load().hx_sumprod_last_kernel_ms(C.byref(ms)))
    return ms.value
