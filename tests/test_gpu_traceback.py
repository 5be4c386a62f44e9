"""Traceback parity (north_star: "traceback alignments bit-identical").

The host-side traceback of the oracle (reference src/forward.cpp:225-314) is run on
matrices filled by the GPU -- exact mode and fast mode -- and on the CPU oracle's matrix;
the best trace, sampled traces (same mt19937 seed) and the resulting alignment paths must
be identical, cell for cell."""
import numpy as np
import pytest

from historian_amd import capi
from oracle import c_oracle
from oracle import historian_oracle as ho
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def engine():
    capi.init(0, c_oracle.table())
    yield
    capi.shutdown()


class ArrayForward(ho.ForwardMatrix):
    """oracle ForwardMatrix whose cell storage is a dense array filled elsewhere."""

    def __init__(self, proto, cells, lp_end):
        # share the (expensive, pure-Python) prepared vectors of the prototype
        self.__dict__.update(proto.__dict__)
        self.arr = cells
        self.lp_end = lp_end

    def cell(self, i, j, s):
        if i >= self.arr.shape[0] or j >= self.arr.shape[1]:
            return H.NEG_INF
        return float(self.arr[i, j, s])

    def xy_cell(self, i, j):
        if i >= self.arr.shape[0] or j >= self.arr.shape[1]:
            return ho._EMPTY_CELL
        return [float(v) for v in self.arr[i, j]]


def traces(fm, seeds=(5489, 7)):
    out = [fm.best_trace()]
    for s in seeds:
        out.append(fm.sample_trace(ho.MT19937(s)))
    return out


def check_case(f, full_align_path=True):
    img = H.job_images(f)
    be, bf = capi.Batch([img]), capi.Batch([img], capi.HX_LSE_FAST)
    be.forward()
    bf.forward()
    want = c_oracle.forward(*img)
    ref = ArrayForward(f, want["cells"], want["lp_end"])
    exact = ArrayForward(f, be.read_matrix(0), float(be.lp_end()[0]))
    fast = ArrayForward(f, bf.read_matrix(0), float(bf.lp_end()[0]))
    # forward log-likelihood tolerance of north_star: 1e-4 relative
    assert exact.lp_end == ref.lp_end
    assert abs(fast.lp_end - ref.lp_end) <= 1e-4 * abs(ref.lp_end)
    t_ref = traces(ref)
    assert traces(exact) == t_ref
    assert traces(fast) == t_ref
    if full_align_path:
        assert fast.trace_align_path(t_ref[0]) == ref.trace_align_path(t_ref[0])
    be.close()
    bf.close()
    return len(t_ref[0])


def test_traceback_identical_small_and_symmetric_branches():
    # equal branch lengths create exact ties between IMD->IDM and IDM->IMD orderings, which
    # bestCell breaks by map order (reference src/forward.cpp:245-255)
    for f in (H.leaf_case(301, 60, 55, tl=.1, tr=.1), H.leaf_case(302, 90, 100, tl=.3, tr=.3),
              H.leaf_case(303, 120, 70, alphabet="arndcqeghilkmfpstwyv", jc=False, tl=.2, tr=.2),
              H.leaf_case(304, 200, 210, band=8)):
        assert check_case(f) > 10


def test_traceback_identical_on_dag_profiles():
    for f in (H.dag_case(31), H.dag_case(43, band=3), H.dag_case(67, n=9, band=2, keep_all=True)):
        check_case(f)


def test_traceback_identical_protein_600():
    f = H.leaf_case(305, 600, 590, alphabet="arndcqeghilkmfpstwyv", jc=False, tl=.2, tr=.3)
    assert check_case(f, full_align_path=False) > 600


def test_near_tie_flags_of_the_best_path_walk():
    # hx_batch_best_trace_ties: per job, whether the walk met a step where another source cell came within 1e-9 (relative) of the
    # winner - where the choice hangs on the last bits of the arithmetic.  An internal-node pair with two equally probable routes
    # (found by tools/sweep_dag_trace_identity.py 240 21: the fast policy's path parts from the exact policy's there) is flagged
    # by the walks of BOTH fills; the exact fill's path is the oracle's; before any walk the call is refused.
    f = H.dag_case(544494, n=150, samples=3)
    img = H.job_images(f)
    paths = {}
    for name, flags in (("exact", capi.HX_LSE_EXACT), ("fast", capi.HX_LSE_FAST)):
        b = capi.Batch([img], flags)
        b.forward()
        with pytest.raises(capi.HxError):
            b.best_trace_ties()
        paths[name] = b.best_trace()[0]
        ties = b.best_trace_ties()
        assert ties.shape == (1,) and ties[0] == 1, (name, ties)
        b.close()
    assert paths["exact"][0] == (0, 0, 0) and paths["exact"][-1][2] == 5
    assert len(paths["fast"]) == len(paths["exact"])
