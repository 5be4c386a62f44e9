"""Next row N4 on the device: per-branch pair DPs (hx_branch.hip, C ABI hx_branch_batch_*) against oracle/branch_oracle.py -
Refiner::BranchMatrix (max-plus) and Sampler::BranchMatrix (the reference's log_sum_exp), reference src/refiner.cpp:10-60,
src/sampler.cpp:1034-1084.  Every cell and lpEnd compared as uint64: both forms are bit-identical to the restatement (max-plus
trivially, the sums because the kernel applies the reference's table operator in the reference's nesting)."""
import math
import random

import numpy as np
import pytest

from historian_amd import capi
from oracle import branch_oracle as bo
from oracle import c_oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def engine():
    capi.init(0, c_oracle.table())
    yield
    capi.shutdown()


def random_branch(seed, nx, ny, C=1, A=4, band=None, one_hot=False):
    rng = random.Random(seed)

    def pwm(n):
        if one_hot:      # leaf-like columns: one residue certain
            rows = []
            for _ in range(n):
                k = rng.randrange(A)
                rows.append([[0. if a == k else -math.inf for a in range(A)] for _ in range(C)])
            return rows
        return [[[math.log(rng.uniform(.01, 1.)) for _ in range(A)] for _ in range(C)] for _ in range(n)]
    x, y = pwm(nx), pwm(ny)
    log_sub = []
    for _ in range(C):
        m = [[rng.uniform(.05, 1.) + (3. if i == j else 0.) for j in range(A)] for i in range(A)]
        log_sub.append([[math.log(v / sum(row)) for v in row] for row in m])
    log_ins = [[math.log(1. / A)] * A for _ in range(C)]
    log_w = [math.log(1. / C)] * C
    T = bo.trans_scores(rng.uniform(.01, .2), rng.uniform(.01, .2), rng.uniform(.3, .9), rng.uniform(.3, .9))
    xe = ye = None
    md = -1
    if band is not None:
        # envelope coordinates as a guide alignment gives them: non-decreasing match counts along either sequence
        xe = np.concatenate([[0], np.cumsum([rng.random() < .9 for _ in range(nx)])]).astype(np.int32)
        ye = np.concatenate([[0], np.cumsum([rng.random() < .9 for _ in range(ny)])]).astype(np.int32)
        md = band
    return x, bo.pre_multiply(y, log_sub), bo.calc_ins_probs(y, log_ins, log_w), T, xe, ye, md


def as_job(case):
    x, ysub, yemit, T, xe, ye, md = case
    C = len(ysub[0]) if ysub else (len(x[0]) if x else 1)
    A = len(ysub[0][0]) if ysub else (len(x[0][0]) if x else 1)
    return (np.array(x, dtype=float).reshape(len(x), C, A), np.array(ysub, dtype=float).reshape(len(ysub), C, A), np.array(yemit, dtype=float),
            T, xe, ye, md)


def dense(bm):
    out = np.full((bm.x_size, bm.y_size, 3), -np.inf)
    for (i, j), c in bm.cells.items():
        out[i, j] = c
    return out


CASES = [(11, 5, 7, 1, 4, None, False), (12, 70, 66, 1, 4, None, False), (13, 130, 90, 2, 4, None, False), (14, 64, 65, 1, 20, None, False),
         (15, 200, 180, 1, 4, 6, False), (16, 90, 140, 1, 4, 0, False), (17, 1, 1, 1, 4, None, False), (18, 0, 3, 1, 4, None, False),
         (19, 150, 150, 1, 20, 10, True), (20, 63, 129, 1, 4, 3, True)]


@pytest.mark.parametrize("viterbi", [True, False])
def test_branch_matrices_bit_for_bit(viterbi):
    cases = [random_branch(*c) for c in CASES]
    b = capi.BranchBatch([as_job(c) for c in cases])
    b.run(viterbi=viterbi)
    lp = b.lp_end()
    for k, case in enumerate(cases):
        x, ysub, yemit, T, xe, ye, md = case
        want = bo.BranchMatrix(x, ysub, yemit, T, None if xe is None else list(xe), None if ye is None else list(ye), md, viterbi=viterbi)
        H.assert_same_bits(b.read_matrix(k), dense(want), "job %d cells (%s)" % (k, "viterbi" if viterbi else "forward"))
        H.assert_same_bits([lp[k]], [want.lp_end], "job %d lpEnd" % k)
    assert b.total_cells() == sum((c[1] + 1) * (c[2] + 1) for c in CASES)
    assert b.kernel_ms() > 0
    b.close()


@pytest.mark.parametrize("waves, windows", [(1, True), (3, True), (16, True), (2, False), (16, False)])
def test_strips_dealt_to_wavefronts_and_band_windows(monkeypatch, waves, windows):
    # a branch's strips on 1 ... 16 wavefronts (the last row of a strip handed to the one below through the matrix), banded
    # strips swept by their step windows or in full: the same bits either way.  Rows 300-400 are five to seven strips, with
    # envelope coordinates that advance unevenly (runs of gaps), a band of 0 and a band wider than a strip.
    monkeypatch.setenv("HX_BRANCH_WAVES", str(waves))
    if not windows:
        monkeypatch.setenv("HX_BRANCH_NO_WINDOWS", "1")
    cases = [random_branch(51, 300, 330, 1, 4, 5, False), random_branch(52, 400, 290, 1, 4, 0, True), random_branch(53, 321, 321, 2, 4, 70, False),
             random_branch(54, 257, 300, 1, 4, None, False), random_branch(55, 129, 64, 1, 4, 2, True)]
    b = capi.BranchBatch([as_job(c) for c in cases])
    for viterbi in (True, False):
        b.run(viterbi=viterbi)
        lp = b.lp_end()
        for k, case in enumerate(cases):
            x, ysub, yemit, T, xe, ye, md = case
            want = bo.BranchMatrix(x, ysub, yemit, T, None if xe is None else list(xe), None if ye is None else list(ye), md, viterbi=viterbi)
            H.assert_same_bits(b.read_matrix(k), dense(want), "job %d cells (%s, %d wavefronts)" % (k, "viterbi" if viterbi else "forward", waves))
            H.assert_same_bits([lp[k]], [want.lp_end], "job %d lpEnd" % k)
    b.close()


def test_refiner_traceback_over_the_device_matrix():
    # Refiner::BranchMatrix::best (src/refiner.cpp:62-104) walks O(path) cells: over the device's matrix it finds the
    # alignment the restatement finds over its own
    case = random_branch(31, 120, 110, 1, 20, 8, True)
    x, ysub, yemit, T, xe, ye, md = case
    b = capi.BranchBatch([as_job(case)])
    b.run(viterbi=True)
    got = b.read_matrix(0)
    want = bo.BranchMatrix(x, ysub, yemit, T, list(xe), list(ye), md, viterbi=True)
    path = want.best()
    dev = bo.BranchMatrix.__new__(bo.BranchMatrix)
    dev.__dict__.update(want.__dict__)
    dev.cells = {(i, j): tuple(got[i, j]) for i in range(got.shape[0]) for j in range(got.shape[1]) if np.isfinite(got[i, j]).any()}
    assert dev.best() == path and sum(path[0]) == 120 and sum(path[1]) == 110
    b.close()


def test_refused_arguments():
    case = as_job(random_branch(41, 4, 4))
    with pytest.raises(capi.HxError):
        capi.BranchBatch([])
    bad = case[:4] + (None, None, 3)          # a band without envelope coordinates
    with pytest.raises(capi.HxError):
        capi.BranchBatch([bad])
    b = capi.BranchBatch([case])
    with pytest.raises(capi.HxError):
        b.lp_end()                            # before run
    b.close()
