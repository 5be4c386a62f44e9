"""Pins oracle/branch_oracle.py (Refiner::BranchMatrix / Sampler::BranchMatrix, reference
src/refiner.cpp:10-104, src/sampler.cpp:1005-1084) by exhaustive enumeration: no reference fixture holds a branch matrix
(every `recon` test of the reference's Makefile passes -norefine), so on tiny pairs every state path is listed and scored in
plain floating point - the Viterbi score must be the best path's, best() must return a path that attains it, the Forward
score must be the log of the summed path probabilities (to the accuracy of the reference's table log_sum_exp)."""
import math
import random

import pytest

from oracle import branch_oracle as bo
from oracle.historian_oracle import NEG_INF


def random_case(seed, nx, ny, C=1, A=4, band=None):
    rng = random.Random(seed)
    def pwm(n):
        return [[[math.log(rng.uniform(.01, 1.)) for _ in range(A)] for _ in range(C)] for _ in range(n)]
    x, y = pwm(nx), pwm(ny)
    log_sub = []
    for _ in range(C):
        m = [[rng.uniform(.05, 1.) for _ in range(A)] for _ in range(A)]
        log_sub.append([[math.log(v / sum(row)) for v in row] for row in m])
    log_ins = [[math.log(1. / A)] * A for _ in range(C)]
    log_w = [math.log(1. / C)] * C
    T = bo.trans_scores(rng.uniform(.01, .3), rng.uniform(.01, .3), rng.uniform(.3, .9), rng.uniform(.3, .9))
    xe = ye = None
    md = -1
    if band is not None:
        xe = [0] + sorted(rng.randint(0, max(nx, ny)) for _ in range(nx))
        ye = [0] + sorted(rng.randint(0, max(nx, ny)) for _ in range(ny))
        md = band
    return x, bo.pre_multiply(y, log_sub), bo.calc_ins_probs(y, log_ins, log_w), T, xe, ye, md


def inside(bm, path):
    i = j = 0
    for s in path:
        pi, pj = i, j
        if s == bo.MATCH:
            i, j = i + 1, j + 1
        elif s == bo.INSERT:
            j += 1
        else:
            i += 1
        # the fill takes a move only when both its source and its destination cell are inside the envelope
        if not (bm.in_envelope(pi, pj) and bm.in_envelope(i, j)):
            return False
    return True


@pytest.mark.parametrize("seed,nx,ny,C,band", [(1, 2, 2, 1, None), (2, 3, 2, 1, None), (3, 2, 4, 2, None), (4, 3, 3, 1, 0),
                                               (5, 4, 3, 1, 1), (6, 1, 3, 1, None), (7, 0, 2, 1, None), (8, 4, 4, 2, 1)])
def test_branch_matrix_against_every_path(seed, nx, ny, C, band):
    x, ysub, yemit, T, xe, ye, md = random_case(seed, nx, ny, C=C, band=band)
    vit = bo.BranchMatrix(x, ysub, yemit, T, xe, ye, md, viterbi=True)
    fwd = bo.BranchMatrix(x, ysub, yemit, T, xe, ye, md, viterbi=False)
    paths = [p for p in bo.enumerate_paths(nx, ny) if inside(vit, p)]
    lps = [bo.path_log_prob(vit, p) for p in paths]
    finite = [lp for lp in lps if lp > NEG_INF]
    if not finite:
        assert vit.lp_end == NEG_INF and fwd.lp_end == NEG_INF
        return
    best = max(finite)
    assert abs(vit.lp_end - best) <= 1e-12 * max(1., abs(best))
    total = best + math.log(sum(math.exp(lp - best) for lp in finite))
    assert abs(fwd.lp_end - total) <= 1e-3
    # the traceback's alignment is a best path
    xrow, yrow = vit.best()
    assert sum(xrow) == nx and sum(yrow) == ny
    states = [bo.MATCH if a and b else (bo.DELETE if a else bo.INSERT) for a, b in zip(xrow, yrow)]
    assert abs(bo.path_log_prob(vit, states) - best) <= 1e-12 * max(1., abs(best))


def test_delete_never_precedes_insert():
    # ProbModel::transProb(Delete, Insert) = 0 (src/model.cpp:435-436)
    T = bo.trans_scores(.1, .1, .5, .5)
    assert T[bo.DELETE][bo.INSERT] == NEG_INF and all(v > NEG_INF for v in T[bo.MATCH])
