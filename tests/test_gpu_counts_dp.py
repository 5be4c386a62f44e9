"""Expected indel events of a pair DP on the device (hx_batch_indel_counts: BackwardMatrix::getCounts restricted to the
IndelCounts members, reference src/forward.cpp:1183-1214, 579-652) against the restatement of oracle/counts_dp_oracle.py,
which exhaustive path enumeration pins (tests/test_oracle_counts_dp.py).  Leaf pairs of a few residues (where the
enumeration itself is the yardstick), DNA and protein pairs of a few dozen to a few hundred residues, banded and unbanded."""
import math

import numpy as np
import pytest

from historian_amd import capi
from oracle import c_oracle, counts_dp_oracle as cd
from oracle import historian_oracle as ho
from tests import helpers as H

pytestmark = pytest.mark.gpu
G = "tests/golden/reference_data/"


@pytest.fixture(scope="module", autouse=True)
def engine():
    capi.init(0, c_oracle.table())
    yield
    capi.shutdown()


def _times(tm):
    return [tm[k] for k in ("l_t", "r_t", "l_ins_wait", "l_del_wait", "r_ins_wait", "r_del_wait")]


def test_tiny_pairs_against_path_enumeration():
    model = ho.RateModel.from_file(G + "testforward.jukescantor.json")
    for xs, ys, t_l, t_r in (("ac", "ag", .1, .2), ("acg", "ag", .3, .1), ("a", "cgt", .2, .2), ("acgt", "act", .05, .4)):
        hmm = ho.PairHMM(ho.ProbModel(model, t_l), ho.ProbModel(model, t_r), model.ins_prob)
        x = ho.Profile.from_seq(1, model.alphabet, xs, 1, "x")
        y = ho.Profile.from_seq(1, model.alphabet, ys, 2, "y")
        fwd = ho.ForwardMatrix(x, y, hmm, 0, ho.GuideAlignmentEnvelope(), fill=False)
        tm = cd.branch_times(model, t_l, t_r)
        b = capi.Batch([H.job_images(fwd)], capi.HX_KEEP_BACKWARD)
        b.forward()
        b.backward()
        got = b.indel_counts(0, _times(tm))
        b.close()
        fwd.fill()
        want = cd.brute_force_indel_counts(fwd, tm)
        for k in cd.KEYS:
            assert abs(got[k] - want[k]) <= 2e-4 * max(1., abs(want[k])), (xs, ys, k, got[k], want[k])


@pytest.mark.parametrize("flags", [0, capi.HX_LSE_FAST])
def test_leaf_pairs_against_the_restatement(flags):
    cases = [H.leaf_case(701, 40, 36), H.leaf_case(702, 90, 100, band=6), H.leaf_case(703, 150, 140, alphabet="arndcqeghilkmfpstwyv", jc=False),
             H.leaf_case(704, 70, 3), H.leaf_case(705, 200, 210, band=10)]
    imgs = [H.job_images(f) for f in cases]
    b = capi.Batch(imgs, capi.HX_KEEP_BACKWARD | flags)
    b.forward()
    b.backward()
    tm = dict(l_t=.2, r_t=.3, l_ins_wait=.09, l_del_wait=.08, r_ins_wait=.14, r_del_wait=.13)
    for k, f in enumerate(cases):
        got = b.indel_counts(k, _times(tm))
        f.fill()
        want = cd.get_indel_counts(ho.BackwardMatrix(f), tm)
        for key in cd.KEYS:
            # exact mode: the same Forward / Backward cells bit for bit, so only exp() and the order of the sum differ
            tol = 1e-9 if flags == 0 else 1e-6
            assert abs(got[key] - want[key]) <= tol * max(1., abs(want[key])), (k, key, got[key], want[key])
    with pytest.raises(capi.HxError):
        b.indel_counts(len(cases), _times(tm))
    b.close()
    nb = capi.Batch(imgs[:1])
    nb.forward()
    with pytest.raises(capi.HxError) as e:               # no Backward fill yet
        nb.indel_counts(0, _times(tm))
    assert e.value.code == -7
    nb.close()
