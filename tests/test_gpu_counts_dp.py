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


def test_substitution_counts_of_a_root_pair_against_the_restatement():
    # The substitution half of BackwardMatrix::getCounts (src/forward.cpp:897-973, 1183-1214) through the device: Forward and
    # Backward fills of the root pair, posterior weights, the cells' alignment columns grouped and sent through the sum-product
    # kernel in one launch (historian_amd/counts.dp_posterior_substitution_counts) - against oracle/counts_dp_oracle, which
    # evaluates one column per cell as the reference does and is pinned by enumeration (tests/test_oracle_counts_dp.py).
    # A four-leaf tree ((a, b) u, (c, d) v) root: both children of the root are internal-node profiles (null states included).
    from historian_amd import counts, hostmodel
    from oracle import sumprod_oracle as so
    model = ho.RateModel.from_file(G + "testforward.jukescantor.json")
    t = .15
    names = ["a", "b", "u", "c", "d", "v", "root"]
    parent = [2, 2, 6, 5, 5, 6, -1]
    tree = so.Tree(parent, [t] * 6 + [0.], names)
    pm = ho.ProbModel(model, t)
    hmm = ho.PairHMM(pm, pm, model.ins_prob)
    seqs = {0: "acgtacgta", 1: "acgaacta", 3: "aggtacgt", 4: "acgtaagtt"}
    leaf = {n: ho.Profile.from_seq(1, model.alphabet, s, n, names[n]) for n, s in seqs.items()}
    gen = ho.MT19937(5489)
    u = ho.ForwardMatrix(leaf[0], leaf[1], hmm, 2, ho.GuideAlignmentEnvelope()).sample_profile(gen, 6, 0, ho.DPMatrix.CollapseChains | ho.DPMatrix.IncludeBestTrace)
    v = ho.ForwardMatrix(leaf[3], leaf[4], hmm, 5, ho.GuideAlignmentEnvelope()).sample_profile(gen, 6, 0, ho.DPMatrix.CollapseChains | ho.DPMatrix.IncludeBestTrace)
    fwd = ho.ForwardMatrix(u, v, hmm, 6, ho.GuideAlignmentEnvelope())
    bwd = ho.BackwardMatrix(fwd)
    sp = so.SumProduct(model, tree)
    want_root, want_eig = cd.get_subst_counts(bwd, sp)

    def columns(prof):
        out = np.full((prof.size(), len(parent)), counts.GAP, dtype=np.int8)
        for s in range(prof.size()):
            for row, ch in cd.profile_align_column(prof, s).items():
                out[s, row] = model.alphabet.lower().find(ch.lower()) if ch != cd.WILDCARD else counts.WILD
        return out
    hm = hostmodel.RateModel.load(G + "testforward.jukescantor.json")
    counter = counts.ColumnCounter(hm, parent, [t] * 6 + [0.], branch_sub=[[np.asarray(m) for m in pm.sub_mat]] * 7)
    b = capi.Batch([H.job_images(fwd)], capi.HX_KEEP_BACKWARD)
    b.forward()
    b.backward()
    got = counts.dp_posterior_substitution_counts(b, 0, columns(u), columns(v), [s.is_null() for s in u.state], [s.is_null() for s in v.state], 6, counter)
    b.close()
    for cpt in range(sp.C):
        assert np.max(np.abs(got["root_counts"][cpt] - want_root[cpt])) <= 1e-9 * max(1., float(np.max(np.abs(want_root[cpt]))))
        assert np.max(np.abs(got["eigen_counts"][cpt] - want_eig[cpt])) <= 1e-8 * max(1., float(np.max(np.abs(want_eig[cpt]))))
    # the columns' residues: four leaves are certain in every column they take part in, so the root counts sum to the expected
    # number of columns with a root
    assert float(np.sum(got["root_counts"])) > 0
