"""The Forward-Backward indel-count restatement (oracle/counts_dp_oracle.py: transitionEigenCounts + getCounts, indel
members) against exhaustive path enumeration on pairs of two to four residues: no reference fixture prints these
numbers, so the enumeration is what pins the restatement (the Forward and Backward fills it weights with are pinned
by the reference's golden files)."""
import math

import pytest

from oracle import counts_dp_oracle as cd
from oracle import historian_oracle as ho

G = "tests/golden/reference_data/"


@pytest.mark.parametrize("xs,ys,t_l,t_r", [("ac", "ag", .1, .2), ("acg", "ag", .3, .1), ("a", "cgt", .2, .2), ("acgt", "act", .05, .4)])
def test_posterior_weighted_counts_equal_the_expectation_over_all_paths(xs, ys, t_l, t_r):
    model = ho.RateModel.from_file(G + "testforward.jukescantor.json")
    hmm = ho.PairHMM(ho.ProbModel(model, t_l), ho.ProbModel(model, t_r), model.ins_prob)
    x = ho.Profile.from_seq(1, model.alphabet, xs, 1, "x")
    y = ho.Profile.from_seq(1, model.alphabet, ys, 2, "y")
    fwd = ho.ForwardMatrix(x, y, hmm, 0, ho.GuideAlignmentEnvelope())
    bwd = ho.BackwardMatrix(fwd)
    tm = cd.branch_times(model, t_l, t_r)
    got = cd.get_indel_counts(bwd, tm)
    want = cd.brute_force_indel_counts(fwd, tm)
    # the table log-sum-exp of the fills is accurate to ~1e-5 relative (logsumexp.txt), the enumeration is exact
    assert abs(want["lp"] - fwd.lp_end) < 1e-4
    for k in cd.KEYS:
        assert abs(got[k] - want[k]) <= 2e-4 * max(1., abs(want[k])), (k, got[k], want[k])
    # sanity: every path absorbs all residues - expected matches + deletions on the x side = len(x) etc. is not a count here,
    # but waiting times are positive and events non-negative
    assert all(got[k] >= 0 for k in cd.KEYS)
    assert got["insTime"] > 0 and got["delTime"] > 0


def _three_leaf_root(xs, y1, y2, t=.2):
    """root of ((y1, y2) inner, xs): x = leaf profile of xs, y = the profile of the inner node (every cell kept), on the tree
    node numbering 0 = y1, 1 = y2, 2 = inner, 3 = x leaf, 4 = root"""
    from oracle import sumprod_oracle as so
    model = ho.RateModel.from_file(G + "testforward.jukescantor.json")
    tree = so.Tree([2, 2, 4, 4, -1], [t, t, t, t, 0.], ["y1", "y2", "inner", "x", "root"])
    pm = ho.ProbModel(model, t)
    hmm = ho.PairHMM(pm, pm, model.ins_prob)
    l1 = ho.Profile.from_seq(1, model.alphabet, y1, 0, "y1")
    l2 = ho.Profile.from_seq(1, model.alphabet, y2, 1, "y2")
    inner = ho.ForwardMatrix(l1, l2, hmm, 2, ho.GuideAlignmentEnvelope()).best_profile()
    x = ho.Profile.from_seq(1, model.alphabet, xs, 3, "x")
    fwd = ho.ForwardMatrix(inner, x, hmm, 4, ho.GuideAlignmentEnvelope())
    return model, tree, fwd


@pytest.mark.parametrize("xs,y1,y2", [("ac", "ag", "a"), ("a", "ac", "gc")])
def test_posterior_weighted_substitution_counts_equal_the_expectation_over_all_paths(xs, y1, y2):
    import numpy as np
    from oracle import sumprod_oracle as so
    model, tree, fwd = _three_leaf_root(xs, y1, y2)
    bwd = ho.BackwardMatrix(fwd)
    sp = so.SumProduct(model, tree)
    root, eig = cd.get_subst_counts(bwd, sp)
    want_root, want_eig = cd.brute_force_subst_counts(fwd, sp)
    for cpt in range(sp.C):
        assert np.max(np.abs(root[cpt] - want_root[cpt])) <= 3e-4 * max(1., float(np.max(np.abs(want_root[cpt]))))
        assert np.max(np.abs(eig[cpt] - want_eig[cpt])) <= 3e-4 * max(1., float(np.max(np.abs(want_eig[cpt]))))
    # every path absorbs every residue once: the root counts add up to the number of columns a path has on average > 0
    assert sum(float(r.sum()) for r in root) > 0
