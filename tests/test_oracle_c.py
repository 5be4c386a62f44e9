"""The plain-C oracle fill (oracle/oracle_fill.c) must agree bit for bit with the
pinned Python oracle: every cell, lpEnd/lpStart and the prepared per-state vectors."""
import numpy as np
import pytest

from oracle import c_oracle
from oracle import historian_oracle as ho
from oracle import ref_mains as rm
from tests import helpers as H


def check(fwd, backward=True):
    x, y, hmm, md = H.job_images(fwd)
    fwd.fill()
    got = c_oracle.forward(x, y, hmm, md)
    H.assert_same_bits(got["cells"], H.oracle_dense(fwd), "forward cells")
    H.assert_same_bits([got["lp_end"]], [fwd.lp_end], "lpEnd")
    H.assert_same_bits(got["insx"], fwd.insx, "insx")
    H.assert_same_bits(got["rootsubx"], fwd.rootsubx, "rootsubx")
    H.assert_same_bits(got["insy"], fwd.insy, "insy")
    H.assert_same_bits(got["rootsuby"], fwd.rootsuby, "rootsuby")
    for i, row in enumerate(fwd.subx):
        if row:
            H.assert_same_bits(got["subx"][i], np.array(row).ravel(), "subx")
    if backward:
        back = ho.BackwardMatrix(fwd)
        gb = c_oracle.backward(x, y, hmm, md)
        H.assert_same_bits(gb["cells"], H.oracle_dense(back), "backward cells")
        H.assert_same_bits([gb["lp_start"]], [back.lp_start()], "lpStart")
    return fwd


def test_table_matches_python():
    H.assert_same_bits(c_oracle.table(), ho.LSE_TABLE, "lse table")


@pytest.mark.parametrize("seed,lx,ly", [(1, 7, 9), (2, 1, 1), (3, 12, 5), (4, 0, 3), (5, 3, 0), (6, 0, 0)])
def test_leaf_dna_jc(seed, lx, ly):
    check(H.leaf_case(seed, lx, ly))


@pytest.mark.parametrize("seed", [11, 12])
def test_leaf_protein_general_model(seed):
    check(H.leaf_case(seed, 9, 11, alphabet="arndcqeghilkmfpstwyv", jc=False, tl=.3, tr=.2))


def test_leaf_mixture_two_components():
    check(H.leaf_case(21, 8, 8, alphabet="ACGT", components=2, jc=False))


@pytest.mark.parametrize("seed", [31, 32, 33, 34])
def test_dag_profiles_unbanded(seed):
    f = check(H.dag_case(seed))
    assert f.lp_end > H.NEG_INF


@pytest.mark.parametrize("seed,band", [(41, 0), (42, 1), (43, 3), (44, 2)])
def test_dag_profiles_banded(seed, band):
    check(H.dag_case(seed, band=band))


@pytest.mark.parametrize("seed,band", [(61, None), (67, 2), (68, 1)])
def test_dag_profiles_with_many_null_states(seed, band):
    f = H.dag_case(seed, n=9, band=band, keep_all=True)
    assert sum(s.is_null() for p in (f.x, f.y) for s in p.state) > 5
    check(f)


def test_dag_mixture():
    check(H.dag_case(51, n=10, components=2))


def test_null_state_case_of_testnullforward():
    # reference t/testnullforward.cpp:28-41
    G = "tests/golden/reference_data/"
    rates = ho.RateModel.from_file(G + "testforward.nosub.json")
    hmm = ho.PairHMM(ho.ProbModel(rates, 1), ho.ProbModel(rates, 1), rates.ins_prob)
    xp = ho.Profile.from_seq(1, rates.alphabet, "acg", 1, "x")
    yp = ho.Profile.from_seq(1, rates.alphabet, "cag", 2, "y")
    xp.state[2].lp_absorb = []
    yp.state[1].lp_absorb = []
    check(ho.ForwardMatrix(xp, yp, hmm, 0, ho.GuideAlignmentEnvelope(), fill=False), backward=False)


def test_quickalign_c_fill_equals_python_restatement():
    import random
    from oracle import quickalign_oracle as q
    G = "tests/golden/reference_data/"
    model = ho.RateModel.from_file(G + "testamino.json")
    model.sub_rate = [m.tolist() for m in model.sub_rate]
    sc = q.QuickAlignScores(model, 1.0)
    rng = random.Random(3)
    A = model.alphabet
    for trial, (lx, ly, sparse) in enumerate([(33, 36, False), (60, 80, True), (120, 90, True), (5, 3, False),
                                              (1, 1, False), (70, 70, True)]):
        x = "".join(rng.choice(A) for _ in range(lx))
        y = "".join((c if rng.random() > .15 else rng.choice(A)) for c in x[:ly])
        y += "".join(rng.choice(A) for _ in range(max(0, ly - len(y))))
        if trial == 1:
            x = x[:20] + "x" + x[21:]          # a character outside the alphabet: emission score 0
        env = q.DiagonalEnvelope(x, y)
        if sparse:
            env.init_sparse(q.KmerIndex(y, A, 3), band_size=8, kmer_threshold=1)
        else:
            env.init_full()
        mx = q.QuickAlignMatrix(env, model, 1.0, scores=sc)
        r = c_oracle.quickalign(q.tokens(x, A), q.tokens(y, A), len(A), sc.submat, sc, env.diagonals if sparse else None)
        assert (r["score"], r["x_end"], r["y_end"]) == (mx.end, mx.x_end, mx.y_end)
        for (i, j), c in mx.cells.items():
            assert all(r["cells"][i, j, k] == c[k] for k in range(3))
        assert int(np.isfinite(r["cells"]).any(axis=2).sum()) == len(mx.cells)


def test_pod_image_traceback_equals_the_object_based_restatement():
    # oracle/trace_oracle.py (used by bench.py and the full-size GPU tests) against historian_oracle's best_trace
    from oracle import trace_oracle
    for f in [H.leaf_case(7, 70, 66), H.leaf_case(3, 0, 5), H.leaf_case(203, 90, 60, band=6), H.dag_case(31), H.dag_case(43, band=3),
              H.dag_case(61, n=9, keep_all=True), H.dag_case(81, n=30, samples=20)]:
        x, y, hmm, md = H.job_images(f)
        fwd = c_oracle.forward(x, y, hmm, md)
        f.fill()
        if f.lp_end > H.NEG_INF:
            assert trace_oracle.best_trace(x, y, hmm, md, fwd) == [tuple(c) for c in f.best_trace()]


def test_map_storage_variant_gives_the_same_likelihood():
    # oracle_fill_map.cpp: the same fill over a std::map per row (the reference's storage), used as a CPU baseline
    for f in [H.leaf_case(7, 70, 66), H.leaf_case(203, 90, 60, band=6), H.dag_case(31), H.dag_case(43, band=3)]:
        x, y, hmm, md = H.job_images(f)
        H.assert_same_bits([c_oracle.forward_map(x, y, hmm, md)], [c_oracle.forward(x, y, hmm, md)["lp_end"]], "lpEnd")
