"""Device-side best-path traceback (hx_batch_best_trace) against the oracle's restatement of
ForwardMatrix::bestTrace (reference src/forward.cpp:278-302) walking the plain-C oracle's matrix: the paths must
be identical cell for cell (additions and comparisons only; ties resolved in CellCoords order)."""
import numpy as np
import pytest

from historian_amd import capi
from oracle import c_oracle
from oracle import historian_oracle as ho
from tests import helpers as H
from tests import recon_helpers as RH

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def engine():
    capi.init(0, c_oracle.table())
    yield
    capi.shutdown()


def oracle_best_trace(f):
    g = RH.ArrayForward(f.x, f.y, f.hmm, f.parent_row_index, f.envelope)
    if not g.lp_end > H.NEG_INF:
        return None
    return [tuple(c) for c in g.best_trace()]


def check_batch(cases, flags=0):
    b = capi.Batch([H.job_images(f) for f in cases], flags)
    b.forward()
    got = b.best_trace()
    b.close()
    for k, f in enumerate(cases):
        want = oracle_best_trace(f)
        assert got[k] == want, "job %d: device path differs from the oracle's" % k
        if want is not None:
            assert want[0][:2] == (0, 0) and want[-1] == (f.x_size - 1, f.y_size - 1, ho.EEE)
    return got


def test_leaf_pairs_including_empty_and_banded():
    cases = [H.leaf_case(1, 30, 34), H.leaf_case(2, 1, 1), H.leaf_case(3, 0, 5), H.leaf_case(4, 6, 0), H.leaf_case(5, 0, 0),
             H.leaf_case(7, 70, 66), H.leaf_case(203, 200, 90, band=12), H.leaf_case(204, 130, 150, band=3)]
    check_batch(cases)
    check_batch(cases, capi.HX_SPARSE_ENVELOPE)


def test_protein_and_mixture_leaf_pairs():
    aa = "arndcqeghilkmfpstwyv"
    check_batch([H.leaf_case(11, 40, 45, alphabet=aa, jc=False, tl=.3, tr=.2),
                 H.leaf_case(12, 33, 30, alphabet=aa, components=2, jc=False)])


def test_general_profiles_with_null_states_bands_and_high_in_degree():
    cases = [H.dag_case(31), H.dag_case(32), H.dag_case(41, band=0), H.dag_case(42, band=1), H.dag_case(43, band=3),
             H.dag_case(51, n=10, components=2), H.dag_case(61, n=9, keep_all=True), H.dag_case(67, n=9, band=2, keep_all=True),
             H.dag_case(81, n=40, samples=25), H.dag_case(83, n=60, samples=30, band=4),
             H.dag_case(72, n=150, band=6, samples=3)]
    check_batch(cases)
    check_batch(cases, capi.HX_FORCE_GENERIC)      # no emission plane: the kernel evaluates the emission itself


def test_mixed_batch_and_zero_likelihood_job():
    cases = [H.leaf_case(7, 70, 66), H.dag_case(31), H.dag_case(45, band=0), H.leaf_case(81, 500, 480)]
    got = check_batch(cases)
    lp = RH.ArrayForward(cases[2].x, cases[2].y, cases[2].hmm, cases[2].parent_row_index, cases[2].envelope).lp_end
    assert (got[2] is None) == (not lp > H.NEG_INF)


def test_null_state_profile_of_testnullforward():
    G = "tests/golden/reference_data/"
    rates = ho.RateModel.from_file(G + "testforward.nosub.json")
    hmm = ho.PairHMM(ho.ProbModel(rates, 1), ho.ProbModel(rates, 1), rates.ins_prob)
    xp = ho.Profile.from_seq(1, rates.alphabet, "acg", 1, "x")
    yp = ho.Profile.from_seq(1, rates.alphabet, "cag", 2, "y")
    xp.state[2].lp_absorb = []
    yp.state[1].lp_absorb = []
    f = ho.ForwardMatrix(xp, yp, hmm, 0, ho.GuideAlignmentEnvelope(), fill=False)
    got = check_batch([f])
    f.fill()
    assert got[0] == [tuple(c) for c in f.best_trace()]


def test_path_buffer_too_small_is_an_error():
    b = capi.Batch([H.job_images(H.leaf_case(1, 30, 34))])
    b.forward()
    with pytest.raises(capi.HxError):
        b.best_trace(cap=5)
    b.close()


def test_fast_mode_paths_are_valid_alignments():
    # the fast log-sum-exp changes cell values in the 6th digit, so the best path may differ near ties; it must still
    # be a legal path from the start cell to the END cell that consumes every state index monotonically
    cases = [H.leaf_case(7, 70, 66), H.leaf_case(81, 500, 480)]
    b = capi.Batch([H.job_images(f) for f in cases], capi.HX_LSE_FAST)
    b.forward()
    got = b.best_trace()
    b.close()
    for f, path in zip(cases, got):
        assert path[0][:2] == (0, 0) and path[-1] == (f.x_size - 1, f.y_size - 1, ho.EEE)
        for a, c in zip(path, path[1:]):
            assert (c[0] - a[0], c[1] - a[1]) in ((1, 1), (1, 0), (0, 1))
