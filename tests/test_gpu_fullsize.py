"""BASELINE.json-size cases, checked through size-independent properties plus the CPU oracle where
it finishes in seconds: a 2x2000-residue protein pair (configs[3] unit), a 2x1000 DNA pair under
JC69 (configs[1]) and a 4-component mixture with 5000-column profiles (configs[4] shape)."""
import numpy as np
import pytest

from historian_amd import capi
from oracle import c_oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu
AA = "arndcqeghilkmfpstwyv"


@pytest.fixture(scope="module", autouse=True)
def engine():
    capi.init(0, c_oracle.table())
    yield
    capi.shutdown()


def fills(f, backward=True):
    img = H.job_images(f)
    be, bf = capi.Batch([img]), capi.Batch([img], capi.HX_LSE_FAST)
    be.forward()
    bf.forward()
    out = dict(img=img, lp_exact=float(be.lp_end()[0]), lp_fast=float(bf.lp_end()[0]))
    if backward:
        be.backward()
        out["lp_start"] = float(be.lp_start()[0])
    out["be"], out["bf"] = be, bf
    return out


def test_protein_2x2000_pair():
    f = H.leaf_case(401, 2000, 2000, alphabet=AA, jc=False, tl=.2, tr=.3)
    r = fills(f)
    want = c_oracle.forward(*r["img"])
    assert r["lp_exact"] == want["lp_end"]                                   # bit-identical
    assert abs(r["lp_fast"] - want["lp_end"]) <= 1e-4 * abs(want["lp_end"])   # north_star tolerance
    assert abs(r["lp_fast"] - want["lp_end"]) <= 1e-9 * abs(want["lp_end"])   # what it actually achieves
    # Forward == Backward.  The two recursions drop different terms under the reference's
    # |a-b| >= 10 truncation, so they agree to ~1e-7 relative, not to rounding; the reference
    # itself only checks gsl_fcmp(..., 0.01) (src/forward.cpp:9,1091).
    assert abs(r["lp_start"] - r["lp_exact"]) <= 1e-5 * abs(r["lp_exact"])
    # spot cells against the oracle, bit for bit
    ij = np.array([[0, 0], [1, 1], [777, 801], [1999, 2000], [2000, 2000], [2000, 1], [64, 63], [63, 64], [1024, 1023]])
    got = r["be"].read_cells(0, ij)
    for k, (i, j) in enumerate(ij):
        H.assert_same_bits(got[k], want["cells"][i, j], "cell (%d,%d)" % (i, j))
    # every cell of the exact fill, bit for bit
    H.assert_same_bits(r["be"].read_matrix(0), want["cells"], "2x2000 forward cells")
    # the device-side best-path traceback through the exact matrix is the reference's path (oracle/trace_oracle.py
    # restates bestTrace over the CPU matrix); the fast fill's matrix gives the same path here
    from oracle import trace_oracle
    path = trace_oracle.best_trace(*r["img"], want)
    assert r["be"].best_trace()[0] == path
    assert r["bf"].best_trace()[0] == path
    # HX_LSE_LINEAR runs this pair on scaled probabilities (hx_linear.hip): exact arithmetic up to fp64 rounding, so it
    # differs from the reference by the reference's own truncation of log-sum-exp terms below e^-10
    bl = capi.Batch([r["img"]], capi.HX_LSE_LINEAR)
    bl.forward()
    lp_lin = float(bl.lp_end()[0])
    assert abs(lp_lin - want["lp_end"]) <= 1e-4 * abs(want["lp_end"])   # north_star tolerance
    assert abs(lp_lin - want["lp_end"]) <= 1e-5 * abs(want["lp_end"])   # what it actually achieves
    true = c_oracle.forward(*r["img"], true_math=True)                  # the oracle's recursion in libm arithmetic
    assert abs(lp_lin - true["lp_end"]) <= 1e-12 * abs(true["lp_end"])
    lin = bl.read_matrix(0)
    assert np.array_equal(np.isneginf(lin), np.isneginf(true["cells"]))
    fin = np.isfinite(lin)
    assert np.max(np.abs(lin[fin] - true["cells"][fin])) < 1e-8
    bl.close()
    r["be"].close()
    r["bf"].close()


def test_dna_2x1000_jc69_pair():
    f = H.leaf_case(402, 1000, 1000, tl=.1, tr=.1)
    r = fills(f)
    want = c_oracle.forward(*r["img"])
    assert r["lp_exact"] == want["lp_end"]
    assert abs(r["lp_start"] - r["lp_exact"]) <= 1e-5 * abs(r["lp_exact"])
    assert abs(r["lp_fast"] - want["lp_end"]) <= 1e-9 * abs(want["lp_end"])
    H.assert_same_bits(r["be"].read_matrix(0), want["cells"], "2x1000 forward cells")
    fast = r["bf"].read_matrix(0)
    fin = np.isfinite(want["cells"])
    assert np.array_equal(np.isneginf(fast), np.isneginf(want["cells"]))
    assert np.max(np.abs(fast[fin] - want["cells"][fin])) < 1e-6
    r["be"].close()
    r["bf"].close()


def test_four_component_mixture_5000_columns():
    # configs[4] shape: 4-component mixture, 5000-column profile (300 rows keep the oracle quick)
    f = H.leaf_case(403, 300, 5000, alphabet=AA, components=4, jc=False, tl=.05, tr=.05)
    r = fills(f)
    want = c_oracle.forward(*r["img"])
    assert r["lp_exact"] == want["lp_end"]
    assert abs(r["lp_fast"] - want["lp_end"]) <= 1e-9 * abs(want["lp_end"])
    assert abs(r["lp_start"] - r["lp_exact"]) <= 1e-5 * abs(r["lp_exact"])
    H.assert_same_bits(r["be"].read_matrix(0), want["cells"], "mixture forward cells")
    # and the transposed shape: 5000 rows x 300 columns (many strips per wave)
    g = H.leaf_case(404, 5000, 300, alphabet=AA, components=4, jc=False, tl=.05, tr=.05)
    r2 = fills(g, backward=False)
    w2 = c_oracle.forward(*r2["img"])
    assert r2["lp_exact"] == w2["lp_end"]
    H.assert_same_bits(r2["be"].read_matrix(0), w2["cells"], "tall mixture forward cells")
    # the scaled-probability fills on both shapes: 5000 columns per strip, and 79 strips over 16 waves (five rounds of the
    # wrap-around link), Forward and Backward, against the oracle's recursion in libm arithmetic
    for img in (r["img"], r2["img"]):
        bl = capi.Batch([img], capi.HX_LSE_LINEAR)
        bl.forward()
        bl.backward()
        tf, tb = c_oracle.forward(*img, true_math=True), c_oracle.backward(*img, true_math=True)
        assert abs(bl.lp_end()[0] - tf["lp_end"]) <= 1e-12 * abs(tf["lp_end"])
        assert abs(bl.lp_start()[0] - tb["lp_start"]) <= 1e-12 * abs(tb["lp_start"])
        for which, want in ((0, tf["cells"]), (1, tb["cells"])):
            got = bl.read_matrix(0, which)
            assert np.array_equal(np.isneginf(got), np.isneginf(want))
            fin = np.isfinite(got)
            assert np.max(np.abs(got[fin] - want[fin])) < 1e-8
        bl.close()
    for b in (r["be"], r["bf"], r2["be"], r2["bf"]):
        b.close()


def test_config4_batch_of_512_pairs_forward_equals_backward_in_every_policy():
    """BASELINE configs[3] at size: 512 independent 2x2000-residue protein pairs (WAG) in ONE batch, Forward and Backward.
    Size-independent properties over all 512 pairs - lpStart == lpEnd (1e-6 relative; to 1e-11 on scaled probabilities),
    the fast and scaled-probability policies within north_star's 1e-4 of the exact one (they are within 1e-9 / 1e-5),
    a second launch reproduces the first bit for bit - and the CPU oracle on three of the pairs: lpEnd bit for bit in
    exact mode, the best path identical in exact and fast mode."""
    import os
    from historian_amd import hostmodel, workload
    from oracle import trace_oracle
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    model = hostmodel.RateModel.load(os.path.join(root, "tests", "golden", "models", "wag.json"))
    hmm = hostmodel.make_hmm(model, .2, .3)
    jobs = [workload.leaf_pair(np.random.default_rng(4000 + k), model, hmm, 2000) for k in range(512)]
    lp = {}
    for name, flags in (("exact", 0), ("fast", capi.HX_LSE_FAST), ("linear", capi.HX_LSE_LINEAR)):
        b = capi.Batch(jobs, flags | capi.HX_KEEP_BACKWARD)
        b.forward()
        b.backward()
        le, ls = b.lp_end().copy(), b.lp_start().copy()
        assert np.all(np.isfinite(le)) and np.all(le < 0)
        assert np.max(np.abs(ls - le) / np.abs(le)) <= (1e-11 if name == "linear" else 1e-6), name
        b.forward()
        H.assert_same_bits(b.lp_end(), le, "second launch, " + name)
        lp[name] = le
        if name != "linear":
            paths = b.best_trace()
            for k in (0, 255, 511):
                x, y, h, md = jobs[k]
                want = c_oracle.forward(x, y, h, md)
                if name == "exact":
                    H.assert_same_bits([le[k]], [want["lp_end"]], "lpEnd of pair %d" % k)
                assert paths[k] == trace_oracle.best_trace(x, y, h, md, want), (name, k)
        b.close()
    assert np.max(np.abs(lp["fast"] - lp["exact"]) / np.abs(lp["exact"])) <= 1e-9
    assert np.max(np.abs(lp["linear"] - lp["exact"]) / np.abs(lp["exact"])) <= 1e-5


def test_config4_batch_banded_as_the_reference_runs_it(monkeypatch):
    """The same batch in the reference's default mode - every fill inside a band of 20 around the guide alignment
    (src/forward.h:92-98, src/alignpath.cpp:282-310) - at size: 512 pairs of 2x2000 residues, Forward and Backward in the
    banded rotating-row sweep (hx_band.hip), every policy.  Properties over all 512 pairs: both fills run the sweep
    (hx_batch_job_kernel), lpStart == lpEnd, fast / scaled probabilities within 1e-9 / 1e-5 of exact, the Backward sweep's
    lpStart bit for bit that of the strip pipeline it replaces (table policies); and the CPU oracle on three of the pairs:
    lpEnd and lpStart bit for bit in exact mode."""
    import os
    from historian_amd import hostmodel, workload
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    model = hostmodel.RateModel.load(os.path.join(root, "tests", "golden", "models", "wag.json"))
    hmm = hostmodel.make_hmm(model, .2, .3)
    jobs = [workload.leaf_pair(np.random.default_rng(4000 + k), model, hmm, 2000, band=20) for k in range(512)]
    lp = {}
    for name, flags in (("exact", 0), ("fast", capi.HX_LSE_FAST), ("linear", capi.HX_LSE_LINEAR)):
        b = capi.Batch(jobs, flags | capi.HX_KEEP_BACKWARD | capi.HX_SPARSE_ENVELOPE)
        kern = [b.job_kernel(k) for k in range(len(jobs))]
        assert all(c == 2 and s for c, s in kern), "every pair in the rotating-row sweep, both directions"
        b.forward()
        b.backward()
        le, ls = b.lp_end().copy(), b.lp_start().copy()
        assert np.all(np.isfinite(le)) and np.all(le < 0)
        assert np.max(np.abs(ls - le) / np.abs(le)) <= (1e-11 if name == "linear" else 1e-6), name
        lp[name] = le
        if name == "exact":
            for k in (0, 255, 511):
                x, y, h, md = jobs[k]
                H.assert_same_bits([le[k]], [c_oracle.forward(x, y, h, md)["lp_end"]], "lpEnd of pair %d" % k)
                H.assert_same_bits([ls[k]], [c_oracle.backward(x, y, h, md)["lp_start"]], "lpStart of pair %d" % k)
        b.close()
        if name != "linear":
            monkeypatch.setenv("HX_BAND_BWD_OLD", "1")
            o = capi.Batch(jobs, flags | capi.HX_KEEP_BACKWARD | capi.HX_SPARSE_ENVELOPE)
            assert not any(s for c, s in (o.job_kernel(k) for k in range(len(jobs))))
            o.forward()
            o.backward()
            H.assert_same_bits(o.lp_start(), ls, "lpStart, sweep vs strip pipeline, " + name)
            o.close()
            monkeypatch.delenv("HX_BAND_BWD_OLD")
    assert np.max(np.abs(lp["fast"] - lp["exact"]) / np.abs(lp["exact"])) <= 1e-9
    assert np.max(np.abs(lp["linear"] - lp["exact"]) / np.abs(lp["exact"])) <= 1e-5


def test_large_banded_batch_two_pairs_per_wavefront_default_policy():
    """The driver-run bench's banded block at size: 2560 pairs of 2x2000 residues, band 20, band-compressed planes, the
    library's default policy - above 1024 pairs the fill is two pairs per wavefront (hx_band2.hip) with its edge kernel on
    the side stream.  128 distinct pairs, each twenty times over (the fill does not know): every copy's lpEnd the same bits;
    against one pair per wavefront (HX_BAND2=0: hx_band.hip, same arithmetic) the same bits; against the libm-arithmetic
    oracle with truncation within 1e-9 relative on sampled pairs; against the exact policy within 1e-9; the device's best
    paths of the first 64 pairs those of the exact policy's matrices; and the in-envelope cells of one pair, read through
    hx_batch_read_cells, within 1e-9 of the oracle's."""
    import os
    from historian_amd import hostmodel, workload
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    model = hostmodel.RateModel.load(os.path.join(root, "tests", "golden", "models", "wag.json"))
    hmm = hostmodel.make_hmm(model, .2, .3)
    base = [workload.leaf_pair(np.random.default_rng(7000 + k), model, hmm, 2000, band=20) for k in range(128)]
    jobs = [base[k % 128] for k in range(2560)]
    b = capi.Batch(jobs, capi.HX_LSE_TRUNC | capi.HX_BAND_COMPRESSED)
    assert b.shared_wavefront_pairs() == 2560
    b.forward()
    lp = b.lp_end().copy()
    assert np.all(np.isfinite(lp)) and np.all(lp < 0)
    for k in range(128, 2560):
        assert lp[k] == lp[k % 128], k
    all_paths, all_len = b.best_trace(raw=True)
    paths, plen = all_paths[:64].copy(), all_len[:64].copy()
    # the oracle in libm arithmetic with the reference's truncation (true_math=2): three pairs
    for k in (0, 77, 127):
        x, y, h, md = base[k]
        want = c_oracle.forward(x, y, h, md, true_math=2)
        assert abs(want["lp_end"] - lp[k]) <= 1e-9 * abs(lp[k]), k
        if k == 77:
            mask = np.isfinite(want["cells"]).any(axis=2)
            ii, jj = np.nonzero(mask)
            sel = np.random.default_rng(1).choice(len(ii), size=20000, replace=False)
            got = b.read_cells(k, np.stack([ii[sel], jj[sel]], axis=1))
            ref = want["cells"][ii[sel], jj[sel]]
            fin = np.isfinite(ref)
            assert np.array_equal(np.isfinite(got), fin)
            assert np.max(np.abs(got[fin] - ref[fin])) < 1e-9
    b.close()
    os.environ["HX_BAND2"] = "0"
    try:
        one = capi.Batch(jobs[:128], capi.HX_LSE_TRUNC | capi.HX_BAND_COMPRESSED)
        assert one.shared_wavefront_pairs() == 0
        one.forward()
        H.assert_same_bits(one.lp_end(), lp[:128], "one pair per wavefront vs two")
        one.close()
    finally:
        del os.environ["HX_BAND2"]
    ex = capi.Batch(jobs[:64], capi.HX_LSE_EXACT | capi.HX_BAND_COMPRESSED)
    ex.forward()
    le = ex.lp_end()
    assert np.max(np.abs(le - lp[:64]) / np.abs(le)) <= 1e-9
    epaths, elen = ex.best_trace(raw=True)
    assert np.array_equal(elen, plen)
    for k in range(64):
        assert np.array_equal(epaths[k, :elen[k]], paths[k, :plen[k]]), "best path of pair %d" % k
    ex.close()
