"""HX_LSE_TRUNC: the scaled-probability fills with the reference's truncation (hx_linear.hip / hx_band.hip trunc_sum).

The reference's log_sum_exp(a, b) is max + T(|a - b|) with T = 0 for differences >= 10 (src/logsumexp.h:42-84): it drops
the smaller term when it is at most e^-10 of the larger, and its n-ary forms are left-nested (:86-100).  The policy
does exactly that on probabilities - no table, no logarithm until the store.  Two yardsticks:
  * the oracle's recursion in libm arithmetic WITH the truncation (c_oracle true_math=2): same -inf pattern, finite
    cells within 1e-9, lpEnd within 1e-12 relative - the kernel is that arithmetic up to fp64 rounding;
  * the pinned oracle (the reference's table arithmetic): cells within 1e-6 (what is left is the interpolation error of
    the reference's table, < 3e-10 per operation, accumulated along the alignment), lpEnd within 1e-9 relative - and the
    best path through the device matrix is the reference's (tests/test_gpu_trace_identity.py)."""
import numpy as np
import pytest

from historian_amd import capi
from oracle import c_oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu

AA = "arndcqeghilkmfpstwyv"


@pytest.fixture(scope="module", autouse=True)
def engine():
    capi.init(0, c_oracle.table())      # host-libm table
    yield
    capi.shutdown()


def check_forward(cases, flags=0, env_only=False, monkeypatch=None):
    imgs = [H.job_images(f) for f in cases]
    bt = capi.Batch(imgs, capi.HX_LSE_TRUNC | flags)
    bt.forward()
    lt = bt.lp_end()
    for k, (x, y, hmm, md) in enumerate(imgs):
        want = c_oracle.forward(x, y, hmm, md, true_math=2)
        ref = c_oracle.forward(x, y, hmm, md)
        got = bt.read_matrix(k, 0)
        sel = H.envelope_mask(cases[k]) if env_only else np.ones(got.shape[:2], dtype=bool)
        assert not np.isnan(got[sel]).any()
        assert np.array_equal(np.isneginf(want["cells"][sel]), np.isneginf(got[sel])), "job %d: -inf pattern" % k
        fin = np.isfinite(want["cells"]) & sel[:, :, None]
        assert np.max(np.abs(want["cells"][fin] - got[fin]), initial=0.) < 1e-9, "job %d" % k
        assert np.array_equal(np.isneginf(ref["cells"][sel]), np.isneginf(got[sel])), "job %d: -inf pattern of the reference" % k
        assert np.max(np.abs(ref["cells"][fin] - got[fin]), initial=0.) < 1e-6, "job %d vs the reference arithmetic" % k
        if np.isfinite(ref["lp_end"]):
            assert abs(want["lp_end"] - lt[k]) <= 1e-12 * abs(lt[k])
            assert abs(ref["lp_end"] - lt[k]) <= 1e-9 * abs(lt[k])
        else:
            assert lt[k] == ref["lp_end"]
    return bt, imgs


@pytest.mark.parametrize("waves", [0, 1, 2, 8])
def test_unbanded_leaf_pairs(waves, monkeypatch):
    if waves:
        monkeypatch.setenv("HX_LINEAR_WAVES", str(waves))
    groups = [[H.leaf_case(301, 40, 45), H.leaf_case(312, 1, 1), H.leaf_case(305, 63, 64, alphabet=AA, jc=False, tl=.3, tr=.2)],
              [H.leaf_case(306, 100, 130), H.leaf_case(307, 127, 20, alphabet=AA, components=2, jc=False)],
              [H.leaf_case(308, 250, 200, alphabet=AA, jc=False), H.leaf_case(313, 0, 4)],
              [H.leaf_case(310, 1100, 700, alphabet=AA, jc=False, tl=.2, tr=.3), H.leaf_case(311, 700, 1500)]]
    for cases in groups:
        bt, imgs = check_forward(cases)
        # the best path through the device's matrix is the path through the exact fill's matrix
        be = capi.Batch(imgs)
        be.forward()
        assert bt.best_trace() == be.best_trace()
        be.close()
        bt.close()


def test_backward_on_unbanded_leaf_pairs():
    cases = [H.leaf_case(306, 100, 130), H.leaf_case(307, 127, 20, alphabet=AA, components=2, jc=False),
             H.leaf_case(309, 500, 300), H.leaf_case(305, 63, 64, alphabet=AA, jc=False, tl=.3, tr=.2)]
    imgs = [H.job_images(f) for f in cases]
    bt = capi.Batch(imgs, capi.HX_LSE_TRUNC | capi.HX_KEEP_BACKWARD)
    bt.forward()
    bt.backward()
    lt, st = bt.lp_end(), bt.lp_start()
    for k, (x, y, hmm, md) in enumerate(imgs):
        want = c_oracle.backward(x, y, hmm, md, true_math=2)
        ref = c_oracle.backward(x, y, hmm, md)
        got = bt.read_matrix(k, 1)
        assert np.array_equal(np.isneginf(want["cells"]), np.isneginf(got))
        fin = np.isfinite(got)
        assert np.max(np.abs(want["cells"][fin] - got[fin]), initial=0.) < 1e-9
        assert np.max(np.abs(ref["cells"][fin] - got[fin]), initial=0.) < 1e-6
        assert abs(ref["lp_start"] - st[k]) <= 1e-9 * abs(st[k])
        # Forward and Backward drop different terms, so the two likelihoods agree only as far as the reference's own do
        # (FWD_BACK_ERROR_TOLERANCE .01, src/forward.cpp:9): the gap is the reference's gap
        ref_gap = ref["lp_start"] - c_oracle.forward(x, y, hmm, md)["lp_end"]
        assert abs((st[k] - lt[k]) - ref_gap) <= 1e-9 * abs(lt[k])
        assert abs(st[k] - lt[k]) <= 1e-4 * abs(lt[k])
    bt.close()


@pytest.mark.parametrize("ppw", [0, -3, 2])
def test_banded_leaf_pairs_in_the_rotating_row_sweep(ppw, monkeypatch):
    if ppw:
        monkeypatch.setenv("HX_BAND_PPW", str(ppw))
    cases = [H.leaf_case(401, 70, 66, band=5), H.leaf_case(402, 200, 90, band=12), H.leaf_case(403, 130, 150, band=3),
             H.leaf_case(404, 300, 330, alphabet=AA, jc=False, band=20), H.leaf_case(405, 40, 45, band=0),
             H.leaf_case(407, 500, 520, band=8)]
    for flags in (0, capi.HX_SPARSE_ENVELOPE, capi.HX_BAND_COMPRESSED):
        bt, imgs = check_forward(cases, flags, env_only=flags != 0)
        be = capi.Batch(imgs, flags & ~capi.HX_BAND_COMPRESSED)
        be.forward()
        assert bt.best_trace() == be.best_trace()
        if flags != capi.HX_BAND_COMPRESSED:
            be.backward()
            bt.backward()
            for k, (x, y, hmm, md) in enumerate(imgs):
                want = c_oracle.backward(x, y, hmm, md, true_math=2)
                got = bt.read_matrix(k, 1)
                inside = np.isfinite(want["cells"])
                assert np.max(np.abs(want["cells"][inside] - got[inside]), initial=0.) < 1e-9, "job %d backward" % k
                if not flags:
                    assert np.array_equal(np.isneginf(want["cells"]), np.isneginf(got))
        be.close()
        bt.close()


def test_general_profiles_run_as_the_table_policy():
    # no truncating kernel for state DAGs yet: those classes take HX_LSE_FAST (which truncates as the reference does)
    cases = [H.dag_case(71, n=90, samples=4), H.dag_case(72, n=150, band=6, samples=3)]
    imgs = [H.job_images(f) for f in cases]
    bt = capi.Batch(imgs, capi.HX_LSE_TRUNC)
    bf = capi.Batch(imgs, capi.HX_LSE_FAST)
    bt.forward()
    bf.forward()
    for k in range(len(imgs)):
        H.assert_same_bits(bt.read_matrix(k, 0), bf.read_matrix(k, 0), "general profile under HX_LSE_TRUNC")
    bt.close()
    bf.close()


@pytest.mark.parametrize("nw", [1, 2, 4])
def test_two_banded_pairs_per_wavefront(nw, monkeypatch):
    # hx_band2.hip: lanes 0-31 sweep one pair, lanes 32-63 another (lane = row mod 32, ds_swizzle rotation inside each half).
    # Forced on a small batch (HX_BAND2=1; by itself it takes batches of more than 512 pairs), an odd number of pairs (the
    # last wavefront has an idle half), pairs of different lengths sharing a wavefront, both scaled-probability policies, the
    # three storage modes, Forward and Backward.  Yardsticks as above; the untruncated policy against true_math=1.
    monkeypatch.setenv("HX_BAND2", "1")
    monkeypatch.setenv("HX_BAND2_NW", str(nw))
    cases = [H.leaf_case(401, 70, 66, band=5), H.leaf_case(402, 200, 90, band=12), H.leaf_case(403, 130, 150, band=3),
             H.leaf_case(404, 300, 330, alphabet=AA, jc=False, band=20), H.leaf_case(405, 40, 45, band=0),
             H.leaf_case(407, 500, 520, band=8), H.leaf_case(408, 33, 31, band=4), H.leaf_case(409, 260, 250, alphabet=AA, jc=False, band=20),
             H.leaf_case(410, 64, 64, band=6)]
    imgs = [H.job_images(f) for f in cases]
    for policy, tm in ((capi.HX_LSE_TRUNC, 2), (capi.HX_LSE_LINEAR, 1)):
        for flags in (0, capi.HX_SPARSE_ENVELOPE, capi.HX_BAND_COMPRESSED):
            bt = capi.Batch(imgs, policy | flags)
            assert bt.shared_wavefront_pairs() == len(cases)
            bt.forward()
            lt = bt.lp_end()
            for k, (x, y, hmm, md) in enumerate(imgs):
                want = c_oracle.forward(x, y, hmm, md, true_math=tm)
                got = bt.read_matrix(k, 0)
                sel = H.envelope_mask(cases[k]) if flags else np.ones(got.shape[:2], dtype=bool)
                assert not np.isnan(got[sel]).any(), "job %d" % k
                assert np.array_equal(np.isneginf(want["cells"][sel]), np.isneginf(got[sel])), "job %d: -inf pattern" % k
                fin = np.isfinite(want["cells"]) & sel[:, :, None]
                assert np.max(np.abs(want["cells"][fin] - got[fin]), initial=0.) < 1e-9, "job %d" % k
                if np.isfinite(want["lp_end"]):
                    assert abs(want["lp_end"] - lt[k]) <= 1e-12 * abs(lt[k])
                else:
                    assert lt[k] == want["lp_end"]
            if flags != capi.HX_BAND_COMPRESSED:
                bt.backward()
                st = bt.lp_start()
                for k, (x, y, hmm, md) in enumerate(imgs):
                    want = c_oracle.backward(x, y, hmm, md, true_math=tm)
                    got = bt.read_matrix(k, 1)
                    inside = np.isfinite(want["cells"])
                    assert np.max(np.abs(want["cells"][inside] - got[inside]), initial=0.) < 1e-9, "job %d backward" % k
                    if not flags:
                        assert np.array_equal(np.isneginf(want["cells"]), np.isneginf(got)), "job %d backward -inf pattern" % k
                    if np.isfinite(want["lp_start"]):
                        assert abs(want["lp_start"] - st[k]) <= 1e-12 * abs(st[k])
            # the same bits as one pair per wavefront (hx_band.hip computes the same operations in the same order)
            monkeypatch.setenv("HX_BAND2", "0")
            b1 = capi.Batch(imgs, policy | flags)
            assert b1.shared_wavefront_pairs() == 0
            b1.forward()
            for k in range(len(imgs)):
                a, c = bt.read_matrix(k, 0), b1.read_matrix(k, 0)
                sel = H.envelope_mask(cases[k]) if flags else np.ones(a.shape[:2], dtype=bool)
                H.assert_same_bits(a[sel], c[sel], "two pairs per wavefront vs one, job %d" % k)
            assert bt.best_trace() == b1.best_trace()
            b1.close()
            monkeypatch.setenv("HX_BAND2", "1")
            bt.close()


@pytest.mark.parametrize("policy", ["trunc", "exact"])
def test_line_groups_change_no_in_envelope_cell(policy, monkeypatch):
    # build_band_rows gives the four rows of a 64-byte group the same owned steps, so that the sweeps write whole cache lines;
    # what a row gains lies outside the envelope.  With and without (HX_BAND_NO_LINE_GROUPS=1): every in-envelope cell the same
    # bits, Forward and Backward, lpEnd / lpStart the same bits, dense planes still -inf outside the envelope - one pair per
    # wavefront and two, pre-filled, sparse and band-compressed planes, bands of 0 to 20, pairs whose ends come close to the
    # last rows and columns (unequal lengths).
    flag = capi.HX_LSE_TRUNC if policy == "trunc" else capi.HX_LSE_EXACT
    cases = [H.leaf_case(501, 70, 66, band=5), H.leaf_case(502, 200, 90, band=12), H.leaf_case(503, 131, 150, band=3),
             H.leaf_case(504, 300, 333, alphabet=AA, jc=False, band=20), H.leaf_case(505, 41, 45, band=0), H.leaf_case(506, 257, 256, band=9)]
    imgs = [H.job_images(f) for f in cases]
    for band2 in (("0", "1") if policy == "trunc" else ("0",)):
        monkeypatch.setenv("HX_BAND2", band2)
        for extra in (0, capi.HX_SPARSE_ENVELOPE, capi.HX_BAND_COMPRESSED):
            got = {}
            for groups in (True, False):
                if groups:
                    monkeypatch.delenv("HX_BAND_NO_LINE_GROUPS", raising=False)
                else:
                    monkeypatch.setenv("HX_BAND_NO_LINE_GROUPS", "1")
                b = capi.Batch(imgs, flag | extra | (0 if extra == capi.HX_BAND_COMPRESSED else capi.HX_KEEP_BACKWARD))
                assert all(b.job_kernel(k)[0] == 2 for k in range(len(imgs)))
                b.forward()
                res = {"lp_end": b.lp_end().copy()}
                masks = []
                for k, f in enumerate(cases):
                    mask = H.envelope_mask(f)
                    masks.append(mask)
                    ii, jj = np.nonzero(mask)
                    res["f%d" % k] = b.read_cells(k, np.stack([ii, jj], axis=1))
                    if extra == 0:
                        full = b.read_matrix(k, 0)
                        assert np.all(np.isneginf(full[~mask])), "pre-filled planes: -inf outside the envelope"
                if extra != capi.HX_BAND_COMPRESSED:
                    b.backward()
                    res["lp_start"] = b.lp_start().copy()
                    for k in range(len(cases)):
                        res["b%d" % k] = b.read_matrix(k, 1)[masks[k]]
                b.close()
                got[groups] = res
            for key in got[True]:
                a, c = np.asarray(got[True][key]), np.asarray(got[False][key])
                what = "%s, band2 %s, flags %d: %s" % (policy, band2, extra, key)
                if policy == "exact":
                    H.assert_same_bits(a, c, what)
                else:
                    # the scaled-probability policies evaluate row 0 beyond the sweep's reach as a prefix sum (hx_bandedge.h):
                    # where the sweep's reach on row 0 changes, the same cell is the same number to an ulp, not to the bit
                    assert np.array_equal(np.isneginf(a), np.isneginf(c)), what
                    fin = np.isfinite(a)
                    assert np.all(np.abs(a[fin] - c[fin]) <= 1e-12 * np.abs(a[fin])), what
