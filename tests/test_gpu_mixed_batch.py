"""A batch is partitioned by kernel class inside hx_batch_forward / hx_batch_backward (leaf pairs with the y side in
LDS, other leaf pairs, chain profiles, general profiles; banded or not), so a tree level that mixes leaf-leaf and
internal-node pairs runs every pair on the kernel that suits it.  Whatever the mix, exact mode stays bit-identical
to the oracle, the results come back in the caller's job order, and lpEnd / lpStart are one contiguous copy.
Also here: the regression shape of round 1's 04:26 abort and the several-devices entry points."""
import numpy as np
import pytest

from historian_amd import capi
from oracle import c_oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def engine():
    capi.init(0, c_oracle.table())
    yield
    capi.shutdown()


def _mixed_cases():
    prot = "arndcqeghilkmfpstwyv"
    return [H.leaf_case(11, 70, 66),                                   # leaf, y side in LDS
            H.dag_case(31),                                            # general profiles
            H.leaf_case(12, 130, 90, band=6),                          # banded leaf
            H.leaf_case(13, 40, 75, alphabet=prot, jc=False),          # protein leaf
            H.dag_case(43, band=3),                                    # banded general profiles
            H.leaf_case(14, 200, 150, alphabet=prot, jc=False, components=2),
            H.leaf_case(15, 3, 5),                                     # tiny
            H.dag_case(67, n=9, band=2, keep_all=True)]


@pytest.mark.parametrize("flags", [capi.HX_LSE_EXACT, capi.HX_LSE_FAST, capi.HX_LSE_LINEAR, capi.HX_SPARSE_ENVELOPE])
def test_mixed_batch_matches_single_job_batches(flags):
    cases = _mixed_cases()
    imgs = [H.job_images(f) for f in cases]
    b = capi.Batch(imgs, flags | capi.HX_KEEP_BACKWARD)
    b.forward()
    b.backward()
    lp_end, lp_start = b.lp_end(), b.lp_start()
    paths = b.best_trace()
    for k, img in enumerate(imgs):
        one = capi.Batch([img], flags | capi.HX_KEEP_BACKWARD)
        one.forward()
        one.backward()
        inside = None
        if flags & capi.HX_SPARSE_ENVELOPE and img[3] >= 0:        # cells outside the envelope are undefined
            inside = np.isfinite(c_oracle.forward(*img)["cells"])
        for which in (0, 1):
            got, want = b.read_matrix(k, which), one.read_matrix(0, which)
            if inside is not None and which == 0:
                got, want = got[inside], want[inside]
            elif inside is not None:
                continue
            H.assert_same_bits(got, want, "job %d matrix %d" % (k, which))
        H.assert_same_bits([lp_end[k]], one.lp_end(), "lpEnd of job %d" % k)
        H.assert_same_bits([lp_start[k]], one.lp_start(), "lpStart of job %d" % k)
        assert paths[k] == one.best_trace()[0]
        one.close()
    b.close()


def test_mixed_batch_exact_is_bit_identical_to_the_oracle():
    imgs = [H.job_images(f) for f in _mixed_cases()]
    b = capi.Batch(imgs, capi.HX_KEEP_BACKWARD)
    b.forward()
    b.backward()
    lp_end, lp_start = b.lp_end(), b.lp_start()
    for k, (x, y, hmm, md) in enumerate(imgs):
        wf, wb = c_oracle.forward(x, y, hmm, md), c_oracle.backward(x, y, hmm, md)
        H.assert_same_bits(b.read_matrix(k, 0), wf["cells"], "forward cells of job %d" % k)
        H.assert_same_bits(b.read_matrix(k, 1), wb["cells"], "backward cells of job %d" % k)
        H.assert_same_bits([lp_end[k]], [wf["lp_end"]], "lpEnd")
        H.assert_same_bits([lp_start[k]], [wb["lp_start"]], "lpStart")
    b.close()


@pytest.mark.parametrize("flags", [capi.HX_LSE_EXACT, capi.HX_LSE_FAST, capi.HX_LSE_LINEAR])
def test_two_tiny_leaf_pairs_with_backward(flags):
    """The shape that was running when a working-tree build aborted inside hx_batch_lp_end in round 1 (a 2-job batch of
    3x3- and 3x5-state leaf pairs with HX_KEEP_BACKWARD): one wavefront per pair, y side in LDS, every arithmetic."""
    imgs = [H.job_images(H.leaf_case(1, 1, 1)), H.job_images(H.leaf_case(2, 1, 3))]
    b = capi.Batch(imgs, flags | capi.HX_KEEP_BACKWARD)
    b.forward()
    b.backward()
    lp_end, lp_start = b.lp_end(), b.lp_start()
    for k, img in enumerate(imgs):
        want = c_oracle.forward(*img)
        if flags == capi.HX_LSE_EXACT:
            H.assert_same_bits(b.read_matrix(k, 0), want["cells"], "cells")
            H.assert_same_bits([lp_end[k]], [want["lp_end"]], "lpEnd")
        assert abs(lp_end[k] - want["lp_end"]) <= 1e-4 * abs(want["lp_end"])
        assert abs(lp_start[k] - lp_end[k]) <= 1e-2 * abs(lp_end[k])
    b.close()


def test_unsupported_shapes_are_refused_not_launched():
    """Nothing aborts across the ABI: a batch no kernel supports comes back as an error code from create / forward."""
    img = H.job_images(H.dag_case(31, band=2))
    with pytest.raises(capi.HxError) as e:
        capi.Batch([img], capi.HX_BAND_COMPRESSED)          # compressed planes: chain (leaf) profiles only
    assert e.value.code == -1
    with pytest.raises(capi.HxError):
        capi.Batch([H.job_images(H.leaf_case(3, 20, 20))], 0, device=7)      # no tables on that device


def test_batches_on_an_explicit_device():
    lib = capi.load()
    assert lib.hx_device_count() >= 1
    img = H.job_images(H.leaf_case(21, 50, 60))
    b = capi.Batch([img], 0, device=0)
    assert lib.hx_batch_device(b._h) == 0
    b.forward()
    H.assert_same_bits(b.lp_end(), [c_oracle.forward(*img)["lp_end"]], "lpEnd on device 0")
    b.close()


def test_many_jobs_one_scalar_copy():
    """33 000 one-residue pairs: more jobs than grid.y admits (the prep kernels are launched in chunks), lpEnd read back
    as one array."""
    img = H.job_images(H.leaf_case(5, 1, 1))
    n = 33000
    b = capi.Batch([img] * n, capi.HX_LSE_FAST)
    b.forward()
    lp = b.lp_end()
    want = c_oracle.forward(*img)["lp_end"]
    assert lp.shape == (n,) and np.all(np.abs(lp - want) < 1e-9)
    b.close()


def test_asynchronous_matrix_reads_equal_the_synchronous_ones():
    """hx_batch_read_matrix_async / hx_batch_wait_read: copies of several jobs in flight at once into page-locked buffers
    (hx_host_alloc), waited for out of order; an unstarted read is refused."""
    import ctypes as C
    cases = [H.leaf_case(801, 120, 100), H.dag_case(33), H.leaf_case(802, 64, 200, band=8)]
    imgs = [H.job_images(f) for f in cases]
    b = capi.Batch(imgs, capi.HX_KEEP_BACKWARD)
    b.forward()
    b.backward()
    lib = capi.load()
    lib.hx_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
    lib.hx_host_free.argtypes = [C.c_void_p]
    bufs = []
    for k in range(len(cases)):
        for which in (0, 1):
            n = b.layout(k, which).matrix_doubles
            p = C.c_void_p()
            assert lib.hx_host_alloc(n * 8, C.byref(p)) == 0
            assert lib.hx_batch_read_matrix_async(b._h, k, which, p) == 0
            bufs.append((k, which, n, p))
    assert lib.hx_batch_wait_read(b._h, 1, 0) == 0
    for k, which, n, p in reversed(bufs):
        assert lib.hx_batch_wait_read(b._h, k, which) == 0
        got = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(n,)).copy()
        want = np.empty(n)
        assert lib.hx_batch_read_matrix(b._h, k, which, want.ctypes.data_as(C.POINTER(C.c_double))) == 0
        H.assert_same_bits(got, want, "async read of job %d matrix %d" % (k, which))
        lib.hx_host_free(p)
    nb = capi.Batch(imgs[:1])
    nb.forward()
    assert lib.hx_batch_wait_read(nb._h, 0, 0) == -7          # nothing was started
    nb.close()
    b.close()
