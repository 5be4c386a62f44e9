"""Randomised sweep of the scaled-probability fills (HX_LSE_LINEAR) against the plain-C oracle with the cell recursion
in libm arithmetic (c_oracle true_math): leaf pairs, with and without a band, default / sparse-envelope /
band-compressed storage, every workgroup shape (HX_LINEAR_WAVES, HX_LINEAR_PPW), Forward and Backward,
plus the device traceback on the compressed planes.  Tolerances: finite cells 1e-9 absolute, lpEnd / lpStart 1e-12
relative, the same -inf pattern.  Not part of the test suite; run on the GPU box:  python tools/stress_linear.py [n_batches]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from historian_amd import capi
from oracle import c_oracle
from tests import helpers as H

capi.init(0, c_oracle.table())
n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 20261005)
t0 = time.time()
n_jobs = n_cells = 0
worst = 0.
for bi in range(n_batches):
    banded = rng.random() < .5
    many = rng.random() < .3
    cases = []
    for _ in range(rng.randint(8, 20) if many else rng.randint(1, 5)):
        if many:
            lx, ly = rng.randint(1, 180), rng.randint(1, 180)
        else:
            lx, ly = rng.choice([1, 5, 63, 64, 65, 130, 200, 333, 700]), rng.choice([2, 7, 64, 100, 129, 260, 520])
        prot = rng.random() < .4
        band = rng.choice([0, 1, 3, 8, 20]) if banded and rng.random() < .85 else None
        cases.append(H.leaf_case(rng.randint(1, 10 ** 6), lx, ly, alphabet="arndcqeghilkmfpstwyv" if prot else "ACGT",
                                 jc=not prot, band=band, components=rng.choice([1, 1, 2]) if prot else 1))
    any_band = any(H.job_images(f)[3] >= 0 for f in cases)
    storage = rng.choice([0, capi.HX_SPARSE_ENVELOPE, capi.HX_BAND_COMPRESSED]) if any_band else 0
    os.environ.pop("HX_LINEAR_WAVES", None); os.environ.pop("HX_LINEAR_PPW", None)
    shape = ""
    if any_band and rng.random() < .5:
        os.environ["HX_LINEAR_PPW"] = "6"; shape = " ppw6"
    elif not any_band and rng.random() < .6:
        w = rng.choice([1, 2, 4, 8, 16]); os.environ["HX_LINEAR_WAVES"] = str(w); shape = " waves%d" % w
    imgs = [H.job_images(f) for f in cases]
    b = capi.Batch(imgs, capi.HX_LSE_LINEAR | storage)
    b.forward()
    do_back = storage != capi.HX_BAND_COMPRESSED      # (no Backward matrices on compressed planes)
    if do_back:
        b.backward()
    lp_end = b.lp_end()
    lp_start = b.lp_start() if do_back else None
    paths = b.best_trace()
    bd = capi.Batch(imgs, capi.HX_LSE_LINEAR) if storage == capi.HX_BAND_COMPRESSED else None
    if bd is not None:
        bd.forward()
        assert paths == bd.best_trace(), "batch %d: traceback on compressed planes" % bi
        bd.close()
    for k, (x, y, hmm, md) in enumerate(imgs):
        wf = c_oracle.forward(x, y, hmm, md, true_math=True)
        gf = b.read_matrix(k, 0)
        inside = np.isfinite(wf["cells"])
        # (sparse-envelope and band-compressed planes: cells outside the envelope are undefined - compare inside it)
        env = None if (storage == 0 or md < 0) else H.envelope_mask(cases[k])
        if env is None:
            assert np.array_equal(np.isneginf(wf["cells"]), np.isneginf(gf)), "batch %d job %d: -inf pattern" % (bi, k)
        else:
            assert np.array_equal(np.isneginf(wf["cells"][env]), np.isneginf(gf[env])), "batch %d job %d: -inf pattern inside the envelope" % (bi, k)
        dev = float(np.max(np.abs(wf["cells"][inside] - gf[inside]), initial=0.))
        assert dev < 1e-9, "batch %d job %d forward: %g" % (bi, k, dev)
        worst = max(worst, dev)
        if np.isfinite(wf["lp_end"]):
            assert abs(wf["lp_end"] - lp_end[k]) <= 1e-12 * abs(lp_end[k]), "lpEnd"
        else:
            assert lp_end[k] == wf["lp_end"] and paths[k] is None
        if do_back:
            wb = c_oracle.backward(x, y, hmm, md, true_math=True)
            gb = b.read_matrix(k, 1)
            if env is None:
                assert np.array_equal(np.isneginf(wb["cells"]), np.isneginf(gb)), "batch %d job %d backward -inf pattern" % (bi, k)
            else:
                assert np.array_equal(np.isneginf(wb["cells"][env]), np.isneginf(gb[env])), "batch %d job %d backward -inf pattern inside the envelope" % (bi, k)
            fin = np.isfinite(wb["cells"])
            dev = float(np.max(np.abs(wb["cells"][fin] - gb[fin]), initial=0.))
            assert dev < 1e-9, "batch %d job %d backward: %g" % (bi, k, dev)
            worst = max(worst, dev)
            if np.isfinite(wb["lp_start"]):
                assert abs(wb["lp_start"] - lp_start[k]) <= 1e-12 * abs(lp_start[k]), "lpStart"
            else:
                assert lp_start[k] == wb["lp_start"]
        n_jobs += 1
        n_cells += wf["cells"].shape[0] * wf["cells"].shape[1]
    b.close()
    print("batch %2d %2d jobs ok%s%s%s  (%.0f s)" % (bi, len(cases), " banded" if any_band else "",
          {0: "", capi.HX_SPARSE_ENVELOPE: " [sparse-envelope]", capi.HX_BAND_COMPRESSED: " [band-compressed]"}[storage], shape, time.time() - t0), flush=True)
print("all within tolerance: %d jobs, %d lattice cells, worst cell deviation %.2e" % (n_jobs, n_cells, worst))
