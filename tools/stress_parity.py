"""Randomised bit-exact parity sweep (exact mode) of the Forward/Backward fills against the plain-C oracle:
leaf pairs and internal-node (DAG) pairs, with and without a band, mixed batches, default and sparse-envelope
storage.  Not part of the test suite (minutes); run on the GPU box:  python tools/stress_parity.py [n_batches]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from historian_amd import capi
from oracle import c_oracle
from tests import helpers as H

capi.init(0, c_oracle.table())
n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 20261004)
t0 = time.time()
n_jobs = n_cells = 0
for bi in range(n_batches):
    kind = rng.choice(["leaf", "dag", "mixed", "leaf-many"])
    cases = []
    if kind in ("leaf", "mixed", "leaf-many"):
        for _ in range(70 if kind == "leaf-many" else rng.randint(1, 6)):
            lx, ly = rng.choice([0, 1, 5, 63, 64, 65, 130, 200, 333]), rng.choice([0, 2, 7, 64, 100, 129, 260])
            if kind == "leaf-many":
                lx, ly = rng.randint(0, 150), rng.randint(0, 150)
            band = rng.choice([None, None, 0, 1, 3, 8, 20]) if kind != "leaf-many" else rng.choice([0, 2, 5, 11])
            prot = rng.random() < .3
            cases.append(H.leaf_case(rng.randint(1, 10 ** 6), lx, ly, alphabet="arndcqeghilkmfpstwyv" if prot else "ACGT",
                                     jc=not prot, band=band))
    if kind in ("dag", "mixed"):
        for _ in range(rng.randint(1, 4)):
            cases.append(H.dag_case(rng.randint(1, 10 ** 6), n=rng.choice([6, 14, 30, 70, 120]), band=rng.choice([None, None, 0, 2, 5]),
                                    samples=rng.choice([2, 6, 15]), keep_all=rng.random() < .2,
                                    components=rng.choice([1, 1, 2])))
    imgs = [H.job_images(f) for f in cases]
    sparse = rng.random() < .5
    b = capi.Batch(imgs, capi.HX_KEEP_BACKWARD | (capi.HX_SPARSE_ENVELOPE if sparse else 0))
    b.forward(); b.backward()
    lp_end, lp_start = b.lp_end(), b.lp_start()
    for k, (x, y, hmm, md) in enumerate(imgs):
        wf, wb = c_oracle.forward(x, y, hmm, md), c_oracle.backward(x, y, hmm, md)
        gf, gb = b.read_matrix(k, 0), b.read_matrix(k, 1)
        if sparse and md >= 0:       # cells outside the envelope are undefined in this mode
            inside = np.isfinite(wf["cells"]).any(axis=2) | np.isfinite(wb["cells"]).any(axis=2)
            gf, gb = np.where(inside[:, :, None], gf, -np.inf), np.where(inside[:, :, None], gb, -np.inf)
            wfc, wbc = np.where(inside[:, :, None], wf["cells"], -np.inf), np.where(inside[:, :, None], wb["cells"], -np.inf)
        else:
            wfc, wbc = wf["cells"], wb["cells"]
        H.assert_same_bits(gf, wfc, "batch %d (%s) job %d forward" % (bi, kind, k))
        H.assert_same_bits(gb, wbc, "batch %d (%s) job %d backward" % (bi, kind, k))
        H.assert_same_bits([lp_end[k]], [wf["lp_end"]], "lpEnd")
        H.assert_same_bits([lp_start[k]], [wb["lp_start"]], "lpStart")
        n_jobs += 1
        n_cells += wf["cells"].shape[0] * wf["cells"].shape[1]
    b.close()
    print("batch %2d %-9s %3d jobs ok%s  (%.0f s)" % (bi, kind, len(cases), " [sparse-envelope]" if sparse else "", time.time() - t0), flush=True)
print("all bit-identical: %d jobs, %d lattice cells" % (n_jobs, n_cells))
