"""Forward and Backward fill kernels on a banded leaf batch (2x2000 aa, WAG, band 20), in-envelope cells counted."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from historian_amd import capi, hostmodel, workload
model = hostmodel.RateModel.load(os.path.join(ROOT, "tests", "golden", "models", "wag.json"))
capi.init(0)
hmm = hostmodel.make_hmm(model, .2, .3)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 512
band = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tr, cells = [], 0
for k in range(P):
    rng = np.random.default_rng(1000 + k)
    x, y, h, md = workload.leaf_pair(rng, model, hmm, 2000, band=band)
    tr.append((x, y, h, md))
    cells += workload.in_envelope_cells(x.env_pos, y.env_pos, band) if band >= 0 else 2001 * 2001
for mode in ("linear", "fast", "exact"):
    b = capi.Batch(tr, {"linear": capi.HX_LSE_LINEAR, "fast": capi.HX_LSE_FAST, "exact": 0}[mode] | capi.HX_KEEP_BACKWARD | capi.HX_SPARSE_ENVELOPE)
    b.forward(); b.backward(); b.sync()
    b.forward(); b.backward(); b.sync()
    print("%-7s band %d, %d pairs, %d in-envelope cells: forward %.3f ms %.1f Gcell/s | backward %.3f ms %.1f Gcell/s | lpEnd %.6f lpStart %.6f" %
          (mode, band, P, cells, b.kernel_ms(0), cells / b.kernel_ms(0) / 1e6, b.kernel_ms(1), cells / b.kernel_ms(1) / 1e6, b.lp_end()[0], b.lp_start()[0]))
    b.close()
