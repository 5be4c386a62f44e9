import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from historian_amd import capi, hostmodel
import torch
model = hostmodel.RateModel.load(os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + '/tests/golden/models/wag.json')
capi.init(0)
hmm = hostmodel.make_hmm(model, .2, .3)
pi = np.asarray(model.root[0]); pi /= pi.sum()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tr = []
for k in range(P):
    rng = np.random.default_rng(1000 + k)
    xs, ys = bench.synth_pair(rng, pi, 2000)
    tr.append((hostmodel.leaf_profile(xs, 20), hostmodel.leaf_profile(ys, 20), hmm, -1))
for mode in ("linear", "fast", "exact"):
    b = capi.Batch(tr, {"linear": capi.HX_LSE_LINEAR, "fast": capi.HX_LSE_FAST, "exact": 0}[mode] | capi.HX_KEEP_BACKWARD)
    b.forward(); b.backward(); b.sync()
    b.forward(); b.backward(); b.sync()
    cells = b.total_cells()
    print(mode, "fwd kernel ms", b.kernel_ms(0), "Gcell/s", cells / b.kernel_ms(0) / 1e6, "| bwd kernel ms", b.kernel_ms(1), "Gcell/s", cells / b.kernel_ms(1) / 1e6,
          "| lpEnd", b.lp_end()[0], "lpStart", b.lp_start()[0])
    n, cellsabove = b.posterior_scan(0, .9, cap=1 << 16)
    print("  posterior>.9 cells in pair 0:", n)
    b.close()
