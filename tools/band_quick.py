"""Kernel time of the banded leaf fill for quick A/B runs: 64 generated 2 x L pairs replicated to the batch size (the pairs
of a batch are independent, so the kernel's time is that of bench.py's batch of distinct pairs), no checks, no CPU leg.
usage: python tools/band_quick.py [pairs=4096] [mode=trunc] [band=20] [length=2000] [steps=5]
Environment as the library reads it (HX_BAND2, HX_BAND2_NW, HX_BAND_PPW, HX_LIB_PATH ...); prints one line."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from historian_amd import capi, hostmodel, workload  # noqa: E402


def main():
    a = sys.argv[1:]
    pairs = int(a[0]) if len(a) > 0 else 4096
    mode = a[1] if len(a) > 1 else "trunc"
    band = int(a[2]) if len(a) > 2 else 20
    length = int(a[3]) if len(a) > 3 else 2000
    steps = int(a[4]) if len(a) > 4 else 5
    flags = {"trunc": capi.HX_LSE_TRUNC, "linear": capi.HX_LSE_LINEAR, "fast": capi.HX_LSE_FAST, "exact": capi.HX_LSE_EXACT}[mode]
    model = hostmodel.RateModel.load(os.path.join(ROOT, "tests", "golden", "models", "wag.json"))
    hmm = hostmodel.make_hmm(model, 0.2, 0.3)
    capi.init(0)
    base = [workload.leaf_pair(np.random.default_rng(1000 + k), model, hmm, length, band=band) for k in range(min(64, pairs))]
    env = [workload.in_envelope_cells(t[0].env_pos, t[1].env_pos, band) if band >= 0 else 0 for t in base]
    triples = [base[k % len(base)] for k in range(pairs)]
    cells = sum(env[k % len(base)] for k in range(pairs))
    b = capi.Batch(triples, flags | (capi.HX_BAND_COMPRESSED if band >= 0 else 0))
    if band < 0:
        cells = b.total_cells()
    b.forward()
    ms = []
    for _ in range(steps):
        b.forward()
        ms.append(b.kernel_ms(0))
    lp = b.lp_end()
    shared = b.shared_wavefront_pairs()
    b.close()
    t = float(np.median(ms))
    tag = " ".join("%s=%s" % (k, os.environ[k]) for k in sorted(os.environ) if k.startswith("HX_"))
    print("pairs %d mode %s band %d: %.3f ms (min %.3f)  %.1f Gcell/s  frac %.3f  shared-wavefront pairs %d  lpEnd[0] %.6f  %s"
          % (pairs, mode, band, t, min(ms), cells / t / 1e6, cells * 40 / (t * 1e-3) / 8e12, shared, lp[0], tag), flush=True)


if __name__ == "__main__":
    main()
