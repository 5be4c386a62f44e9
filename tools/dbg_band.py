import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from historian_amd import capi
from oracle import c_oracle
from tests import helpers as H
capi.init(0, c_oracle.table())
for case in (H.leaf_case(401, 70, 66, band=5), H.leaf_case(402, 200, 90, band=12)):
    img = H.job_images(case)
    for flags in (0, capi.HX_SPARSE_ENVELOPE, capi.HX_BAND_COMPRESSED):
        b = capi.Batch([img], capi.HX_LSE_LINEAR | flags); b.forward()
        want = c_oracle.forward(*img, true_math=True)
        mf = b.read_matrix(0, 0)
        bad = np.argwhere(np.isneginf(want["cells"]) != np.isneginf(mf))
        print("flags", flags, "shape", mf.shape, "n bad", len(bad))
        for q in bad[:12]: print("  ", tuple(q), want["cells"][tuple(q)], mf[tuple(q)])
        inside = np.isfinite(want["cells"]) & np.isfinite(mf)
        print("  max dev", np.max(np.abs(want["cells"][inside]-mf[inside])), "lp", b.lp_end()[0], want["lp_end"])
        b.close()
