import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
os.environ["HX_BAND2"] = "1"; os.environ.setdefault("HX_BAND2_NW", "1")
from historian_amd import capi
from oracle import c_oracle
from tests import helpers as H
capi.init(0, c_oracle.table())
AA = "arndcqeghilkmfpstwyv"
cases = [H.leaf_case(401, 70, 66, band=5), H.leaf_case(402, 200, 90, band=12), H.leaf_case(403, 130, 150, band=3),
         H.leaf_case(404, 300, 330, alphabet=AA, jc=False, band=20)]
imgs = [H.job_images(f) for f in cases]
for pol, tm in ((capi.HX_LSE_TRUNC, 2), (capi.HX_LSE_LINEAR, 1)):
    b = capi.Batch(imgs, pol)
    print("shared", b.shared_wavefront_pairs())
    b.forward()
    for k, (x, y, hmm, md) in enumerate(imgs):
        want = c_oracle.forward(x, y, hmm, md, true_math=tm)["cells"]
        got = b.read_matrix(k, 0)
        bad = np.argwhere(np.isneginf(want) != np.isneginf(got))
        fin = np.isfinite(want) & np.isfinite(got)
        print("policy", pol, "job", k, "inf-pattern mismatches", len(bad), bad[:12].tolist(), "max diff", np.max(np.abs(want[fin] - got[fin]), initial=0))
        for (i, j, s) in bad[:6]:
            print("    cell", i, j, s, "want", want[i, j, s], "got", got[i, j, s])
    b.close()
