import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from historian_amd import capi
from oracle import c_oracle
from tests import helpers as H
capi.init(0, c_oracle.table())
aa = "arndcqeghilkmfpstwyv"
for f in [H.leaf_case(301, 40, 45), H.leaf_case(305, 63, 64, alphabet=aa, jc=False, tl=.3, tr=.2), H.leaf_case(306, 100, 130),
          H.leaf_case(308, 250, 200, alphabet=aa, jc=False), H.leaf_case(309, 500, 300), H.leaf_case(310, 1100, 700, alphabet=aa, jc=False, tl=.2, tr=.3)]:
    img = [H.job_images(f)]
    be = capi.Batch(img); bf = capi.Batch(img, capi.HX_LSE_FAST)
    be.forward(); bf.forward()
    me, mf = be.read_matrix(0, 0), bf.read_matrix(0, 0)
    same_inf = np.array_equal(np.isneginf(me), np.isneginf(mf))
    fin = np.isfinite(me) & np.isfinite(mf)
    d = np.abs(me[fin] - mf[fin])
    print(me.shape, "inf pattern same", same_inf, "nan", int(np.isnan(mf).sum()), "max dev %.3e" % d.max(), "lpEnd", be.lp_end()[0], bf.lp_end()[0], flush=True)
    if not same_inf:
        bad = np.argwhere(np.isneginf(me) != np.isneginf(mf))
        print(bad[:10], me[tuple(bad[0])], mf[tuple(bad[0])])
    be.close(); bf.close()
