import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from historian_amd import capi
from oracle import c_oracle
from tests import helpers as H
capi.init(0, c_oracle.table())
seeds = [int(a) for a in sys.argv[1:]] or [32]
for seed in seeds:
    f = H.dag_case(seed)
    img = H.job_images(f)
    x, y, hmm, md = img
    want = c_oracle.forward(x, y, hmm, md, true_math=True)["cells"]
    b = capi.Batch([img], capi.HX_LSE_LINEAR); b.forward(); got = b.read_matrix(0, 0); b.close()
    bad = np.argwhere(np.isneginf(want) != np.isneginf(got))
    print("seed", seed, "shape", want.shape, "mismatching -inf pattern:", len(bad))
    for i, j, s in bad[:12]:
        print("  cell", i, j, "state", s, "want", want[i, j, s], "got", got[i, j, s])
    if len(bad):
        i, j, s = bad[0]
        print("  x state", i, "null", bool(x.is_null[i]), "in:", [(int(x.trans_src[t]), float(x.trans_lp[t])) for t in x.in_idx[x.in_off[i]:x.in_off[i+1]]])
        print("  y state", j, "null", bool(y.is_null[j]), "in:", [(int(y.trans_src[t]), float(y.trans_lp[t])) for t in y.in_idx[y.in_off[j]:y.in_off[j+1]]])
    fin = np.isfinite(want) & np.isfinite(got)
    print("  max abs diff on finite cells", np.max(np.abs(want[fin] - got[fin]), initial=0.))
