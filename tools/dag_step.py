"""Per-step latency of the general-profile Forward pipeline: leaf pairs pushed through it (HX_FORCE_DAG=1)."""
import os, sys
os.environ["HX_FORCE_DAG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from historian_amd import capi
from oracle import c_oracle
from tests import helpers as H
capi.init(0, c_oracle.table())
cases = ((60, 4000, 1), (60, 4000, 256), (60, 4000, 1024), (500, 4000, 1), (500, 4000, 256))
if len(sys.argv) > 3:
    cases = ((int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])),)
for (lx, ly, n) in cases:
    f = H.leaf_case(5, lx, ly, alphabet="ACDEFGHIKLMNPQRSTVWY", jc=False)
    img = H.job_images(f)
    for name, flags in (("exact", capi.HX_LSE_EXACT), ("fast", capi.HX_LSE_FAST)):
        b = capi.Batch([img] * n, flags)
        b.forward(); b.sync(); b.forward(); b.sync()
        ms = b.kernel_ms(0)
        steps = ly + 1 + 63 + 64 * ((lx + 1 + 63) // 64 - 1)
        print("%4d x %4d  jobs %5d  %-5s  %8.3f ms   %6.2f us/step (critical path %d steps)  %7.2f Gcell/s" %
              (lx, ly, n, name, ms, ms * 1e3 / steps, steps, b.total_cells() / ms / 1e6))
        b.close()
