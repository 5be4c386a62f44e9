"""Per-branch pair DPs (next row N4, hx_branch.hip) on a batch the size of a refinement sweep: every branch of a 64-leaf tree
(126 branches), parent and child profiles of `length` positions over the 20-letter alphabet, band 20 around the diagonal or
none.  Prints kernel time and cells/s (24 B/cell: three fp64 states) for the Viterbi and the log_sum_exp form.
    python tools/branch_bench.py [length] [branches] [band]       (on the GPU box)"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from historian_amd import capi, hostmodel

length = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 126
band = int(sys.argv[3]) if len(sys.argv) > 3 else -1
capi.init(0, hostmodel.lse_table())
rng = np.random.default_rng(3)
A = 20
jobs = []
T = [[math.log(v) if v > 0 else -math.inf for v in row] for row in
     [[.9 * .9, .1, .9 * .1, .9], [.4 * .9, .6, .4 * .1, .4], [.4, 0., .6, .4]]]
for k in range(n):
    x = np.log(rng.dirichlet(np.ones(A) * .3, size=length)).reshape(length, 1, A)
    y = np.log(rng.dirichlet(np.ones(A) * .3, size=length)).reshape(length, 1, A)
    env = np.arange(length + 1, dtype=np.int32) if band >= 0 else None
    jobs.append((x, y, np.full(length, math.log(1. / A)), T, env, env, band))
b = capi.BranchBatch(jobs)
cells = b.total_cells()
if band >= 0:
    cells = n * sum(min(length, i + band) - max(0, i - band) + 1 for i in range(length + 1))
for viterbi in (True, False):
    b.run(viterbi=viterbi)
    b.lp_end()
    t0 = time.perf_counter()
    b.run(viterbi=viterbi)
    lp = b.lp_end()
    wall = time.perf_counter() - t0
    ms = b.kernel_ms()
    print("%d branches of %d x %d, band %s, %s: fill kernel %.2f ms = %.2f Gcell/s (%.3f of the HBM roofline at 24 B/cell), with emission "
          "and clearing %.2f ms; lpEnd[0] %.4f" % (n, length, length, band if band >= 0 else "none", "Viterbi" if viterbi else "log_sum_exp", ms,
                                                   cells / ms / 1e6, cells * 24 / (ms * 1e-3) / 8e12, wall * 1e3, lp[0]))
b.close()
