"""Guide-alignment Viterbi throughput: a batch of independent pair DPs
(QuickAlignMatrix fills, full envelope) on one GPU; cells/s, fraction of the HBM roofline at the
algorithmic 24 B/cell (3 fp64 states written once), and the plain-C oracle on one host core beside it.
Usage: quickalign_bench.py [pairs] [length] [steps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from historian_amd import capi, hostmodel
from oracle import c_oracle, historian_oracle as ho, quickalign_oracle as q

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
length = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
model = ho.RateModel.from_file(os.path.join(ROOT, "tests", "golden", "models", "wag.json"))
model.sub_rate = [m.tolist() for m in model.sub_rate]
sc = q.QuickAlignScores(model, 0.5)
scores = [getattr(sc, n) for n in capi.QuickBatch.SCORE_NAMES]
a = len(model.alphabet)
pi = np.asarray(model.ins_prob[0], dtype=float); pi /= pi.sum()
sys.path.insert(0, ROOT)
import bench
jobs = []
for k in range(pairs):
    rng = np.random.default_rng(5000 + k)
    x, y = bench.synth_pair(rng, pi, length)
    jobs.append((x.astype(np.int32), y.astype(np.int32), a, sc.submat, scores, None))
capi.init(0, c_oracle.table())
b = capi.QuickBatch(jobs)
b.run(); b.results()
ms = []
t0 = time.perf_counter()
for _ in range(steps):
    b.run()
    ms.append(b.kernel_ms())
score, xe, ye = b.results()
dt = time.perf_counter() - t0
cells = b.total_cells()
k_ms = float(np.mean(ms))
n_cpu = min(pairs, 8)
t1 = time.perf_counter()
for k in range(n_cpu):
    r = c_oracle.quickalign(jobs[k][0], jobs[k][1], a, sc.submat, sc, None)
    assert r["score"] == score[k] and (r["x_end"], r["y_end"]) == (int(xe[k]), int(ye[k]))
cpu_dt = time.perf_counter() - t1
print(json.dumps({"metric": "guide-alignment Viterbi cells/s", "value": cells * steps / dt, "unit": "cells/s",
                  "config": {"workload": "%d independent %dx%d protein pairs, WAG, t=0.5, full DiagonalEnvelope" % (pairs, length, length),
                             "cells_per_step": cells},
                  "ms_per_step": dt / steps * 1e3, "dtype": "f64",
                  "roofline": {"bound": "hbm", "achieved": cells * 24 / (k_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                               "frac": cells * 24 / (k_ms * 1e-3) / 1e9 / 8000.0, "kernel": "hx::k_quickalign", "kernel_ms": k_ms,
                               "bytes_per_cell": 24},
                  "cpu_baseline": {"value": n_cpu * length * length / cpu_dt, "unit": "cells/s", "cores": 1, "kind": "port",
                                   "sample": "first %d pairs, oracle/oracle_quickalign.c, %.1f s (scores and end cells equal the GPU's)" % (n_cpu, cpu_dt)}}))
b.close()
