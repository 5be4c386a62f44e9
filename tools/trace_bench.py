"""Device-side best-path traceback on the headline workload: Forward fill of a batch of
independent leaf pairs, then hx_batch_best_trace; reports the traceback time next to the fill time and to what
copying the matrices to the host would cost.  Usage: trace_bench.py [pairs] [length] [band]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from historian_amd import capi, hostmodel
import bench

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
length = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
band = int(sys.argv[3]) if len(sys.argv) > 3 else -1
model = hostmodel.RateModel.load(os.path.join(ROOT, "tests", "golden", "models", "wag.json"))
a, c = len(model.alphabet), model.components()
capi.init(0, hostmodel.lse_table())
hmm = hostmodel.make_hmm(model, .2, .3)
pi = np.asarray(model.root[0], dtype=float); pi /= pi.sum()
triples = []
for k in range(pairs):
    rng = np.random.default_rng(1000 + k)
    if band < 0:
        xs, ys = bench.synth_pair(rng, pi, length)
        triples.append((hostmodel.leaf_profile(xs, a, c), hostmodel.leaf_profile(ys, a, c), hmm, -1))
    else:
        xs, ys, (xrow, yrow) = bench.synth_pair(rng, pi, length, want_guide=True)
        ex, ey = bench.envelope_coordinates(xrow, yrow)
        triples.append((hostmodel.leaf_profile(xs, a, c, ex), hostmodel.leaf_profile(ys, a, c, ey), hmm, band))
out = {}
for mode, flags in (("linear", capi.HX_LSE_LINEAR), ("fast", capi.HX_LSE_FAST), ("exact", 0)):
    b = capi.Batch(triples, flags | (0 if band < 0 else capi.HX_BAND_COMPRESSED if mode == "linear" else capi.HX_SPARSE_ENVELOPE))
    b.forward(); b.sync()
    t0 = time.perf_counter(); b.forward(); b.sync(); t_fill = time.perf_counter() - t0
    b.best_trace(raw=True)
    t0 = time.perf_counter(); cells, n_cells = b.best_trace(raw=True); t_trace = time.perf_counter() - t0
    t0 = time.perf_counter(); m = b.read_matrix(0, 0); t_read = time.perf_counter() - t0
    steps = int(n_cells.sum())
    out[mode] = {"fill_ms": t_fill * 1e3, "best_trace_ms": t_trace * 1e3, "path_cells": steps,
                 "us_per_step": t_trace * 1e6 / (steps / pairs), "read_one_matrix_ms": t_read * 1e3,
                 "matrix_bytes_all_pairs": int(b.total_cells()) * 40}
    b.close()
print(json.dumps({"workload": "%d leaf pairs %dx%d, WAG, band %d" % (pairs, length, length, band), **out}))
