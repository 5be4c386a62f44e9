"""CPU baselines of SURVEY section 8(d) on the host cores of the GPU box, for the two pair DPs:
(i) the plain-C oracle on one core, (ii) the Forward fill over the reference's own cell storage (a std::map per row,
oracle_fill_map.cpp) on one core, (iii) the plain-C oracle farmed over all cores (independent pairs; ctypes releases
the GIL).  Usage: cpu_baseline.py [threads]"""
import json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from historian_amd import hostmodel
from oracle import c_oracle, historian_oracle as ho, quickalign_oracle as q

threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
model = hostmodel.RateModel.load(os.path.join(ROOT, "tests", "golden", "models", "wag.json"))
a, c = len(model.alphabet), model.components()
hmm = hostmodel.make_hmm(model, 0.2, 0.3)
pi = np.asarray(model.root[0], dtype=float); pi /= pi.sum()
c_oracle.load()
L = 1000
pairs = []
for k in range(max(threads, 8)):
    rng = np.random.default_rng(1000 + k)
    xs, ys = bench.synth_pair(rng, pi, L)
    pairs.append((hostmodel.leaf_profile(xs, a, c), hostmodel.leaf_profile(ys, a, c), xs, ys))
omodel = ho.RateModel.from_file(os.path.join(ROOT, "tests", "golden", "models", "wag.json"))
omodel.sub_rate = [m.tolist() for m in omodel.sub_rate]
sc = q.QuickAlignScores(omodel, 0.5)

def fwd(k):
    x, y, _, _ = pairs[k]
    return c_oracle.forward(x, y, hmm, -1)["lp_end"]

def qa(k):
    _, _, xs, ys = pairs[k]
    return c_oracle.quickalign(xs.astype(np.int32), ys.astype(np.int32), a, sc.submat, sc, None)["score"]

out = {}
for name, fn, cells in (("forward", fwd, (L + 1) ** 2), ("viterbi", qa, L * L)):
    t0 = time.perf_counter(); fn(0); fn(1); one = (time.perf_counter() - t0) / 2
    n = len(pairs)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(fn, range(n)))
    allc = time.perf_counter() - t0
    out[name] = {"one_core_cells_per_s": cells / one, "all_cores_cells_per_s": n * cells / allc, "threads": threads,
                 "pairs": n, "length": L}
t0 = time.perf_counter()
lp_map = c_oracle.forward_map(pairs[0][0], pairs[0][1], hmm, -1)
out["forward"]["one_core_map_storage_cells_per_s"] = (L + 1) ** 2 / (time.perf_counter() - t0)
assert lp_map == fwd(0)
print(json.dumps(out))
