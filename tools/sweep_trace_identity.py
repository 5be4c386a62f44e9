"""Traceback identity of the device fills against the reference's table arithmetic (north_star: "traceback alignments
bit-identical"; reference src/forward.cpp:245-255,283-302).

For N random leaf pairs (lengths 50-2000, DNA / protein / 4-component mixture, banded and unbanded, symmetric branch
lengths included) the device-side best path (hx_batch_best_trace) of a fill in each arithmetic policy - exact, fast,
linear, trunc - is compared with oracle/trace_oracle.best_trace over the matrix of the pinned plain-C oracle
(oracle_fill.c, the reference's table operator with its truncation).  Reports, per policy, how many paths differ and
the largest relative lpEnd difference; the differing pairs are listed by their seed.

    python tools/sweep_trace_identity.py [n_pairs] [seed] [out.json]         (on the GPU box)

The CPU oracle runs in a process pool started before the device is initialised."""
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

MODELS = {"dna": "jc", "protein": "wag", "mixture": "prot4"}
_cache = {}


def model_of(kind):
    from historian_amd import hostmodel
    if kind not in _cache:
        _cache[kind] = hostmodel.RateModel.load(os.path.join(ROOT, "tests", "golden", "models", MODELS[kind] + ".json"))
    return _cache[kind]


def hmm_of(kind, tl, tr):
    from historian_amd import hostmodel
    key = (kind, tl, tr)
    if key not in _cache:
        _cache[key] = hostmodel.make_hmm(model_of(kind), tl, tr)
    return _cache[key]


def make_specs(n, seed):
    rng = np.random.default_rng(seed)
    specs = []
    times = [.05, .1, .2, .3, .5]
    for k in range(n):
        kind = ["dna", "protein", "protein", "mixture"][int(rng.integers(4))]
        lx = int(np.exp(rng.uniform(np.log(50), np.log(2000))))
        if kind == "mixture":
            lx = min(lx, 1200)          # (the CPU oracle's per-cell emission is 80 table look-ups)
        ly = lx if rng.random() < .5 else max(20, int(lx * rng.uniform(.8, 1.2)))
        tl = float(times[int(rng.integers(len(times)))])
        tr = tl if rng.random() < .4 else float(times[int(rng.integers(len(times)))])     # symmetric branches: 40 %+
        band = int(rng.choice([-1, -1, 20, 20, 5, 40]))
        sub = float(rng.choice([.05, .2, .4]))
        indel = float(rng.choice([.005, .02, .05]))
        specs.append(dict(id=k, seed=int(rng.integers(1 << 31)), kind=kind, lx=lx, ly=ly, tl=tl, tr=tr, band=band, sub=sub,
                          indel=indel))
    return specs


def job_of(spec):
    from historian_amd import workload
    return workload.leaf_pair(np.random.default_rng(spec["seed"]), model_of(spec["kind"]),
                              hmm_of(spec["kind"], spec["tl"], spec["tr"]), spec["lx"], spec["ly"], spec["band"],
                              spec["sub"], spec["indel"])


def cpu_reference(spec):
    """Worker: the reference's Forward fill (table arithmetic) and bestTrace on the CPU."""
    from oracle import c_oracle, trace_oracle
    x, y, h, md = job_of(spec)
    r = c_oracle.forward(x, y, h, md)
    if not np.isfinite(r["lp_end"]):
        return spec["id"], r["lp_end"], None
    path = trace_oracle.best_trace(x, y, h, md, r)
    return spec["id"], r["lp_end"], np.asarray(path, dtype=np.int32)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20261004
    out_path = sys.argv[3] if len(sys.argv) > 3 else None
    specs = make_specs(n, seed)
    workers = max(1, min(14, (os.cpu_count() or 2) - 2))
    pool = mp.get_context("spawn").Pool(workers)
    t0 = time.time()
    pending = pool.imap_unordered(cpu_reference, specs, chunksize=4)

    from historian_amd import capi
    from oracle import c_oracle
    capi.init(0, c_oracle.table())
    modes = {"exact": capi.HX_LSE_EXACT, "fast": capi.HX_LSE_FAST, "linear": capi.HX_LSE_LINEAR, "trunc": capi.HX_LSE_TRUNC}
    if os.environ.get("HX_SWEEP_MODES"):
        modes = {m: modes[m] for m in os.environ["HX_SWEEP_MODES"].split(",")}
    dev = {m: {} for m in modes}
    # device side: batches of pairs of one band class (a batch is banded or not), sorted by size
    order = sorted(range(n), key=lambda k: (specs[k]["band"] >= 0, specs[k]["lx"]))
    bs = 32
    for b0 in range(0, n, bs):
        ids = [k for k in order[b0:b0 + bs]]
        groups = {}
        for k in ids:
            groups.setdefault(specs[k]["band"] >= 0, []).append(k)
        for banded, gids in groups.items():
            jobs = [job_of(specs[k]) for k in gids]
            for m, flag in modes.items():
                storage = 0
                if banded:
                    storage = [capi.HX_BAND_COMPRESSED, capi.HX_SPARSE_ENVELOPE, 0][(b0 // bs) % 3]
                batch = capi.Batch(jobs, flag | storage)
                batch.forward()
                lp = batch.lp_end()
                paths = batch.best_trace()
                batch.close()
                for k, p, l in zip(gids, paths, lp):
                    dev[m][k] = (float(l), None if p is None else np.asarray(p, dtype=np.int32))
        print("device: %d of %d pairs (%.0f s)" % (min(b0 + bs, n), n, time.time() - t0), flush=True)

    ref = {}
    for i, (k, lp, path) in enumerate(pending):
        ref[k] = (lp, path)
        if (i + 1) % 100 == 0:
            print("cpu oracle: %d of %d pairs (%.0f s)" % (i + 1, n, time.time() - t0), flush=True)
    pool.close()

    report = {"pairs": n, "seed": seed, "cpu_workers": workers, "seconds": None, "modes": {}}
    for m in modes:
        differ, worst, zero = [], 0., 0
        for k in range(n):
            lp_ref, p_ref = ref[k]
            lp_dev, p_dev = dev[m][k]
            if p_ref is None:
                zero += 1
                assert p_dev is None and lp_dev == lp_ref
                continue
            worst = max(worst, abs(lp_dev - lp_ref) / abs(lp_ref))
            if p_dev is None or p_dev.shape != p_ref.shape or not np.array_equal(p_dev, p_ref):
                n_diff = -1 if (p_dev is None or p_dev.shape != p_ref.shape) else int((p_dev != p_ref).any(axis=1).sum())
                differ.append(dict(specs[k], cells_differing=n_diff, path_cells=int(len(p_ref))))
        report["modes"][m] = {"paths_differing": len(differ), "lp_end_max_rel_diff": worst, "zero_likelihood_pairs": zero,
                              "differing": differ}
        print("%-6s: %d of %d best paths differ from the reference's; lpEnd max rel. diff %.3g" % (m, len(differ), n, worst))
    report["seconds"] = time.time() - t0
    by = {}
    for s in specs:
        key = "%s/%s" % (s["kind"], "banded" if s["band"] >= 0 else "unbanded")
        by[key] = by.get(key, 0) + 1
    report["pairs_by_class"] = by
    report["symmetric_branch_pairs"] = sum(1 for s in specs if s["tl"] == s["tr"])
    if out_path:
        with open(out_path, "w") as f:
            json.dump(report, f, indent=1)
    capi.shutdown()


if __name__ == "__main__":
    main()
