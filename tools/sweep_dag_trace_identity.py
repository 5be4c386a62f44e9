"""Traceback identity on GENERAL profiles (internal tree nodes): the device-side best path (hx_batch_best_trace) of a fill in the
fast table policy - what the default mode runs on general profiles - against the exact policy's, whose cells are the pinned
oracle's bit for bit (tests/test_gpu_parity.py).  N random pairs of internal-node profiles (oracle-built from sampled paths of
two leaf pairs: tests/helpers.dag_case; 30-300 ancestral residues, 3-25 samples, DNA and protein, one- and two-component models,
every fourth pair with a band).  Reports how many best paths differ, how many walks of the fast fill met a near tie (hx_batch_best_trace_ties:
the pairs a caller refills under the exact policy), whether every differing pair is among them, and the largest relative lpEnd
difference.

    python tools/sweep_dag_trace_identity.py [n_pairs] [seed] [out.json]         (on the GPU box)"""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from historian_amd import capi  # noqa: E402
from oracle import c_oracle  # noqa: E402
from tests import helpers as H  # noqa: E402


def main():
    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    out = sys.argv[3] if len(sys.argv) > 3 else None
    rng = random.Random(seed)
    capi.init(0, c_oracle.table())
    t0 = time.time()
    differing, unflagged, worst, done, cells, flagged = [], [], 0.0, 0, 0, 0
    batch = []

    def flush():
        nonlocal worst, done, cells, flagged
        if not batch:
            return
        imgs = [H.job_images(f) for _, f in batch]
        res = {}
        for name, flags in (("exact", capi.HX_LSE_EXACT), ("fast", capi.HX_LSE_FAST)):
            b = capi.Batch(imgs, flags)
            b.forward()
            res[name] = (b.lp_end(), b.best_trace(), b.best_trace_ties())
            cells += b.total_cells() if name == "exact" else 0
            b.close()
        for k, (tag, _) in enumerate(batch):
            le, lf = res["exact"][0][k], res["fast"][0][k]
            if np.isfinite(le):
                worst = max(worst, abs(lf - le) / abs(le))
            flagged += int(res["fast"][2][k] != 0)
            if res["exact"][1][k] != res["fast"][1][k]:
                differing.append(tag)
                if not res["fast"][2][k]:          # a difference the near-tie flag of the fast walk did not announce
                    unflagged.append(tag)
            done += 1
        batch.clear()

    for k in range(n_pairs):
        s = rng.randint(1, 10 ** 6)
        n = rng.choice([30, 60, 100, 150, 220, 300])
        samples = rng.choice([3, 6, 10, 15, 25])
        comps = rng.choice([1, 1, 2])
        band = rng.choice([None, None, None, 5])
        protein = rng.random() < .5           # (protein pairs have too many distinct columns for a class-pair table: per-cell emission terms)
        batch.append(({"seed": s, "n": n, "samples": samples, "components": comps, "band": band, "protein": protein},
                      H.dag_case(s, n=n, samples=samples, components=comps, band=band,
                                 alphabet="arndcqeghilkmfpstwyv" if protein else "ACGT")))
        if len(batch) == 16:
            flush()
            print("  %d pairs, %d differing best paths, %.0f s" % (done, len(differing), time.time() - t0), flush=True)
    flush()
    report = {"pairs": done, "lattice_cells": int(cells), "differing_best_paths_fast_vs_exact": len(differing), "differing": differing,
              "pairs_flagged_near_tie_by_the_fast_walk": flagged, "differing_and_not_flagged": len(unflagged), "not_flagged": unflagged,
              "lp_end_max_rel_diff_fast_vs_exact": worst, "seed": seed}
    print(json.dumps(report))
    if out:
        with open(out, "w") as fh:
            json.dump(report, fh, indent=1)


if __name__ == "__main__":
    main()
