"""Throughput of the general (DAG) kernels on real internal-node profiles: the gp120 family's
internal nodes (oracle-built profiles), replicated into a batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from historian_amd import capi
from oracle import c_oracle, historian_oracle as ho
from tests import helpers as H, recon_helpers as R

G = os.path.join(ROOT, "tests", "golden", "reference_data") + os.sep
LG = os.path.join(ROOT, "tests", "golden", "models", "lg.json")
tree, seqs, guide = R.load_family(G + "gp120.tree.nh", G + "gp120.fa", G + "gp120.guide.fa")
res, rows = R.oracle_reconstruct(LG, tree, seqs, guide, max_distance_from_guide=20)
model = ho.RateModel.from_file(LG); model.sub_rate = [m.tolist() for m in model.sub_rate]
closest = ho.closest_leaves(tree)
capi.init(0, c_oracle.table())
root = tree.root()
for band in ((-1,) if "unbanded" in sys.argv else (20, -1)):
    imgs = []
    env_cells = 0
    for node in range(tree.nodes()):
        if tree.is_leaf(node): continue
        lc, rc = tree.child[node]
        if tree.is_leaf(lc) and tree.is_leaf(rc): continue
        lp = ho.ProbModel(model, tree.branch_length[lc], [ho.sub_prob_matrix_ss(sr, tree.branch_length[lc]) for sr in model.sub_rate])
        rp = ho.ProbModel(model, tree.branch_length[rc], [ho.sub_prob_matrix_ss(sr, tree.branch_length[rc]) for sr in model.sub_rate])
        hmm = ho.PairHMM(lp, rp, model.ins_prob)
        env = ho.GuideAlignmentEnvelope(guide, closest[lc], closest[rc], band) if band >= 0 else ho.GuideAlignmentEnvelope()
        f = ho.ForwardMatrix(res["prof"][lc], res["prof"][rc], hmm, node, env, fill=False)
        imgs.append(H.job_images(f))
        if band >= 0:
            # in-envelope cells of the pair (what a banded fill computes): vectorised |envelope coordinate difference| <= band,
            # plus the always-stored edge row / column
            img = imgs[-1]
            ex, ey = np.asarray(img[0].env_pos[:f.x_size - 1]), np.asarray(img[1].env_pos[:f.y_size - 1])
            inside = np.abs(ex[:, None] - ey[None, :]) <= band
            inside |= np.asarray(f.x_near_start[:f.x_size - 1], dtype=bool)[:, None] | np.asarray(f.y_near_end[:f.y_size - 1], dtype=bool)[None, :]
            env_cells += int(inside.sum())
        else:
            env_cells += (f.x_size - 1) * (f.y_size - 1)
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    if "perjob" in sys.argv:
        for k, img in enumerate(imgs):
            b = capi.Batch([img], capi.HX_LSE_LINEAR if "linear" in sys.argv else capi.HX_LSE_FAST)
            b.forward(); b.sync(); b.forward(); b.sync()
            x, y = img[0], img[1]
            steps = (y.n_states - 1) + 63 + 72 * ((x.n_states - 1 + 63) // 64 - 1)
            fwd_ms = b.kernel_ms(0)
            b.backward(); b.sync(); b.backward(); b.sync()
            print("  job %d  %4d x %4d  fast forward %7.3f ms   %5.2f us/step over %d critical-path steps | backward %7.3f ms" %
                  (k, x.n_states, y.n_states, fwd_ms, fwd_ms * 1e3 / steps, steps, b.kernel_ms(1)))
            b.close()
        continue
    sizes = [(x.n_states, y.n_states) for x, y, _, _ in imgs]
    print("band", band, "jobs", len(imgs) * reps, "sizes", sizes)
    for name, flags in (("exact", capi.HX_LSE_EXACT), ("fast", capi.HX_LSE_FAST), ("linear", capi.HX_LSE_LINEAR)) + (() if "fwdonly" in sys.argv else (("barrier-exact", capi.HX_FORCE_GENERIC),)):
        batch = capi.Batch(imgs * reps, flags)
        batch.forward(); batch.sync()
        batch.forward(); batch.sync()
        cells = batch.total_cells()
        in_env = env_cells * reps
        fms = batch.kernel_ms(0)
        if "fwdonly" in sys.argv:
            print("  %-14s forward %8.3f ms %7.2f Gcell/s over the lattice, %6.2f Gcell/s in-envelope (%d of %d cells)" %
                  (name, fms, cells / fms / 1e6, in_env / fms / 1e6, in_env, cells))
            batch.close()
            continue
        batch.backward(); batch.sync()
        batch.backward(); batch.sync()
        bms = batch.kernel_ms(1)
        print("  %-14s forward %8.3f ms %7.2f Gcell/s   backward %8.3f ms %7.2f Gcell/s (lattice cells %d; in-envelope %d: forward %.2f, backward %.2f Gcell/s)" %
              (name, fms, cells / fms / 1e6, bms, cells / bms / 1e6, cells, in_env, in_env / fms / 1e6, in_env / bms / 1e6))
        batch.close()
