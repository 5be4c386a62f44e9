"""Level batching in the reconstruction driver: a balanced synthetic family through bin/hxrecon with
`batch 1` (every ready node's fill in one device batch) and `batch 0` (one node at a time, as the
reference does).  Usage: recon_batch_bench.py [n_leaves] [length] [model] [guide]
With `guide` the unbanded reconstruction's leaf rows become the guide alignment of a second run with the reference's
default band of 20: what `historian` does after its own guide-alignment step (every fill banded)."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import recon_helpers as R
n_leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 32
length = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
model = sys.argv[3] if len(sys.argv) > 3 else "wag"
MODEL = os.path.join(ROOT, "tests", "golden", "models", model + ".json")
alphabet = "arndcqeghilkmfpstwyv"
tree, seqs = R.balanced_family(n_leaves, length, alphabet, seed=21, branch=.05)
exe = os.path.join(ROOT, "historian_amd", "bin", "hxrecon")
results = {}
with tempfile.TemporaryDirectory() as d:
    for mode in ("linear", "fast", "exact"):
        for batch in (1, 0):
            job = os.path.join(d, "job_%d.txt" % batch)
            R.write_job(job, MODEL, tree, seqs, {}, os.path.join(d, "s.fa"), os.path.join(d, "g.fa"), samples=10, batch=batch)
            env = dict(os.environ, HX_TIMING="1")
            if mode != "exact":
                env["HX_FILL_MODE"] = mode
            out = subprocess.run([exe, job], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=1500)
            assert out.returncode == 0, out.stderr.decode()[-2000:]
            results[(mode, batch)] = out.stdout
            print("%d leaves x %d residues, %s, %s, batch %d" % (n_leaves, length, model, mode, batch))
            lines = out.stderr.decode().strip().splitlines()
            for line in lines if os.environ.get("HX_TIMING_LEVELS") else lines[-2:]:
                print("   ", line)
        assert results[(mode, 1)] == results[(mode, 0)], "batched and sequential reconstructions differ"
print("batched == sequential output: identical")
if len(sys.argv) > 4 and sys.argv[4] == "guide":
    rows = {}
    for line in results[("linear", 1)].decode().splitlines():
        f = line.split()
        if f and f[0] == "row":
            rows[f[2]] = f[3] if len(f) > 3 else ""
    guide = {n: [c not in "-.*" for c in rows[nm]] for n, (nm, s) in seqs.items()}
    for n, (nm, s) in seqs.items():
        assert sum(guide[n]) == len(s), nm
    with tempfile.TemporaryDirectory() as d:
        for mode in ("linear", "fast", "exact"):
            for batch in (1, 0):
                job = os.path.join(d, "job_%d.txt" % batch)
                R.write_job(job, MODEL, tree, seqs, guide, os.path.join(d, "s.fa"), os.path.join(d, "g.fa"), samples=10, batch=batch, band=20)
                env = dict(os.environ, HX_TIMING="1")
                if mode != "exact":
                    env["HX_FILL_MODE"] = mode
                out = subprocess.run([exe, job], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=1500)
                assert out.returncode == 0, out.stderr.decode()[-2000:]
                results[(mode, batch, "g")] = out.stdout
                print("%d leaves x %d residues, %s, %s, batch %d, guide + band 20" % (n_leaves, length, model, mode, batch))
                lines = out.stderr.decode().strip().splitlines()
                for line in lines if os.environ.get("HX_TIMING_LEVELS") else lines[-2:]:
                    print("   ", line)
            assert results[(mode, 1, "g")] == results[(mode, 0, "g")], "batched and sequential reconstructions differ"
    print("banded: batched == sequential output: identical")
