"""The posterior-profile mode of the reconstruction (reference `-profminpost`: Forward + Backward at every node, posterior threshold
scan, best traces through the cells above it) through bin/hxrecon, 32 leaves x 1000 residues: where the time goes.
    python tools/recon_posterior_bench.py"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import recon_helpers as R
MODEL = os.path.join(ROOT, "tests", "golden", "models", "wag.json")
tree, seqs = R.balanced_family(32, 1000, "arndcqeghilkmfpstwyv", seed=21, branch=.05)
exe = os.path.join(ROOT, "historian_amd", "bin", "hxrecon")
with tempfile.TemporaryDirectory() as d:
    for mode in (sys.argv[1:] or ("fast", "exact")):       # also: linear
        job = os.path.join(d, "job.txt")
        R.write_job(job, MODEL, tree, seqs, {}, os.path.join(d, "s.fa"), os.path.join(d, "g.fa"), posterior=.01, batch=1)
        env = dict(os.environ, HX_TIMING="1", HX_FILL_MODE=mode)
        out = subprocess.run([exe, job], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=1500)
        print(mode, out.returncode)
        for line in out.stderr.decode().strip().splitlines()[-2:]:
            print("   ", line[:900])
