import os, subprocess, sys, tempfile
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from tests import recon_helpers as R
MODEL = os.path.join(ROOT, "tests", "golden", "models", "wag.json")
tree, seqs = R.balanced_family(32, 1000, "arndcqeghilkmfpstwyv", seed=21, branch=.05)
exe = os.path.join(ROOT, "historian_amd", "bin", "hxrecon")
with tempfile.TemporaryDirectory() as d:
    for samples in (0, 10, 30):
        job = os.path.join(d, "job.txt")
        R.write_job(job, MODEL, tree, seqs, {}, os.path.join(d, "s.fa"), os.path.join(d, "g.fa"), samples=samples, batch=1)
        env = dict(os.environ, HX_TIMING="1", HX_FILL_MODE="fast")
        out = subprocess.run([exe, job], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=1500)
        print("samples", samples)
        for line in out.stderr.decode().strip().splitlines()[-2:]:
            print("   ", line[:330])
