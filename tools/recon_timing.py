"""End-to-end timing of the gp120 reconstruction through the C++ mirror + GPU fills (bin/hxrecon, HX_TIMING=1)."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import recon_helpers as R
G = os.path.join(ROOT, "tests", "golden", "reference_data") + os.sep
LG = os.path.join(ROOT, "tests", "golden", "models", "lg.json")
tree, seqs, guide = R.load_family(G + "gp120.tree.nh", G + "gp120.fa", G + "gp120.guide.fa")
exe = os.path.join(ROOT, "historian_amd", "bin", "hxrecon")
with tempfile.TemporaryDirectory() as d:
    for band in (20, -1):
        for mode in ("exact", "fast"):
            job = os.path.join(d, "job_%d.txt" % band)
            opts = {"band": band} if band >= 0 else {}
            R.write_job(job, LG, tree, seqs, guide if band >= 0 else {}, os.path.join(d, "s.fa"), os.path.join(d, "g.fa"), **opts)
            env = dict(os.environ, HX_TIMING="1")
            if mode == "fast":
                env["HX_FILL_MODE"] = "fast"
            for rep in range(2):      # the second run has the image paged in
                t0 = time.time()
                out = subprocess.run([exe, job], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
                dt = time.time() - t0
            print("band", band, mode, "process wall %.2f s" % dt)
            for line in out.stderr.decode().strip().splitlines()[-2:]:
                print("   ", line)
