"""BASELINE configs[4] as a whole: a balanced 64-leaf synthetic family of 5000-residue sequences under the 4-component mixture
(prot4.json), no band, progressive reconstruction through bin/hxrecon with every tree level's ready nodes in one device batch -
32 leaf pairs of 5000 x 5000 x 4 components, then 16 / 8 / 4 / 2 / 1 internal-node pairs whose profiles come from the best trace
plus 10 sampled traces of the level below.  Prints the per-level and total timing lines of the host mirror (HX_TIMING), and
checks what can be checked at this size: every leaf row of the final alignment spells its sequence, all rows have one length,
and the runs in the two truncating policies (trunc, fast) give the same alignment.
    python tools/config5_levels.py [n_leaves] [length] [modes]          (on the GPU box)"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import recon_helpers as R

n_leaves = int(sys.argv[1]) if len(sys.argv) > 1 else 64
length = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["trunc", "fast"]
MODEL = os.path.join(ROOT, "tests", "golden", "models", "prot4.json")
alphabet = "arndcqeghilkmfpstwyv"
tree, seqs = R.balanced_family(n_leaves, length, alphabet, seed=21, branch=.05)
exe = os.path.join(ROOT, "historian_amd", "bin", "hxrecon")
rows_of = {}
with tempfile.TemporaryDirectory() as d:
    for mode in modes:
        job = os.path.join(d, "job.txt")
        R.write_job(job, MODEL, tree, seqs, {}, os.path.join(d, "s.fa"), os.path.join(d, "g.fa"), samples=10, batch=1, maxstates=0)
        env = dict(os.environ, HX_TIMING="1", HX_TIMING_LEVELS="1", HX_FILL_MODE=mode)
        t0 = time.time()
        out = subprocess.run([exe, job], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=3000)
        assert out.returncode == 0, out.stderr.decode()[-3000:]
        got = R.parse_hxrecon(out.stdout.decode())
        print("== %d leaves x %d residues, prot4 (4 components), no band, HX_FILL_MODE=%s: %.1f s wall, lpFinalFwd %.4f" %
              (n_leaves, length, mode, time.time() - t0, got["lpFinalFwd"]), flush=True)
        for line in out.stdout.decode().splitlines():          # (a -DHX_DAG_TRACE build of the library: the kernels' phase lines)
            if line.startswith("trace "):
                print("   ", line, flush=True)
        for line in out.stderr.decode().strip().splitlines():
            if line.startswith("timing") or line.startswith("profile stats"):
                print("   ", line, flush=True)
        rows = got["rows"]
        assert len({len(v) for v in rows.values()}) == 1, "ragged alignment"
        for n, (nm, s) in seqs.items():
            assert rows[n].replace("-", "") == s, "row of %s does not spell its sequence" % nm
        rows_of[mode] = rows
        print("    alignment: %d rows x %d columns; every leaf row spells its sequence" % (len(rows), len(next(iter(rows.values())))), flush=True)
if len(rows_of) > 1:
    first = rows_of[modes[0]]
    for m in modes[1:]:
        print("    alignment of %s == alignment of %s: %s" % (m, modes[0], rows_of[m] == first))
