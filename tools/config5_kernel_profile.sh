export TMPDIR=/tmp
python - <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from tests import recon_helpers as R
ROOT = os.environ["GRAFT_REPO_ROOT"]
tree, seqs = R.balanced_family(64, 5000, "arndcqeghilkmfpstwyv", seed=21, branch=.05)
os.makedirs("/tmp/c5", exist_ok=True)
R.write_job("/tmp/c5/job.txt", os.path.join(ROOT, "tests", "golden", "models", "prot4.json"), tree, seqs, {}, "/tmp/c5/s.fa", "/tmp/c5/g.fa", samples=10, batch=1, maxstates=0)
PY
HX_FILL_MODE=trunc timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_c5 -- historian_amd/bin/hxrecon /tmp/c5/job.txt > /tmp/c5/out.txt 2> gpurun_out/kt_c5.log
f=$(ls -t $(find gpurun_out/kt_c5 -name "*kernel_stats.csv") | head -1); cut -c1-170 $f | head -14
