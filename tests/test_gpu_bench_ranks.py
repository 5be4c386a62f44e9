"""The multi-rank flow of bench.py (dealing the pairs, broadcast of the constant block, barriers, max / sum over ranks, ONE
JSON line from rank 0) rehearsed with two processes on the one GPU of the test box: HX_BENCH_REHEARSE puts every rank on device 0
and runs the collectives over gloo.  The driver's scaling run uses RCCL with one GPU per rank; this checks everything around it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("scaling,pairs,port", [("weak", 6, 29531), ("strong", 7, 29532)])
def test_two_ranks_on_one_gpu(scaling, pairs, port):
    env = dict(os.environ, HX_BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--pairs", str(pairs), "--len", "300", "--scaling", scaling, "--no-cpu-baseline", "--single-mode"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                                   # rank 0 alone reports
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == scaling
    assert line["metric"] == "forward-DP cells/s" and line["value"] > 0 and line["higher_is_better"] is True
    total_pairs = 2 * pairs if scaling == "weak" else pairs         # weak: per rank; strong: in total, dealt 4 + 3
    cells = total_pairs * 301 * 301
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 - cells) <= 1e-6 * cells, (line["value"], line["ms_per_step"], cells)
