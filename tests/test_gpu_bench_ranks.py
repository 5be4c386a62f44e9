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


def test_gpus_flag_starts_the_ranks_itself():
    # `python bench.py --gpus 2` with no launcher around it: bench.py starts two rank processes (torch.distributed.run)
    # before touching the GPU and rank 0's single line comes back, with the banded block (weak) and the strong block
    # (configs[3] as written: 512 pairs in total dealt to the ranks)
    env = dict(os.environ, HX_BENCH_REHEARSE="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--pairs", "6",
           "--len", "200", "--no-cpu-baseline", "--banded-pairs", "4,10"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["fill_mode"] == "trunc"
    cells = 2 * 6 * 201 * 201
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 - cells) <= 1e-6 * cells
    assert line["strong"]["pairs_total"] == 512 and line["strong"]["pairs_per_gpu"] == 256
    s_cells = 512 * 201 * 201
    assert abs(line["strong"]["value"] * line["strong"]["ms_per_step"] * 1e-3 - s_cells) <= 1e-6 * s_cells
    assert [b["pairs_per_gpu"] for b in line["banded_mode"]["batches"]] == [4, 10]
    assert all(b["value"] > 0 and 0 < b["roofline_frac"] < 1 for b in line["banded_mode"]["batches"])
    for m in ("fast_mode", "exact_mode", "scaled_probability_mode"):
        assert line[m]["value"] > 0


def test_rank_count_and_gpus_flag_must_agree():
    env = dict(os.environ, HX_BENCH_REHEARSE="1", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--pairs", "2", "--len", "50"], cwd=ROOT, env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode != 0 and b"must agree" in out.stderr
