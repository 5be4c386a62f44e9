"""The device farm of the C++ host mirror (host job queue, jobs sorted by cell count, longest first, dealt to
the devices; families and the ready nodes of a tree level are the independent units, reference
src/recon.cpp:942-945,1368-1372).

CPU: the dealing rule (lptAssign) on known cases through bin/testfarm - no device is touched.
GPU: several families farmed by `hxrecon -devices ...` (one host thread per listed device; the one-GPU box lists device 0
twice, so two threads share it) give, family for family, the output of one-at-a-time runs; a single family whose tree
levels are dealt to two (identical) devices gives the single-device output."""
import os
import subprocess

import pytest

from tests import recon_helpers as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "historian_amd", "bin")
JC = os.path.join(ROOT, "tests", "golden", "models", "jc.json")


def test_lpt_dealing_known_cases():
    out = subprocess.run([os.path.join(BIN, "testfarm")], stdout=subprocess.PIPE, check=True).stdout.decode().splitlines()
    want = ["classic: 0 1 2 2 0 1 1 0 2 0 | load 16 15 14",          # the textbook LPT example: makespan 16
            "equal: 0 1 2 3 4 5 6 7 | load 4 4 4 4 4 4 4 4",
            "one-big: 0 1 2 3 1 2 3 1 2 | load 100 3 3 2",
            "fewer-jobs: 1 0 | load 9 3 0 0 0 0 0 0",
            "single-device: 0 0 0 | load 9",
            "tree-level: 1 1 1 0 1 0 | load 9.64e+06 9.25e+06"]
    assert out == want


def _families(tmp_path, n):
    jobs = []
    for k in range(n):
        tree, seqs = R.balanced_family(4 if k % 2 else 8, 40 + 25 * k, "acgt", seed=20 + k)
        job = str(tmp_path / ("family%d.txt" % k))
        R.write_job(job, JC, tree, seqs, {}, str(tmp_path / ("seqs%d.fa" % k)), str(tmp_path / ("guide%d.fa" % k)),
                    maxstates=0, samples=4, seed=5489)
        jobs.append(job)
    return jobs


def _run(args):
    out = subprocess.run([os.path.join(BIN, "hxrecon")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()
    return out.stdout.decode()


@pytest.mark.gpu
def test_families_farmed_over_devices_match_one_at_a_time(tmp_path):
    jobs = _families(tmp_path, 5)
    single = [_run([j]) for j in jobs]
    for devices in ("0", "0,0", "0,0,0"):
        farmed = _run(["-devices", devices] + jobs)
        parts = farmed.split("family ")[1:]
        assert len(parts) == len(jobs)
        for k, part in enumerate(parts):
            head, body = part.split("\n", 1)
            assert int(head) == k and body == single[k], "family %d with devices %s" % (k, devices)


@pytest.mark.gpu
def test_tree_levels_dealt_to_two_devices(tmp_path):
    job = _families(tmp_path, 1)[0]
    assert _run(["-devices", "0,0", job]) == _run([job])
