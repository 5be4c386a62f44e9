"""Whole-tree parity (BASELINE config 3): progressive reconstruction of the gp120 family
(10 leaves, LG, guide band 20, 9 internal pair DPs over DAG profiles built from the best trace
plus 10 sampled traces) by the C++ host mirror with the fills on the GPU, against the oracle's
restatement of the same driver loop (reference src/recon.cpp:917-1052) with CPU fills.

Exact mode: root Forward log-likelihood bit-identical, final alignment identical -- which
requires every sampled traceback of every node (shared mt19937, node order) to be identical.
The oracle's root log-likelihood, -7731.36, is the value SURVEY.md section 6 reports for the
reference itself on these inputs."""
import os
import subprocess

import pytest

from tests import recon_helpers as R

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden", "reference_data") + os.sep
LG = os.path.join(ROOT, "tests", "golden", "models", "lg.json")
HXRECON = os.path.join(ROOT, "historian_amd", "bin", "hxrecon")


def run_case(tmp_path, max_len, band, fast=False, samples=10, mode=None):
    tree, seqs, guide = R.load_family(G + "gp120.tree.nh", G + "gp120.fa", G + "gp120.guide.fa", max_len=max_len)
    job = str(tmp_path / "job.txt")
    R.write_job(job, LG, tree, seqs, guide, str(tmp_path / "seqs.fa"), str(tmp_path / "guide.fa"), band=band,
                samples=samples, maxstates=0, seed=5489)
    env = dict(os.environ)
    if fast:
        env["HX_FILL_MODE"] = "fast"
    if mode:
        env["HX_FILL_MODE"] = mode
    out = subprocess.run([HXRECON, job], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert out.returncode == 0, out.stderr.decode()
    got = R.parse_hxrecon(out.stdout.decode())
    res, rows = R.oracle_reconstruct(LG, tree, seqs, guide, max_distance_from_guide=band, profile_samples=samples)
    return got, res, rows


def test_gp120_truncated_exact(tmp_path):
    got, res, rows = run_case(tmp_path, 120, 20)
    assert got["lpFinalFwd"] == res["lp_final_fwd"]            # bit-identical
    assert got["lpFinalTrace"] == res["lp_final_trace"]
    assert got["rows"] == rows
    assert got["bands"] == res["bands"]


def test_gp120_truncated_band_retry(tmp_path):
    # band 0 on this family gives zero likelihood at some nodes: the driver doubles the band (recon.cpp:956-975)
    got, res, rows = run_case(tmp_path, 120, 0)
    assert got["bands"] == res["bands"]
    assert any(b != 0 for b in res["bands"].values())
    assert got["lpFinalFwd"] == res["lp_final_fwd"]
    assert got["rows"] == rows


def test_gp120_full_exact(tmp_path):
    got, res, rows = run_case(tmp_path, None, 20)
    assert "%.2f" % res["lp_final_fwd"] == "-7731.36"          # the reference's own number
    assert got["lpFinalFwd"] == res["lp_final_fwd"]
    assert got["lpFinalTrace"] == res["lp_final_trace"]
    assert got["rows"] == rows


def test_gp120_full_fast_mode(tmp_path):
    # internal (DAG) profiles use the exact general kernels in both modes; the leaf-vs-leaf nodes use the
    # fast chain kernel: log-likelihood within north_star's 1e-4 relative, alignment identical
    got, res, rows = run_case(tmp_path, None, 20, fast=True)
    assert abs(got["lpFinalFwd"] - res["lp_final_fwd"]) <= 1e-4 * abs(res["lp_final_fwd"])
    assert got["rows"] == rows


def test_gp120_full_linear_mode(tmp_path):
    # HX_FILL_MODE=linear: batches made of leaf-vs-leaf nodes only run on scaled probabilities (the banded kernel, one
    # wavefront per pair), everything else as in fast mode.  The scaled-probability fill does not truncate small terms as
    # the reference does, so cells move in the 5th decimal and sampled paths could differ at near-ties: required are the
    # log-likelihood within north_star's 1e-4 relative and a complete alignment of the right sequences; on this family
    # the alignment comes out identical.
    got, res, rows = run_case(tmp_path, None, 20, mode="linear")
    assert abs(got["lpFinalFwd"] - res["lp_final_fwd"]) <= 1e-4 * abs(res["lp_final_fwd"])
    assert {k: v.replace("-", "") for k, v in got["rows"].items() if k in rows} == {k: v.replace("-", "") for k, v in rows.items()}
    assert len({len(v) for v in got["rows"].values()}) == 1
    assert got["rows"] == rows


def test_mixture_family_unbanded_exact(tmp_path):
    # BASELINE configs[4] shape in miniature: 4-component mixture (prot4), balanced 8-leaf tree, no guide band
    prot4 = os.path.join(ROOT, "tests", "golden", "models", "prot4.json")
    import json
    alphabet = json.load(open(prot4))["alphabet"]
    tree, seqs = R.balanced_family(8, 60, alphabet, seed=11)
    job = str(tmp_path / "job.txt")
    R.write_job(job, prot4, tree, seqs, {}, str(tmp_path / "seqs.fa"), str(tmp_path / "guide.fa"), samples=5,
                maxstates=0, seed=5489)
    out = subprocess.run([HXRECON, job], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()
    got = R.parse_hxrecon(out.stdout.decode())
    res, rows = R.oracle_reconstruct(prot4, tree, seqs, {}, profile_samples=5)
    assert got["lpFinalFwd"] == res["lp_final_fwd"]
    assert got["lpFinalTrace"] == res["lp_final_trace"]
    assert got["rows"] == rows


def test_posterior_profiles_with_batched_fills(tmp_path):
    # usePosteriorsForProfile (reference src/recon.cpp:978-1013): Backward fills, threshold scan and posterior
    # profiles, with the four leaf-pair nodes (then the two above) sharing one device batch each
    alphabet = "acgt"
    jc = os.path.join(ROOT, "tests", "golden", "models", "jc.json")
    tree, seqs = R.balanced_family(8, 50, alphabet, seed=3)
    outs = {}
    for batch in (1, 0):
        job = str(tmp_path / ("job%d.txt" % batch))
        R.write_job(job, jc, tree, seqs, {}, str(tmp_path / "seqs.fa"), str(tmp_path / "guide.fa"), maxstates=0,
                    posterior=0.01, batch=batch)
        out = subprocess.run([HXRECON, job], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert out.returncode == 0, out.stderr.decode()
        outs[batch] = out.stdout.decode()
    assert outs[0] == outs[1]
    got = R.parse_hxrecon(outs[1])
    res, rows = R.oracle_reconstruct(jc, tree, seqs, {}, min_post_prob=0.01)
    assert got["lpFinalFwd"] == res["lp_final_fwd"]
    assert got["lpFinalTrace"] == res["lp_final_trace"]
    assert got["rows"] == rows


def test_batched_and_sequential_fills_give_the_same_reconstruction(tmp_path):
    tree, seqs, guide = R.load_family(G + "gp120.tree.nh", G + "gp120.fa", G + "gp120.guide.fa", max_len=120)
    outs = {}
    for batch in (1, 0):
        job = str(tmp_path / ("job%d.txt" % batch))
        R.write_job(job, LG, tree, seqs, guide, str(tmp_path / "seqs.fa"), str(tmp_path / "guide.fa"), band=5, samples=10,
                    maxstates=0, seed=5489, batch=batch)
        out = subprocess.run([HXRECON, job], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert out.returncode == 0, out.stderr.decode()
        outs[batch] = out.stdout.decode()
    assert outs[0] == outs[1]


def test_best_path_profiles_never_copy_a_matrix(tmp_path):
    # profsamples 0: every node's profile is its best path (what the reference's -fast preset amounts to).  The
    # tracebacks run on the device (hx_batch_best_trace) and the profile annotations come from a cell gather, so no
    # Forward matrix crosses PCIe - and the alignment is still the oracle's, with host or device tracebacks.
    import re
    tree, seqs, guide = R.load_family(G + "gp120.tree.nh", G + "gp120.fa", G + "gp120.guide.fa", max_len=200)
    job = str(tmp_path / "job.txt")
    R.write_job(job, LG, tree, seqs, guide, str(tmp_path / "seqs.fa"), str(tmp_path / "guide.fa"), band=10,
                samples=0, maxstates=0, seed=5489)
    res, rows = R.oracle_reconstruct(LG, tree, seqs, guide, max_distance_from_guide=10, profile_samples=0)
    outs = {}
    for host_traceback in (False, True):
        env = dict(os.environ, HX_TIMING="1")
        if host_traceback:
            env["HX_HOST_TRACEBACK"] = "1"
        out = subprocess.run([HXRECON, job], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
        assert out.returncode == 0, out.stderr.decode()
        got = R.parse_hxrecon(out.stdout.decode())
        assert got["lpFinalFwd"] == res["lp_final_fwd"]
        assert got["lpFinalTrace"] == res["lp_final_trace"]
        assert got["rows"] == rows
        outs[host_traceback] = out.stdout.decode()
        reads = int(re.search(r"matrix D2H [0-9.]+ s in (\d+) reads", out.stderr.decode()).group(1))
        traces = int(re.search(r"tracebacks [0-9.]+ s in (\d+) calls", out.stderr.decode()).group(1))
        assert (reads > 0) == host_traceback and (traces > 0) == (not host_traceback)
    assert outs[False] == outs[True]


def test_near_ties_resolved_by_an_exact_fill_on_request(tmp_path):
    # HX_TIE_REFILL=1: a best trace whose walk met a near tie between two source cells (hx_batch_best_trace_ties, or the same test
    # in the host walk) is taken again from a fresh fill of that pair under the exact policy.  Best-path profiles (profsamples 0)
    # in the default policy, device and host walks: the oracle's alignment, and the run says how many traces it took again.
    import re
    tree, seqs, guide = R.load_family(G + "gp120.tree.nh", G + "gp120.fa", G + "gp120.guide.fa", max_len=200)
    job = str(tmp_path / "job.txt")
    R.write_job(job, LG, tree, seqs, guide, str(tmp_path / "seqs.fa"), str(tmp_path / "guide.fa"), band=10,
                samples=0, maxstates=0, seed=5489)
    res, rows = R.oracle_reconstruct(LG, tree, seqs, guide, max_distance_from_guide=10, profile_samples=0)
    refills = {}
    for host_traceback in (False, True):
        env = dict(os.environ, HX_TIMING="1", HX_TIE_REFILL="1", HX_FILL_MODE="trunc")
        if host_traceback:
            env["HX_HOST_TRACEBACK"] = "1"
        out = subprocess.run([HXRECON, job], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
        assert out.returncode == 0, out.stderr.decode()
        got = R.parse_hxrecon(out.stdout.decode())
        assert got["rows"] == rows
        assert abs(got["lpFinalFwd"] - res["lp_final_fwd"]) <= 1e-9 * abs(res["lp_final_fwd"])
        refills[host_traceback] = int(re.search(r"near tie \(HX_TIE_REFILL=1\): (\d+),", out.stderr.decode()).group(1))
    assert refills[False] == refills[True]          # the device walk and the host walk flag the same pairs
    # without the variable no trace is taken again
    out = subprocess.run([HXRECON, job], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, HX_TIMING="1", HX_FILL_MODE="trunc"), timeout=600)
    assert out.returncode == 0, out.stderr.decode()
    assert int(re.search(r"near tie \(HX_TIE_REFILL=1\): (\d+),", out.stderr.decode()).group(1)) == 0
    assert R.parse_hxrecon(out.stdout.decode())["rows"] == rows
    # sampling mode on a synthetic family of 16 leaves x 600 residues (ten sampled traces per node: internal nodes are DAGs with
    # routes of equal probability): some walks do meet a near tie, their traces come from an exact fill, and the alignment is the
    # one the exact policy gives (which the tests above hold to the oracle bit for bit)
    WAG = os.path.join(ROOT, "tests", "golden", "models", "wag.json")
    tree, seqs = R.balanced_family(16, 600, "arndcqeghilkmfpstwyv", seed=21, branch=.05)
    R.write_job(job, WAG, tree, seqs, {}, str(tmp_path / "s2.fa"), str(tmp_path / "g2.fa"), samples=10, batch=1, maxstates=0)
    outs = {}
    for mode, refill in (("exact", False), ("trunc", True)):
        env = dict(os.environ, HX_TIMING="1", HX_FILL_MODE=mode)
        if refill:
            env["HX_TIE_REFILL"] = "1"
        out = subprocess.run([HXRECON, job], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
        assert out.returncode == 0, out.stderr.decode()
        outs[mode] = (R.parse_hxrecon(out.stdout.decode())["rows"], out.stderr.decode())
    assert outs["trunc"][0] == outs["exact"][0]
    taken_again = [line for line in outs["trunc"][1].splitlines() if "near tie (HX_TIE_REFILL=1)" in line]
    assert taken_again and int(taken_again[0].split("):")[1].split(",")[0]) >= 1, taken_again


def test_posterior_profiles_in_linear_fill_mode(tmp_path):
    # the same posterior-profile reconstruction with HX_FILL_MODE=linear: the leaf-pair nodes' Forward and Backward
    # matrices are the scaled-probability fills' (interleaved layout, read back by the mirror); the root log-likelihood
    # stays within north_star's tolerance of the oracle's and every sequence is aligned completely
    alphabet = "acgt"
    jc = os.path.join(ROOT, "tests", "golden", "models", "jc.json")
    tree, seqs = R.balanced_family(8, 150, alphabet, seed=5)
    job = str(tmp_path / "job.txt")
    R.write_job(job, jc, tree, seqs, {}, str(tmp_path / "seqs.fa"), str(tmp_path / "guide.fa"), maxstates=0, posterior=0.01, batch=1)
    out = subprocess.run([HXRECON, job], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600,
                         env=dict(os.environ, HX_FILL_MODE="linear"))
    assert out.returncode == 0, out.stderr.decode()
    got = R.parse_hxrecon(out.stdout.decode())
    res, rows = R.oracle_reconstruct(jc, tree, seqs, {}, min_post_prob=0.01)
    assert abs(got["lpFinalFwd"] - res["lp_final_fwd"]) <= 1e-4 * abs(res["lp_final_fwd"])
    assert {k: v.replace("-", "") for k, v in got["rows"].items() if k in rows} == {k: v.replace("-", "") for k, v in rows.items()}
    assert len({len(v) for v in got["rows"].values()}) == 1


# ---- the reference's own whole-pipeline outputs (`testhist`, reference Makefile:304-308) through the GPU ----
from tests import test_oracle_testhist as TH


@pytest.mark.parametrize("mode", ["exact", "fast", "trunc"])
@pytest.mark.parametrize("name", sorted(TH.CASES))
def test_hxrecon_reproduces_the_references_testhist_output(tmp_path, name, mode):
    # posterior-profile mode (cases 1, 2: Forward + Backward + threshold scan at every node) and sampling mode with
    # 100 traces per node on the shared generator (cases 3, 4: 43 sequences, 42 pair DPs): the C++ mirror with every
    # fill on the device prints the reference's file byte for byte
    case = TH.CASES[name]
    tree, seqs, guide = TH.load_case(case)
    kw = case["kw"]
    opts = dict(band=kw["max_distance_from_guide"], maxstates=0, seed=5489)
    if "min_post_prob" in kw:
        opts["posterior"] = kw["min_post_prob"]
    else:
        opts["samples"] = kw["profile_samples"]
    job = str(tmp_path / "job.txt")
    R.write_job(job, G + case["model"], tree, seqs, guide, str(tmp_path / "seqs.fa"), str(tmp_path / "guide.fa"), **opts)
    env = dict(os.environ, HX_FILL_MODE=mode)
    out = subprocess.run([HXRECON, job], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert out.returncode == 0, out.stderr.decode()
    got = R.parse_hxrecon(out.stdout.decode())
    assert R.fasta_rows(tree, got["rows"]) == open(G + name.split()[0]).read()
