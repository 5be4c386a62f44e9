"""The C++ host mirror of the reference interface (historian_amd/csrc/host), driven by mains
that mirror the reference's own t/test*.cpp: same command lines, stdout diffed byte for byte
against the reference's golden files (reference Makefile:206-208, 239-257).  The Forward /
Backward fills inside run on the GPU through the C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "historian_amd", "bin") + os.sep
G = os.path.join(ROOT, "tests", "golden", "reference_data") + os.sep
JC, NOSUB = G + "testforward.jukescantor.json", G + "testforward.nosub.json"


def run(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([BIN + args[0]] + [str(a) for a in args[1:]], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          env=e, check=True, timeout=300).stdout.decode()


def test_testseqprofile_host_only():
    assert run(["testseqprofile", "ACGT", "AAGCT"]) == open(G + "testseqprofile.aagct.json").read()


def test_testlogsumexp_host_only():
    want = open(G + "logsumexp.txt").read()
    assert run(["testlogsumexp", "-fast"]) == want
    assert run(["testlogsumexp", "-slow"]) == want


GPU_CASES = {
    "testforward.id100.len2.nosub.out": ["testforward", "-all", "-matrix", G + "testforward.id100.len2.fa", NOSUB, 1],
    "testforward.len2.nosub.best.out": ["testforward", "-hubs", "-best", G + "testforward.len2.fa", NOSUB, 1],
    "testforward.len2.jc.best.out": ["testforward", "-hubs", "-best", G + "testforward.len2.fa", JC, 1],
    "testforward.len2-4.xdel.out": ["testforward", "-hubs", "-best", G + "testforward.len2-4.fa", JC, .1, .01],
    "testforward.len2-4.yins.out": ["testforward", "-hubs", "-best", G + "testforward.len2-4.fa", JC, .01, 1],
    "testforward.len2-4.n10.all.out": ["testforward", "-all", "10", G + "testforward.len2-4.fa", JC, .1],
    "testforward.len2-4.n10.hubs.out": ["testforward", "-hubs", "10", G + "testforward.len2-4.fa", JC, .1],
    "testnullforward.nosub.out": ["testnullforward", NOSUB, 1],
    "testbackward.len2.out": ["testbackward", G + "testforward.len2.fa", JC, 1],
    "testbackward.len2-4.out": ["testbackward", G + "testforward.len2-4.fa", JC, 1],
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(GPU_CASES))
def test_reference_golden_file_through_the_gpu(name):
    assert run(GPU_CASES[name]) == open(G + name).read()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(GPU_CASES))
def test_reference_golden_file_through_the_gpu_fast_mode(name):
    # 6-decimal fixtures: the fast fill mode reproduces them too.  (testbackward prints cells with
    # posterior 1 in priority-queue order; with a fast Forward and an exact Backward fill those exact
    # ties become 1e-12 near-ties, so their order is compared as a set.)
    got, want = run(GPU_CASES[name], {"HX_FILL_MODE": "fast"}), open(G + name).read()
    if name.startswith("testbackward"):
        assert sorted(got.splitlines()) == sorted(want.splitlines())
    else:
        assert got == want


@pytest.mark.gpu
def test_testquickalign_golden_file_through_the_gpu():
    # reference Makefile:278-279
    got = run(["testquickalign", G + "PF16593.pair.fa", G + "testamino.json", 1])
    assert got == open(G + "testquickalign.out.fa").read()
