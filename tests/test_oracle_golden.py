"""Pins the oracle: the CPU restatement must reproduce, byte for byte, every golden
file the reference's own tests hold for the hot path (reference Makefile:206-208,
239-257).  The golden files under tests/golden/reference_data/ are data files copied
from the reference's data/ directory."""
import os
import re

import pytest

from oracle import ref_mains as rm

G = os.path.join(os.path.dirname(__file__), "golden", "reference_data") + os.sep

JC = G + "testforward.jukescantor.json"
NOSUB = G + "testforward.nosub.json"

CASES = {
    # reference Makefile:243-250
    "testforward.id100.len2.nosub.out": lambda: rm.testforward("-all", "-matrix", G + "testforward.id100.len2.fa", NOSUB, 1),
    "testforward.len2.nosub.best.out": lambda: rm.testforward("-hubs", "-best", G + "testforward.len2.fa", NOSUB, 1),
    "testforward.len2.jc.best.out": lambda: rm.testforward("-hubs", "-best", G + "testforward.len2.fa", JC, 1),
    "testforward.len2-4.xdel.out": lambda: rm.testforward("-hubs", "-best", G + "testforward.len2-4.fa", JC, .1, .01),
    "testforward.len2-4.yins.out": lambda: rm.testforward("-hubs", "-best", G + "testforward.len2-4.fa", JC, .01, 1),
    "testforward.len2-4.n10.all.out": lambda: rm.testforward("-all", "10", G + "testforward.len2-4.fa", JC, .1),
    "testforward.len2-4.n10.hubs.out": lambda: rm.testforward("-hubs", "10", G + "testforward.len2-4.fa", JC, .1),
    # reference Makefile:252-253
    "testnullforward.nosub.out": lambda: rm.testnullforward(NOSUB, 1),
    # reference Makefile:255-257
    "testbackward.len2.out": lambda: rm.testbackward(G + "testforward.len2.fa", JC, 1),
    "testbackward.len2-4.out": lambda: rm.testbackward(G + "testforward.len2-4.fa", JC, 1),
    # reference Makefile:239-240
    "testseqprofile.aagct.json": lambda: rm.testseqprofile("ACGT", "AAGCT"),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_reference_golden_file(name):
    want = open(G + name).read()
    assert CASES[name]() == want


def test_logsumexp_fast_and_slow_match_golden():
    # reference Makefile:206-208: golden written by -slow, checked against -fast
    want = open(G + "logsumexp.txt").read()
    assert rm.testlogsumexp(slow=True) == want
    assert rm.testlogsumexp(slow=False) == want


def test_cumlp_equals_fwdlp_on_51_cells():
    # reference perl/testcumlp.pl data/testforward.id100.len2.nosub.out 51 (Makefile:244)
    text = CASES["testforward.id100.len2.nosub.out"]()
    n, clp = 0, None
    for line in text.splitlines():
        m = re.search(r'"cumLogProb": "([^"]+)"', line)
        if m:
            clp = m.group(1)
            continue
        m = re.search(r'"fwdLogProb": "([^"]+)"', line)
        if m and clp is not None:
            assert clp == m.group(1)
            n += 1
        clp = None
    assert n == 51


def test_testquickalign_fixture():
    # reference Makefile:278-279: bin/testquickalign data/PF16593.pair.fa data/testamino.json 1
    from oracle import quickalign_oracle as q
    got = q.testquickalign_main(G + "PF16593.pair.fa", G + "testamino.json", 1)
    assert got == open(G + "testquickalign.out.fa").read()


@pytest.mark.parametrize("files,want", [(["testmerge1.xy.fa", "testmerge1.xz.fa"], "testmerge1.xyz.fa"),
                                        (["testmerge1.xy.fa", "testmerge1.ayz.fa"], "testmerge1.xyaz.fa"),
                                        (["testmerge1.xz.fa", "testmerge1.ayz.fa"], "testmerge1.xzay.fa"),
                                        (["testmerge1.axyz.fa", "testmerge1.xz.fa"], "testmerge1.axyz.fa")])
def test_testmerge_fixtures(files, want):
    # reference Makefile:231-235
    from oracle import quickalign_oracle as q
    assert q.testmerge_main([G + f for f in files]) == open(G + want).read()


def test_testmerge_inconsistent_alignments_fail():
    # reference Makefile:236
    from oracle import quickalign_oracle as q
    with pytest.raises(AssertionError):
        q.testmerge_main([G + f for f in ("testmerge1.xy.fa", "testmerge1.xz.fa", "testmerge1-fail.ayz.fa")])
