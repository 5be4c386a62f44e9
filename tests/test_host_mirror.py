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
    assert run(["hxtest", "seqprofile", "ACGT", "AAGCT"]) == open(G + "testseqprofile.aagct.json").read()


def test_substitution_matrix_is_the_oracles_bit_for_bit():
    # exp(R t) is an input of every DP parity definition: the mirror's restatement of GSL's series (hx_host_base.cpp) and the
    # oracle's (historian_oracle.sub_prob_matrix_ss) perform the same IEEE operations in the same order
    from oracle import historian_oracle as ho
    for model_file in (G + "testamino.json", os.path.join(ROOT, "tests", "golden", "models", "prot4.json")):
        model = ho.RateModel.from_file(model_file)
        for t in (1e-9, 0.0254, 0.17, 1.0, 12.0):
            got = [[float.fromhex(v) for v in line.split()] for line in run(["hxtest", "expm", model_file, t]).splitlines()]
            want = [row for sr in model.sub_rate for row in ho.sub_prob_matrix_ss(sr.tolist(), t)]
            assert got == want, (model_file, t)


def test_testlogsumexp_host_only():
    want = open(G + "logsumexp.txt").read()
    assert run(["hxtest", "logsumexp", "-fast"]) == want
    assert run(["hxtest", "logsumexp", "-slow"]) == want


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
@pytest.mark.parametrize("mode", ["fast", "trunc"])
@pytest.mark.parametrize("name", sorted(GPU_CASES))
def test_reference_golden_file_through_the_gpu_fast_mode(name, mode):
    # 6-decimal fixtures: the fast fill mode and the library's default, the truncating scaled-probability mode, reproduce
    # them too.  (testbackward prints cells with posterior 1 in priority-queue order; with a fast Forward and an exact
    # Backward fill those exact ties become 1e-12 near-ties, so their order is compared as a set.)
    got, want = run(GPU_CASES[name], {"HX_FILL_MODE": mode}), open(G + name).read()
    if name.startswith("testbackward"):
        assert sorted(got.splitlines()) == sorted(want.splitlines())
    else:
        assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in sorted(GPU_CASES) if ".best." in n or n.endswith(("xdel.out", "yins.out", "nosub.out"))])
def test_reference_golden_file_through_the_gpu_linear_mode(name):
    # HX_FILL_MODE=linear: the scaled-probability fills do not truncate small terms as the reference's operator does,
    # so the 6-decimal numbers of the golden files may move in the 5th decimal (north_star: 1e-4 relative on
    # log-likelihoods).  Everything that is not a number - state names, the best alignment, the structure of the
    # profile - must be the reference's, and every number within 2e-4 of it.
    import re
    got, want = run(GPU_CASES[name], {"HX_FILL_MODE": "linear"}), open(G + name).read()
    num = re.compile(r"-?\d+\.\d+(?:[eE][-+]?\d+)?")
    assert num.sub("#", got) == num.sub("#", want)
    for a, b in zip(num.findall(got), num.findall(want)):
        assert abs(float(a) - float(b)) <= 2e-4 * max(1.0, abs(float(b))), (name, a, b)


@pytest.mark.gpu
def test_testquickalign_golden_file_through_the_gpu():
    # reference Makefile:278-279
    got = run(["hxtest", "quickalign", G + "PF16593.pair.fa", G + "testamino.json", 1])
    assert got == open(G + "testquickalign.out.fa").read()


MERGE_CASES = [(["testmerge1.xy.fa", "testmerge1.xz.fa"], "testmerge1.xyz.fa"),
               (["testmerge1.xy.fa", "testmerge1.ayz.fa"], "testmerge1.xyaz.fa"),
               (["testmerge1.xz.fa", "testmerge1.ayz.fa"], "testmerge1.xzay.fa"),
               (["testmerge1.axyz.fa", "testmerge1.xz.fa"], "testmerge1.axyz.fa")]


@pytest.mark.parametrize("files,want", MERGE_CASES)
def test_testmerge_golden_files_host_only(files, want):
    # reference Makefile:231-235
    assert run(["testmerge"] + [G + f for f in files]) == open(G + want).read()


def test_testmerge_inconsistent_alignments_abort():
    # reference Makefile:236: expected to fail with empty stdout
    out = subprocess.run([BIN + "testmerge", G + "testmerge1.xy.fa", G + "testmerge1.xz.fa", G + "testmerge1-fail.ayz.fa"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert out.returncode != 0 and out.stdout == b""


def _span_against_oracle(extra_args, sparse_params):
    from oracle import historian_oracle as ho
    from oracle import quickalign_oracle as q
    from oracle.ref_mains import read_fasta
    e = dict(os.environ, HX_DEBUG_SPAN="1")
    out = subprocess.run([BIN + "testspan", "-dense"] + extra_args + [G + "PF16593.fa", G + "testamino.json", "1"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, check=True, timeout=300)
    got = out.stdout.decode()
    model = ho.RateModel.from_file(G + "testamino.json")
    model.sub_rate = [m.tolist() for m in model.sub_rate]
    # the DP's inputs (substitution log-odds, gap scores) exactly as the mirror computed them: this model has no
    # "rootprob", and the mirror's Householder solve of the equilibrium differs from numpy's in the last bits
    sc = q.QuickAlignScores(model, 1.0)
    for line in out.stderr.decode().splitlines():
        f = line.split()
        if f[0] == "scores":
            for name, v in zip(("m2m", "m2i", "m2d", "i2i", "i2m", "i2d", "d2d", "d2m", "gap_open", "gap_extend", "no_gap"), f[1:]):
                setattr(sc, name, float.fromhex(v))
        elif f[0] == "submat":
            sc.submat[int(f[1])][int(f[2])] = float.fromhex(f[3])
    ag = q.AlignGraph(read_fasta(G + "PF16593.fa"), model, 1.0, sparse_params=sparse_params, scores=sc)
    assert got == "".join(">%s\n%s\n" % ns for ns in ag.mst_gapped())
    # ... and the same edges, with bit-identical scores, in the same order
    mst = [(int(f[1]), int(f[2]), float.fromhex(f[3])) for f in (l.split() for l in out.stderr.decode().splitlines()) if f[0] == "mst"]
    assert mst == ag.mst_edges and len(mst) == len(ag.seqs) - 1
    return ag


@pytest.mark.gpu
def test_testspan_all_pairs_equals_the_oracle_graph():
    # AlignGraph (src/span.cpp) with the all-vs-all graph and full envelopes: 903 pairwise fills as one device
    # batch, maximum spanning tree, merge -- against the oracle's restatement of the same
    _span_against_oracle(["-kmatchoff"], None)


@pytest.mark.gpu
def test_testspan_all_pairs_sparse_envelopes():
    # k-mer seeded DiagonalEnvelope::initSparse (3-mers, threshold 1, band 8) in the mirror and in the oracle
    from oracle import quickalign_oracle as q
    from oracle.ref_mains import read_fasta
    ag = _span_against_oracle(["-kmatch", "3", "-kmatchn", "1", "-kmatchband", "8"],
                              dict(kmer_len=3, band_size=8, kmer_threshold=1, max_size=0))
    seqs = read_fasta(G + "PF16593.fa")
    env = q.DiagonalEnvelope(seqs[0][1], seqs[1][1])
    env.init_sparse(q.KmerIndex(seqs[1][1], ag.model.alphabet, 3), 8, 1, 40, 0)
    assert len(env.diagonals) < len(seqs[0][1]) + len(seqs[1][1]) - 1     # the envelopes really are sparse


@pytest.mark.gpu
def test_testspan_random_graph_runs_and_is_a_valid_alignment():
    # the reference's own Makefile skips testspan ("inconsistent platform-dependent behavior": the random graph
    # comes from std::uniform_int_distribution); here: the output is a flush alignment of the input sequences
    from oracle.ref_mains import read_fasta
    got = run(["testspan", G + "PF16593.fa", G + "testamino.json", 1])
    rows = {}
    name = None
    for line in got.splitlines():
        if line.startswith(">"):
            name = line[1:]
        else:
            rows[name] = rows.get(name, "") + line
    seqs = dict(read_fasta(G + "PF16593.fa"))
    assert set(rows) == set(seqs) and len({len(r) for r in rows.values()}) == 1
    assert all(rows[n].replace("-", "") == seqs[n] for n in seqs)


@pytest.mark.gpu
def test_branch_matrices_of_the_mirror_against_the_oracle():
    # next row N4: Refiner::BranchMatrix / Sampler::BranchMatrix of the host mirror (fills on the device) on the reference's
    # PF16593 pair as parent and child of one branch - Viterbi and Forward log-likelihoods bit for bit, the best alignment
    # equal to the restatement's (oracle/branch_oracle.py, pinned by enumeration in tests/test_oracle_branch.py)
    from oracle import branch_oracle as bo
    from oracle import historian_oracle as ho
    from oracle.ref_mains import read_fasta
    (_, xs), (_, ys) = read_fasta(G + "PF16593.pair.fa")
    # (a model whose root distribution is given: testamino.json leaves it to a least-squares solve, which the oracle and the
    # mirror perform with different routines - equal to rounding, not bit for bit)
    lg = os.path.join(ROOT, "tests", "golden", "models", "lg.json")
    model = ho.RateModel.from_file(lg)
    t = 0.7
    pm = ho.ProbModel(model, t, [ho.sub_prob_matrix_ss(sr.tolist(), t) for sr in model.sub_rate])
    lpm = ho.LogProbModel(pm)
    neg = float("-inf")
    def pwm(seq):
        return [[[0. if model.alphabet[a] == ch.lower() else neg for a in range(len(model.alphabet))]] for ch in seq]
    x, y = pwm(xs), pwm(ys)
    log_sub = [[[ho.safe_log(v) for v in row] for row in m] for m in pm.sub_mat]
    ysub, yemit = bo.pre_multiply(y, log_sub), bo.calc_ins_probs(y, lpm.log_ins_prob, lpm.log_cpt_weight)
    T = bo.trans_scores(pm.ins, pm.dele, pm.ins_ext, pm.del_ext)
    vit = bo.BranchMatrix(x, ysub, yemit, T, viterbi=True)
    fwd = bo.BranchMatrix(x, ysub, yemit, T, viterbi=False)
    out = run(["hxtest", "branch", G + "PF16593.pair.fa", lg, t]).split()
    assert float.fromhex(out[1]) == vit.lp_end and float.fromhex(out[3]) == fwd.lp_end
    xrow, yrow = vit.best()
    def gapped(seq, row):
        it = iter(seq)
        return "".join(next(it) if b else "-" for b in row)
    assert out[4:] == [gapped(xs, xrow), gapped(ys, yrow)]
