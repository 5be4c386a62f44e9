"""Restatements of the reference's hot-path test mains (t/test*.cpp) on top of the
Python oracle.  Each returns the exact text the reference program prints to stdout,
so tests can diff it byte-for-byte against the reference's committed golden files
(tests/golden/reference_data/, copied data files).

TEST INFRASTRUCTURE ONLY (see historian_oracle.py header).
"""
import math
from . import historian_oracle as ho


def read_fasta(path):
    seqs = []
    name, buf = None, []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line.startswith(">"):
                if name is not None:
                    seqs.append((name, "".join(buf)))
                name, buf = line[1:].split()[0], []
            elif line:
                buf.append(line)
    if name is not None:
        seqs.append((name, "".join(buf)))
    return seqs


def _setup(seqs_path, model_path, xt, yt):
    seqs = read_fasta(seqs_path)
    assert len(seqs) == 2
    rates = ho.RateModel.from_file(model_path)
    xprobs = ho.ProbModel(rates, xt)
    yprobs = ho.ProbModel(rates, yt)
    hmm = ho.PairHMM(xprobs, yprobs, rates.ins_prob)
    xprof = ho.Profile.from_seq(1, rates.alphabet, seqs[0][1], 1, seqs[0][0])
    yprof = ho.Profile.from_seq(1, rates.alphabet, seqs[1][1], 2, seqs[1][0])
    return rates, hmm, xprof, yprof


def all_cells(fwd, xprof, yprof):
    cells = {fwd.start_cell, fwd.end_cell}
    for xpos in range(xprof.size() - 1):
        for ypos in range(yprof.size() - 1):
            for s in ho.STATES:
                if xpos > 0 or ypos > 0:
                    cells.add((xpos, ypos, s))
    return cells


def testforward(strat, what, seqs_path, model_path, xt, yt=None):
    """t/testforward.cpp:8-69"""
    yt = xt if yt is None else yt
    strategy = {"-all": ho.DPMatrix.KeepAll, "-hubs": ho.DPMatrix.CollapseChains}[strat]
    rates, hmm, xprof, yprof = _setup(seqs_path, model_path, xt, yt)
    fwd = ho.ForwardMatrix(xprof, yprof, hmm, 0, ho.GuideAlignmentEnvelope())
    if what == "-matrix":
        prof = fwd.make_profile(all_cells(fwd, xprof, yprof), strategy)
    elif what == "-best":
        prof = fwd.best_profile(strategy)
    else:
        gen = ho.MT19937()
        prof = fwd.sample_profile(gen, int(what), 0, strategy)
    prof.calc_sum_path_absorb_probs([0.], hmm.log_root)
    return prof.to_json()


def cout_double(d):
    """ostream << double at default precision (6 significant digits, %g)."""
    return "%g" % d


def testbackward(seqs_path, model_path, xt, yt=None):
    """t/testbackward.cpp:8-42"""
    yt = xt if yt is None else yt
    rates, hmm, xprof, yprof = _setup(seqs_path, model_path, xt, yt)
    fwd = ho.ForwardMatrix(xprof, yprof, hmm, 0, ho.GuideAlignmentEnvelope())
    back = ho.BackwardMatrix(fwd)
    out = "Forward score: " + cout_double(fwd.lp_end) + "\n"
    out += "Backward score: " + cout_double(back.lp_start()) + "\n"
    for lpp, c in back.cells_above_post_prob_threshold(.5):
        out += "P" + back.cell_name(c) + " = " + cout_double(math.exp(lpp)) + "\n"
    return out


def testnullforward(model_path, xt, yt=None):
    """t/testnullforward.cpp:8-55"""
    yt = xt if yt is None else yt
    rates = ho.RateModel.from_file(model_path)
    xprobs = ho.ProbModel(rates, xt)
    yprobs = ho.ProbModel(rates, yt)
    hmm = ho.PairHMM(xprobs, yprobs, rates.ins_prob)
    xprof = ho.Profile.from_seq(1, rates.alphabet, "acg", 1, "x")
    yprof = ho.Profile.from_seq(1, rates.alphabet, "cag", 2, "y")
    xprof.state[2].lp_absorb = []
    yprof.state[1].lp_absorb = []
    fwd = ho.ForwardMatrix(xprof, yprof, hmm, 0, ho.GuideAlignmentEnvelope())
    prof = fwd.make_profile(all_cells(fwd, xprof, yprof), ho.DPMatrix.KeepAll)
    prof.calc_sum_path_absorb_probs([0.], hmm.log_root)
    return prof.to_json()


def testseqprofile(alphabet, seq):
    """t/testseqprofile.cpp:6-20"""
    return ho.Profile.from_seq(1, alphabet, seq, 0).to_json()


def testlogsumexp(slow):
    """t/testlogsumexp.cpp:8-21"""
    out = []
    x = 0.
    while x < 2:
        y = 0.
        while y < 2:
            v = ho.log_sum_exp_slow(x, y) if slow else ho.log_sum_exp(x, y)
            out.append("%s %s %s\n" % (cout_double(x), cout_double(y), cout_double(v)))
            y += .1
        x += .1
    return "".join(out)
