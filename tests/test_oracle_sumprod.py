"""The sum-product oracle (SumProduct up/down passes, branch posteriors, eigen-basis substitution counts)
against every fixture the reference holds for it that does not need the indel / DP machinery: testsumprod (posteriors of a
three-node column) and testaligncount (root counts, substitution counts and wait times of fixed alignments; the
reference's Makefile checks -sub and -eigen against the same file).  Byte for byte."""
import os

from oracle import historian_oracle as ho
from oracle import sumprod_oracle as so
from oracle.ref_mains import read_fasta
from tests.recon_helpers import parse_newick

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_data") + os.sep


def load(model_file, fasta, newick):
    model = ho.RateModel.from_file(G + model_file)
    with open(G + newick) as f:
        rt = parse_newick(f.read())
    tree = so.Tree(rt.parent, rt.branch_length, rt.name)
    rows = dict(read_fasta(G + fasta))
    gapped = {n: rows[tree.name[n]] for n in range(tree.nodes())}     # Tree::reorderSeqs: rows by node name
    return model, tree, gapped


def test_testsumprod_fixture():
    model, tree, gapped = load("testnj.jukescantor.json", "testaligncount.fa", "testaligncount.nh")
    with open(G + "testsumprod.out") as f:
        assert so.main_testsumprod(model, tree, gapped) == f.read()


def test_testaligncount_fixtures():
    model, tree, gapped = load("testnj.jukescantor.json", "testaligncount.fa", "testaligncount.nh")
    with open(G + "testaligncount.out") as f:
        assert so.main_testaligncount(model, tree, gapped) == f.read()
    model, tree, gapped = load("testcount.jukescantor.json", "testaligncount2.fa", "testcount.nh")
    with open(G + "testaligncount2.out.json") as f:
        assert so.main_testaligncount(model, tree, gapped) == f.read()


def test_count_of_a_fixed_reconstruction_fixtures():
    """`historian count -recon` (reference Makefile testcount): indel counts of every branch of a given reconstruction,
    substitution counts of its columns, the log-likelihood - single component and the two-component cyclic mixture."""
    for model_file, fasta, newick, expected in (
            ("testcount.jukescantor.json", "testcount.fa", "testcount.nh", "testcount.out.json"),
            ("testcount.jukescantor.json", "testcount.historian.fa", "testcount.nh", "testcount.count.json"),
            ("testrates.mix2.json", "testcount.mix2.fa", "testcount.mix2.nh", "testcount.mix2.count.json")):
        model, tree, gapped = load(model_file, fasta, newick)
        indel, root, counts = so.count_reconstruction(model, tree, gapped)
        with open(G + expected) as f:
            assert so.write_event_counts(model, indel, root, counts) == f.read(), expected
