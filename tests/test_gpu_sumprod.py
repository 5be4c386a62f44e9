"""Counts mode on the device: hx_sumprod_columns against the sum-product oracle.

The reference's fixtures first - the alignments of t/testaligncount (whose expected outputs pin the oracle byte for byte in
tests/test_oracle_sumprod.py) go through the device and must print the same JSON - then seeded protein columns on a
nine-leaf tree with the 4-component mixture, gaps and wildcards included, column by column.  Sums over columns are atomic
on the device, so counts agree to rounding (1e-10 relative to the largest entry), not bit for bit; column likelihoods use
the device's log() in place of libm's and agree to 1e-12 relative."""
import json
import os

import numpy as np
import pytest

from historian_amd import capi, counts, hostmodel
from oracle import c_oracle
from oracle import historian_oracle as ho
from oracle import sumprod_oracle as so
from oracle.ref_mains import read_fasta
from tests.recon_helpers import parse_newick

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden", "reference_data") + os.sep
PROT4 = os.path.join(ROOT, "tests", "golden", "models", "prot4.json")


@pytest.fixture(scope="module", autouse=True)
def engine():
    capi.init(0, c_oracle.table())
    yield
    capi.shutdown()


def _fixture(model_file, fasta, newick):
    with open(G + newick) as f:
        rt = parse_newick(f.read())
    tree = so.Tree(rt.parent, rt.branch_length, rt.name)
    rows = dict(read_fasta(G + fasta))
    gapped = {n: rows[tree.name[n]] for n in range(tree.nodes())}
    return ho.RateModel.from_file(G + model_file), hostmodel.RateModel.load(G + model_file), tree, gapped


@pytest.mark.parametrize("files,expected", [
    (("testnj.jukescantor.json", "testaligncount.fa", "testaligncount.nh"), "testaligncount.out"),
    (("testcount.jukescantor.json", "testaligncount2.fa", "testcount.nh"), "testaligncount2.out.json")])
def test_reference_alignment_count_fixtures(files, expected):
    omodel, model, tree, gapped = _fixture(*files)
    cc = counts.ColumnCounter(model, tree.parent, tree.branch_length)
    tok = counts.tokenize_columns(model.alphabet, [gapped[n] for n in range(tree.nodes())])
    got = cc.run(tok, want_root_post=True)
    with open(G + expected) as f:
        assert so.write_sub_counts(omodel, got["root_counts"], got["counts"]) + "\n" == f.read()
    # the root posteriors t/testsumprod prints, and the column likelihoods
    sp = so.SumProduct(omodel, tree)
    for col, seq in enumerate(so.columns_of(tree, gapped)):
        sp.init_column(seq)
        sp.fill_up()
        sp.fill_down()
        assert abs(got["col_log_like"][col] - sp.col_log_like) <= 1e-12 * abs(sp.col_log_like)
        np.testing.assert_allclose(np.exp(got["root_post"][col]), np.exp(sp.log_node_post_prob(sp.column_root())), rtol=1e-9, atol=1e-300)


def _random_tree(rng, leaves):
    """parent / branch length arrays of a random binary tree, children before parents, root last"""
    parent, length, live = [], [], []
    for _ in range(leaves):
        live.append(len(parent))
        parent.append(-1)
        length.append(float(rng.uniform(.02, .6)))
    while len(live) > 1:
        i, j = sorted(rng.choice(len(live), 2, replace=False))
        a, b = live[i], live[j]
        node = len(parent)
        parent.append(-1)
        length.append(float(rng.uniform(.02, .6)))
        parent[a] = parent[b] = node
        live = [v for k, v in enumerate(live) if k not in (i, j)] + [node]
    return parent, length


def _random_columns(rng, parent, alphabet, n_cols):
    """Columns whose ungapped nodes form one subtree: pick the column's root, keep or drop each child subtree."""
    n = len(parent)
    child = [[] for _ in range(n)]
    for r, p in enumerate(parent):
        if p >= 0:
            child[p].append(r)
    rows = [[] for _ in range(n)]
    for _ in range(n_cols):
        col = ["-"] * n
        root = int(rng.integers(0, n)) if rng.random() < .3 else n - 1
        stack = [root]
        while stack:
            r = stack.pop()
            col[r] = "x" if rng.random() < .05 else alphabet[int(rng.integers(0, len(alphabet)))]
            if not child[r]:
                continue
            keep = [c for c in child[r] if rng.random() < .85]
            # an internal node with a single kept child is fine for the passes but has no sibling for the counts:
            # the reference's accumulateEigenCounts requires both children, so keep both or none
            if len(keep) == len(child[r]):
                stack.extend(keep)
        for r in range(n):
            rows[r].append(col[r])
    return ["".join(r) for r in rows]


@pytest.mark.parametrize("outer", ["matrix cores", "vector units"])
def test_protein_mixture_columns_on_a_nine_leaf_tree(outer, monkeypatch):
    # (the outer-product sums of real eigen bases run on the f64 matrix cores; HX_SUMPROD_NO_MFMA keeps the vector-ALU kernel)
    if outer == "vector units":
        monkeypatch.setenv("HX_SUMPROD_NO_MFMA", "1")
    with open(PROT4) as f:
        omodel = ho.RateModel(json.load(f))
    model = hostmodel.RateModel.load(PROT4)
    rng = np.random.default_rng(9)
    parent, length = _random_tree(rng, 9)
    tree = so.Tree(parent, length, ["n%d" % k for k in range(len(parent))])
    rows = _random_columns(rng, parent, omodel.alphabet, 300)
    weight = rng.uniform(.1, 2., 300)
    sp = so.SumProduct(omodel, tree)
    c, a = sp.C, sp.A
    assert c == 4 and a == 20
    # the device gets the oracle's exp(R t) so that the comparison is of the passes, not of two matrix exponentials
    sub = [[sp.branch_sub[cpt][r] if parent[r] >= 0 else np.zeros((a, a)) for cpt in range(c)] for r in range(len(parent))]
    cc = counts.ColumnCounter(model, parent, length, branch_sub=sub)
    got = cc.run(counts.tokenize_columns(model.alphabet, rows), weight, want_root_post=True)
    root = [np.zeros(a) for _ in range(c)]
    eig = [np.zeros((a, a), dtype=complex) for _ in range(c)]
    gapped = dict(enumerate(rows))
    for col, seq in enumerate(so.columns_of(tree, gapped)):
        sp.init_column(seq)
        sp.fill_up()
        sp.fill_down()
        sp.accumulate_eigen_counts(root, eig, weight[col])
        assert abs(got["col_log_like"][col] - sp.col_log_like) <= 1e-12 * abs(sp.col_log_like), col
        np.testing.assert_allclose(np.exp(got["root_post"][col]), np.exp(sp.log_node_post_prob(sp.column_root())), rtol=1e-8, atol=1e-300)
    want = sp.eigen.get_sub_counts(eig)
    for cpt in range(c):
        np.testing.assert_allclose(got["root_counts"][cpt], root[cpt], rtol=1e-10, atol=1e-12 * root[cpt].max())
        # the two decompositions come from the same numpy call on the same matrix: compare in the eigen basis too
        np.testing.assert_allclose(got["eigen_counts"][cpt], eig[cpt], rtol=0, atol=1e-10 * np.abs(eig[cpt]).max())
        np.testing.assert_allclose(got["counts"][cpt], want[cpt], rtol=0, atol=1e-10 * np.abs(want[cpt]).max())
    assert capi.sumprod_kernel_ms() > 0


def test_complex_eigenvectors_and_column_chunks(monkeypatch):
    """An irreversible model (cyclic rates: complex eigenvalues) takes the four-part basis; a 1 MB scratch budget splits
    the columns into several chunks whose sums must add up."""
    js = {"alphabet": "acgt", "insrate": .01, "delrate": .01, "insextprob": .5, "delextprob": .5,
          "rootprob": {"a": .1, "c": .2, "g": .3, "t": .4},
          "subrate": {"a": {"c": 1., "g": .1}, "c": {"g": 1.2, "t": .05}, "g": {"t": .9, "a": .02}, "t": {"a": 1.1, "c": .3}}}
    omodel, model = ho.RateModel(js), hostmodel.RateModel(js)
    rng = np.random.default_rng(4)
    parent, length = _random_tree(rng, 6)
    tree = so.Tree(parent, length, ["n%d" % k for k in range(len(parent))])
    rows = _random_columns(rng, parent, "acgt", 700)
    cc = counts.ColumnCounter(model, parent, length)
    assert np.abs(np.asarray(cc.eigen.evec).imag).max() > 1e-3
    tok = counts.tokenize_columns("acgt", rows)
    whole = cc.run(tok)
    monkeypatch.setenv("HX_SUMPROD_SCRATCH_MB", "1")
    got = cc.run(tok)
    want_root, want, eig, sp = so.counts_for_alignment(omodel, tree, dict(enumerate(rows)))
    for res in (whole, got):
        np.testing.assert_allclose(res["root_counts"][0], want_root[0], rtol=1e-10)
        np.testing.assert_allclose(res["eigen_counts"][0], eig[0], rtol=0, atol=1e-10 * np.abs(eig[0]).max())
        np.testing.assert_allclose(res["counts"][0], want[0], rtol=0, atol=1e-9 * np.abs(want[0]).max())
    np.testing.assert_array_equal(whole["col_log_like"], got["col_log_like"])


def _random_model(rng, alphabet, reversible):
    a = len(alphabet)
    pi = rng.dirichlet(np.ones(a) * 3.)
    if reversible:
        sym = rng.uniform(.05, 1.5, (a, a))
        sym = (sym + sym.T) / 2.
        rate = sym * pi[None, :]                              # detailed balance: pi_i R_ij = pi_j R_ji (real eigenvectors)
    else:
        rate = rng.uniform(.02, 1.5, (a, a))
        for k in range(a):
            rate[k, (k + 1) % a] += 2.                            # a cyclic drift: complex eigenvalues
    return {"alphabet": alphabet, "insrate": .02, "delrate": .02, "insextprob": .4, "delextprob": .4,
            "rootprob": {alphabet[i]: float(pi[i]) for i in range(a)},
            "subrate": {alphabet[i]: {alphabet[j]: float(rate[i, j]) for j in range(a) if j != i} for i in range(a)}}


@pytest.mark.parametrize("alphabet,reversible", [("acgtu", True), ("acgtu", False), ("abcdefghijklmnopqrstu", True),
                                                 ("abcdefghijklmnopqrstuvwyz012345678", True), ("ab", True)])
def test_alphabets_without_a_kernel_of_their_own(alphabet, reversible):
    """Alphabet sizes other than 4 and 20 take the any-alphabet instantiation (message vectors in private memory): an odd
    size (the scratch stores vector entries in pairs: one padding entry), 21 and 34 symbols (two and three 16 x 16 tiles per
    side of the matrix-core product, rows past the alphabet zero), 2 symbols; real and complex eigenvectors.  Column
    likelihoods, root counts and substitution counts against the oracle."""
    rng = np.random.default_rng(len(alphabet) * 2 + reversible)
    js = _random_model(rng, alphabet, reversible)
    omodel, model = ho.RateModel(js), hostmodel.RateModel(js)
    parent, length = _random_tree(rng, 7)
    tree = so.Tree(parent, length, ["n%d" % k for k in range(len(parent))])
    rows = _random_columns(rng, parent, alphabet, 333)                # (not a multiple of 64: a partly filled block)
    cc = counts.ColumnCounter(model, parent, length)
    assert (np.abs(np.asarray(cc.eigen.evec).imag).max() > 1e-6) == (not reversible)
    got = cc.run(counts.tokenize_columns(alphabet, rows))
    want_root, want, eig, sp = so.counts_for_alignment(omodel, tree, dict(enumerate(rows)))
    np.testing.assert_allclose(got["root_counts"][0], want_root[0], rtol=1e-9)
    np.testing.assert_allclose(got["eigen_counts"][0], eig[0], rtol=0, atol=1e-9 * np.abs(eig[0]).max())
    np.testing.assert_allclose(got["counts"][0], want[0], rtol=0, atol=1e-8 * np.abs(want[0]).max())
    for col, seq in enumerate(so.columns_of(tree, dict(enumerate(rows)))):
        if col % 37:
            continue
        sp.init_column(seq)
        sp.fill_up()
        assert abs(got["col_log_like"][col] - sp.col_log_like) <= 1e-11 * abs(sp.col_log_like) + 1e-13, col      # (an all-wildcard column: 0)


@pytest.mark.parametrize("model_file,fasta,newick,expected", [
    ("testcount.jukescantor.json", "testcount.fa", "testcount.nh", "testcount.out.json"),
    ("testcount.jukescantor.json", "testcount.historian.fa", "testcount.nh", "testcount.count.json"),
    ("testrates.mix2.json", "testcount.mix2.fa", "testcount.mix2.nh", "testcount.mix2.count.json")])
def test_count_of_a_reconstruction_prints_the_reference_files(model_file, fasta, newick, expected):
    """`historian count -recon` end to end (reference Makefile testcount): the counts file, character for character -
    single-component, and the two-component cyclic mixture whose eigenvectors are complex."""
    _, model, tree, gapped = _fixture(model_file, fasta, newick)
    indel, root, sub = counts.count_reconstruction(model, tree.parent, tree.branch_length, [gapped[n] for n in range(tree.nodes())])
    with open(G + expected) as f:
        assert counts.event_counts_json(model.alphabet, indel, root, sub) == f.read()


def test_refused_arguments():
    model = hostmodel.RateModel.load(G + "testnj.jukescantor.json")
    cc = counts.ColumnCounter(model, [2, 2, -1], [.1, .2, 0.])
    tok = np.zeros((4, 3), dtype=np.int8)
    cc.run(tok)
    bad = counts.ColumnCounter(model, [2, 2, -1], [.1, .2, 0.])
    bad.parent = np.array([-1, 0, 0], dtype=np.int32)          # a parent before its children
    with pytest.raises(capi.HxError) as e:
        bad.run(tok)
    assert e.value.code == -5
    bad_tok = tok.copy()
    bad_tok[2, 1] = 4                                          # not a token of a four-letter alphabet
    with pytest.raises(capi.HxError) as e:
        cc.run(bad_tok)
    assert e.value.code == -8
    three = counts.ColumnCounter(model, [3, 3, 3, -1], [.1, .2, .3, 0.])
    with pytest.raises(capi.HxError):
        three.run(np.zeros((4, 4), dtype=np.int8))
