"""Helpers for the whole-tree reconstruction parity test (config 3: gp120)."""
import os

import numpy as np

from oracle import c_oracle
from oracle import historian_oracle as ho
from oracle.ref_mains import read_fasta
from tests import helpers as H


def parse_newick(text):
    """Minimal Newick reader -> ho.ReconTree with nodes in post-order (children before parents,
    left to right, root last).  Internal nodes get the name 'node<k>'."""
    text = text.strip().rstrip(";")
    pos = 0
    parent, blen, name = [], [], []

    def parse_node():
        nonlocal pos
        kids = []
        if text[pos] == "(":
            pos += 1
            while True:
                kids.append(parse_node())
                if text[pos] == ",":
                    pos += 1
                    continue
                assert text[pos] == ")"
                pos += 1
                break
        start = pos
        while pos < len(text) and text[pos] not in ",():":
            pos += 1
        nm = text[start:pos]
        length = 0.
        if pos < len(text) and text[pos] == ":":
            pos += 1
            start = pos
            while pos < len(text) and text[pos] not in ",()":
                pos += 1
            length = float(text[start:pos])
        idx = len(parent)
        parent.append(-1)
        blen.append(length)
        name.append(nm if nm else "node%d" % idx)
        for k in kids:
            parent[k] = idx
        return idx

    parse_node()
    return ho.ReconTree(parent, blen, name)


class ArrayForward(ho.ForwardMatrix):
    """oracle ForwardMatrix whose cells come from the plain-C oracle fill (dense array)."""

    def __init__(self, x, y, hmm, node, env):
        super().__init__(x, y, hmm, node, env, fill=False)
        r = c_oracle.forward(*H.job_images(self))
        self.arr = r["cells"]
        self.lp_end = r["lp_end"]

    def cell(self, i, j, s):
        if i >= self.arr.shape[0] or j >= self.arr.shape[1]:
            return H.NEG_INF
        return float(self.arr[i, j, s])

    def xy_cell(self, i, j):
        if i >= self.arr.shape[0] or j >= self.arr.shape[1]:
            return ho._EMPTY_CELL
        return [float(v) for v in self.arr[i, j]]


class ArrayBackward(ho.BackwardMatrix):
    """oracle BackwardMatrix whose cells come from the plain-C oracle fill."""

    def __init__(self, fwd):
        super().__init__(fwd, fill=False)
        self.arr = c_oracle.backward(*H.job_images(fwd))["cells"]

    def cell(self, i, j, s):
        if i >= self.arr.shape[0] or j >= self.arr.shape[1]:
            return H.NEG_INF
        return float(self.arr[i, j, s])

    def xy_cell(self, i, j):
        if i >= self.arr.shape[0] or j >= self.arr.shape[1]:
            return ho._EMPTY_CELL
        return [float(v) for v in self.arr[i, j]]


def load_family(tree_path, seqs_path, guide_path, max_len=None, leaves=None):
    """tree + {leaf node: (name, seq)} + guide AlignPath; optionally truncated (first max_len guide
    columns) and pruned to the first `leaves` leaves of a caterpillar for quick tests."""
    tree = parse_newick(open(tree_path).read())
    ung = dict(read_fasta(seqs_path))
    gap = dict(read_fasta(guide_path))
    if max_len is not None:
        gap = {k: v[:max_len] for k, v in gap.items()}
        ung = {k: "".join(c for c in gap[k] if c not in "-.") for k in gap}
    seqs, guide = {}, {}
    for n in range(tree.nodes()):
        if tree.is_leaf(n):
            nm = tree.name[n]
            seqs[n] = (nm, ung[nm])
            guide[n] = [c not in "-." for c in gap[nm]]
    return tree, seqs, guide


def write_job(path, model_path, tree, seqs, guide, seqs_fa, guide_fa, **opts):
    with open(seqs_fa, "w") as f:
        for n, (nm, s) in seqs.items():
            f.write(">%s\n%s\n" % (nm, s))
    if guide:
        with open(guide_fa, "w") as f:
            for n, (nm, s) in seqs.items():
                k, row = 0, []
                for b in guide[n]:
                    row.append(s[k] if b else "-")
                    k += 1 if b else 0
                f.write(">%s\n%s\n" % (nm, "".join(row)))
    with open(path, "w") as f:
        f.write("model %s\nseqs %s\n" % (model_path, seqs_fa))
        if guide:
            f.write("guide %s\n" % guide_fa)
        for k, v in opts.items():
            f.write("%s %s\n" % (k, v))
        f.write("tree %d\n" % tree.nodes())
        for n in range(tree.nodes()):
            f.write("%d %r %s\n" % (tree.parent[n], tree.branch_length[n], tree.name[n]))


def oracle_reconstruct(model_path, tree, seqs, guide, **kw):
    model = ho.RateModel.from_file(model_path)
    model.sub_rate = [m.tolist() for m in model.sub_rate]
    res = ho.reconstruct(model, tree, seqs, guide, forward_factory=ArrayForward, backward_factory=ArrayBackward, **kw)
    rows = ho.gapped_rows(tree, seqs, res["path"])
    return res, rows


def fasta_rows(tree, rows):
    """the reference's `-output fasta` of a reconstruction: one record per tree node in node order; an unnamed
    internal node is called by its subtree (Tree::seqName, reference src/tree.cpp:464-477: default ostream format)"""
    def seq_name(n):
        if tree.is_leaf(n) or not tree.name[n].startswith("node"):
            return tree.name[n]
        return "(" + ",".join("%s:%g" % (seq_name(c), tree.branch_length[c]) for c in tree.child[n]) + ")"
    return "".join(">%s\n%s\n" % (seq_name(n), rows[n]) for n in sorted(rows))


def parse_hxrecon(text):
    out = {"rows": {}, "bands": {}}
    for line in text.splitlines():
        f = line.split()
        if f[0] in ("lpFinalFwd", "lpFinalTrace"):
            out[f[0]] = float.fromhex(f[1])
        elif f[0] == "band":
            out["bands"][int(f[1])] = int(f[2])
        elif f[0] == "row":
            out["rows"][int(f[1])] = f[3] if len(f) > 3 else ""
    return out


def balanced_family(n_leaves, length, alphabet, seed, branch=.05):
    """Balanced binary tree with n_leaves (power of two) synthetic sequences evolved down the tree
    (substitutions and indels); post-order ReconTree + {leaf: (name, seq)}; no guide alignment."""
    import random
    rng = random.Random(seed)
    parent, blen, name, seqs = [], [], [], {}

    def build(depth, anc):
        if depth == 0:
            idx = len(parent)
            parent.append(-1); blen.append(branch); name.append("leaf%d" % idx)
            seqs[idx] = (name[idx], anc if anc else alphabet[0])
            return idx
        kids = [build(depth - 1, H.mutate(rng, anc, alphabet, .08, .02)) for _ in range(2)]
        idx = len(parent)
        parent.append(-1); blen.append(branch); name.append("node%d" % idx)
        for k in kids:
            parent[k] = idx
        return idx

    depth = n_leaves.bit_length() - 1
    build(depth, H.random_seq(rng, alphabet, length))
    return ho.ReconTree(parent, blen, name), seqs
