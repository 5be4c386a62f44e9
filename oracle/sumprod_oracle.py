"""TEST INFRASTRUCTURE ONLY (see oracle/README in DESIGN.md section 3): CPU restatement of the reference's column
sum-product and substitution-count accumulation - SURVEY section 8(f) row N3.

  SumProduct.init_column / fill_up / fill_down            reference src/sumprod.cpp:58-198
  log_branch_post_prob / log_node_post_prob               src/sumprod.cpp:208-217, 252-257
  accumulate_root_counts / accumulate_eigen_counts        src/sumprod.cpp:264-271, 294-372
  EigenModel (eigenSubCount, getSubCounts)                src/model.cpp:1135-1200, 1329-1373
  write_sub_counts                                        src/model.cpp:657-700
  main_testsumprod / main_testaligncount                  t/testsumprod.cpp, t/testaligncount.cpp

The reference diagonalises the rate matrix with GSL (gsl_eigen_nonsymmv + complex LU), which is not vendored and not in
this image; here numpy.linalg.eig / inv take its place.  Eigenvectors are only defined up to scale and order, but every
quantity built from them below (exp(Rt), the expected counts) is basis independent, so the printed results agree at the
fixtures' precision (6 significant digits).  PINNED by the reference's own fixtures, byte for byte
(tests/test_oracle_sumprod.py): data/testsumprod.out, data/testaligncount.out (-sub and -eigen give the same file),
data/testaligncount2.out.json."""
import math

import numpy as np

from oracle import historian_oracle as ho

RESCALE_THRESHOLD = 1e-30           # SUMPROD_RESCALE_THRESHOLD, src/sumprod.cpp:8
EPSILON = 1e-6                      # EIGENMODEL_EPSILON, src/model.cpp:22
GAP, WILD = "-", "*"


def _fcmp_equal(a, b, eps):
    """gsl_fcmp(a, b, eps) == 0"""
    m = a if abs(a) > abs(b) else b
    _, e = math.frexp(m)
    return abs(a - b) <= math.ldexp(eps, e)


class Tree:
    """Node arrays in the reference's order (children before parents, root last: src/tree.cpp, via tests'
    parse_newick).  Only what SumProduct reads."""

    def __init__(self, parent, branch_length, name):
        self.parent, self.branch_length, self.name = list(parent), list(branch_length), list(name)
        self.child = [[] for _ in parent]
        for n, p in enumerate(parent):
            if p >= 0:
                self.child[p].append(n)

    def nodes(self):
        return len(self.parent)

    def siblings(self, n):
        return [c for c in self.child[self.parent[n]] if c != n]

    def preorder(self):
        out = []

        def visit(n):
            out.append(n)
            for c in self.child[n]:
                visit(c)
        for n in range(self.nodes()):
            if self.parent[n] < 0:
                visit(n)
        return out


class EigenModel:
    """src/model.cpp:1135-1200"""

    def __init__(self, model):
        self.model = model
        self.ev, self.evec, self.evec_inv = [], [], []
        for r in model.sub_rate:
            w, v = np.linalg.eig(np.asarray(r, dtype=float))
            self.ev.append(w.astype(complex))
            self.evec.append(v.astype(complex))
            self.evec_inv.append(np.linalg.inv(v.astype(complex)))

    def eigen_sub_count(self, t):
        """J[k][l] = integral_0^t exp(ev_k s) exp(ev_l (t - s)) ds  (src/model.cpp:1329-1350)"""
        out = []
        for ev in self.ev:
            a = len(ev)
            e = np.exp(ev * t)
            j = np.zeros((a, a), dtype=complex)
            for k in range(a):
                for l in range(a):
                    same = k == l or (_fcmp_equal(ev[k].real, ev[l].real, EPSILON) and _fcmp_equal(ev[k].imag, ev[l].imag, EPSILON))
                    j[k, l] = e[k] * t if same else (e[k] - e[l]) / (ev[k] - ev[l])
            out.append(j)
        return out

    def get_sub_counts(self, eigen_counts):
        """src/model.cpp:1352-1373: counts[i][j] = Re(sum_k evecInv[k][i] sum_l eigenCounts[k][l] evec[j][l]) * (i == j ? 1 : R_ij)"""
        out = []
        for cpt, ec in enumerate(eigen_counts):
            r = np.asarray(self.model.sub_rate[cpt], dtype=float)
            c = np.real(self.evec_inv[cpt].T @ ec @ self.evec[cpt].T)
            scale = r.copy()
            np.fill_diagonal(scale, 1.)
            out.append(c * scale)
        return out


class SumProduct:
    """src/sumprod.h, src/sumprod.cpp.  Linear-space messages with per-node log scale factors:
    F (a node's subtree given its state), E (the same seen from its parent), G (everything outside the subtree)."""

    def __init__(self, model, tree):
        self.model, self.tree = model, tree
        self.C, self.A, self.N = model.components(), len(model.alphabet), tree.nodes()
        self.pre = tree.preorder()
        self.post = self.pre[::-1]
        self.eigen = EigenModel(model)
        self.ins_prob = [np.asarray(p, dtype=float) for p in model.ins_prob]
        self.log_cpt_weight = [math.log(w) for w in model.cpt_weight]
        self.branch_sub = [[None] * self.N for _ in range(self.C)]
        self.branch_esc = [[None] * self.N for _ in range(self.C)]
        for r in range(self.N - 1):
            sub = ho.ProbModel(model, tree.branch_length[r]).sub_mat
            esc = self.eigen.eigen_sub_count(tree.branch_length[r])
            for cpt in range(self.C):
                self.branch_sub[cpt][r] = np.asarray(sub[cpt], dtype=float)
                self.branch_esc[cpt][r] = esc[cpt]
        z = lambda: [[np.zeros(self.A) for _ in range(self.N)] for _ in range(self.C)]
        self.E, self.F, self.G = z(), z(), z()
        self.logE = [[0.] * self.N for _ in range(self.C)]
        self.logF = [[0.] * self.N for _ in range(self.C)]
        self.logG = [[0.] * self.N for _ in range(self.C)]
        self.cpt_log_like = [0.] * self.C
        self.col_log_like = -math.inf

    def is_gap(self, r):
        return self.col[r] == GAP

    def tokenize(self, ch):
        a = self.model.alphabet
        k = a.find(ch)
        if k < 0:
            k = a.find(ch.lower() if ch.isupper() else ch.upper())
        return k

    def init_column(self, seq):
        """seq: {row: char}  (src/sumprod.cpp:58-86)"""
        self.col = [GAP] * self.N
        self.ungapped, self.roots = [], []
        for r in range(self.N):
            if r in seq:
                self.col[r] = seq[r] if self.tokenize(seq[r]) >= 0 else WILD
                self.ungapped.append(r)
        for r in range(self.N):
            if self.is_gap(r):
                for cpt in range(self.C):
                    self.E[cpt][r] = np.ones(self.A)
                    self.logE[cpt][r] = 0.
            else:
                rp = self.tree.parent[r]
                if rp < 0 or self.is_gap(rp):
                    self.roots.append(r)

    def column_root(self):
        assert len(self.roots) == 1, "column needs exactly one root"
        return self.roots[0]

    def fill_up(self):
        """tip-to-root messages (src/sumprod.cpp:99-161)"""
        self.col_log_like = -math.inf
        t = self.tree
        for cpt in range(self.C):
            self.cpt_log_like[cpt] = 0.
            for r in self.post:
                self.logF[cpt][r] = sum(self.logE[cpt][c] for c in t.child[r])
                if self.is_gap(r):
                    continue
                ch = self.col[r]
                if ch == WILD:
                    f = np.ones(self.A)
                    for c in t.child[r]:
                        f = f * self.E[cpt][c]
                    fmax = max(0., float(f.max()))
                    if fmax < RESCALE_THRESHOLD:
                        f = f / fmax
                        self.logF[cpt][r] += math.log(fmax)
                    self.F[cpt][r] = f
                else:
                    tok = self.tokenize(ch)
                    ftok = 1.
                    for c in t.child[r]:
                        ftok *= self.E[cpt][c][tok]
                    if ftok < RESCALE_THRESHOLD:
                        self.logF[cpt][r] += math.log(ftok)
                        ftok = 1.
                    f = np.zeros(self.A)
                    f[tok] = ftok
                    self.F[cpt][r] = f
                rp = t.parent[r]
                if rp < 0 or self.is_gap(rp):
                    self.cpt_log_like[cpt] += self.logF[cpt][r] + math.log(float(np.dot(self.F[cpt][r], self.ins_prob[cpt])))
                else:
                    self.logE[cpt][r] = self.logF[cpt][r]
                    self.E[cpt][r] = self.branch_sub[cpt][r] @ self.F[cpt][r]
            self.col_log_like = ho.log_sum_exp(self.col_log_like, self.log_cpt_weight[cpt] + self.cpt_log_like[cpt])

    def fill_down(self):
        """root-to-tip messages (src/sumprod.cpp:163-198)"""
        t = self.tree
        for cpt in range(self.C):
            if not self.ungapped:
                continue
            for r in self.pre:
                if self.is_gap(r):
                    continue
                rp = t.parent[r]
                if rp < 0 or self.is_gap(rp):
                    self.G[cpt][r] = self.ins_prob[cpt].copy()
                    self.logG[cpt][r] = 0.
                else:
                    sibs = t.siblings(r)
                    self.logG[cpt][r] = self.logG[cpt][rp] + sum(self.logE[cpt][s] for s in sibs)
                    w = self.G[cpt][rp].copy()
                    for s in sibs:
                        if not self.is_gap(s):
                            w = w * self.E[cpt][s]
                    self.G[cpt][r] = w @ self.branch_sub[cpt][r]

    def log_node_post_prob(self, node):
        lpp = []
        for i in range(self.A):
            lp = -math.inf
            for cpt in range(self.C):
                lp = ho.log_sum_exp(lp, self.log_cpt_weight[cpt] + self.logF[cpt][node] + _log(self.F[cpt][node][i]) +
                                    self.logG[cpt][node] + _log(self.G[cpt][node][i]) - self.col_log_like)
            lpp.append(min(lp, 0.))
        return lpp

    def log_branch_post_prob(self, cpt, node, a, b):
        t = self.tree
        parent, sib = t.parent[node], t.siblings(node)[0]
        return (self.log_cpt_weight[cpt] + self.logG[cpt][parent] + _log(self.G[cpt][parent][a]) + _log(self.branch_sub[cpt][node][a][b]) +
                self.logF[cpt][node] + _log(self.F[cpt][node][b]) + self.logE[cpt][sib] + _log(self.E[cpt][sib][a]) - self.col_log_like)

    def accumulate_root_counts(self, root_counts, weight=1.):
        root = self.column_root()
        for cpt in range(self.C):
            norm = math.exp(self.log_cpt_weight[cpt] + self.logF[cpt][root] - self.col_log_like)
            root_counts[cpt] += weight * self.ins_prob[cpt] * self.F[cpt][root] * norm

    def accumulate_eigen_counts(self, root_counts, eigen_counts, weight=1.):
        """src/sumprod.cpp:294-372: eigenCounts[k][l] += Dbasis[k] * eigenSubCount[k][l] * Ubasis[l] * weight / norm"""
        self.accumulate_root_counts(root_counts, weight)
        root, t = self.column_root(), self.tree
        for node in self.ungapped:
            if node == root:
                continue
            parent, sib = t.parent[node], t.siblings(node)[0]
            for cpt in range(self.C):
                u0 = self.F[cpt][node]
                d0 = self.G[cpt][parent] * self.E[cpt][sib]
                max_u, max_d = float(u0.max()), float(d0.max())
                norm = math.exp(self.col_log_like - self.log_cpt_weight[cpt] - self.logF[cpt][node] - self.logG[cpt][parent] -
                                self.logE[cpt][sib]) / (max_u * max_d)
                ub = self.eigen.evec_inv[cpt] @ (u0 / max_u)
                db = (d0 / max_d) @ self.eigen.evec[cpt]
                eigen_counts[cpt] += np.outer(db, ub) * self.branch_esc[cpt][node] * (weight / norm)


def _log(x):
    return math.log(x) if x > 0 else -math.inf


def columns_of(tree, gapped):
    """gapped: {node: aligned row}; yields {row: char} per alignment column (AlignColSumProduct, src/sumprod.cpp:374-396)"""
    width = len(next(iter(gapped.values())))
    for col in range(width):
        yield {r: gapped[r][col] for r in range(tree.nodes()) if gapped[r][col] not in "-."}


def counts_for_alignment(model, tree, gapped):
    """root counts [C][A], substitution counts + wait times [C][A][A] of a fixed alignment (t/testaligncount.cpp -eigen)"""
    sp = SumProduct(model, tree)
    a = len(model.alphabet)
    root = [np.zeros(a) for _ in range(sp.C)]
    eig = [np.zeros((a, a), dtype=complex) for _ in range(sp.C)]
    for seq in columns_of(tree, gapped):
        sp.init_column(seq)
        sp.fill_up()
        sp.fill_down()
        sp.accumulate_eigen_counts(root, eig)
    return root, sp.eigen.get_sub_counts(eig), eig, sp


def _g(x):
    """C++ ostream << double at default precision (%g)"""
    s = "%g" % x
    return "0" if s == "-0" else s


def write_sub_counts(model, root, counts):
    """AlphabetOwner::writeSubCounts for a single component (src/model.cpp:657-700)"""
    assert len(root) == 1
    alph, r, c = model.alphabet, root[0], counts[0]
    out = ["{", " \"root\":", "  {" + ",".join("\n   \"%s\": %s" % (alph[i], _g(r[i])) for i in range(len(alph))), "  },", " \"sub\":"]
    rows = []
    for i in range(len(alph)):
        rows.append("\n   \"%s\": {%s }" % (alph[i], ",".join(" \"%s\": %s" % (alph[j], _g(c[i][j])) for j in range(len(alph)) if j != i)))
    out.append("  {" + ",".join(rows))
    out += ["  },", " \"wait\":", "  {" + ",".join("\n   \"%s\": %s" % (alph[i], _g(c[i][i])) for i in range(len(alph))), "  }", "}"]
    return "\n".join(out)


def main_testsumprod(model, tree, gapped):
    """t/testsumprod.cpp: branch and root posteriors of every column"""
    sp = SumProduct(model, tree)
    out = []
    for col, seq in enumerate(columns_of(tree, gapped)):
        sp.init_column(seq)
        sp.fill_up()
        sp.fill_down()
        out.append("Column #%d" % col)
        root = sp.column_root()
        for node in sp.ungapped:
            if node == root:
                continue
            p = tree.parent[node]
            for cpt in range(sp.C):
                for a in range(sp.A):
                    for b in range(sp.A):
                        out.append("P( %s = %s%d , %s = %s%d ) = %s" % (tree.name[p], model.alphabet[a], cpt, tree.name[node], model.alphabet[b], cpt,
                                                                        _g(math.exp(sp.log_branch_post_prob(cpt, node, a, b)))))
        lnpp = sp.log_node_post_prob(root)
        for a in range(sp.A):
            out.append("P( %s = %s ) = %s" % (tree.name[root], model.alphabet[a], _g(math.exp(lnpp[a]))))
        out.append("")
    return "\n".join(out) + "\n"


def main_testaligncount(model, tree, gapped):
    root, counts, _, _ = counts_for_alignment(model, tree, gapped)
    return write_sub_counts(model, root, counts) + "\n"


# ---- counts of a fixed reconstruction: `historian count -recon` (src/recon.cpp:1284-1291, src/model.cpp:847-923) ----

def decay_wait_time(rate, t):
    """IndelCounts::decayWaitTime (src/model.cpp:1106-1108)"""
    return 1 / rate - t / (math.exp(rate * t) - 1)


def _trans_prob(pm, src, dest):
    """ProbModel::transProb (src/model.cpp:400-447); states M, I, D, E"""
    ins, dele, ie, de = pm.ins, pm.dele, pm.ins_ext, pm.del_ext
    if src == "M":
        return {"M": (1 - ins) * (1 - dele), "I": ins, "D": (1 - ins) * dele, "E": 1 - ins}[dest]
    if src == "I":
        return {"M": (1 - ie) * (1 - dele), "I": ie, "D": (1 - ie) * dele, "E": 1 - ie}[dest]
    return {"M": 1 - de, "E": 1 - de, "I": 0., "D": de}[dest]


def indel_counts_of_branch(model, t, parent_row, child_row, c, weight=1.):
    """IndelCounts::accumulateIndelCounts on one branch (src/model.cpp:847-893).  parent_row, child_row: per-column
    presence flags; c: dict ins del insExt delExt insTime delTime lp, updated in place."""
    ins_wait, del_wait = decay_wait_time(model.ins_rate, t), decay_wait_time(model.del_rate, t)
    pm = ho.ProbModel(model, t, sub_mat=[])
    state = "M"
    for p, ch in zip(parent_row, child_row):
        if p and ch:
            nxt = "M"
        elif p:
            nxt = "D"
        elif ch:
            nxt = "I"
        else:
            continue
        if nxt == "M":
            if state == nxt:
                c["insTime"] += weight * t
                c["delTime"] += weight * t
        elif nxt == "I":
            if state == nxt:
                c["insExt"] += weight
            else:
                c["ins"] += weight
                c["insTime"] += weight * ins_wait
        else:
            if state == nxt:
                c["delExt"] += weight
            else:
                c["del"] += weight
                c["delTime"] += weight * del_wait
        c["lp"] += _log(_trans_prob(pm, state, nxt)) * weight
        state = nxt
    c["lp"] += _log(_trans_prob(pm, state, "E")) * weight


def count_reconstruction(model, tree, gapped):
    """EigenCounts::accumulateCounts + transform for one dataset with a reconstruction: indel counts of every branch,
    substitution counts of every column, log-likelihood = indel path + column likelihoods.
    -> (indel dict, root counts [C][A], counts [C][A][A])"""
    c = dict(ins=0., insExt=0., insTime=0., delExt=0., delTime=0., lp=0.)
    c["del"] = 0.
    present = {n: [ch not in "-." for ch in gapped[n]] for n in range(tree.nodes())}
    for node in range(tree.nodes() - 1):
        indel_counts_of_branch(model, tree.branch_length[node], present[tree.parent[node]], present[node], c)
    sp = SumProduct(model, tree)
    a = len(model.alphabet)
    root = [np.zeros(a) for _ in range(sp.C)]
    eig = [np.zeros((a, a), dtype=complex) for _ in range(sp.C)]
    for seq in columns_of(tree, gapped):
        sp.init_column(seq)
        sp.fill_up()
        sp.fill_down()
        sp.accumulate_eigen_counts(root, eig)
        c["lp"] += sp.col_log_like
    return c, root, sp.eigen.get_sub_counts(eig)


def _sub_counts_component(alph, r, c, indent):
    """AlphabetOwner::writeSubCountsComponent (src/model.cpp:675-705)"""
    ind = " " * indent
    n = range(len(alph))
    out = ind + "{\n" + ind + " \"root\":\n" + ind + "  {"
    out += ",".join("\n" + ind + "   \"%s\": %s" % (alph[i], _g(r[i])) for i in n)
    out += "\n" + ind + "  },\n" + ind + " \"sub\":\n" + ind + "  {"
    out += ",".join("\n" + ind + "   \"%s\": {" % alph[i] + ",".join(" \"%s\": %s" % (alph[j], _g(c[i][j])) for j in n if j != i) + " }" for i in n)
    out += "\n" + ind + "  },\n" + ind + " \"wait\":\n" + ind + "  {"
    out += ",".join("\n" + ind + "   \"%s\": %s" % (alph[i], _g(c[i][i])) for i in n)
    out += "\n" + ind + "  }\n" + ind + "}"
    return out


def write_event_counts(model, indel, root, counts):
    """EventCounts::writeJson (src/model.cpp:933-956, 657-673)"""
    alph = model.alphabet
    out = "{\n \"alphabet\": \"%s\",\n \"indel\":\n" % alph
    out += "  {\n" + "".join("   \"%s\": %s%s\n" % (k, _g(indel[k]), "," if k not in ("insTime", "delTime") else "")
                            for k in ("ins", "del", "insExt", "delExt", "insTime", "delTime")) + "  }"
    out += ",\n \"sub\":\n"
    if len(root) > 1:
        out += "  {\n   \"mixture\": [\n"
        out += ",\n".join(_sub_counts_component(alph, root[k], counts[k], 4) for k in range(len(root))) + "\n"
        out += "   ]\n  }"
    else:
        out += _sub_counts_component(alph, root[0], counts[0], 2)
    out += ",\n \"logLikelihood\": %s\n}\n" % _g(indel["lp"])
    return out
