"""Counts mode on a fixed alignment: the host side of hx_sumprod_columns.

Mirrors the reference's EigenModel (src/model.cpp:1135-1373) and EigenCounts::accumulateSubstitutionCounts
(src/sumprod.cpp:430-459): the rate matrices are diagonalised once on the host (numpy in place of GSL), every column
of the alignment goes through the device's sum-product kernel in one launch, and getSubCounts turns the summed
eigen-basis matrix into wait times (diagonal) and substitution counts (off-diagonal).  The device path is the only
path: without the HIP library capi.load() raises."""
import math

import numpy as np

from . import capi

GAP, WILD = -2, -1
EPSILON = 1e-6        # EIGENMODEL_EPSILON: eigenvalues this close are treated as degenerate (src/model.cpp:1339)


def _fcmp_equal(a, b, eps):
    m = a if abs(a) > abs(b) else b
    return abs(a - b) <= math.ldexp(eps, math.frexp(m)[1])


class EigenModel:
    def __init__(self, model):
        self.model = model
        self.ev, self.evec, self.evec_inv = [], [], []
        for r in model.sub_rate:
            w, v = np.linalg.eig(np.asarray(r, dtype=float))
            v = v.astype(complex)
            self.ev.append(w.astype(complex))
            self.evec.append(v)
            self.evec_inv.append(np.linalg.inv(v))

    def sub_prob(self, t):
        """exp(R t) of every component (EigenModel::getSubProbMatrix)"""
        out = []
        for ev, v, vi in zip(self.ev, self.evec, self.evec_inv):
            p = np.real((v * np.exp(ev * t)) @ vi)
            out.append(np.clip(p, 0., 1.))
        return out

    def eigen_sub_count(self, t):
        """J[k][l] = integral over the branch of exp(ev_k s) exp(ev_l (t - s))  (EigenModel::eigenSubCount)"""
        out = []
        for ev in self.ev:
            e = np.exp(ev * t)
            a = len(ev)
            dk, dl = np.meshgrid(ev, ev, indexing="ij")
            ek, el = np.meshgrid(e, e, indexing="ij")
            same = np.eye(a, dtype=bool)
            for k in range(a):
                for l in range(a):
                    if k != l and _fcmp_equal(ev[k].real, ev[l].real, EPSILON) and _fcmp_equal(ev[k].imag, ev[l].imag, EPSILON):
                        same[k, l] = True
            with np.errstate(divide="ignore", invalid="ignore"):
                j = np.where(same, ek * t, (ek - el) / np.where(same, 1., dk - dl))
            out.append(j)
        return out

    def sub_counts(self, eigen_counts):
        """EigenModel::getSubCounts: [C][A][A], wait times on the diagonal, substitution counts off it"""
        out = []
        for cpt, ec in enumerate(eigen_counts):
            scale = np.array(self.model.sub_rate[cpt], dtype=float)
            np.fill_diagonal(scale, 1.)
            out.append(np.real(self.evec_inv[cpt].T @ ec @ self.evec[cpt].T) * scale)
        return out


def tokenize_columns(alphabet, rows):
    """rows: [N] gapped strings of equal length, one per tree node (gap characters '-' and '.').
    -> int8 [n_cols][N] tokens: -2 gap, -1 wildcard (a character outside the alphabet, either case)."""
    n, width = len(rows), len(rows[0])
    lut = np.full(256, WILD, dtype=np.int8)
    for k, ch in enumerate(alphabet):
        lut[ord(ch.lower())] = lut[ord(ch.upper())] = k
    lut[ord("-")] = lut[ord(".")] = GAP
    out = np.empty((width, n), dtype=np.int8)
    for r, row in enumerate(rows):
        if len(row) != width:
            raise ValueError("alignment rows differ in length")
        out[:, r] = lut[np.frombuffer(row.encode("latin-1"), dtype=np.uint8)]
    return out


class ColumnCounter:
    """Sum-product over alignment columns on one tree (children-before-parents node order, root last)."""

    def __init__(self, model, parent, branch_length, branch_sub=None):
        self.model, self.parent = model, np.asarray(parent, dtype=np.int32)
        self.eigen = EigenModel(model)
        c, a, n = model.components(), len(model.alphabet), len(parent)
        self.ins_prob = np.asarray(model.root, dtype=float).reshape(c, a)
        self.log_cpt_weight = np.log(np.asarray(model.cpt_weight, dtype=float))
        self.branch_sub = np.zeros((c, n, a, a))
        self.esc = np.zeros((c, n, a, a), dtype=complex)
        for r in range(n):
            if parent[r] < 0:
                continue
            sub = branch_sub[r] if branch_sub is not None else self.eigen.sub_prob(branch_length[r])
            esc = self.eigen.eigen_sub_count(branch_length[r])
            for cpt in range(c):
                self.branch_sub[cpt, r] = sub[cpt]
                self.esc[cpt, r] = esc[cpt]

    def run(self, tokens, weight=None, want_root_post=False):
        """-> dict(col_log_like [n_cols], root_counts [C][A], eigen_counts [C][A][A] complex, counts [C][A][A], root_post)"""
        cll, root, eig, post = capi.sumprod_columns(self.parent, self.ins_prob, self.log_cpt_weight, self.branch_sub,
                                                    np.asarray(self.eigen.evec), np.asarray(self.eigen.evec_inv), self.esc, tokens,
                                                    weight, want_root_post)
        return dict(col_log_like=cll, root_counts=root, eigen_counts=eig, counts=self.eigen.sub_counts(eig), root_post=post)


# ---- `historian count -recon`: event counts of a fixed reconstruction (src/recon.cpp:1284-1291) ----

def decay_wait_time(rate, t):
    """IndelCounts::decayWaitTime"""
    return 1 / rate - t / (math.exp(rate * t) - 1)


def _branch_indels(model, t, parent_in, child_in, c):
    """IndelCounts::accumulateIndelCounts of one branch (src/model.cpp:847-893): a three-state walk over the columns
    where either end of the branch has a residue.  Host work: O(columns) byte comparisons per branch."""
    ins, dele = 1 - math.exp(-model.ins_rate * t), 1 - math.exp(-model.del_rate * t)
    ie, de = model.ins_ext, model.del_ext
    ins_wait, del_wait = decay_wait_time(model.ins_rate, t), decay_wait_time(model.del_rate, t)
    trans = {("M", "M"): (1 - ins) * (1 - dele), ("M", "I"): ins, ("M", "D"): (1 - ins) * dele, ("M", "E"): 1 - ins,
             ("I", "M"): (1 - ie) * (1 - dele), ("I", "I"): ie, ("I", "D"): (1 - ie) * dele, ("I", "E"): 1 - ie,
             ("D", "M"): 1 - de, ("D", "E"): 1 - de, ("D", "I"): 0., ("D", "D"): de}
    used = parent_in | child_in
    kinds = np.where(parent_in & child_in, 0, np.where(parent_in, 2, 1))[used]        # 0 match, 1 insert, 2 delete
    state = "M"
    for k in kinds:
        nxt = "MID"[k]
        if nxt == "M":
            if state == "M":
                c["insTime"] += t
                c["delTime"] += t
        elif nxt == "I":
            if state == "I":
                c["insExt"] += 1
            else:
                c["ins"] += 1
                c["insTime"] += ins_wait
        else:
            if state == "D":
                c["delExt"] += 1
            else:
                c["del"] += 1
                c["delTime"] += del_wait
        p = trans[(state, nxt)]
        c["lp"] += math.log(p) if p > 0 else -math.inf
        state = nxt
    p = trans[(state, "E")]
    c["lp"] += math.log(p) if p > 0 else -math.inf


def count_reconstruction(model, parent, branch_length, rows):
    """EigenCounts::accumulateCounts + transform for a dataset that comes with its reconstruction.
    rows: gapped strings, one per tree node (ancestors included, '*' where the residue is unknown).
    -> (indel dict, root counts [C][A], substitution counts and wait times [C][A][A])"""
    tok = tokenize_columns(model.alphabet, rows)
    present = tok != GAP
    c = {"ins": 0., "del": 0., "insExt": 0., "delExt": 0., "insTime": 0., "delTime": 0., "lp": 0.}
    for node, p in enumerate(parent):
        if p >= 0:
            _branch_indels(model, branch_length[node], present[:, p], present[:, node], c)
    res = ColumnCounter(model, parent, branch_length).run(tok)
    c["lp"] += float(np.sum(res["col_log_like"]))
    return c, res["root_counts"], res["counts"]


def _g(x):
    s = "%g" % x
    return "0" if s == "-0" else s


def event_counts_json(alphabet, indel, root, sub):
    """EventCounts::writeJson (src/model.cpp:933-956): the counts file of `historian count`, character for character"""
    def component(r, c, indent):
        ind, n = " " * indent, range(len(alphabet))
        body = [ind + "{", ind + " \"root\":",
                ind + "  {" + ",".join("\n%s   \"%s\": %s" % (ind, alphabet[i], _g(r[i])) for i in n), ind + "  },", ind + " \"sub\":",
                ind + "  {" + ",".join("\n%s   \"%s\": {%s }" % (ind, alphabet[i], ",".join(" \"%s\": %s" % (alphabet[j], _g(c[i][j])) for j in n if j != i))
                                       for i in n), ind + "  },", ind + " \"wait\":",
                ind + "  {" + ",".join("\n%s   \"%s\": %s" % (ind, alphabet[i], _g(c[i][i])) for i in n), ind + "  }", ind + "}"]
        return "\n".join(body)
    lines = ["{", " \"alphabet\": \"%s\"," % alphabet, " \"indel\":", "  {"]
    for k in ("ins", "del", "insExt", "delExt", "insTime", "delTime"):
        lines.append("   \"%s\": %s%s" % (k, _g(indel[k]), "" if k.endswith("Time") else ","))
    lines += ["  },", " \"sub\":"]
    if len(root) > 1:
        lines += ["  {", "   \"mixture\": [", ",\n".join(component(root[k], sub[k], 4) for k in range(len(root))), "   ]", "  },"]
    else:
        lines.append(component(root[0], sub[0], 2) + ",")
    lines += [" \"logLikelihood\": %s" % _g(indel["lp"]), "}", ""]
    return "\n".join(lines)


# ---- `historian count` without -recon: the substitution half of BackwardMatrix::getCounts (src/forward.cpp:897-973, 1183-1214) ----

def dp_posterior_substitution_counts(batch, job, x_cols, y_cols, x_null, y_null, parent_row, counter):
    """Expected substitution events of a pair DP whose Forward and Backward matrices are on the device (capi.Batch with
    HX_KEEP_BACKWARD, both fills done): every in-envelope cell's alignment column - ForwardMatrix::getAlignmentColumn, the
    columns of the two profile states plus, for an absorbing cell, the parent as a wildcard - through the tree's sum-product
    (ColumnCounter: the device kernel of hx_sumprod.hip), weighted with the cell's posterior probability
    exp(F + B - lpEnd).  The reference evaluates one column per cell (and caches those of cells that change one side only,
    cachedCellEigenCounts); here the cells are grouped by column first - an IMD cell's column depends on its row only, an IDM
    cell's on its column, insertions likewise - so the device sees (Nx-2)(Ny-2) + O(Nx + Ny) weighted columns in one launch.

    x_cols [Nx][N], y_cols [Ny][N]: int8 tokens of every profile state's alignment column over the N tree nodes
    (Profile::alignColumn: residue token, -1 wildcard for an ancestor's row, -2 where the state has no residue of that
    row; null states: what their alignPath holds); x_null / y_null [Nx], [Ny]: ProfileState::isNull(); parent_row: the tree
    node the pair's parent profile belongs to.  -> dict(root_counts, eigen_counts, counts) as ColumnCounter.run."""
    fm, bm = batch.read_matrix(job, 0), batch.read_matrix(job, 1)
    lp_end = batch.lp_end()[job]
    with np.errstate(invalid="ignore"):
        w = np.exp(fm + bm - lp_end)                       # [Nx-1][Ny-1][5]; -inf cells (outside the envelope) -> 0
    w[~np.isfinite(w)] = 0.
    w[0, :, :] = 0.                                        # getAlignmentColumn: no column for xpos == 0 or ypos == 0
    w[:, 0, :] = 0.
    x_cols, y_cols = np.asarray(x_cols, dtype=np.int8), np.asarray(y_cols, dtype=np.int8)
    nx1, ny1 = w.shape[0], w.shape[1]
    xc, yc = x_cols[:nx1], y_cols[:ny1]
    xn, yn = np.asarray(x_null, dtype=bool)[:nx1], np.asarray(y_null, dtype=bool)[:ny1]
    IMM, IMD, IDM, IMI, IIW = 0, 1, 2, 3, 4
    tokens, weight = [], []

    def add(cols, ws):
        keep = (ws > 0) & (cols != GAP).any(axis=1)
        if keep.any():
            tokens.append(cols[keep])
            weight.append(ws[keep])

    def with_parent(cols):
        out = cols.copy()
        out[:, parent_row] = WILD
        return out

    # IMM, both states absorbing: the two columns merged (a row present in both keeps x's), the parent a wildcard
    ii, jj = np.nonzero((w[:, :, IMM] > 0) & ~xn[:, None] & ~yn[None, :])
    if len(ii):
        merged = np.where(xc[ii] != GAP, xc[ii], yc[jj])
        merged[:, parent_row] = WILD
        add(merged, w[ii, jj, IMM])
    # per-row columns: IMD (with the parent when x absorbs), IIW, and the IMM / IMD cells of null x states
    wx_imd, wx_iiw = w[:, :, IMD].sum(axis=1), w[:, :, IIW].sum(axis=1)
    wx_imm_null = (w[:, :, IMM] * xn[:, None]).sum(axis=1)
    add(with_parent(xc), np.where(xn, 0., wx_imd))
    add(xc, wx_iiw + np.where(xn, wx_imd + wx_imm_null, 0.))
    # per-column columns: IDM (with the parent when y absorbs), IMI, and the IMM cells of a null y state under an emitting x
    wy_idm, wy_imi = w[:, :, IDM].sum(axis=0), w[:, :, IMI].sum(axis=0)
    wy_imm_null = (w[:, :, IMM] * (~xn)[:, None] * yn[None, :]).sum(axis=0)
    add(with_parent(yc), np.where(yn, 0., wy_idm))
    add(yc, wy_imi + np.where(yn, wy_idm, 0.) + wy_imm_null)
    if not tokens:
        raise ValueError("no cell of the pair has a posterior probability")
    return counter.run(np.concatenate(tokens), np.concatenate(weight))
