"""CPU restatement (test infrastructure only) of the indel-event part of the reference's Forward-Backward counts:
ForwardMatrix::transitionEigenCounts (src/forward.cpp:579-652, the indelCounts members) and BackwardMatrix::getCounts
(src/forward.cpp:1183-1214) - the expected numbers of insertions, deletions, extensions and the waiting times of a pair DP,
every transition between two cells weighted with its posterior probability.

PARITY UNPINNED by reference fixtures: the reference's only users of getCounts (`historian count` without -recon, `fit`)
need GSL's eigen solver, and no test of the reference prints these numbers.  What pins this restatement instead is
brute_force_indel_counts below: on pairs small enough to enumerate, the posterior-weighted sum over transitions must equal
the expectation over ALL paths of the counts along each path (tests/test_oracle_counts_dp.py).  Covers profiles whose
transitions carry no event counts of their own (leaf profiles; profiles built without CountIndelEvents): the
`x.getTrans(...)->counts` terms of the reference are zero there.

The substitution part is restated below (get_subst_counts: getAlignmentColumn, accumulateEigenCounts, the weighting of
getCounts; src/forward.cpp:897-973, 1183-1214) on top of oracle/sumprod_oracle.SumProduct, which the reference's
testsumprod / testaligncount / testcount fixtures pin byte for byte; what this file adds - which column a cell stands for and
with which weight it counts - is pinned by the same enumeration (brute_force_subst_counts)."""
import math

from . import historian_oracle as ho

IMM, IMD, IDM, IMI, IIW, EEE = ho.IMM, ho.IMD, ho.IDM, ho.IMI, ho.IIW, ho.EEE
KEYS = ("ins", "del", "insExt", "delExt", "insTime", "delTime")


def branch_times(model, t_l, t_r):
    """hmm.l.t, hmm.r.t, and the ProbModel wait times insWait / delWait of the two branches (src/model.cpp:374-391, 1106-1108)"""
    def wait(rate, t):
        return 1 / rate - t / (math.exp(rate * t) - 1)
    return dict(l_t=t_l, r_t=t_r, l_ins_wait=wait(model.ins_rate, t_l), l_del_wait=wait(model.del_rate, t_l),
                r_ins_wait=wait(model.ins_rate, t_r), r_del_wait=wait(model.del_rate, t_r))


def transition_indel_counts(tm, src_state, dest_state, x_null, y_null):
    """src/forward.cpp:585-649 (dest.state switch); x_null / y_null of the DESTINATION's profile states"""
    c = dict.fromkeys(KEYS, 0.)
    s, d = src_state, dest_state
    if d == IMM:
        if not x_null and not y_null:
            if s in (IMM, IMD):
                c["insTime"] += tm["l_t"]; c["delTime"] += tm["l_t"]
            if s in (IMM, IDM):
                c["insTime"] += tm["r_t"]; c["delTime"] += tm["r_t"]
    elif d == IMD:
        if not x_null:
            if s in (IMM, IMD):
                c["insTime"] += tm["l_t"]; c["delTime"] += tm["l_t"]
            if s == d:
                c["delExt"] += 1
            else:
                c["del"] += 1; c["delTime"] += tm["r_del_wait"]
    elif d == IIW:
        if not x_null:
            if s == d:
                c["insExt"] += 1
            else:
                c["ins"] += 1; c["insTime"] += tm["l_ins_wait"]
    elif d == IDM:
        if not y_null:
            if s in (IMM, IDM):
                c["insTime"] += tm["r_t"]; c["delTime"] += tm["r_t"]
            if s == d:
                c["delExt"] += 1
            else:
                c["del"] += 1; c["delTime"] += tm["l_del_wait"]
    elif d == IMI:
        if not y_null:
            if s == d:
                c["insExt"] += 1
            else:
                c["ins"] += 1; c["insTime"] += tm["r_ins_wait"]
    return c


def get_indel_counts(bwd, tm):
    """BackwardMatrix::getCounts, indel part (src/forward.cpp:1183-1214): bwd a filled ho.BackwardMatrix"""
    fwd = bwd.fwd
    out = dict.fromkeys(KEYS, 0.)
    out["lp"] = fwd.lp_end
    for i in range(fwd.x_size - 1):
        x_null = fwd.x.state[i].is_null()
        for j in range(fwd.y_size - 1):
            if not fwd.in_envelope(i, j):
                continue
            y_null = fwd.y.state[j].is_null()
            for s in (IMM, IMD, IDM, IMI, IIW):
                lp_dest = bwd.cell(i, j, s)
                for src, lp in fwd.source_transitions((i, j, s)).items():
                    w = math.exp(fwd.cellc(src) + lp + lp_dest - fwd.lp_end) if min(fwd.cellc(src), lp, lp_dest) > -math.inf else 0.
                    if w == 0.:
                        continue
                    c = transition_indel_counts(tm, src[2], s, x_null, y_null)
                    for k in KEYS:
                        out[k] += c[k] * w
    return out


def brute_force_indel_counts(fwd, tm):
    """The same expectation by enumeration: every path from the start cell to the end cell (depth-first over
    source_transitions, from the end), its probability exp(sum of transition log-weights), its event counts; the
    expectation is sum(P(path) * counts(path)) / sum(P(path)).  Exponential: pairs of a few residues only.  Transitions into
    the end cell carry no events (getCounts stops at xSize-2, ySize-2)."""
    end = (fwd.x_size - 1, fwd.y_size - 1, EEE)
    tot = [0.]
    acc = dict.fromkeys(KEYS, 0.)

    def walk(cell, lp, counts):
        if cell[0] == 0 and cell[1] == 0:
            if cell[2] != ho.SSS:
                return                         # only the start cell (0, 0, SSS) begins a path (src/forward.cpp:73)
            p = math.exp(lp)
            tot[0] += p
            for k in KEYS:
                acc[k] += p * counts[k]
            return
        for src, tlp in fwd.source_transitions(cell).items():
            if tlp == -math.inf:
                continue
            nc = counts
            if cell[2] != EEE:
                c = transition_indel_counts(tm, src[2], cell[2], fwd.x.state[cell[0]].is_null(), fwd.y.state[cell[1]].is_null())
                nc = {k: counts[k] + c[k] for k in KEYS}
            walk(src, lp + tlp, nc)

    walk(end, 0., dict.fromkeys(KEYS, 0.))
    out = {k: acc[k] / tot[0] for k in KEYS}
    out["lp"] = math.log(tot[0])
    return out


# ---- the substitution part: BackwardMatrix::getCounts with a SumProduct (src/forward.cpp:897-973, 1183-1214) -------------
WILDCARD = "*"


def profile_align_column(prof, s):
    """Profile::alignColumn (src/profile.cpp:100-110): {row: residue} of the rows state s has a residue in; rows without
    sequence coordinates (ancestors) show the wildcard"""
    st = prof.state[s]
    col = {}
    for row, path in st.align_path.items():
        if path and path[0]:
            col[row] = prof.seq[row][st.seq_coords[row] - 1] if row in st.seq_coords else WILDCARD
    return col


def get_alignment_column(fwd, cell):
    """ForwardMatrix::getAlignmentColumn (src/forward.cpp:937-973)"""
    i, j, s = cell
    col = {}
    if not (0 < i < fwd.x_size - 1 and 0 < j < fwd.y_size - 1):
        return col
    xs, ys = fwd.x.state[i], fwd.y.state[j]
    if s == IMM:
        if not xs.is_null() and not ys.is_null():
            col = profile_align_column(fwd.x, i)
            for r, ch in profile_align_column(fwd.y, j).items():
                col.setdefault(r, ch)                 # (map::insert keeps what is there)
            col[fwd.parent_row_index] = WILDCARD
        elif xs.is_emit_or_start() and ys.is_null():
            col = profile_align_column(fwd.y, j)
        elif xs.is_null():
            col = profile_align_column(fwd.x, i)
    elif s == IMD:
        col = profile_align_column(fwd.x, i)
        if not xs.is_null():
            col[fwd.parent_row_index] = WILDCARD
    elif s == IDM:
        col = profile_align_column(fwd.y, j)
        if not ys.is_null():
            col[fwd.parent_row_index] = WILDCARD
    elif s == IIW:
        col = profile_align_column(fwd.x, i)
    elif s == IMI:
        col = profile_align_column(fwd.y, j)
    return col


def _column_counts(sp, col, root, eig, weight):
    """ForwardMatrix::accumulateEigenCounts (src/forward.cpp:917-926)"""
    if col:
        sp.init_column(col)
        sp.fill_up()
        sp.fill_down()
        sp.accumulate_eigen_counts(root, eig, weight)


def get_subst_counts(bwd, sp):
    """The substitution half of BackwardMatrix::getCounts: every in-envelope cell's alignment column through the tree's
    sum-product, weighted with the cell's posterior probability.  (The reference caches the per-column result of cells that
    change one side only - cachedCellEigenCounts - which changes nothing but the cost.)  -> root counts [C][A], eigen-basis
    counts [C][A][A] (complex)"""
    import numpy as np
    fwd = bwd.fwd
    a = len(sp.model.alphabet)
    root = [np.zeros(a) for _ in range(sp.C)]
    eig = [np.zeros((a, a), dtype=complex) for _ in range(sp.C)]
    for i in range(fwd.x_size - 1):
        for j in range(fwd.y_size - 1):
            if not fwd.in_envelope(i, j):
                continue
            for s in (IMM, IMD, IDM, IMI, IIW):
                lp = fwd.cell(i, j, s) + bwd.cell(i, j, s) - fwd.lp_end
                if lp > -math.inf:
                    _column_counts(sp, get_alignment_column(fwd, (i, j, s)), root, eig, math.exp(lp))
    return root, eig


def brute_force_subst_counts(fwd, sp):
    """The same expectation by enumeration of every path (as brute_force_indel_counts): a path counts the columns of the
    cells it visits."""
    import numpy as np
    end = (fwd.x_size - 1, fwd.y_size - 1, EEE)
    a = len(sp.model.alphabet)
    root = [np.zeros(a) for _ in range(sp.C)]
    eig = [np.zeros((a, a), dtype=complex) for _ in range(sp.C)]
    tot = [0.]

    def walk(cell, lp, visited):
        if cell[0] == 0 and cell[1] == 0:
            if cell[2] != ho.SSS:
                return
            p = math.exp(lp)
            tot[0] += p
            for c in visited + [cell]:
                _column_counts(sp, get_alignment_column(fwd, c), root, eig, p)
            return
        for src, tlp in fwd.source_transitions(cell).items():
            if tlp > -math.inf:
                walk(src, lp + tlp, visited + ([cell] if cell[2] != EEE else []))

    walk(end, 0., [])
    return [r / tot[0] for r in root], [e / tot[0] for e in eig]
