"""Test infrastructure only (see oracle/README or DESIGN.md section 3): the reference's argmax traceback
(ForwardMatrix::bestTrace, src/forward.cpp:278-302, with sourceCells :309-314, sourceTransitionsWithoutEmitOrAbsorb
:326-398, lpCellEmitOrAbsorb :404-440 and bestCell :245-255) restated over the POD job image and the dense matrix
that oracle_fill.c produces - for pairs too large for the object-based restatement in historian_oracle.py.
Checked against historian_oracle.ForwardMatrix.best_trace in tests/test_oracle_c.py."""
import math

from oracle import c_oracle

NEG_INF = -math.inf
IMM, IMD, IDM, IMI, IIW, EEE = range(6)
_SOURCES = {IMM: (0, 1, 2, 3, 4), EEE: (0, 1, 2, 3, 4), IMD: (0, 1, 2, 3), IDM: (0, 1, 2, 4), IMI: (0, 3), IIW: (0, 4, 3)}


def _in_transitions(p, i):
    return [(int(p.trans_src[t]), float(p.trans_lp[t])) for t in p.in_idx[p.in_off[i]:p.in_off[i + 1]]]


def best_trace(x, y, hmm, max_distance, fwd):
    """fwd: the dict c_oracle.forward returns for this job.  Returns [(xpos, ypos, state), ...] start cell first."""
    cells, lp_end = fwd["cells"], fwd["lp_end"]
    assert lp_end > NEG_INF, "Forward likelihood is zero; traceback fail"
    nx, ny = x.n_states, y.n_states
    T = hmm.lp_trans
    x_ready = [x.nout_off[i + 1] == x.nout_off[i] for i in range(nx)]
    y_ready = [y.nout_off[j + 1] == y.nout_off[j] for j in range(ny)]
    x_empty, y_empty = bool(all(x.is_null)), bool(all(y.is_null))       # Profile::isEmpty(): every state is null
    lse = c_oracle.load().orc_log_sum_exp
    ca = hmm.alph_size * hmm.components

    def emission(i, j):         # computeLogProbAbsorb, src/forward.h:112-124
        sx, sy = fwd["subx"][i], fwd["suby"][j]
        lip = NEG_INF
        for cpt in range(hmm.components):
            inner = NEG_INF
            for a in range(hmm.alph_size):
                k = cpt * hmm.alph_size + a
                inner = lse(inner, float(hmm.log_root.reshape(-1)[k]) + (float(sx[k]) + float(sy[k])))
            lip = lse(lip, inner)
        return lip

    def cell(i, j, s):          # (oracle_fill.c leaves -inf in every cell outside the envelope)
        if i >= nx - 1 or j >= ny - 1:
            return NEG_INF
        return float(cells[i, j, s])

    path = [(nx - 1, ny - 1, EEE)]
    dx, dy, ds = path[0]
    while dx > 0 or dy > 0:
        xnull, ynull = bool(x.is_null[dx]), bool(y.is_null[dy])
        xin, yin = _in_transitions(x, dx), _in_transitions(y, dy)
        x_eos = (not xnull) or len(xin) == 0
        clp = {}
        lp_abs = 0.
        if ds in (IMD, IIW):
            if xnull:
                if (y_ready[dy] or y_empty) and dx < nx - 1:
                    for sx, lp in xin:
                        clp[(sx, dy, ds)] = lp
            elif y_ready[dy] or y_empty:
                for sx, lp in xin:
                    for s in _SOURCES[ds]:
                        clp[(sx, dy, s)] = float(T[s][ds]) + lp
            if not xnull:
                lp_abs = float(fwd["rootsubx"][dx] if ds == IMD else fwd["insx"][dx])
        elif ds in (IDM, IMI):
            if ynull:
                if dy < ny - 1:
                    for sy, lp in yin:
                        clp[(dx, sy, ds)] = lp
            elif x_ready[dx] or x_empty:
                for sy, lp in yin:
                    for s in _SOURCES[ds]:
                        clp[(dx, sy, s)] = float(T[s][ds]) + lp
            if not ynull:
                lp_abs = float(fwd["rootsuby"][dy] if ds == IDM else fwd["insy"][dy])
        elif ds == IMM:
            if ynull and x_eos:
                if dy < ny - 1:
                    for sy, lp in yin:
                        clp[(dx, sy, ds)] = lp
            elif xnull:
                if (y_ready[dy] or y_empty) and dx < nx - 1:
                    for sx, lp in xin:
                        clp[(sx, dy, ds)] = lp
            elif not ynull:
                for sx, lpx in xin:
                    for sy, lpy in yin:
                        for s in _SOURCES[ds]:
                            clp[(sx, sy, s)] = float(T[s][ds]) + lpx + lpy
                lp_abs = emission(dx, dy)
        else:
            for sx, lpx in xin:
                for sy, lpy in yin:
                    for s in _SOURCES[ds]:
                        clp[(sx, sy, s)] = float(T[s][ds]) + lpx + lpy
        assert clp, "traceback failure"
        best, pbest = None, NEG_INF
        for c in sorted(clp):
            v = (clp[c] + lp_abs) + cell(*c)
            if v > pbest:
                pbest, best = v, c
        if best is None:
            best = (0, 0, EEE)
        path.insert(0, best)
        dx, dy, ds = best
    return path
