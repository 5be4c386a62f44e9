"""Seeded synthetic workloads for bench.py, the tools/ sweeps and the tests: pairs of leaf sequences related
through substitutions and indels, with their true pairwise alignment as the guide alignment of a
GuideAlignmentEnvelope band (reference src/alignpath.cpp:282-310).  Plumbing, not product."""
import numpy as np

from . import hostmodel


def synth_pair(rng, pi, length, want_guide=False, ly=None, sub=.2, indel=.02):
    """x ~ pi (`length` residues); y = x with ~sub substitutions (~pi) and ~indel deletions / insertions, cut or
    padded to `ly` residues (default: `length`).  With want_guide, also the true pairwise alignment as two boolean
    rows (the guide alignment)."""
    a = len(pi)
    ly = length if ly is None else ly
    x = rng.choice(a, size=length, p=pi)
    keep = rng.random(length) >= indel
    y = x[keep]
    subs = rng.random(len(y)) < sub
    y = np.where(subs, rng.choice(a, size=len(y), p=pi), y)
    ins_at = np.flatnonzero(rng.random(len(y)) < indel)
    y = np.insert(y, ins_at, rng.choice(a, size=len(ins_at), p=pi))
    pad = max(0, ly - len(y))
    if pad:
        y = np.concatenate([y, rng.choice(a, size=pad, p=pi)])
    if not want_guide:
        return x, y[:ly]
    xrow, yrow = [], []
    ins = set(int(k) for k in ins_at)
    k = 0                                   # index into the kept (pre-insertion) y residues
    for p in range(length):
        if keep[p]:
            if k in ins:
                xrow.append(False); yrow.append(True)
            xrow.append(True); yrow.append(True)
            k += 1
        else:
            xrow.append(True); yrow.append(False)
    xrow += [False] * pad
    yrow += [True] * pad
    # y is cut to `ly` residues: later y residues leave the alignment
    seen = 0
    for c in range(len(yrow)):
        if yrow[c]:
            seen += 1
            if seen > ly:
                yrow[c] = False
    cols = [c for c in range(len(xrow)) if xrow[c] or yrow[c]]
    return x, y[:ly], (np.array([xrow[c] for c in cols]), np.array([yrow[c] for c in cols]))


def envelope_coordinates(xrow, yrow):
    """Per-state envelope coordinate of the two leaf profiles under a pairwise guide alignment
    (reference src/alignpath.cpp:282-310 + src/forward.cpp:26-35: cumulativeMatches[rowPosToCol[pos]];
    START has position 0, END the position of the last residue)."""
    cm = np.concatenate([[0], np.cumsum(xrow & yrow)])

    def coords(row):
        pos2col = np.concatenate([[0], np.flatnonzero(row) + 1])
        e = cm[pos2col]
        return np.concatenate([e, e[-1:]]).astype(np.int32)
    return coords(xrow), coords(yrow)


def in_envelope_cells(ex, ey, band):
    """Number of in-envelope cells of a leaf pair (reference src/forward.h:92-98): within the band, or at an edge
    (x START row; y column of the last residue, the source of the transition into END).  The envelope coordinates
    are non-decreasing, so a row's band is a range of columns."""
    exr, eyc = ex[:-1].astype(np.int64), ey[:-1].astype(np.int64)
    in_band = np.searchsorted(eyc, exr + band, "right") - np.searchsorted(eyc, exr - band, "left")
    return int(in_band.sum()) + (len(eyc) - int(in_band[0])) + int((np.abs(exr[1:] - eyc[-1]) > band).sum())


def leaf_pair(rng, model, hmm, lx, ly=None, band=-1, sub=.2, indel=.02):
    """One pair-DP job image (x profile, y profile, hmm, max_distance) of two related leaf sequences drawn from the
    model's first root distribution; with band >= 0 the envelope follows the pair's true alignment."""
    a, c = len(model.alphabet), model.components()
    pi = np.asarray(model.root[0], dtype=float)
    pi = pi / pi.sum()
    if band < 0:
        xs, ys = synth_pair(rng, pi, lx, ly=ly, sub=sub, indel=indel)
        return hostmodel.leaf_profile(xs, a, c), hostmodel.leaf_profile(ys, a, c), hmm, -1
    xs, ys, (xrow, yrow) = synth_pair(rng, pi, lx, want_guide=True, ly=ly, sub=sub, indel=indel)
    ex, ey = envelope_coordinates(xrow, yrow)
    return hostmodel.leaf_profile(xs, a, c, ex), hostmodel.leaf_profile(ys, a, c, ey), hmm, band
