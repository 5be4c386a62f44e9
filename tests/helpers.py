"""Test-side glue: oracle objects -> C-ABI images, seeded workload generators."""
import random

import numpy as np

from historian_amd import capi
from oracle import historian_oracle as ho

NEG_INF = float("-inf")


def profile_image(prof, env=None, env_row=None):
    """oracle Profile -> capi.ProfileImage (POD image of the reference Profile)."""
    n = prof.size()
    C, A = prof.components, prof.alph_size
    is_null = np.array([1 if s.is_null() else 0 for s in prof.state], dtype=np.uint8)
    lpa = np.full((n, C, A), NEG_INF)
    for i, s in enumerate(prof.state):
        if not s.is_null():
            lpa[i] = np.array(s.lp_absorb)
    env_pos = None
    if env is not None and env.initialized():
        pos2col = env.row1_pos_to_col if env_row == env.row1 else env.row2_pos_to_col
        env_pos = np.zeros(n, dtype=np.int32)
        for i in range(1, n):
            env_pos[i] = env.cumulative_matches[pos2col[prof.state[i].seq_coords[env_row]]]
        env_pos[0] = env.cumulative_matches[pos2col[0]]
    return capi.ProfileImage([t.src for t in prof.trans], [t.dest for t in prof.trans],
                             [t.lp_trans for t in prof.trans],
                             [s.in_ for s in prof.state], [s.absorb_out for s in prof.state],
                             [s.null_out for s in prof.state], is_null, lpa, env_pos)


def hmm_image(hmm):
    """oracle PairHMM -> capi.HmmImage"""
    log = np.vectorize(ho.safe_log)
    return capi.HmmImage(np.array(hmm.trans_matrix()), np.array(hmm.log_root),
                         log(np.array(hmm.l.sub_mat)), log(np.array(hmm.r.sub_mat)),
                         np.array(hmm.logl.log_ins_prob), np.array(hmm.logr.log_ins_prob),
                         np.array(hmm.logl.log_cpt_weight), np.array(hmm.logr.log_cpt_weight))


def job_images(fwd):
    """(ProfileImage x, ProfileImage y, HmmImage, max_distance) for an oracle DPMatrix."""
    env = fwd.envelope
    x = profile_image(fwd.x, env, env.row1)
    y = profile_image(fwd.y, env, env.row2)
    return x, y, hmm_image(fwd.hmm), env.max_distance


def oracle_dense(dp):
    """oracle DPMatrix sparse cells -> dense [R][Cc][5] (-inf where absent)."""
    out = np.full((dp.x_size - 1, dp.y_size - 1, 5), NEG_INF)
    for (i, j), c in dp.cells.items():
        if i < dp.x_size - 1 and j < dp.y_size - 1:
            out[i, j] = c
    return out


def envelope_mask(dp):
    """[R][Cc] bool: the cells of an oracle DPMatrix that lie inside its envelope (reference src/forward.h:92-98)."""
    return np.array([[dp.in_envelope(i, j) for j in range(dp.y_size - 1)] for i in range(dp.x_size - 1)], dtype=bool)


def same_bits(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def assert_same_bits(a, b, what=""):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    bad = np.argwhere(a.view(np.uint64) != b.view(np.uint64))
    # -0.0 never arises; NaN never arises
    assert len(bad) == 0, "%s: %d of %d values differ, first at %s: %r vs %r" % (
        what, len(bad), a.size, tuple(bad[0]), a[tuple(bad[0])], b[tuple(bad[0])])


# ---------------------------------------------------------------------------
# seeded synthetic workloads
# ---------------------------------------------------------------------------
def random_seq(rng, alphabet, n, weights=None):
    return "".join(rng.choices(alphabet, weights=weights, k=n))


def mutate(rng, seq, alphabet, sub=.1, indel=.03):
    out = []
    for ch in seq:
        r = rng.random()
        if r < indel:
            continue
        if r < 2 * indel:
            out.append(rng.choice(alphabet))
        out.append(rng.choice(alphabet) if rng.random() < sub else ch)
    return "".join(out)


def jc_model(alphabet="ACGT", ins=.01, dele=.01, ext=.66):
    a = alphabet
    return ho.RateModel({"alphabet": a, "insrate": ins, "delrate": dele, "insextprob": ext, "delextprob": ext,
                         "subrate": {c: {d: 1 for d in a if d != c} for c in a}})


def random_reversible_model(rng, alphabet, components=1, ins=.02, dele=.02, ext=.6):
    """A random GTR-like mixture: symmetric exchangeabilities x stationary frequencies."""
    a = len(alphabet)
    cpts = []
    for _ in range(components):
        pi = np.array([rng.random() + .2 for _ in range(a)])
        pi /= pi.sum()
        ex = np.zeros((a, a))
        for i in range(a):
            for j in range(i + 1, a):
                ex[i, j] = ex[j, i] = rng.random() + .05
        scale = rng.random() + .5
        sub = {alphabet[i]: {alphabet[j]: float(scale * ex[i, j] * pi[j]) for j in range(a) if j != i} for i in range(a)}
        cpts.append({"subrate": sub, "rootprob": {alphabet[i]: float(pi[i]) for i in range(a)},
                     "weight": rng.random() + .5})
    js = {"alphabet": alphabet, "insrate": ins, "delrate": dele, "insextprob": ext, "delextprob": ext}
    if components == 1:
        js.update({k: v for k, v in cpts[0].items() if k != "weight"})
    else:
        js["mixture"] = cpts
    return ho.RateModel(js)


def make_hmm(model, tl, tr):
    return ho.PairHMM(ho.ProbModel(model, tl), ho.ProbModel(model, tr), model.ins_prob)


def leaf(model, seq, row, name=None):
    return ho.Profile.from_seq(model.components(), model.alphabet, seq, row, name or ("s%d" % row))


def internal_profile(model, sx, sy, rows, parent_row, seed, samples=6, tl=.1, tr=.15, keep_all=False):
    """A real internal-node profile (DAG with null/wait/ready states) made the way
    Reconstructor::reconstruct makes them (reference src/recon.cpp:1010)."""
    hmm = make_hmm(model, tl, tr)
    x, y = leaf(model, sx, rows[0]), leaf(model, sy, rows[1])
    fwd = ho.ForwardMatrix(x, y, hmm, parent_row, ho.GuideAlignmentEnvelope())
    gen = ho.MT19937(seed)
    strat = (0 if keep_all else ho.DPMatrix.CollapseChains) | ho.DPMatrix.IncludeBestTrace
    return fwd.sample_profile(gen, samples, 0, strat)


def left_justified_guide(seqs):
    """A (bad but valid) guide alignment: all sequences left-justified."""
    cols = max(len(s) for s in seqs.values())
    return {row: [True] * len(s) + [False] * (cols - len(s)) for row, s in seqs.items()}


def dag_case(seed, n=14, alphabet="ACGT", components=1, band=None, samples=6, keep_all=False):
    """Forward DP of two internal (DAG) profiles; returns the oracle ForwardMatrix (unfilled)."""
    rng = random.Random(seed)
    model = jc_model(alphabet) if components == 1 and seed % 2 == 0 else random_reversible_model(rng, alphabet, components)
    anc = random_seq(rng, alphabet, n)
    s = [mutate(rng, anc, alphabet, .15, .06) for _ in range(4)]
    s = [t if t else alphabet[0] for t in s]
    p1 = internal_profile(model, s[0], s[1], (0, 1), 4, seed * 7 + 1, samples, keep_all=keep_all)
    p2 = internal_profile(model, s[2], s[3], (2, 3), 5, seed * 7 + 2, samples, keep_all=keep_all)
    hmm = make_hmm(model, .2, .05)
    if band is None:
        env = ho.GuideAlignmentEnvelope()
    else:
        guide = left_justified_guide({0: s[0], 1: s[1], 2: s[2], 3: s[3]})
        env = ho.GuideAlignmentEnvelope(guide, 0, 2, band)
    return ho.ForwardMatrix(p1, p2, hmm, 6, env, fill=False)


def leaf_case(seed, lx, ly, alphabet="ACGT", components=1, jc=True, tl=.1, tr=.1, band=None):
    rng = random.Random(seed)
    model = jc_model(alphabet) if jc else random_reversible_model(rng, alphabet, components)
    sx = random_seq(rng, alphabet, lx)
    sy = mutate(rng, sx, alphabet)[:ly] if lx else ""
    while len(sy) < ly:
        sy += rng.choice(alphabet)
    hmm = make_hmm(model, tl, tr)
    env = ho.GuideAlignmentEnvelope()
    if band is not None:
        env = ho.GuideAlignmentEnvelope(left_justified_guide({1: sx, 2: sy}), 1, 2, band)
    return ho.ForwardMatrix(leaf(model, sx, 1, "x"), leaf(model, sy, 2, "y"), hmm, 0, env, fill=False)
