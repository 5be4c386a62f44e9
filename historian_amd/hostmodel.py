"""Host-side builders for the inputs of the C ABI (Python plumbing used by bench.py,
smoke() and the tests; the C++ mirror of the reference classes lives in csrc/host/).

What is restated here is what the reference computes *before* the DP, once per branch
(reference src/recon.cpp:946-948): ProbModel fields (src/model.cpp:374-391), the 24
PairHMM log weights (src/pairhmm.cpp:17-43) and leaf profiles (src/profile.cpp:23-76).
exp(Rt) is scipy's expm: the reference takes it from un-vendored GSL
(src/model.cpp:329), so the substitution matrix is an *input* of the parity definition.
"""
import json
import math

import numpy as np

from . import capi

NEG_INF = float("-inf")


def lse_table():
    """log(1+exp(-n*1e-4)), n = 0..100001, by HOST libm (reference src/logsumexp.cpp:8-16)."""
    return np.array([math.log(1. + math.exp(-(n * .0001))) for n in range(capi.HX_LSE_TABLE_ENTRIES)])


def _log(x):
    x = np.asarray(x, dtype=np.float64)
    out = np.full(x.shape, NEG_INF)
    nz = x > 0
    # math.log (libm) element-wise: numpy's SIMD log may differ from libm in the last ulp
    out[nz] = [math.log(v) for v in x[nz]]
    return out


class RateModel:
    """Field meanings of the reference's rate-model JSON (src/model.cpp:172-232)."""

    def __init__(self, js):
        self.alphabet = js["alphabet"]
        self.ins_rate, self.del_rate = js["insrate"], js["delrate"]
        self.ins_ext, self.del_ext = js["insextprob"], js["delextprob"]
        cpts = js["mixture"] if "mixture" in js else [js]
        a = len(self.alphabet)
        self.sub_rate, self.root, w = [], [], []
        for c in cpts:
            r = np.zeros((a, a))
            for i, si in enumerate(self.alphabet):
                for j, sj in enumerate(self.alphabet):
                    if i != j and si in c["subrate"] and sj in c["subrate"][si]:
                        r[i, j] += c["subrate"][si][sj]
                        r[i, i] -= c["subrate"][si][sj]
            if "rootprob" in c:
                pi = np.array([c["rootprob"].get(s, 0.) for s in self.alphabet])
            else:
                m = np.vstack([r.T, np.ones((1, a))])
                rhs = np.zeros(a + 1)
                rhs[a] = 1
                pi = np.maximum(np.linalg.lstsq(m, rhs, rcond=None)[0], 0.)
                pi = pi / pi.sum()
            self.sub_rate.append(r)
            self.root.append(pi)
            w.append(c.get("weight", 1))
        self.cpt_weight = np.array(w, dtype=float) / sum(w)

    @staticmethod
    def load(path):
        with open(path) as f:
            return RateModel(json.load(f))

    def components(self):
        return len(self.sub_rate)

    def sub_prob(self, t):
        from scipy.linalg import expm
        return [expm(r * t) for r in self.sub_rate]


def branch_params(model, t):
    """ProbModel(model, t) scalars (reference src/model.cpp:374-391)."""
    return dict(ins=1 - math.exp(-model.ins_rate * t), dele=1 - math.exp(-model.del_rate * t),
                ins_ext=model.ins_ext, del_ext=model.del_ext)


def pair_hmm_trans(l, r):
    """The 24 PairHMM log transition weights as a [5][6] table, -inf where absent
    (reference src/pairhmm.cpp:17-43, 46-110)."""
    li, ld, lie, lde = l["ins"], l["dele"], l["ins_ext"], l["del_ext"]
    ri, rd, rie, rde = r["ins"], r["dele"], r["ins_ext"], r["del_ext"]
    lni, lnd, lnie, lnde = 1 - li, 1 - ld, 1 - lie, 1 - lde
    rni, rnd, rnie, rnde = 1 - ri, 1 - rd, 1 - rie, 1 - rde
    lg = lambda v: math.log(v) if v > 0 else NEG_INF
    I, D, M, S, W, E = capi.IMM, capi.IMD, capi.IDM, capi.IMI, capi.IIW, capi.EEE
    t = np.full((5, 6), NEG_INF)
    t[I, S] = lg(ri)
    t[I, W] = lg(li * rni)
    t[I, I] = lg(lni * rni * lnd * rnd)
    t[I, D] = lg(lni * rni * lnd * rd)
    t[I, M] = lg(lni * rni * ld * rnd)
    t[I, E] = lg(lni * rni)
    t[D, I] = lg(lni * lnd * rnde)
    t[D, D] = lg(lni * lnd * rde)
    t[D, M] = lg(lni * ld * rnde)
    t[D, E] = lg(lni * rnde)
    t[M, I] = lg(rni * lnde * rnd)
    t[M, D] = lg(rni * lnde * rd)
    t[M, M] = lg(rni * lde * rnd)
    t[M, E] = lg(rni * lnde)
    t[S, S] = lg(rie)
    t[S, W] = lg(li * rnie)
    t[S, I] = lg(lni * rnie * lnd * rnd)
    t[S, D] = lg(lni * rnie * lnd * rd)
    t[S, E] = lg(lni * rnie)
    t[W, W] = lg(lie)
    t[W, I] = lg(lnie * lnd * rnd)
    t[W, M] = lg(lnie * ld * rnd)
    t[W, E] = lg(lnie)
    return t


def make_hmm(model, t_l, t_r, sub_l=None, sub_r=None):
    """PairHMM image for one internal node (reference src/recon.cpp:946-948)."""
    sub_l = model.sub_prob(t_l) if sub_l is None else sub_l
    sub_r = model.sub_prob(t_r) if sub_r is None else sub_r
    log_cptw = _log(model.cpt_weight)
    log_ins = np.stack([_log(p) for p in model.root])
    log_root = log_ins + log_cptw[:, None]           # src/pairhmm.cpp:11-15
    return capi.HmmImage(pair_hmm_trans(branch_params(model, t_l), branch_params(model, t_r)), log_root,
                         np.stack([_log(m) for m in sub_l]), np.stack([_log(m) for m in sub_r]),
                         log_ins, log_ins, log_cptw, log_cptw)


def leaf_profile(tokens, alph_size, components=1, env_pos=None):
    """Leaf profile image (reference src/profile.cpp:23-76): states START, one emit state
    per residue (lpAbsorb 0 at the residue's token, -inf elsewhere; a negative token is a
    wildcard: all 0), END."""
    tokens = np.asarray(tokens, dtype=np.int64)
    n = len(tokens) + 2
    is_null = np.zeros(n, dtype=np.uint8)
    is_null[0] = is_null[-1] = 1
    lpa = np.full((n, components, alph_size), NEG_INF)
    for pos, tok in enumerate(tokens):
        if tok < 0:
            lpa[pos + 1] = 0.
        else:
            lpa[pos + 1, :, tok] = 0.
    return capi.ProfileImage.chain(is_null, lpa, env_pos)
