"""CPU restatement of Historian's pair-HMM Forward/Backward DP hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under historian_amd/ (the product) may import,
call, link or execute anything in this directory; only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg do, and there only as the checker.

Parity status: PINNED.  This restatement reproduces byte-for-byte every golden
file the reference's own tests hold for this path (tests/test_oracle_golden.py):
  testforward x7  (Makefile:243-250), testnullforward (Makefile:252-253),
  testbackward x2 (Makefile:255-257), testseqprofile (Makefile:239-240),
  testlogsumexp   (Makefile:206-208).
The reference itself is not buildable here (needs GSL, absent from the image), so
there is no oracle/_ref; see DESIGN.md.

Every function cites the reference file:line it follows (paths relative to the
reference checkout).  Pure-Python loops: small cases only.  Larger parity cases and
the CPU baseline use the plain-C fill in oracle/oracle_fill.c, which is validated
against this file.
"""
import math
import json

NEG_INF = float("-inf")

# ----------------------------------------------------------------------------
# src/logsumexp.h:22-28, src/logsumexp.cpp:8-16 -- lookup table
# ----------------------------------------------------------------------------
LSE_MAX = 10
LSE_PRECISION = .0001
LSE_ENTRIES = int(LSE_MAX / LSE_PRECISION) + 1          # 100001


def log_sum_exp_unary_slow(x):
    """src/logsumexp.cpp:47-49"""
    return math.log(1. + math.exp(-x))


def build_lse_table():
    """src/logsumexp.cpp:8-16.  One guard entry is appended: (int)(x/1e-4) can
    reach 100000 for x just below 10 and the reference then reads lookup[n+1]
    one past its allocation (src/logsumexp.h:53-57); we define that entry as the
    natural continuation of the table."""
    return [log_sum_exp_unary_slow(n * LSE_PRECISION) for n in range(LSE_ENTRIES + 1)]


LSE_TABLE = build_lse_table()


def log_sum_exp_unary(x):
    """src/logsumexp.h:42-64 (table + linear interpolation, 0 for x>=10/nan/inf)"""
    if x >= LSE_MAX or math.isnan(x) or math.isinf(x):
        return 0.
    if x < 0:
        return -x
    n = int(x / LSE_PRECISION)
    f0 = LSE_TABLE[n]
    dx = x - (n * LSE_PRECISION)
    f1 = LSE_TABLE[n + 1]
    df = f1 - f0
    return f0 + df * (dx / LSE_PRECISION)


def log_sum_exp(a, b, *rest):
    """src/logsumexp.h:66-100 (binary op; n-ary forms are left-nested)"""
    if a == b:
        mx, diff = a, 0.
    elif a < b:
        mx, diff = b, b - a
    else:
        mx, diff = a, a - b
    ret = mx + log_sum_exp_unary(diff)
    for c in rest:
        ret = log_sum_exp(ret, c)
    return ret


def log_sum_exp_slow(a, b):
    """src/logsumexp.cpp:22-37"""
    mn, mx = (a, b) if a < b else (b, a)
    if mn == NEG_INF:
        return mx
    return mx + log_sum_exp_unary_slow(mx - mn)


def log_inner_product(v1, v2):
    """src/logsumexp.h:132-137"""
    lip = NEG_INF
    for a, b in zip(v1, v2):
        lip = log_sum_exp(lip, a + b)
    return lip


def log_inner_product_nested(vv1, vv2):
    """src/logsumexp.h:146-151"""
    lip = NEG_INF
    for v1, v2 in zip(vv1, vv2):
        lip = log_sum_exp(lip, log_inner_product(v1, v2))
    return lip


def safe_log(x):
    """C log(): log(0) = -inf (Python raises instead)."""
    return math.log(x) if x > 0 else NEG_INF


# ----------------------------------------------------------------------------
# std::to_string(double), JsonUtil::toString  (src/jsonutil.cpp:151-176)
# ----------------------------------------------------------------------------
def to_string(d):
    if d == NEG_INF:
        return "-inf"
    if d == float("inf"):
        return "inf"
    return "%f" % d


def json_double(d):
    if d == NEG_INF:
        return '"-inf"'
    if d == float("inf"):
        return '"inf"'
    return "%f" % d


def json_tags(tags, indent):
    """JsonUtil::toString(map<string,string>, indent)  src/jsonutil.cpp:158-176"""
    if not tags:
        return "{ }"
    s = ""
    first = True
    n = len(tags)
    for k in sorted(tags):
        if first:
            s += "{ " if n == 1 else ("\n" + " " * indent + "{")
        else:
            s += ","
        first = False
        if n > 1:
            s += "\n" + " " * (indent + 1)
        s += '"' + k + '": "' + tags[k] + '"'
    s += " " if n == 1 else ("\n" + " " * indent)
    s += "}"
    return s


# ----------------------------------------------------------------------------
# AlignPath algebra (src/alignpath.cpp:33-81); AlignPath = dict row -> list[bool]
# ----------------------------------------------------------------------------
def align_path_columns(a):
    cols = None
    for row in sorted(a):
        if cols is None:
            cols = len(a[row])
        else:
            assert cols == len(a[row]), "Alignment path is not flush"
    return cols or 0


def align_path_residues_in_row(r):
    return sum(1 for b in r if b)


def align_path_union(a1, a2):
    a = {k: list(v) for k, v in a1.items()}
    for k, v in a2.items():
        if k not in a:              # std::map::insert does not overwrite
            a[k] = list(v)
    return a


def align_path_concat(a1, a2, a3=None):
    if a3 is not None:
        return align_path_concat(align_path_concat(a1, a2), a3)
    a = {k: list(v) for k, v in a1.items()}
    c1, c2 = align_path_columns(a1), align_path_columns(a2)
    for k in a:
        if k not in a2:
            a[k].extend([False] * c2)
    for row, rpath in a2.items():
        lpath = a.setdefault(row, [])
        if not lpath:
            lpath.extend([False] * c1)
        lpath.extend(rpath)
    return a


def ensure_align_path_has_row(a, r):
    cols = align_path_columns(a)
    if r not in a:
        a[r] = [False] * cols


class GuideAlignmentEnvelope:
    """src/alignpath.h:43-62, src/alignpath.cpp:282-310"""

    def __init__(self, guide=None, row1=0, row2=0, max_distance=-1):
        self.max_distance = max_distance
        self.row1, self.row2 = row1, row2
        self.cumulative_matches = []
        self.row1_pos_to_col = []
        self.row2_pos_to_col = []
        if guide is None:
            self.max_distance = -1
            return
        cols = align_path_columns(guide)
        matches = 0
        self.row1_pos_to_col.append(0)
        self.row2_pos_to_col.append(0)
        self.cumulative_matches.append(0)
        for col in range(cols):
            if guide[row1][col]:
                self.row1_pos_to_col.append(col + 1)
            if guide[row2][col]:
                self.row2_pos_to_col.append(col + 1)
            if guide[row1][col] and guide[row2][col]:
                matches += 1
            self.cumulative_matches.append(matches)

    def initialized(self):
        return self.max_distance >= 0

    def in_range(self, pos1, pos2):
        if not self.initialized():
            return True
        d = (self.cumulative_matches[self.row1_pos_to_col[pos1]]
             - self.cumulative_matches[self.row2_pos_to_col[pos2]])
        return abs(d) <= self.max_distance


# ----------------------------------------------------------------------------
# Rate / probability models: only the *field meanings* of src/model.cpp:374-391,
# 492-504 are restated.  exp(Rt) lives in un-vendored GSL (gsl_linalg_exponential_ss,
# src/model.cpp:329); here it is scipy.linalg.expm (or the JC closed form), and the
# resulting subMat is an explicit *input* of every DP parity definition.
# ----------------------------------------------------------------------------
class RateModel:
    """src/model.cpp:172-232 (JSON field meanings)"""

    def __init__(self, js):
        self.alphabet = js["alphabet"]
        self.ins_rate = js["insrate"]
        self.ins_ext_prob = js["insextprob"]
        self.del_rate = js["delrate"]
        self.del_ext_prob = js["delextprob"]
        self.sub_rate, self.ins_prob, self.cpt_weight = [], [], []
        cpts = js["mixture"] if "mixture" in js else [js]
        for c in cpts:
            self._read_component(c)
        norm = sum(self.cpt_weight)
        self.cpt_weight = [w / norm for w in self.cpt_weight]

    @staticmethod
    def from_file(path):
        with open(path) as f:
            return RateModel(json.load(f))

    def components(self):
        return len(self.sub_rate)

    def _read_component(self, jm):
        import numpy as np
        A = len(self.alphabet)
        sr = np.zeros((A, A))
        rm = jm["subrate"]
        for i, si in enumerate(self.alphabet):
            if si in rm:
                for j, sj in enumerate(self.alphabet):
                    if j != i and sj in rm[si]:
                        sr[i, j] += rm[si][sj]
                        sr[i, i] -= rm[si][sj]
        if "rootprob" in jm:
            ip = np.array([jm["rootprob"].get(s, 0.) for s in self.alphabet], dtype=float)
        else:
            ip = eqm_prob_vector(sr)
        self.cpt_weight.append(jm.get("weight", 1))
        self.ins_prob.append(ip)
        self.sub_rate.append(sr)


def eqm_prob_vector(sr):
    """src/model.cpp:282-320: least-squares solve of [R^T; 1] pi = [0; 1]
    (reference: GSL QR; here numpy lstsq), clamp at 0, renormalise."""
    import numpy as np
    A = sr.shape[0]
    M = np.vstack([sr.T, np.ones((1, A))])
    b = np.zeros(A + 1)
    b[A] = 1
    eqm = np.linalg.lstsq(M, b, rcond=None)[0]
    eqm = np.maximum(eqm, 0.)
    return eqm / eqm.sum()


def sub_prob_matrix(sr, t):
    """exp(R t).  Uniform-off-diagonal matrices (Jukes-Cantor like) use the closed
    form (accurate for the 1e-9 rates of testforward.nosub.json); otherwise expm."""
    import numpy as np
    A = sr.shape[0]
    off = sr[~np.eye(A, dtype=bool)]
    if np.all(off == off[0]):
        r = float(off[0])
        e1 = math.expm1(-A * r * t)                 # e^{-A r t} - 1
        pij = -e1 / A
        m = np.full((A, A), pij)
        np.fill_diagonal(m, 1. + e1 * (A - 1) / A)
        return m
    from scipy.linalg import expm
    return expm(sr * t)


class ProbModel:
    """src/model.cpp:374-391"""

    def __init__(self, model, t, sub_mat=None):
        self.alphabet = model.alphabet
        self.t = t
        self.ins = 1 - math.exp(-model.ins_rate * t)
        self.dele = 1 - math.exp(-model.del_rate * t)
        self.ins_ext = model.ins_ext_prob
        self.del_ext = model.del_ext_prob
        self.cpt_weight = list(model.cpt_weight)
        self.ins_vec = [[float(v) for v in ip] for ip in model.ins_prob]
        if sub_mat is None:
            sub_mat = [sub_prob_matrix(sr, t) for sr in model.sub_rate]
        self.sub_mat = [[[float(v) for v in row] for row in m] for m in sub_mat]

    def components(self):
        return len(self.sub_mat)

    def alphabet_size(self):
        return len(self.alphabet)


class LogProbModel:
    """src/model.cpp:492-504"""

    def __init__(self, pm):
        self.log_cpt_weight = [safe_log(w) for w in pm.cpt_weight]
        self.log_ins_prob = [[safe_log(v) for v in iv] for iv in pm.ins_vec]


# ----------------------------------------------------------------------------
# PairHMM  (src/pairhmm.h:14-18,45-54; src/pairhmm.cpp:5-140)
# ----------------------------------------------------------------------------
IMM, IMD, IDM, IMI, IIW, EEE = 0, 1, 2, 3, 4, 5
SSS = 0
TOTAL_STATES = 5
STATES = (IMM, IMD, IDM, IMI, IIW)

_SOURCES = {
    IMM: (IMM, IMD, IDM, IMI, IIW),
    EEE: (IMM, IMD, IDM, IMI, IIW),
    IMD: (IMM, IMD, IDM, IMI),
    IDM: (IMM, IMD, IDM, IIW),
    IMI: (IMM, IMI),
    IIW: (IMM, IIW, IMI),
}


def state_name(s, x_at_start, y_at_start):
    """src/pairhmm.cpp:142-153"""
    if s == IMM:
        return "SSS" if (x_at_start and y_at_start) else "IMM"
    if s == IMD:
        return "IMD"
    if s == IDM:
        return "IDM"
    if s == IMI:
        return "SSI" if x_at_start else "IMI"
    if s == IIW:
        return "SIW" if y_at_start else "IIW"
    if s == EEE:
        return "EEE"
    raise ValueError(s)


class PairHMM:
    def __init__(self, l, r, root):
        """src/pairhmm.cpp:5-44.  root = list (per component) of prob vectors."""
        self.l, self.r = l, r
        self.logl, self.logr = LogProbModel(l), LogProbModel(r)
        self.log_root = [[safe_log(v) for v in rv] for rv in root]
        for cpt in range(l.components()):
            self.log_root[cpt] = [lr + self.logl.log_cpt_weight[cpt] for lr in self.log_root[cpt]]
        lIns, lDel, lInsExt, lDelExt = l.ins, l.dele, l.ins_ext, l.del_ext
        rIns, rDel, rInsExt, rDelExt = r.ins, r.dele, r.ins_ext, r.del_ext
        lNoIns, lNoDel, lNoInsExt, lNoDelExt = 1 - lIns, 1 - lDel, 1 - lInsExt, 1 - lDelExt
        rNoIns, rNoDel, rNoInsExt, rNoDelExt = 1 - rIns, 1 - rDel, 1 - rInsExt, 1 - rDelExt
        log = safe_log
        T = {}
        T[IMM, IMI] = log(rIns)
        T[IMM, IIW] = log(lIns * rNoIns)
        T[IMM, IMM] = log(lNoIns * rNoIns * lNoDel * rNoDel)
        T[IMM, IMD] = log(lNoIns * rNoIns * lNoDel * rDel)
        T[IMM, IDM] = log(lNoIns * rNoIns * lDel * rNoDel)
        T[IMM, EEE] = log(lNoIns * rNoIns)

        T[IMD, IMM] = log(lNoIns * lNoDel * rNoDelExt)
        T[IMD, IMD] = log(lNoIns * lNoDel * rDelExt)
        T[IMD, IDM] = log(lNoIns * lDel * rNoDelExt)
        T[IMD, EEE] = log(lNoIns * rNoDelExt)

        T[IDM, IMM] = log(rNoIns * lNoDelExt * rNoDel)
        T[IDM, IMD] = log(rNoIns * lNoDelExt * rDel)
        T[IDM, IDM] = log(rNoIns * lDelExt * rNoDel)
        T[IDM, EEE] = log(rNoIns * lNoDelExt)

        T[IMI, IMI] = log(rInsExt)
        T[IMI, IIW] = log(lIns * rNoInsExt)
        T[IMI, IMM] = log(lNoIns * rNoInsExt * lNoDel * rNoDel)
        T[IMI, IMD] = log(lNoIns * rNoInsExt * lNoDel * rDel)
        T[IMI, EEE] = log(lNoIns * rNoInsExt)

        T[IIW, IIW] = log(lInsExt)
        T[IIW, IMM] = log(lNoInsExt * lNoDel * rNoDel)
        T[IIW, IDM] = log(lNoInsExt * lDel * rNoDel)
        T[IIW, EEE] = log(lNoInsExt)
        self.T = T

    def components(self):
        return len(self.log_root)

    def alphabet_size(self):
        return self.l.alphabet_size()

    def lp_trans(self, src, dest):
        """src/pairhmm.cpp:46-110 (-inf for absent transitions)"""
        return self.T.get((src, dest), NEG_INF)

    @staticmethod
    def sources(dest):
        """src/pairhmm.cpp:117-140"""
        return _SOURCES[dest]

    def trans_matrix(self):
        """6x5... flattened [src][dest] 5x6 table, -inf where absent."""
        return [[self.lp_trans(s, d) for d in range(6)] for s in range(5)]


# ----------------------------------------------------------------------------
# Profile  (src/profile.h:13-76, src/profile.cpp)
# ----------------------------------------------------------------------------
class ProfileTransition:
    __slots__ = ("src", "dest", "lp_trans", "align_path")

    def __init__(self, src=0, dest=0, lp_trans=NEG_INF, align_path=None):
        self.src, self.dest, self.lp_trans = src, dest, lp_trans
        self.align_path = align_path if align_path is not None else {}

    def copy(self):
        return ProfileTransition(self.src, self.dest, self.lp_trans,
                                 {k: list(v) for k, v in self.align_path.items()})


class ProfileState:
    def __init__(self, components=0, alph_size=0):
        self.name = ""
        self.meta = {}
        self.in_ = []
        self.null_out = []
        self.absorb_out = []
        # lpAbsorb[cpt][a]; empty list <=> null state (src/profile.h:32)
        self.lp_absorb = [[NEG_INF] * alph_size for _ in range(components)]
        self.align_path = {}
        self.seq_coords = {}

    def is_null(self):
        return len(self.lp_absorb) == 0

    def is_emit(self):
        return len(self.lp_absorb) != 0

    def is_start(self):
        return len(self.in_) == 0

    def is_emit_or_start(self):
        return self.is_emit() or self.is_start()

    def is_ready(self):
        return len(self.null_out) == 0

    def is_wait(self):
        return len(self.absorb_out) == 0

    def copy(self):
        s = ProfileState()
        s.name = self.name
        s.meta = dict(self.meta)
        s.in_ = list(self.in_)
        s.null_out = list(self.null_out)
        s.absorb_out = list(self.absorb_out)
        s.lp_absorb = [list(v) for v in self.lp_absorb]
        s.align_path = {k: list(v) for k, v in self.align_path.items()}
        s.seq_coords = dict(self.seq_coords)
        return s


WILDCARD_CHAR = "*"


def tokenize(c, alphabet):
    """src/fastseq.cpp:10-16"""
    p = alphabet.find(c)
    if p < 0:
        p = alphabet.find(c.lower() if c.isupper() else c.upper())
    return p


class Profile:
    def __init__(self, components=0, alph_size=0, row_index=0):
        self.components = components
        self.alph_size = alph_size
        self.name = ""
        self.meta = {}
        self.state = []
        self.trans = []
        self.seq = {}
        self.equiv_absorb_state = {}
        self.root_row_index = row_index

    @staticmethod
    def from_seq(components, alphabet, seq, row_index, name=""):
        """Leaf profile, src/profile.cpp:23-76"""
        p = Profile(components, len(alphabet), row_index)
        L = len(seq)
        p.state = [ProfileState(components, len(alphabet)) for _ in range(L + 2)]
        p.trans = [ProfileTransition() for _ in range(L + 1)]
        p.name = name
        p.state[0] = ProfileState()
        p.state[-1] = ProfileState()
        p.state[0].name = "START"
        p.state[0].seq_coords[row_index] = 0
        p.state[-1].name = "END"
        p.state[-1].seq_coords[row_index] = L
        for pos in range(L + 1):
            t = p.trans[pos]
            t.src, t.dest, t.lp_trans = pos, pos + 1, 0.
            if pos == L:
                p.state[pos].null_out.append(pos)
            else:
                p.state[pos].absorb_out.append(pos)
            p.state[pos + 1].in_.append(pos)
            if pos < L:
                st = p.state[pos + 1]
                st.name = seq[pos] + str(pos + 1)
                st.align_path.setdefault(row_index, []).append(True)
                st.seq_coords[row_index] = pos + 1
                for lpa in st.lp_absorb:
                    if seq[pos] == WILDCARD_CHAR:
                        for k in range(len(lpa)):
                            lpa[k] = 0.
                    else:
                        tok = tokenize(seq[pos], alphabet)
                        if tok < 0:
                            for k in range(len(lpa)):
                                lpa[k] = 0.
                        else:
                            lpa[tok] = 0.
        p.seq[row_index] = seq
        p.assert_transitions_consistent()
        p.assert_all_states_wait_or_ready()
        p.example_path_to_end()
        return p

    def size(self):
        return len(self.state)

    def end(self):
        return self.state[-1]

    def is_empty(self):
        return all(s.is_null() for s in self.state)

    def left_multiply(self, sub):
        """src/profile.cpp:78-91.  Returns only the multiplied lpAbsorb table
        (list indexed by state; [] for null states)."""
        out = []
        for st in self.state:
            if st.is_null():
                out.append([])
                continue
            rows = []
            for cpt in range(self.components):
                row = []
                for c in range(self.alph_size):
                    lp = NEG_INF
                    for d in range(self.alph_size):
                        lp = log_sum_exp(lp, safe_log(sub[cpt][c][d]) + st.lp_absorb[cpt][d])
                    row.append(lp)
                rows.append(row)
            out.append(rows)
        return out

    def get_trans(self, src, dest):
        """src/profile.cpp:93-98"""
        for t in self.state[dest].in_:
            if self.trans[t].src == src:
                return self.trans[t]
        return None

    def calc_sum_path_absorb_probs(self, log_cpt_weight, log_ins_prob, tag="cumLogProb"):
        """src/profile.cpp:112-131"""
        n = len(self.state)
        cum = [NEG_INF] * n
        cum[0] = 0.
        for pos in range(1, n):
            lp_abs = 0.
            st = self.state[pos]
            if not st.is_null():
                lp_abs = NEG_INF
                for cpt in range(self.components):
                    lp_abs = log_sum_exp(lp_abs, log_cpt_weight[cpt]
                                         + log_inner_product(log_ins_prob[cpt], st.lp_absorb[cpt]))
            for ti in st.in_:
                t = self.trans[ti]
                assert t.src < pos, "not toposorted"
                cum[pos] = log_sum_exp(cum[pos], cum[t.src] + t.lp_trans + lp_abs)
            if tag is not None:
                st.meta[tag] = to_string(cum[pos])
        return cum[-1]

    # -- integrity checks (src/profile.cpp:321-361) --
    def assert_transitions_consistent(self):
        for i, s in enumerate(self.state):
            for t in s.in_:
                assert self.trans[t].dest == i
            for t in s.null_out:
                assert self.trans[t].src == i
            for t in s.absorb_out:
                assert self.trans[t].src == i

    def assert_all_states_wait_or_ready(self):
        for s in self.state:
            assert s.is_ready() or s.is_wait(), "state %s neither wait nor ready" % s.name

    def example_path_to_end(self):
        n = len(self.state)
        from_start = [False] * n
        from_start[0] = True
        for i in range(n):
            if from_start[i]:
                s = self.state[i]
                for t in s.null_out + s.absorb_out:
                    assert self.trans[t].dest > i, "transition violates toposort"
                    from_start[self.trans[t].dest] = True
        assert from_start[-1], "No path from start to end"

    def add_ready_states(self):
        """src/profile.cpp:268-319"""
        n0 = self.size()
        old2new = [0] * n0
        prof = Profile(self.components, self.alph_size, self.root_row_index)
        prof.name = self.name
        prof.meta = dict(self.meta)
        prof.seq = dict(self.seq)
        prof.trans = [t.copy() for t in self.trans]
        prof_state = [s.copy() for s in self.state]
        n = 0
        for s in range(n0):
            old2new[s] = n
            n += 1
            if not self.state[s].is_ready() and not self.state[s].is_wait():
                ready = ProfileState()
                old_ready_idx = len(prof_state)
                new_ready_idx = n
                n += 1
                ready_trans_idx = len(prof.trans)
                prof_state[s].name += ";"
                ready.name = self.state[s].name + "."
                ready.meta = dict(self.state[s].meta)
                ready.seq_coords = dict(self.state[s].seq_coords)
                prof_state[s].absorb_out, ready.absorb_out = ready.absorb_out, prof_state[s].absorb_out
                for t in ready.absorb_out:
                    prof.trans[t].src = old_ready_idx
                rt = ProfileTransition(s, old_ready_idx, 0.)
                prof_state[s].null_out.append(ready_trans_idx)
                ready.in_.append(ready_trans_idx)
                prof_state.append(ready)
                prof.trans.append(rt)
                old2new.append(new_ready_idx)
        prof.state = [None] * len(prof_state)
        for s in range(len(prof_state)):
            prof.state[old2new[s]] = prof_state[s]
        for t in prof.trans:
            t.src = old2new[t.src]
            t.dest = old2new[t.dest]
        for a, b in self.equiv_absorb_state.items():
            prof.equiv_absorb_state[old2new[a]] = old2new[b]
        prof.assert_transitions_consistent()
        prof.assert_all_states_wait_or_ready()
        prof.example_path_to_end()
        return prof

    # -- JSON writer (src/profile.cpp:133-211), byte-exact --
    def to_json(self):
        out = []
        w = out.append
        w("{\n")
        if self.name:
            w(' "name": "' + self.name + '",\n')
        if self.meta:
            w(' "meta": ' + json_tags(self.meta, 2) + ",\n")
        w(' "alphSize": %d,\n' % self.alph_size)
        w(' "state": [\n')
        for s, st in enumerate(self.state):
            w("  {\n")
            w('   "n": %d,\n' % s)
            if st.name:
                w('   "name": "' + st.name + '",\n')
            if st.meta:
                w('   "meta": ' + json_tags(st.meta, 4) + ",\n")
            if st.align_path:
                w('   "path": ' + align_path_json(st.align_path) + ",\n")
            if st.seq_coords:
                w('   "seqPos": [')
                for k, row in enumerate(sorted(st.seq_coords)):
                    w((", " if k else " ") + "[ %d, %d ]" % (row, st.seq_coords[row]))
                w(" ],\n")
            if not st.is_null():
                w('   "lpAbsorb": [')
                for cpt in range(self.components):
                    w((", " if cpt > 0 else "") + "[")
                    for a in range(self.alph_size):
                        w((", " if a > 0 else " ") + json_double(st.lp_absorb[cpt][a]))
                    w(" ]")
                w("],\n")
            w('   "trans": [')
            s_out = sorted(set(st.null_out) | set(st.absorb_out))
            first = True
            for ti in s_out:
                tr = self.trans[ti]
                if not first:
                    w(",\n             ")
                first = False
                w(' { "to": %d,' % tr.dest)
                w(' "lpTrans": ' + json_double(tr.lp_trans))
                if tr.align_path:
                    w(', "path": ' + align_path_json(tr.align_path))
                w(" }")
            w(" ]\n")
            w("  }")
            if s < len(self.state) - 1:
                w(",")
            w("\n")
        w(" ]\n")
        w("}\n")
        return "".join(out)


def align_path_json(a):
    """src/profile.cpp:133-145"""
    s = "["
    for row in sorted(a):
        if len(s) > 1:
            s += ","
        s += " [ " + str(row) + ', "'
        for col in a[row]:
            s += "*" if col else "-"
        s += '" ]'
    s += " ]"
    return s


def pair_parent_name(lname, ltime, rname, rtime):
    """src/tree.cpp:479-484 (ostream default float format == %g)"""
    return "(%s:%g,%s:%g)" % (lname, ltime, rname, rtime)


# ----------------------------------------------------------------------------
# mt19937 + libstdc++ uniform_real_distribution<double> (generate_canonical, 2 draws)
# ----------------------------------------------------------------------------
class MT19937:
    def __init__(self, seed=5489):
        self.mt = [0] * 624
        self.mt[0] = seed & 0xffffffff
        for i in range(1, 624):
            self.mt[i] = (1812433253 * (self.mt[i - 1] ^ (self.mt[i - 1] >> 30)) + i) & 0xffffffff
        self.idx = 624

    def _twist(self):
        mt = self.mt
        for i in range(624):
            y = (mt[i] & 0x80000000) | (mt[(i + 1) % 624] & 0x7fffffff)
            mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ (0x9908b0df if (y & 1) else 0)
        self.idx = 0

    def next_u32(self):
        if self.idx >= 624:
            self._twist()
        y = self.mt[self.idx]
        self.idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9d2c5680
        y ^= (y << 15) & 0xefc60000
        y ^= y >> 18
        return y & 0xffffffff

    def canonical(self):
        s = float(self.next_u32())
        s = s + float(self.next_u32()) * 4294967296.0
        r = s / 18446744073709551616.0
        if r >= 1.0:
            r = math.nextafter(1.0, 0.0)
        return r

    def uniform_real(self, a, b):
        return self.canonical() * (b - a) + a


# ----------------------------------------------------------------------------
# DPMatrix / ForwardMatrix / BackwardMatrix  (src/forward.h, src/forward.cpp)
# ----------------------------------------------------------------------------
class DPMatrix:
    KeepAll, CollapseChains = 0, 1
    CountSubstEvents, CountIndelEvents = 2, 4
    IncludeBestTrace, KeepGapsOpen = 8, 16

    def __init__(self, x, y, hmm, env):
        """src/forward.cpp:11-66"""
        self.x, self.y, self.hmm = x, y, hmm
        self.alph_size = hmm.alphabet_size()
        self.x_empty, self.y_empty = x.is_empty(), y.is_empty()
        self.x_size, self.y_size = x.size(), y.size()
        self.subx = x.left_multiply(hmm.l.sub_mat)
        self.suby = y.left_multiply(hmm.r.sub_mat)
        self.cells = {}                      # (i,j) -> [5 doubles]; absent == -inf
        C = hmm.components()
        self.insx = [NEG_INF] * self.x_size
        self.insy = [NEG_INF] * self.y_size
        self.rootsubx = [NEG_INF] * self.x_size
        self.rootsuby = [NEG_INF] * self.y_size
        self.start_cell = (0, 0, SSS)
        self.end_cell = (self.x_size - 1, self.y_size - 1, EEE)
        self.envelope = env
        self.lp_end = NEG_INF
        self.x_closest_leaf_pos = [0] * self.x_size
        self.y_closest_leaf_pos = [0] * self.y_size
        self.x_near_start = [False] * self.x_size
        self.y_near_end = [False] * self.y_size
        if env.initialized():
            for i in range(1, self.x_size):
                self.x_closest_leaf_pos[i] = x.state[i].seq_coords[env.row1]
            for j in range(1, self.y_size):
                self.y_closest_leaf_pos[j] = y.state[j].seq_coords[env.row2]
        for i in range(1, self.x_size - 1):
            if not x.state[i].is_null():
                for cpt in range(C):
                    self.insx[i] = log_sum_exp(self.insx[i], hmm.logl.log_cpt_weight[cpt]
                                               + log_inner_product(hmm.logl.log_ins_prob[cpt], x.state[i].lp_absorb[cpt]))
                    self.rootsubx[i] = log_sum_exp(self.rootsubx[i],
                                                   log_inner_product(hmm.log_root[cpt], self.subx[i][cpt]))
        for j in range(1, self.y_size - 1):
            if not y.state[j].is_null():
                for cpt in range(C):
                    self.insy[j] = log_sum_exp(self.insy[j], hmm.logr.log_cpt_weight[cpt]
                                               + log_inner_product(hmm.logr.log_ins_prob[cpt], y.state[j].lp_absorb[cpt]))
                    self.rootsuby[j] = log_sum_exp(self.rootsuby[j],
                                                   log_inner_product(hmm.log_root[cpt], self.suby[j][cpt]))
        self.x_near_start[0] = True
        for i in range(self.x_size):
            if self.x_near_start[i]:
                for t in x.state[i].null_out:
                    self.x_near_start[x.trans[t].dest] = True
        for yt in y.end().in_:
            self.y_near_end[y.trans[yt].src] = True

    # accessors (src/forward.h:68-88)
    def cell(self, i, j, s):
        c = self.cells.get((i, j))
        return NEG_INF if c is None else c[s]

    def xy_cell(self, i, j):
        c = self.cells.get((i, j))
        return _EMPTY_CELL if c is None else c

    def cellc(self, c):
        return self.cell(c[0], c[1], c[2])

    def lp_start(self):
        return self.cell(0, 0, IMM)

    def at_edge(self, i, j):
        return self.x_near_start[i] or self.y_near_end[j]

    def in_envelope(self, i, j):
        """src/forward.h:92-98"""
        return self.at_edge(i, j) or self.envelope.in_range(self.x_closest_leaf_pos[i], self.y_closest_leaf_pos[j])

    def cell_name(self, c):
        """src/forward.cpp:467-469"""
        return "(" + state_name(c[2], c[0] == 0, c[1] == 0) + "," + self.x.state[c[0]].name + "," + self.y.state[c[1]].name + ")"

    def absorb_scratch(self, i, j):
        """src/forward.h:112-119"""
        C, A = self.hmm.components(), self.hmm.alphabet_size()
        return [[self.subx[i][cpt][n] + self.suby[j][cpt][n] for n in range(A)] for cpt in range(C)]

    def compute_log_prob_absorb(self, i, j):
        """src/forward.h:121-124"""
        return log_inner_product_nested(self.hmm.log_root, self.absorb_scratch(i, j))

    def lp_cell_emit_or_absorb(self, c):
        """src/forward.cpp:404-440"""
        xs, ys = self.x.state[c[0]], self.y.state[c[1]]
        s = c[2]
        lp = 0.
        if s == IMD:
            if not xs.is_null():
                lp = self.rootsubx[c[0]]
        elif s == IIW:
            if not xs.is_null():
                lp = self.insx[c[0]]
        elif s == IDM:
            if not ys.is_null():
                lp = self.rootsuby[c[1]]
        elif s == IMI:
            if not ys.is_null():
                lp = self.insy[c[1]]
        elif s == IMM:
            if not xs.is_null() and not ys.is_null():
                lp = self.compute_log_prob_absorb(c[0], c[1])
        return lp

    def is_absorbing(self, c):
        """src/forward.cpp:471-475"""
        xs, ys = self.x.state[c[0]], self.y.state[c[1]]
        return ((c[2] == IMM and not xs.is_null() and not ys.is_null())
                or (c[2] == IMD and not xs.is_null())
                or (c[2] == IDM and not ys.is_null()))

    def changes_x(self, c):
        xs, ys = self.x.state[c[0]], self.y.state[c[1]]
        return ((c[2] == IMM and (xs.is_null() or not ys.is_null()))
                or c[2] == IMD or c[2] == IIW or c[2] == EEE)

    def changes_y(self, c):
        xs = self.x.state[c[0]]
        return ((c[2] == IMM and xs.is_emit_or_start())
                or c[2] == IDM or c[2] == IMI or c[2] == EEE)

    def equiv_absorb_cells(self, c):
        """src/forward.cpp:491-502"""
        xs, ys = self.x.state[c[0]], self.y.state[c[1]]
        eq = []
        if c[2] == IIW and not xs.is_null():
            eq.append((c[0], c[1], IMD))
        elif c[2] == IMI and not ys.is_null():
            eq.append((c[0], c[1], IDM))
        elif self.changes_x(c) and xs.is_null() and c[0] in self.x.equiv_absorb_state:
            eq.append((self.x.equiv_absorb_state[c[0]], c[1], IMD))
        elif self.changes_y(c) and ys.is_null() and c[1] in self.y.equiv_absorb_state:
            eq.append((c[0], self.y.equiv_absorb_state[c[1]], IDM))
        return eq

    @staticmethod
    def best_cell(clp):
        """src/forward.cpp:245-255: strict > over map order"""
        best, pbest = None, NEG_INF
        assert clp, "traceback failure"
        for c in sorted(clp):
            if clp[c] > pbest:
                pbest, best = clp[c], c
        if best is None:
            best = (0, 0, EEE)      # default-constructed CellCoords (forward.h:32)
        return best

    @staticmethod
    def sample_cell(clp, gen):
        """src/forward.cpp:225-243"""
        keys = sorted(clp)
        lpmax = NEG_INF
        for c in keys:
            lpmax = max(lpmax, clp[c])
        ptot = 0.
        for c in keys:
            ptot += math.exp(clp[c] - lpmax)
        p = gen.uniform_real(0., ptot)
        for c in keys:
            p -= math.exp(clp[c] - lpmax)
            if p <= 0:
                return c
        raise RuntimeError("sampleCell fail")


_EMPTY_CELL = [NEG_INF] * 5


class ForwardMatrix(DPMatrix):
    def __init__(self, x, y, hmm, parent_row_index, env, fill=True):
        super().__init__(x, y, hmm, env)
        self.parent_row_index = parent_row_index
        if fill:
            self.fill()

    def fill(self):
        """src/forward.cpp:68-223"""
        x, y, T = self.x, self.y, self.hmm.T
        lse = log_sum_exp
        self.cells[(0, 0)] = [0., NEG_INF, NEG_INF, NEG_INF, NEG_INF]
        for i in range(self.x_size - 1):
            xs = x.state[i]
            for j in range(self.y_size - 1):
                ys = y.state[j]
                if not self.in_envelope(i, j):
                    continue
                dest = self.cells.setdefault((i, j), [NEG_INF] * 5)
                imm, imd, idm, imi, iiw = dest
                if not xs.is_null():
                    if ys.is_ready() or self.y_empty:
                        for xt in xs.in_:
                            tr = x.trans[xt]
                            src = self.xy_cell(tr.src, j)
                            imd = lse(imd, lse(src[IMM] + T[IMM, IMD], src[IMD] + T[IMD, IMD],
                                               src[IDM] + T[IDM, IMD], src[IMI] + T[IMI, IMD]) + tr.lp_trans)
                            iiw = lse(iiw, lse(src[IMM] + T[IMM, IIW], src[IMI] + T[IMI, IIW],
                                               src[IIW] + T[IIW, IIW]) + tr.lp_trans)
                        imd += self.rootsubx[i]
                        iiw += self.insx[i]
                else:
                    if ys.is_ready() or self.y_empty:
                        for xt in xs.in_:
                            tr = x.trans[xt]
                            src = self.xy_cell(tr.src, j)
                            imd = lse(imd, src[IMD] + tr.lp_trans)
                            iiw = lse(iiw, src[IIW] + tr.lp_trans)
                if not ys.is_null():
                    if xs.is_ready() or self.x_empty:
                        for yt in ys.in_:
                            tr = y.trans[yt]
                            src = self.xy_cell(i, tr.src)
                            idm = lse(idm, lse(src[IMM] + T[IMM, IDM], src[IMD] + T[IMD, IDM],
                                               src[IDM] + T[IDM, IDM], src[IIW] + T[IIW, IDM]) + tr.lp_trans)
                            imi = lse(imi, lse(src[IMM] + T[IMM, IMI], src[IMI] + T[IMI, IMI]) + tr.lp_trans)
                        idm += self.rootsuby[j]
                        imi += self.insy[j]
                else:
                    for yt in ys.in_:
                        tr = y.trans[yt]
                        src = self.xy_cell(i, tr.src)
                        idm = lse(idm, src[IDM] + tr.lp_trans)
                        imi = lse(imi, src[IMI] + tr.lp_trans)
                if not xs.is_null() and not ys.is_null():
                    for xt in xs.in_:
                        xtr = x.trans[xt]
                        for yt in ys.in_:
                            ytr = y.trans[yt]
                            src = self.xy_cell(xtr.src, ytr.src)
                            imm = lse(imm, lse(src[IMM] + T[IMM, IMM], src[IMD] + T[IMD, IMM],
                                               src[IDM] + T[IDM, IMM], src[IMI] + T[IMI, IMM],
                                               src[IIW] + T[IIW, IMM]) + xtr.lp_trans + ytr.lp_trans)
                    imm += self.compute_log_prob_absorb(i, j)
                elif ys.is_null() and xs.is_emit_or_start():
                    for yt in ys.in_:
                        tr = y.trans[yt]
                        imm = lse(imm, self.cell(i, tr.src, IMM) + tr.lp_trans)
                else:
                    if ys.is_ready() or self.y_empty:
                        for xt in xs.in_:
                            tr = x.trans[xt]
                            imm = lse(imm, self.cell(tr.src, j, IMM) + tr.lp_trans)
                dest[IMM], dest[IMD], dest[IDM], dest[IMI], dest[IIW] = imm, imd, idm, imi, iiw
        self.lp_end = self.compute_lp_end()

    def compute_lp_end(self):
        """src/forward.cpp:205-220"""
        x, y, T = self.x, self.y, self.hmm.T
        lp_end = NEG_INF
        for xt in x.end().in_:
            xtr = x.trans[xt]
            for yt in y.end().in_:
                ytr = y.trans[yt]
                c = self.xy_cell(xtr.src, ytr.src)
                lp_end = log_sum_exp(lp_end,
                                     log_sum_exp(c[IMM] + T[IMM, EEE], c[IMD] + T[IMD, EEE], c[IDM] + T[IDM, EEE],
                                                 c[IMI] + T[IMI, EEE], c[IIW] + T[IIW, EEE])
                                     + xtr.lp_trans + ytr.lp_trans)
        return lp_end

    # ---- traceback (src/forward.cpp:257-314) ----
    def source_transitions_without_emit_or_absorb(self, dest):
        """src/forward.cpp:326-398"""
        clp = {}
        x, y, hmm = self.x, self.y, self.hmm
        dx, dy, ds = dest
        xs, ys = x.state[dx], y.state[dy]
        if ds in (IMD, IIW):
            if xs.is_null():
                if ys.is_ready() or self.y_empty:
                    if dx < self.x_size - 1:
                        for xt in xs.in_:
                            clp[(x.trans[xt].src, dy, ds)] = x.trans[xt].lp_trans
            else:
                if ys.is_ready() or self.y_empty:
                    for xt in xs.in_:
                        for s in hmm.sources(ds):
                            clp[(x.trans[xt].src, dy, s)] = hmm.lp_trans(s, ds) + x.trans[xt].lp_trans
        elif ds in (IDM, IMI):
            if ys.is_null():
                if dy < self.y_size - 1:
                    for yt in ys.in_:
                        clp[(dx, y.trans[yt].src, ds)] = y.trans[yt].lp_trans
            else:
                if xs.is_ready() or self.x_empty:
                    for yt in ys.in_:
                        for s in hmm.sources(ds):
                            clp[(dx, y.trans[yt].src, s)] = hmm.lp_trans(s, ds) + y.trans[yt].lp_trans
        elif ds == IMM:
            if ys.is_null() and xs.is_emit_or_start():
                if dy < self.y_size - 1:
                    for yt in ys.in_:
                        clp[(dx, y.trans[yt].src, ds)] = y.trans[yt].lp_trans
            elif xs.is_null():
                if ys.is_ready() or self.y_empty:
                    if dx < self.x_size - 1:
                        for xt in xs.in_:
                            clp[(x.trans[xt].src, dy, ds)] = x.trans[xt].lp_trans
            elif not xs.is_null() and not ys.is_null():
                for xt in xs.in_:
                    for yt in ys.in_:
                        for s in hmm.sources(ds):
                            clp[(x.trans[xt].src, y.trans[yt].src, s)] = (hmm.lp_trans(s, ds) + x.trans[xt].lp_trans
                                                                         + y.trans[yt].lp_trans)
        elif ds == EEE:
            if dx == self.x_size - 1 and dy == self.y_size - 1:
                for xt in x.end().in_:
                    for yt in y.end().in_:
                        for s in hmm.sources(ds):
                            clp[(x.trans[xt].src, y.trans[yt].src, s)] = (hmm.lp_trans(s, ds) + x.trans[xt].lp_trans
                                                                         + y.trans[yt].lp_trans)
        else:
            raise ValueError
        return clp

    def source_transitions(self, dest):
        """src/forward.cpp:316-324"""
        clp = self.source_transitions_without_emit_or_absorb(dest)
        lp_abs = self.lp_cell_emit_or_absorb(dest)
        for k in clp:
            clp[k] += lp_abs
        return clp

    def source_cells(self, dest):
        """src/forward.cpp:309-314"""
        sc = self.source_transitions(dest)
        for k in sc:
            sc[k] += self.cellc(k)
        return sc

    def sample_trace(self, gen):
        """src/forward.cpp:257-276"""
        assert self.lp_end > NEG_INF, "Forward likelihood is zero; traceback fail"
        path = [self.end_cell]
        clp = self.source_cells(self.end_cell)
        while True:
            cur = self.sample_cell(clp, gen)
            path.insert(0, cur)
            if cur[0] == 0 and cur[1] == 0:
                break
            clp = self.source_cells(cur)
        return path

    def best_trace(self, end=None):
        """src/forward.cpp:278-302"""
        if end is None:
            assert self.lp_end > NEG_INF, "Forward likelihood is zero; traceback fail"
            end = self.end_cell
        path = [end]
        if end[0] > 0 or end[1] > 0:
            clp = self.source_cells(end)
            while True:
                cur = self.best_cell(clp)
                path.insert(0, cur)
                if cur[0] == 0 and cur[1] == 0:
                    break
                clp = self.source_cells(cur)
        return path

    def best_align_path(self):
        return self.trace_align_path(self.best_trace())

    # ---- profile construction helpers ----
    def eliminated_log_prob_insert(self, c):
        """src/forward.cpp:504-527"""
        if c[2] == IIW:
            return 0. if self.x.state[c[0]].is_null() else self.insx[c[0]]
        if c[2] == IMI:
            return 0. if self.y.state[c[1]].is_null() else self.insy[c[1]]
        return 0.

    def cell_seq_coords(self, c):
        """src/forward.cpp:534-539"""
        coords = dict(self.x.state[c[0]].seq_coords)
        coords.update(self.y.state[c[1]].seq_coords)
        return coords

    def cell_align_path(self, c):
        """src/forward.cpp:541-570"""
        xs, ys = self.x.state[c[0]], self.y.state[c[1]]
        s = c[2]
        if s == IMM:
            if not xs.is_null() and not ys.is_null():
                ap = align_path_union(xs.align_path, ys.align_path)
            elif xs.is_emit_or_start():
                ap = {k: list(v) for k, v in ys.align_path.items()}
            else:
                ap = {k: list(v) for k, v in xs.align_path.items()}
        elif s in (IMD, IIW):
            ap = {k: list(v) for k, v in xs.align_path.items()}
        elif s in (IDM, IMI):
            ap = {k: list(v) for k, v in ys.align_path.items()}
        elif s == EEE:
            ap = {}
        else:
            raise ValueError
        if self.is_absorbing(c):
            ap.setdefault(self.parent_row_index, []).append(True)
        return ap

    def transition_align_path(self, src, dest):
        """src/forward.cpp:572-579"""
        path = {}
        if src[0] != dest[0]:
            path = {k: list(v) for k, v in self.x.get_trans(src[0], dest[0]).align_path.items()}
        if src[1] != dest[1]:
            path = align_path_concat(path, self.y.get_trans(src[1], dest[1]).align_path)
        return path

    def trace_align_path(self, path):
        """src/forward.cpp:654-684 (consistency asserts omitted except flushness)"""
        p = {}
        pv = list(path)
        for n in range(len(pv) - 1):
            cap = self.cell_align_path(pv[n])
            tap = self.transition_align_path(pv[n], pv[n + 1])
            p = align_path_concat(p, cap, tap)
        p = align_path_concat(p, self.cell_align_path(pv[-1]))
        ensure_align_path_has_row(p, self.parent_row_index)
        ensure_align_path_has_row(p, self.x.root_row_index)
        ensure_align_path_has_row(p, self.y.root_row_index)
        align_path_columns(p)
        return p

    def make_profile(self, cells, strategy=DPMatrix.CollapseChains):
        """src/forward.cpp:686-843 (event counts -- strategy bits 2,4 -- not restated)"""
        hmm = self.hmm
        cells = set(cells)
        prof = Profile(hmm.components(), self.alph_size, self.parent_row_index)
        prof.name = pair_parent_name(self.x.name, hmm.l.t, self.y.name, hmm.r.t)
        prof.meta["node"] = str(self.parent_row_index)
        assert self.start_cell in cells, "Missing SSS"
        assert self.end_cell in cells, "Missing EEE"
        ordered = sorted(cells)
        prof_state_index = {}
        out_count = {}
        for dest in ordered:
            for src in self.source_transitions(dest):
                out_count[src] = out_count.get(src, 0) + 1
        for c in ordered:
            if (self.is_absorbing(c) or c == self.start_cell or c == self.end_cell
                    or out_count.get(c, 0) > 1 or (strategy & self.KeepGapsOpen) != 0
                    or (strategy & self.CollapseChains) == 0):
                prof_state_index[c] = len(prof.state)
                st = ProfileState()
                if self.is_absorbing(c):
                    if c[2] == IMM:
                        st.lp_absorb = self.absorb_scratch(c[0], c[1])
                    elif c[2] == IMD:
                        st.lp_absorb = [list(v) for v in self.subx[c[0]]]
                    elif c[2] == IDM:
                        st.lp_absorb = [list(v) for v in self.suby[c[1]]]
                st.align_path = self.cell_align_path(c)
                st.seq_coords = self.cell_seq_coords(c)
                st.name = self.cell_name(c)
                st.meta["fwdLogProb"] = to_string(self.lp_end if c[2] == EEE else self.cell(c[0], c[1], c[2]))
                prof.state.append(st)
        if strategy & self.KeepGapsOpen:
            for c in ordered:
                if not self.is_absorbing(c) and c in prof_state_index:
                    eq = self.equiv_absorb_cells(c)
                    if eq and eq[0] in prof_state_index:
                        prof.equiv_absorb_state[prof_state_index[c]] = prof_state_index[eq[0]]
        # effective transitions: eff[src][destIdx] = [lpPath, lpBestAlignPath, bestAlignPath]
        eff = {}
        for it in reversed(ordered):
            slp = self.source_transitions_without_emit_or_absorb(it)
            lp_ins = self.eliminated_log_prob_insert(it)
            if it in prof_state_index:
                idx = prof_state_index[it]
                for src in sorted(slp):
                    lp = slp[src] + lp_ins
                    eff.setdefault(src, {})[idx] = [lp, lp, self.transition_align_path(src, it)]
            else:
                cell_eff = eff.setdefault(it, {})
                cap = self.cell_align_path(it)
                for src in sorted(slp):
                    src_lp = slp[src]
                    src_eff = eff.setdefault(src, {})
                    for dest_idx in sorted(cell_eff):
                        cde = cell_eff[dest_idx]
                        sde = src_eff.setdefault(dest_idx, [NEG_INF, NEG_INF, {}])
                        lp_path = src_lp + lp_ins + cde[0]
                        sde[0] = log_sum_exp(sde[0], lp_path)
                        lp_best = src_lp + lp_ins + cde[1]
                        tap = self.transition_align_path(src, it)
                        if lp_best > sde[1]:
                            sde[1] = lp_best
                            sde[2] = align_path_concat(tap, cap, cde[2])
        for c in sorted(prof_state_index):
            src_idx = prof_state_index[c]
            for dest_idx in sorted(eff.get(c, {})):
                e = eff[c][dest_idx]
                ti = len(prof.trans)
                prof.trans.append(ProfileTransition(src_idx, dest_idx, e[0], e[2]))
                if prof.state[dest_idx].is_null():
                    prof.state[src_idx].null_out.append(ti)
                else:
                    prof.state[src_idx].absorb_out.append(ti)
                prof.state[dest_idx].in_.append(ti)
        prof.seq = dict(self.x.seq)
        for k, v in self.y.seq.items():
            prof.seq.setdefault(k, v)
        prof.assert_transitions_consistent()
        prof.example_path_to_end()
        prof = prof.add_ready_states()
        return prof

    def sample_profile(self, gen, profile_samples, max_cells=0, strategy=DPMatrix.CollapseChains,
                       min_len=0, max_len=None):
        """src/forward.cpp:845-889"""
        cell_count = {}
        assert (strategy & self.IncludeBestTrace) or profile_samples > 0
        n_traces = 0
        if strategy & self.IncludeBestTrace:
            for c in self.best_trace():
                cell_count[c] = 2
            n_traces += 1
        n_accepted = 0
        while n_accepted < profile_samples and (max_cells == 0 or len(cell_count) < max_cells):
            sampled = self.sample_trace(gen)
            anc_len = sum(1 for c in sampled if c[2] in (IMM, IDM, IMD))
            if anc_len < min_len or (max_len is not None and anc_len > max_len):
                break
            for c in sampled:
                cell_count[c] = cell_count.get(c, 0) + 1
            n_traces += 1
            n_accepted += 1
        threshold = 2 if (n_traces > 1 and max_cells > 0 and len(cell_count) >= max_cells) else 1
        prof_cells = {c for c, n in cell_count.items() if n >= threshold}
        return self.make_profile(prof_cells, strategy)

    def best_profile(self, strategy=DPMatrix.CollapseChains):
        return self.make_profile(set(self.best_trace()), strategy)


class BackwardMatrix(DPMatrix):
    def __init__(self, fwd, fill=True):
        super().__init__(fwd.x, fwd.y, fwd.hmm, fwd.envelope)
        self.fwd = fwd
        if fill:
            self.fill()

    def fill(self):
        """src/forward.cpp:975-1097"""
        x, y, T = self.x, self.y, self.hmm.T
        lse = log_sum_exp
        self.lp_end = 0.
        for xt in x.end().in_:
            xtr = x.trans[xt]
            for yt in y.end().in_:
                ytr = y.trans[yt]
                if self.in_envelope(xtr.src, ytr.src):
                    src = self.cells.setdefault((xtr.src, ytr.src), [NEG_INF] * 5)
                    for s in STATES:
                        src[s] = xtr.lp_trans + ytr.lp_trans + T[s, EEE]
        for i in range(self.x_size - 2, -1, -1):
            xs = x.state[i]
            for j in range(self.y_size - 2, -1, -1):
                ys = y.state[j]
                if not self.in_envelope(i, j):
                    continue
                srcc = self.cells.setdefault((i, j), [NEG_INF] * 5)
                imm, imd, idm, imi, iiw = srcc
                for xt in xs.absorb_out:
                    xtr = x.trans[xt]
                    for yt in ys.absorb_out:
                        ytr = y.trans[yt]
                        d = (xtr.lp_trans + ytr.lp_trans + self.compute_log_prob_absorb(xtr.dest, ytr.dest)
                             + self.cell(xtr.dest, ytr.dest, IMM))
                        imm = lse(imm, T[IMM, IMM] + d)
                        imd = lse(imd, T[IMD, IMM] + d)
                        idm = lse(idm, T[IDM, IMM] + d)
                        imi = lse(imi, T[IMI, IMM] + d)
                        iiw = lse(iiw, T[IIW, IMM] + d)
                if ys.is_ready() or self.y_empty:
                    for xt in xs.absorb_out:
                        xtr = x.trans[xt]
                        dc = self.xy_cell(xtr.dest, j)
                        d1 = xtr.lp_trans + self.rootsubx[xtr.dest] + dc[IMD]
                        d2 = xtr.lp_trans + self.insx[xtr.dest] + dc[IIW]
                        imm = lse(imm, T[IMM, IMD] + d1)
                        imd = lse(imd, T[IMD, IMD] + d1)
                        idm = lse(idm, T[IDM, IMD] + d1)
                        imi = lse(imi, T[IMI, IMD] + d1)
                        imm = lse(imm, T[IMM, IIW] + d2)
                        imi = lse(imi, T[IMI, IIW] + d2)
                        iiw = lse(iiw, T[IIW, IIW] + d2)
                if xs.is_ready() or self.x_empty:
                    for yt in ys.absorb_out:
                        ytr = y.trans[yt]
                        dc = self.xy_cell(i, ytr.dest)
                        d1 = ytr.lp_trans + self.rootsuby[ytr.dest] + dc[IDM]
                        d2 = ytr.lp_trans + self.insy[ytr.dest] + dc[IMI]
                        imm = lse(imm, T[IMM, IDM] + d1)
                        imd = lse(imd, T[IMD, IDM] + d1)
                        idm = lse(idm, T[IDM, IDM] + d1)
                        iiw = lse(iiw, T[IIW, IDM] + d1)
                        imm = lse(imm, T[IMM, IMI] + d2)
                        imi = lse(imi, T[IMI, IMI] + d2)
                if ys.is_ready() or self.y_empty:
                    for xt in xs.null_out:
                        xtr = x.trans[xt]
                        dc = self.xy_cell(xtr.dest, j)
                        imd = lse(imd, xtr.lp_trans + dc[IMD])
                        iiw = lse(iiw, xtr.lp_trans + dc[IIW])
                        imm = lse(imm, xtr.lp_trans + dc[IMM])
                for yt in ys.null_out:
                    ytr = y.trans[yt]
                    dc = self.xy_cell(i, ytr.dest)
                    idm = lse(idm, ytr.lp_trans + dc[IDM])
                    imi = lse(imi, ytr.lp_trans + dc[IMI])
                    if xs.is_emit_or_start():
                        imm = lse(imm, ytr.lp_trans + dc[IMM])
                srcc[IMM], srcc[IMD], srcc[IDM], srcc[IMI], srcc[IIW] = imm, imd, idm, imi, iiw

    def cell_post_prob(self, c):
        """src/forward.cpp:1172-1174"""
        return math.exp(self.fwd.cellc(c) + self.cellc(c) - self.fwd.lp_end)

    def dest_transitions(self, src):
        """src/forward.cpp:1224-1285"""
        x, y, hmm = self.x, self.y, self.hmm
        sx, sy, ss = src
        xs, ys = x.state[sx], y.state[sy]
        clp = {}
        for xt in xs.absorb_out:
            xtr = x.trans[xt]
            for yt in ys.absorb_out:
                ytr = y.trans[yt]
                clp[(xtr.dest, ytr.dest, IMM)] = hmm.lp_trans(ss, IMM) + xtr.lp_trans + ytr.lp_trans
        if ys.is_ready() or self.y_empty:
            for xt in xs.absorb_out:
                xtr = x.trans[xt]
                clp[(xtr.dest, sy, IMD)] = hmm.lp_trans(ss, IMD) + xtr.lp_trans
                clp[(xtr.dest, sy, IIW)] = hmm.lp_trans(ss, IIW) + xtr.lp_trans
        if xs.is_ready() or self.x_empty:
            for yt in ys.absorb_out:
                ytr = y.trans[yt]
                clp[(sx, ytr.dest, IDM)] = hmm.lp_trans(ss, IDM) + ytr.lp_trans
                clp[(sx, ytr.dest, IMI)] = hmm.lp_trans(ss, IMI) + ytr.lp_trans
        if (ys.is_ready() or self.y_empty) and ss in (IMD, IIW, IMM):
            for xt in xs.null_out:
                xtr = x.trans[xt]
                if xtr.dest != self.x_size - 1:
                    clp[(xtr.dest, sy, ss)] = xtr.lp_trans
        if ss in (IDM, IMI) or (xs.is_emit_or_start() and ss == IMM):
            for yt in ys.null_out:
                ytr = y.trans[yt]
                if ytr.dest != self.y_size - 1:
                    clp[(sx, ytr.dest, ss)] = ytr.lp_trans
        for xt in xs.null_out:
            xtr = x.trans[xt]
            if xtr.dest == self.x_size - 1:
                for yt in ys.null_out:
                    ytr = y.trans[yt]
                    if ytr.dest == self.y_size - 1:
                        clp[(xtr.dest, ytr.dest, EEE)] = xtr.lp_trans + ytr.lp_trans + hmm.lp_trans(ss, EEE)
        for k in clp:
            clp[k] += self.lp_cell_emit_or_absorb(k)
        return clp

    def dest_cells(self, src):
        """src/forward.cpp:1216-1222"""
        clp = self.dest_transitions(src)
        for k in clp:
            if k[2] != EEE:
                clp[k] += self.cellc(k)
        return clp

    def best_trace(self, start):
        """src/forward.cpp:1287-1300"""
        path = []
        cur = start
        while cur[0] < self.x_size - 1 and cur[1] < self.y_size - 1:
            cur = self.best_cell(self.dest_cells(cur))
            path.append(cur)
        path.append(self.end_cell)
        return path

    def cells_above_post_prob_threshold(self, min_post_prob):
        """src/forward.cpp:1302-1319.  Returns the priority_queue's pop order:
        descending logPostProb.  std::priority_queue is not stable, so the order
        of exact ties is an artefact of libstdc++'s heap: emulated with the same
        push sequence and heapq semantics replaced by an explicit binary max-heap
        (std::push_heap/pop_heap sift rules)."""
        lpp_thr = safe_log(min_post_prob)
        fwd_end = self.fwd.lp_end
        heap = _StdMaxHeap()
        for i in range(self.x_size - 2, -1, -1):
            for j in range(self.y_size - 2, -1, -1):
                if self.in_envelope(i, j):
                    b = self.xy_cell(i, j)
                    f = self.fwd.xy_cell(i, j)
                    for s in STATES:
                        lpp = b[s] + f[s] - fwd_end
                        if lpp >= lpp_thr:
                            heap.push((lpp, (i, j, s)))
        out = []
        while heap.a:
            out.append(heap.pop())
        return out

    def add_cells(self, cells, max_cells, fwd_trace, back_trace, keep_gaps_open):
        """src/forward.cpp:1343-1371"""
        new_cells = []
        for c in reversed(fwd_trace):
            if c in cells:
                break
            new_cells.append(c)
        for c in back_trace:
            if c in cells:
                break
            new_cells.append(c)
        if max_cells > 0 and len(cells) > 0 and len(cells) + len(new_cells) > max_cells:
            return False
        cells.update(new_cells)
        if keep_gaps_open:
            for nc in new_cells:
                for eqv in self.equiv_absorb_cells(nc):
                    if eqv not in cells and self.cell_post_prob(eqv) > 0 and self.in_envelope(eqv[0], eqv[1]):
                        self.add_trace(eqv, cells, max_cells, False)
        return True

    def add_trace(self, cell, cells, max_cells, keep_gaps_open):
        """src/forward.cpp:1373-1378"""
        fwd_trace = self.fwd.best_trace(cell)
        back_trace = self.best_trace(cell)
        return self.add_cells(cells, max_cells, fwd_trace, back_trace, keep_gaps_open)

    def best_profile(self, strategy=DPMatrix.CollapseChains):
        """src/forward.cpp:1321-1325"""
        cells = set()
        self.add_trace(self.end_cell, cells, 0, (strategy & self.KeepGapsOpen) != 0)
        return self.fwd.make_profile(cells, strategy)

    def post_prob_profile(self, min_post_prob, max_cells=0, strategy=DPMatrix.CollapseChains):
        """src/forward.cpp:1327-1341"""
        bc = self.cells_above_post_prob_threshold(min_post_prob)   # already in pop order
        cells = set()
        kgo = (strategy & self.KeepGapsOpen) != 0
        if not bc or (strategy & self.IncludeBestTrace):
            self.add_cells(cells, 0, self.fwd.best_trace(), [], kgo)
        k = 0
        while (max_cells == 0 or len(cells) < max_cells) and k < len(bc):
            best = bc[k][1]
            if best in cells:
                k += 1
            elif not self.add_trace(best, cells, max_cells, kgo):
                break
        return self.fwd.make_profile(cells, strategy)


class _StdMaxHeap:
    """std::priority_queue<CellPostProb> with operator< on logPostProb only
    (src/forward.h:191-198): libstdc++ push_heap / pop_heap (__adjust_heap)."""

    def __init__(self):
        self.a = []

    def push(self, v):
        a = self.a
        a.append(v)
        hole = len(a) - 1
        parent = (hole - 1) // 2
        while hole > 0 and a[parent][0] < v[0]:
            a[hole] = a[parent]
            hole = parent
            parent = (hole - 1) // 2
        a[hole] = v

    def pop(self):
        a = self.a
        top = a[0]
        last = a.pop()
        n = len(a)
        if n == 0:
            return top
        # __adjust_heap(first, hole=0, len=n, value=last)
        hole = 0
        child = 0
        while child < (n - 1) // 2:
            child = 2 * (child + 1)
            if a[child][0] < a[child - 1][0]:
                child -= 1
            a[hole] = a[child]
            hole = child
        if (n & 1) == 0 and child == (n - 2) // 2:
            child = 2 * (child + 1)
            a[hole] = a[child - 1]
            hole = child - 1
        parent = (hole - 1) // 2
        while hole > 0 and a[parent][0] < last[0]:
            a[hole] = a[parent]
            hole = parent
            parent = (hole - 1) // 2
        a[hole] = last
        return top


# ----------------------------------------------------------------------------
# Progressive reconstruction driver: Reconstructor::prepareRecon / reconstruct
# (reference src/recon.cpp:864-915, 917-1052), default options (sampled profiles,
# IncludeBestTrace | CollapseChains, band-doubling retry), root reconstruction only.
# ----------------------------------------------------------------------------
# (terms of the series, squarings) by the size of the matrix: double-precision row of the table in
# GSL's linalg/exponential.c (Moler & Van Loan, "Nineteen dubious ways to compute the exponential of a
# matrix", method 3), for sup-norms below 0.01, 0.1, 1, 10, 100, 1000
_MVL_DOUBLE = ((5, 1), (5, 4), (7, 5), (9, 7), (10, 10), (8, 14))


def sub_prob_matrix_ss(sr, t):
    """exp(R t) as the reference obtains it: gsl_linalg_exponential_ss(R t, GSL_PREC_DOUBLE)
    (src/model.cpp:322-334).  GSL is a third-party dependency absent from /root/reference and from
    this image (README.md:28 asks for "2.2.1 or later"; no lock file pins it), so its published
    algorithm is restated: the matrix is divided by 2^j, a k-term Taylor series is evaluated in
    Horner form (eB = 1 + B/k, then eB = 1 + B eB / c for c = k-1 .. 1), and the result is squared
    j times, with (k, j) from the table above by the largest |element|; products accumulate in
    increasing inner index (the row-major loop of GSL's own BLAS), every multiply and add rounded
    separately.  RateModel::getSubProbMatrix of the C++ host mirror performs the same IEEE
    operations in the same order, so both sides of a whole-tree parity test feed the DP
    bit-identical matrices.  Anchor: with these matrices the sampling-mode reconstructions of the
    reference's `testhist` target (Makefile:307-308) come out byte for byte; a 24-term series with
    other scaling, accurate to the same 1e-14, flips an exact tie at the root of that family by one
    ulp (tests/test_oracle_testhist.py)."""
    n = len(sr)
    rt = [[float(sr[i][j]) * t for j in range(n)] for i in range(n)]
    norm = 0.
    for row in rt:
        for v in row:
            norm = max(norm, abs(v))
    if norm < 0.01:
        terms, squarings = _MVL_DOUBLE[0]
    elif norm < 0.1:
        terms, squarings = _MVL_DOUBLE[1]
    elif norm < 1.:
        terms, squarings = _MVL_DOUBLE[2]
    elif norm < 10.:
        terms, squarings = _MVL_DOUBLE[3]
    elif norm < 100.:
        terms, squarings = _MVL_DOUBLE[4]
    elif norm < 1000.:
        terms, squarings = _MVL_DOUBLE[5]
    else:
        terms, squarings = _MVL_DOUBLE[5]
        squarings += int(math.ceil(math.log(1.01 * norm / 1000.) / math.log(2.)))
    shrink = 1. / math.exp(math.log(2.) * squarings)
    b = [[v * shrink for v in row] for row in rt]

    def matmul(p, q):
        c = [[0.] * n for _ in range(n)]
        for i in range(n):
            ci = c[i]
            for k in range(n):
                pik = p[i][k]
                if pik != 0:
                    qk = q[k]
                    for j in range(n):
                        ci[j] += pik * qk[j]
        return c

    first = 1. / terms
    eb = [[v * first for v in row] for row in b]
    for i in range(n):
        eb[i][i] += 1.
    for count in range(terms - 1, 0, -1):
        eb = matmul(b, eb)
        inv = 1. / count
        eb = [[v * inv for v in row] for row in eb]
        for i in range(n):
            eb[i][i] += 1.
    for _ in range(squarings):
        eb = matmul(eb, eb)
    return eb


class ReconTree:
    """Post-order node arrays (children before parents, root last; reference tree.cpp:214-218)."""

    def __init__(self, parent, branch_length, name):
        self.parent, self.branch_length, self.name = list(parent), list(branch_length), list(name)
        self.child = [[] for _ in parent]
        for n, p in enumerate(parent):
            if p >= 0:
                assert p > n, "tree nodes are not sorted in postorder"
                self.child[p].append(n)

    def nodes(self):
        return len(self.parent)

    def is_leaf(self, n):
        return not self.child[n]

    def root(self):
        return len(self.parent) - 1


def closest_leaves(tree):
    """reference src/recon.cpp:885-906"""
    closest, dist = [], []
    for node in range(tree.nodes()):
        if tree.is_leaf(node):
            closest.append(node)
            dist.append(0.)
        else:
            cl, dcl = -1, 0.
            for nc, c in enumerate(tree.child[node]):
                dc = dist[c] + tree.branch_length[c]
                if nc == 0 or dc < dcl:
                    cl, dcl = closest[c], dc
            closest.append(cl)
            dist.append(dcl)
    return closest


def reconstruct(model, tree, seqs, guide, max_distance_from_guide=20, profile_samples=10, max_profile_states=0,
                seed=5489, forward_factory=None, sub_prob=sub_prob_matrix_ss, min_post_prob=None,
                backward_factory=None):
    """seqs: dict leaf node -> (name, sequence); guide: AlignPath over leaf node rows (or {}).
    forward_factory(x, y, hmm, node, env) -> filled ForwardMatrix (default: the Python fill).
    min_post_prob: if given, non-root profiles are posterior profiles (usePosteriorsForProfile,
    reference src/recon.cpp:978-1013) built from backward_factory(fwd) (default: the Python fill).
    Returns dict(path, lp_final_fwd, lp_final_trace, bands, prof)."""
    if forward_factory is None:
        forward_factory = lambda x, y, hmm, node, env: ForwardMatrix(x, y, hmm, node, env)
    closest = closest_leaves(tree)
    gen = MT19937(seed)
    strategy = DPMatrix.CollapseChains | DPMatrix.IncludeBestTrace
    prof, bands = {}, {}
    path, lp_final_fwd, lp_final_trace = {}, NEG_INF, NEG_INF
    log_cptw = [safe_log(w) for w in model.cpt_weight]
    log_root = [[safe_log(v) for v in rv] for rv in model.ins_prob]
    for node in range(tree.nodes()):
        if tree.is_leaf(node):
            name, s = seqs[node]
            prof[node] = Profile.from_seq(model.components(), model.alphabet, s, node, name)
            continue
        lc, rc = tree.child[node]
        lprobs = ProbModel(model, tree.branch_length[lc], [sub_prob(sr, tree.branch_length[lc]) for sr in model.sub_rate])
        rprobs = ProbModel(model, tree.branch_length[rc], [sub_prob(sr, tree.branch_length[rc]) for sr in model.sub_rate])
        hmm = PairHMM(lprobs, rprobs, model.ins_prob)
        max_dist = max_distance_from_guide
        while True:
            env = (GuideAlignmentEnvelope() if not guide
                   else GuideAlignmentEnvelope(guide, closest[lc], closest[rc], max_dist))
            fwd = forward_factory(prof[lc], prof[rc], hmm, node, env)
            if fwd.lp_end > NEG_INF:
                break
            assert max_dist >= 0, "Zero forward likelihood even in the absence of guide alignment constraints"
            if max_dist * 2 > align_path_columns(guide):
                max_dist = -1
            elif max_dist == 0:
                max_dist = 1
            else:
                max_dist *= 2
        bands[node] = max_dist
        if node == tree.root():
            path = fwd.best_align_path()
            node_prof = fwd.best_profile()
            lp_final_fwd = fwd.lp_end
        elif min_post_prob is not None:
            bwd = (backward_factory or BackwardMatrix)(fwd)
            node_prof = bwd.post_prob_profile(min_post_prob, max_profile_states, strategy)
        else:
            node_prof = fwd.sample_profile(gen, profile_samples, max_profile_states, strategy)
        lp_trace = node_prof.calc_sum_path_absorb_probs(log_cptw, log_root, None)
        if node == tree.root():
            lp_final_trace = lp_trace
        prof[node] = node_prof
    return dict(path=path, lp_final_fwd=lp_final_fwd, lp_final_trace=lp_final_trace, bands=bands, prof=prof)


def gapped_rows(tree, seqs, path):
    """Alignment(ungapped, path).gapped() (reference src/alignpath.cpp:232-280, recon.cpp:1410-1421):
    leaves show their residues, internal nodes the wildcard character."""
    rows = {}
    for node in sorted(path):
        s = seqs[node][1] if node in seqs else None
        out, k = [], 0
        for b in path[node]:
            if b:
                out.append(s[k] if s is not None else WILDCARD_CHAR)
                k += 1
            else:
                out.append("-")
        rows[node] = "".join(out)
    return rows
