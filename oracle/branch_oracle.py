"""TEST INFRASTRUCTURE ONLY - CPU restatement of the per-branch pair DPs of SURVEY.md section 8(f) row N4: the three-state
(Match / Insert / Delete) alignment of a parent sequence profile x with a child sequence profile y across one tree branch,
inside a GuideAlignmentEnvelope, in its two forms:

  * Refiner::BranchMatrix (reference src/refiner.cpp:10-60): Viterbi (max-plus) fill, `best()` traceback (:62-104) - what
    `historian reconstruct -refine` runs on every branch;
  * Sampler::BranchMatrix (reference src/sampler.cpp:1034-1084): the same lattice with the reference's log_sum_exp - the
    MCMC sampler's branch move.

and the pieces either is built from: TreeAlignFuncs::SparseDPMatrix<3>::inEnvelope (src/sampler.h:146-149),
BranchMatrixBase (src/sampler.cpp:1005-1032: ySub = preMultiply, yEmit = calcInsProbs, the eleven transition scores),
preMultiply / calcInsProbs (src/sampler.cpp:452-476), ProbModel::transProb (src/model.cpp:400-447), lpEmit / getColumn
(src/sampler.cpp:1166-1183).

PARITY UNPINNED BY REFERENCE FIXTURES: the reference's Makefile runs every `recon` test with -norefine and no test drives
the sampler, so no golden file holds a branch matrix.  The restatement is pinned instead by exhaustive enumeration of all
alignments of tiny pairs (tests/test_oracle_branch.py): the Viterbi score is the best path's score, the best() path attains
it, and the Forward score is the log of the summed path probabilities (to the accuracy of the table log_sum_exp)."""
import math

from oracle.historian_oracle import NEG_INF, log_inner_product_nested, log_sum_exp, safe_log

MATCH, INSERT, DELETE, END = 0, 1, 2, 3      # ProbModel::State (src/model.h:135-137; Start = Match = 0)


def trans_prob(ins, dele, ins_ext, del_ext, src, dest):
    """ProbModel::transProb (src/model.cpp:400-447)"""
    if src == MATCH:
        return {MATCH: (1 - ins) * (1 - dele), INSERT: ins, DELETE: (1 - ins) * dele, END: 1 - ins}[dest]
    if src == INSERT:
        return {MATCH: (1 - ins_ext) * (1 - dele), INSERT: ins_ext, DELETE: (1 - ins_ext) * dele, END: 1 - ins_ext}[dest]
    return {MATCH: 1 - del_ext, END: 1 - del_ext, INSERT: 0., DELETE: del_ext}[dest]


def trans_scores(ins, dele, ins_ext, del_ext):
    """lpTrans for every (src, dest): [3][4] with -inf for Delete -> Insert (src/sampler.cpp:1016-1028, 1162-1164)"""
    return [[safe_log(trans_prob(ins, dele, ins_ext, del_ext, s, d)) for d in (MATCH, INSERT, DELETE, END)]
            for s in (MATCH, INSERT, DELETE)]


def pre_multiply(child, log_sub):
    """TreeAlignFuncs::preMultiply (src/sampler.cpp:452-463): pwm[pos][cpt][i] = (+)_j logSub[cpt][i][j] + child[pos][cpt][j]"""
    out = []
    for lpp in child:
        pre = []
        for cpt, sub in enumerate(log_sub):
            row = []
            for i in range(len(sub)):
                acc = NEG_INF
                for j in range(len(lpp[cpt])):
                    acc = log_sum_exp(acc, sub[i][j] + lpp[cpt][j])
                row.append(acc)
            pre.append(row)
        out.append(pre)
    return out


def calc_ins_probs(child, log_ins, log_cpt_weight):
    """TreeAlignFuncs::calcInsProbs (src/sampler.cpp:465-476)"""
    out = []
    for lpp in child:
        lp = NEG_INF
        for cpt in range(len(log_ins)):
            for i in range(len(lpp[cpt])):
                lp = log_sum_exp(lp, log_cpt_weight[cpt] + log_ins[cpt][i] + lpp[cpt][i])
        out.append(lp)
    return out


class BranchMatrix:
    """x_seq, y_sub: PosWeightMatrix [pos][cpt][tok] of log weights (y already pre-multiplied); y_emit[pos]; T = trans_scores();
    x_env / y_env: per position (0 .. len) the envelope coordinate (cumulativeMatches at the position's guide column), or None
    with max_dist < 0 for no band.  viterbi=True: Refiner::BranchMatrix; False: Sampler::BranchMatrix."""

    def __init__(self, x_seq, y_sub, y_emit, T, x_env=None, y_env=None, max_dist=-1, viterbi=True):
        self.x_seq, self.y_sub, self.y_emit, self.T = x_seq, y_sub, y_emit, T
        self.x_size, self.y_size = len(x_seq) + 1, len(y_sub) + 1
        self.x_env, self.y_env, self.max_dist = x_env, y_env, max_dist
        self.viterbi = viterbi
        self.cells = {}
        self.fill()

    def in_envelope(self, i, j):
        """SparseDPMatrix::inEnvelope (src/sampler.h:146-149) with GuideAlignmentEnvelope::inRange (src/alignpath.h:56-61)"""
        if i == 0 or j == 0 or i == self.x_size - 1 or j == self.y_size - 1 or self.max_dist < 0:
            return True
        return abs(self.x_env[i] - self.y_env[j]) <= self.max_dist

    def cell(self, i, j, s):
        return self.cells.get((i, j), (NEG_INF, NEG_INF, NEG_INF))[s]

    def log_match(self, i, j):
        """BranchMatrixBase::logMatch (src/sampler.h:207-209)"""
        return log_inner_product_nested(self.x_seq[i - 1], self.y_sub[j - 1])

    def combine(self, *terms):
        if self.viterbi:
            return max(terms)
        return log_sum_exp(*terms)

    def fill(self):
        """src/refiner.cpp:16-56 / src/sampler.cpp:1040-1080: a cell that is not computed keeps -inf (XYCell's constructor)"""
        T = self.T
        for i in range(self.x_size):
            for j in range(self.y_size):
                if not self.in_envelope(i, j):
                    continue
                m, ins, dele = NEG_INF, NEG_INF, NEG_INF
                if i == 0 and j == 0:
                    m = 0.                                                  # lpStart() = 0
                if i > 0 and self.in_envelope(i - 1, j):
                    s = self.cells.get((i - 1, j), (NEG_INF,) * 3)
                    dele = self.combine(s[MATCH] + T[MATCH][DELETE], s[INSERT] + T[INSERT][DELETE], s[DELETE] + T[DELETE][DELETE])
                if j > 0 and self.in_envelope(i, j - 1):
                    s = self.cells.get((i, j - 1), (NEG_INF,) * 3)
                    ins = self.y_emit[j - 1] + self.combine(s[MATCH] + T[MATCH][INSERT], s[INSERT] + T[INSERT][INSERT])
                if i > 0 and j > 0 and self.in_envelope(i - 1, j - 1):
                    s = self.cells.get((i - 1, j - 1), (NEG_INF,) * 3)
                    m = self.log_match(i, j) + self.combine(s[MATCH] + T[MATCH][MATCH], s[INSERT] + T[INSERT][MATCH], s[DELETE] + T[DELETE][MATCH])
                self.cells[(i, j)] = (m, ins, dele)
        e = self.cells.get((self.x_size - 1, self.y_size - 1), (NEG_INF,) * 3)
        self.lp_end = self.combine(e[MATCH] + T[MATCH][END], e[INSERT] + T[INSERT][END], e[DELETE] + T[DELETE][END])

    def lp_emit(self, i, j, s):
        """BranchMatrixBase::lpEmit (src/sampler.cpp:1166-1173)"""
        if s == MATCH:
            return self.log_match(i, j) if i > 0 and j > 0 else NEG_INF
        if s == INSERT:
            return self.y_emit[j - 1] if j > 0 else NEG_INF
        return 0.

    def best(self):
        """Refiner::BranchMatrix::best (src/refiner.cpp:62-104): (x row, y row) of the best alignment, first column first"""
        i, j, state = self.x_size - 1, self.y_size - 1, END
        xp, yp = [], []
        while i > 0 or j > 0:
            x = state == DELETE or (state == MATCH and i > 0 and j > 0)
            y = state == INSERT or (state == MATCH and i > 0 and j > 0)
            if x or y:
                xp.append(x)
                yp.append(y)
            si, sj = (i - 1 if x else i), (j - 1 if y else j)
            e = 0. if state == END else self.lp_emit(i, j, state)
            best_lp, best_s, found = NEG_INF, None, False
            for s in (MATCH, INSERT, DELETE):
                lp = self.cell(si, sj, s) + self.T[s][state] + e
                if lp > best_lp:
                    best_lp, best_s, found = lp, s, True
            assert found, "could not find traceback state"
            i, j, state = si, sj, best_s
        return xp[::-1], yp[::-1]


def enumerate_paths(nx, ny):
    """all state paths of the three-state pair model that emit nx parent and ny child positions: lists of states"""
    out = []

    def rec(i, j, path):
        if i == nx and j == ny:
            out.append(list(path))
            return
        if i < nx and j < ny:
            rec(i + 1, j + 1, path + [MATCH])
        if j < ny:
            rec(i, j + 1, path + [INSERT])
        if i < nx:
            rec(i + 1, j, path + [DELETE])
    rec(0, 0, [])
    return out


def path_log_prob(bm, path):
    """log-probability of one state path, in plain floating point (no table)"""
    lp, i, j, prev = 0., 0, 0, MATCH
    for s in path:
        if s == MATCH:
            i, j = i + 1, j + 1
        elif s == INSERT:
            j += 1
        else:
            i += 1
        lp += bm.T[prev][s] + bm.lp_emit(i, j, s)
        prev = s
    return lp + bm.T[prev][END]
