"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's guide-alignment pair DP
(SURVEY section 8f, row N1): the k-mer seeded DiagonalEnvelope and the 3-state Viterbi
QuickAlignMatrix with its traceback.

Follows, function by function:
  src/fastseq.cpp:146-163,255-266     kmerValid, makeKmer, KmerIndex
  src/diagenv.cpp:104-226             DiagonalEnvelope::initFull / initSparse / initStorage
  src/diagenv.h:57-99                 storage-diagonal bookkeeping (only the *set* semantics matter here:
                                      a cell outside the envelope reads as -inf)
  src/quickalign.cpp:7-99             QuickAlignMatrix constructor: scores and fill
  src/quickalign.cpp:147-207          alignPath() traceback
  src/quickalign.h:57-66              startGapScore / endGapScore (SeqIdx is `unsigned int`: the
                                      reference's `xLen-i-2` wraps for i = xLen-1, reproduced here)
  t/testquickalign.cpp                main()

Pinned by the reference's own fixture data/testquickalign.out.fa (tests/test_oracle_golden.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

from oracle import historian_oracle as ho

NEG_INF = float("-inf")
DEFAULT_KMER_LENGTH = 6
DEFAULT_BAND_SIZE = 64
MIN_KMERS_FOR_SPARSE_ENVELOPE = 2


def tokens(seq, alphabet):
    """FastSeq::unvalidatedTokens: -1 for characters outside the alphabet"""
    return [ho.tokenize(c, alphabet) for c in seq]


def kmer_valid(k, tok, pos):
    return all(tok[pos + j] >= 0 for j in range(k))


def make_kmer(k, tok, pos, alphabet_size):
    """src/fastseq.cpp:153-163"""
    kmer, mul = 0, 1
    for j in range(k):
        kmer += mul * tok[pos + k - j - 1]
        mul *= alphabet_size
    return kmer


class KmerIndex:
    """src/fastseq.cpp:255-266"""

    def __init__(self, seq, alphabet, kmer_len):
        self.alphabet, self.kmer_len = alphabet, kmer_len
        self.locations = {}
        tok = tokens(seq, alphabet)
        for j in range(0, len(seq) - kmer_len + 1):
            if kmer_valid(kmer_len, tok, j):
                self.locations.setdefault(make_kmer(kmer_len, tok, j, len(alphabet)), []).append(j)


class DiagonalEnvelope:
    """Set of diagonals d = i - j (1 <= i <= xLen, 1 <= j <= yLen) that the DP visits."""

    def __init__(self, x, y):
        self.x, self.y = x, y
        self.x_len, self.y_len = len(x), len(y)
        self.diagonals = []

    def min_diagonal(self):
        return 1 - self.y_len

    def max_diagonal(self):
        return self.x_len - 1

    def init_full(self):
        """src/diagenv.cpp:104-111"""
        self.diagonals = list(range(self.min_diagonal(), self.max_diagonal() + 1))

    def init_sparse(self, y_kmer_index, band_size=DEFAULT_BAND_SIZE, kmer_threshold=-1, cell_size=8, max_size=0):
        """src/diagenv.cpp:113-199"""
        kmer_len = y_kmer_index.kmer_len
        if kmer_threshold >= 0:
            min_len = MIN_KMERS_FOR_SPARSE_ENVELOPE * (kmer_len + kmer_threshold)
            if self.x_len < min_len or self.y_len < min_len:
                return self.init_full()
        elif self.x_len * self.y_len * cell_size < max_size:
            return self.init_full()
        x_tok = tokens(self.x, y_kmer_index.alphabet)
        a = len(y_kmer_index.alphabet)
        diag_count = {}
        for i in range(0, self.x_len - kmer_len + 1):
            if kmer_valid(kmer_len, x_tok, i):
                for j in y_kmer_index.locations.get(make_kmer(kmer_len, x_tok, i, a), ()):
                    diag_count[i - j] = diag_count.get(i - j, 0) + 1
        distrib = {}
        for d, c in diag_count.items():
            distrib.setdefault(c, set()).add(d)
        diags, storage = {0}, {0}
        half = band_size // 2
        diag_size = min(self.x_len, self.y_len) * cell_size
        for count in sorted(distrib, reverse=True):
            if kmer_threshold >= 0 and count < kmer_threshold:
                break
            more, more_storage = set(diags), set(storage)
            for seed in distrib[count]:
                # (the reference stores seed diagonals as unsigned in this set; negative ones wrap and the
                # std::max/min against int bounds below then clamps them -- reproduced by the cast)
                seed = _as_int32_of_unsigned(seed)
                d_min = max(self.min_diagonal(), seed - half)
                d_max = min(self.max_diagonal(), seed + half)
                more.update(range(d_min, d_max + 1))
                more_storage.update(range(d_min - 1, d_max + 2))
            if kmer_threshold < 0 and len(more_storage) * diag_size >= max_size:
                break
            diags, storage = more, more_storage
        self.diagonals = sorted(diags)

    def contains(self, i, j):
        import bisect
        d = i - j
        k = bisect.bisect_left(self.diagonals, d)
        return k < len(self.diagonals) and self.diagonals[k] == d


def _as_int32_of_unsigned(d):
    """`set<unsigned int>` element read back as `(int) seedDiag`: identity for values that fit an int"""
    u = d & 0xFFFFFFFF
    return u - (1 << 32) if u >= (1 << 31) else u


class QuickAlignScores:
    """The score constants of src/quickalign.cpp:26-54."""

    def __init__(self, model, time, sub_prob=ho.sub_prob_matrix_ss):
        pm = ho.ProbModel(model, time, [sub_prob(sr, time) for sr in model.sub_rate])
        log_ins = [ho.safe_log(v) for v in pm.ins_vec[0]]
        a = len(model.alphabet)
        self.submat = [[ho.safe_log(pm.sub_mat[0][i][j]) - log_ins[j] for j in range(a)] for i in range(a)]
        gap_prob = pm.ins + (1 - pm.ins) * pm.dele
        no_gap_prob = 1 - gap_prob
        gap_ext = 1 / ((pm.ins / gap_prob) / pm.ins_ext + (1 - pm.ins / gap_prob) / pm.del_ext)
        no_gap_ext = 1 - gap_ext
        self.no_gap = math.log(no_gap_prob)
        self.gap_open = math.log(gap_prob) + math.log(no_gap_ext)
        self.gap_extend = math.log(gap_ext)
        self.m2i = math.log(gap_prob)
        self.m2d = math.log(no_gap_prob * gap_prob)
        self.m2m = math.log(no_gap_prob * no_gap_prob)
        self.i2i = math.log(gap_ext)
        self.i2d = math.log(no_gap_ext * gap_prob)
        self.i2m = math.log(no_gap_ext * no_gap_prob)
        self.d2d = math.log(gap_ext)
        self.d2m = math.log(no_gap_ext)


START, MATCH, INSERT, DELETE = 0, 1, 2, 3


class QuickAlignMatrix:
    def __init__(self, env, model, time, scores=None, fill=True):
        self.env = env
        self.x, self.y = env.x, env.y
        self.x_len, self.y_len = env.x_len, env.y_len
        self.x_tok = tokens(self.x, model.alphabet)
        self.y_tok = tokens(self.y, model.alphabet)
        self.sc = scores or QuickAlignScores(model, time)
        self.start = 0.0
        self.end = NEG_INF
        self.x_end = self.y_end = 0
        self.cells = {}          # (i, j) -> [mat, ins, del]; absent = -inf (the reference's `dummy`)
        if fill:
            self.fill()
        self.result = self.end

    # --- score helpers (src/quickalign.h:43-66) ---
    def match_emit(self, i, j):
        xt, yt = self.x_tok[i - 1], self.y_tok[j - 1]
        return 0.0 if (xt < 0 or yt < 0) else self.sc.submat[xt][yt]

    def _gap(self, n_unsigned):
        return self.sc.gap_open + float(n_unsigned & 0xFFFFFFFF) * self.sc.gap_extend

    def start_gap(self, i, j):
        return ((self.sc.no_gap if i == 1 else self._gap(i - 2))
                + (self.sc.no_gap if j == 1 else self._gap(j - 2)))

    def end_gap(self, i, j):
        return ((self.sc.no_gap if i == self.x_len else self._gap(self.x_len - i - 2))
                + (self.sc.no_gap if j == self.y_len else self._gap(self.y_len - j - 2)))

    def get(self, i, j, k):
        c = self.cells.get((i, j))
        return NEG_INF if c is None else c[k]

    def fill(self):
        """src/quickalign.cpp:63-96"""
        sc = self.sc
        diags = self.env.diagonals
        for j in range(1, self.y_len + 1):
            for d in diags:
                i = d + j
                if i < 1 or i > self.x_len:
                    continue
                mat = max(max(self.get(i - 1, j - 1, 0) + sc.m2m, self.get(i - 1, j - 1, 2) + sc.d2m),
                          self.get(i - 1, j - 1, 1) + sc.i2m)
                mat = max(mat, self.start + self.start_gap(i, j))
                mat += self.match_emit(i, j)
                ins = max(self.get(i, j - 1, 1) + sc.i2i, self.get(i, j - 1, 0) + sc.m2i)
                dele = max(max(self.get(i - 1, j, 1) + sc.i2d, self.get(i - 1, j, 2) + sc.d2d),
                           self.get(i - 1, j, 0) + sc.m2d)
                self.cells[(i, j)] = [mat, ins, dele]
                ij_end = mat + self.end_gap(i, j)
                if ij_end > self.end:
                    self.x_end, self.y_end, self.end = i, j, ij_end

    def align_path(self):
        """src/quickalign.cpp:147-207: rows 0 (x) and 1 (y) as lists of bool"""
        assert self.result > NEG_INF, "Can't do Viterbi traceback if final score is -infinity"
        sc = self.sc
        i, j, state = self.x_end, self.y_end, MATCH
        assert i > 0 and j > 0
        p0 = [True] * (self.x_len - self.x_end) + [False] * (self.y_len - self.y_end)
        p1 = [False] * (self.x_len - self.x_end) + [True] * (self.y_len - self.y_end)
        r0, r1 = [], []          # built back to front
        while state != START:
            best, nxt = NEG_INF, state

            def upd(cand, st):
                nonlocal best, nxt
                if cand > best:
                    best, nxt = cand, st
            if state == MATCH:
                emit = self.match_emit(i, j)
                i -= 1
                j -= 1
                r0.append(True)
                r1.append(True)
                upd(self.get(i, j, 0) + sc.m2m + emit, MATCH)
                upd(self.get(i, j, 1) + sc.i2m + emit, INSERT)
                upd(self.get(i, j, 2) + sc.d2m + emit, DELETE)
                upd(self.start + self.start_gap(i + 1, j + 1) + emit, START)
                assert best == self.get(i + 1, j + 1, 0), "Traceback error at (%d,%d,Match)" % (i + 1, j + 1)
            elif state == INSERT:
                j -= 1
                r0.append(False)
                r1.append(True)
                upd(self.get(i, j, 0) + sc.m2i, MATCH)
                upd(self.get(i, j, 1) + sc.i2i, INSERT)
                assert best == self.get(i, j + 1, 1), "Traceback error at (%d,%d,Insert)" % (i, j + 1)
            else:
                i -= 1
                r0.append(True)
                r1.append(False)
                upd(self.get(i, j, 0) + sc.m2d, MATCH)
                upd(self.get(i, j, 1) + sc.i2d, INSERT)
                upd(self.get(i, j, 2) + sc.d2d, DELETE)
                assert best == self.get(i + 1, j, 2), "Traceback error at (%d,%d,Delete)" % (i + 1, j)
            state = nxt
        head0 = [False] * j + [True] * i
        head1 = [True] * j + [False] * i
        row0 = head0 + r0[::-1] + p0
        row1 = head1 + r1[::-1] + p1
        assert sum(row0) == self.x_len and sum(row1) == self.y_len and len(row0) == len(row1)
        return row0, row1

    def gapped(self):
        """Alignment(seqs, path).gapped() (src/alignpath.cpp:254-280)"""
        out = []
        for seq, row in zip((self.x, self.y), self.align_path()):
            k, g = 0, []
            for b in row:
                if b:
                    g.append(seq[k])
                    k += 1
                else:
                    g.append("-")
            out.append("".join(g))
        return out


def testquickalign_main(seq_file, model_file, time):
    """t/testquickalign.cpp; returns what the reference writes to stdout (writeFastaSeqs)."""
    from oracle.ref_mains import read_fasta
    seqs = read_fasta(seq_file)
    assert len(seqs) == 2, "Sequence file must have exactly two sequences"
    model = ho.RateModel.from_file(model_file)
    model.sub_rate = [m.tolist() for m in model.sub_rate]
    env = DiagonalEnvelope(seqs[0][1], seqs[1][1])
    env.init_full()
    mx = QuickAlignMatrix(env, model, float(time))
    out = []
    for (name, _), g in zip(seqs, mx.gapped()):
        out.append(">%s\n%s\n" % (name, g))
    return "".join(out)


# ----------------------------------------------------------------------------------------------------
# Around the pair DP: alignment merging and the alignment graph (reference src/alignpath.cpp:9-20,
# 93-217,232-280; src/span.cpp).  AlignPath = {row: [bool]}.
# ----------------------------------------------------------------------------------------------------
def is_gap(c):
    return c in "-."


def alignment_from_gapped(gapped):
    """Alignment::Alignment(gapped) (src/alignpath.cpp:232-248): [(name, gapped seq)] -> (ungapped [(name, seq)], path)"""
    cols = None
    for name, s in gapped:
        if cols is None:
            cols = len(s)
        assert cols == len(s), "Alignment is not flush"
    ungapped, path = [], {}
    for row, (name, s) in enumerate(gapped):
        path[row] = [not is_gap(c) for c in s]
        ungapped.append((name, "".join(c for c in s if not is_gap(c))))
    return ungapped, path


def align_path_columns(path):
    cols = None
    for row, p in path.items():
        if cols is None:
            cols = len(p)
        assert cols == len(p), "Alignment path is not flush"
    return cols or 0


def gapped_from_path(ungapped, path):
    """Alignment(ungapped, path).gapped() (src/alignpath.cpp:254-280)"""
    out = [("", "")] * len(ungapped)
    for row, p in path.items():
        name, s = ungapped[row]
        k, g = 0, []
        for b in p:
            if b:
                g.append(s[k])
                k += 1
            else:
                g.append("-")
        out[row] = (name, "".join(g))
    return out


class AlignSeqMap:
    """src/alignpath.cpp:93-150"""

    def __init__(self, alignments):
        self.alignments = alignments
        self.seq_len = {}
        self.align_cols = []
        self.align_col_row_to_pos = {}
        self.row_pos_align_to_col = {}
        for align in alignments:
            if len(align) == 0:
                self.align_cols.append(0)
                continue
            self.align_cols.append(align_path_columns(align))
            for row in sorted(align):
                n = sum(align[row])
                if row not in self.seq_len:
                    self.seq_len[row] = n
                else:
                    assert self.seq_len[row] == n, "Incompatible number of residues for row #%d of alignment" % row
        for n_align, align in enumerate(alignments):
            row_pos = {row: 0 for row in align}
            for col in range(self.align_cols[n_align]):
                all_gaps = True
                for row in sorted(align):
                    if align[row][col]:
                        pos = row_pos[row]
                        row_pos[row] += 1
                        self.align_col_row_to_pos.setdefault(n_align, {}).setdefault(col, {})[row] = pos
                        self.row_pos_align_to_col.setdefault(row, {}).setdefault(pos, {})[n_align] = col
                        all_gaps = False
                assert not all_gaps, "Column %d of alignment %d in AlignSeqMap is empty" % (col, n_align)

    def linked_columns(self, n_align, col):
        ac, queue = {}, {n_align: col}
        while len(queue) > len(ac):
            for na in sorted(queue):
                if na not in ac:
                    c = queue[na]
                    ac[na] = c
                    for row, pos in sorted(self.align_col_row_to_pos[na][c].items()):
                        for lna, lcol in sorted(self.row_pos_align_to_col[row][pos].items()):
                            if lna in ac:
                                assert ac[lna] == lcol, "Inconsistent alignments"
                            queue.setdefault(lna, lcol)
        return ac


def align_path_merge(alignments):
    """src/alignpath.cpp:153-203"""
    amap = AlignSeqMap(alignments)
    a = {row: [] for row in amap.seq_len}
    next_col = [0] * len(alignments)
    while True:
        all_done = none_ready = True
        for n in range(len(alignments)):
            if next_col[n] < amap.align_cols[n]:
                all_done = False
                linked = amap.linked_columns(n, next_col[n])
                ready = all(next_col[na] == c for na, c in sorted(linked.items()))
                if ready:
                    none_ready = False
                    if linked:
                        for row in a:
                            a[row].append(False)
                        for na, c in sorted(linked.items()):
                            for row in alignments[na]:
                                if alignments[na][row][c]:
                                    a[row][-1] = True
                            next_col[na] += 1
                    else:
                        next_col[n] += 1
                    break
        if none_ready and not all_done:
            raise AssertionError("align_path_merge fail, no alignments ready")
        if all_done:
            break
    align_path_columns(a)
    return a


def testmerge_main(files):
    """t/testmerge.cpp; returns the FASTA the reference writes (raises where the reference aborts)"""
    from oracle.ref_mains import read_fasta
    name_to_row, ungapped, paths = {}, [], []
    for f in files:
        gapped = read_fasta(f)
        ug, p = alignment_from_gapped(gapped)
        path = {}
        for n, (name, _) in enumerate(gapped):
            if name not in name_to_row:
                name_to_row[name] = len(ungapped)
                ungapped.append(ug[n])
            path[name_to_row[name]] = p[n]
        paths.append(path)
    merged = align_path_merge(paths)
    return "".join(">%s\n%s\n" % ns for ns in gapped_from_path(ungapped, merged))


class AlignGraph:
    """src/span.cpp, all-vs-all graph (buildDenseGraph).  The reference's sparse random graph draws its
    edges with std::uniform_int_distribution, whose output is standard-library specific (the reference's
    own Makefile skips testspan as platform dependent); it is not restated here."""

    def __init__(self, seqs, model, time, sparse_params=None, fill=None, scores=None):
        """seqs: [(name, seq)]; sparse_params: None = full envelopes, else dict(kmer_len, band_size,
        kmer_threshold, max_size) for DiagonalEnvelope.init_sparse; fill(env) -> QuickAlignMatrix-like."""
        self.seqs, self.model, self.time = seqs, model, time
        self.scores = scores or QuickAlignScores(model, time)
        self.edges = [ho._StdMaxHeap() for _ in seqs]
        self.edge_path = [dict() for _ in seqs]
        n_edge = 0
        for src in range(len(seqs) - 1):
            for dest in range(src + 1, len(seqs)):
                env = DiagonalEnvelope(seqs[src][1], seqs[dest][1])
                if sparse_params:
                    env.init_sparse(KmerIndex(seqs[dest][1], model.alphabet, sparse_params["kmer_len"]),
                                    sparse_params["band_size"], sparse_params["kmer_threshold"], 40, sparse_params["max_size"])
                else:
                    env.init_full()
                mx = fill(env) if fill else QuickAlignMatrix(env, model, time, scores=self.scores)
                r0, r1 = mx.align_path()
                self.edge_path[src][dest] = {src: r0, dest: r1}
                # (lp, tie-free payload): std::priority_queue<Edge> orders by lp only
                self.edges[src].push((mx.end, (src, dest, n_edge)))
                self.edges[dest].push((mx.end, (src, dest, n_edge)))
                n_edge += 1

    def min_span_tree(self):
        """src/span.cpp:122-143"""
        n = len(self.seqs)
        set_idx = list(range(n))
        sets = [{i} for i in range(n)]
        n_sets = n
        paths = []
        self.mst_edges = []

        def same(e):
            return set_idx[e[0]] == set_idx[e[1]]
        while n_sets > 1:
            best, found = None, False
            for src in sorted(sets[0]):
                h = self.edges[src]
                while h.a and same(h.a[0][1]):
                    h.pop()
                if h.a and (not found or best[0] < h.a[0][0]):
                    best, found = h.a[0], True
            assert found, "Found no valid edge"
            r1, r2 = best[1][0], best[1][1]
            self.mst_edges.append((r1, r2, best[0]))
            paths.append(self.edge_path[r1][r2])
            i1, i2 = sorted((set_idx[r1], set_idx[r2]))
            for m in sets[i2]:
                set_idx[m] = i1
            sets[i1] |= sets[i2]
            sets[i2] = set()
            n_sets -= 1
        return paths

    def mst_gapped(self):
        merged = align_path_merge(self.min_span_tree())
        return gapped_from_path(self.seqs, merged)
