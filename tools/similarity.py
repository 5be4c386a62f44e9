"""Token-level overlap of the host mirror's sources with the reference: the share of a file's 10-token shingles
(comments, whitespace and #include lines removed, so re-wrapping and re-indenting hide nothing) that occur anywhere
in /root/reference/src or /root/reference/t, and the longest run of consecutive shared tokens - the measure VERDICT r02
applied.    python tools/similarity.py [files...]          (needs /root/reference; not run on the GPU box)"""
import glob
import os
import re
import sys

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 10
TOKEN = re.compile(r"[A-Za-z_]\w*|\d+\.?\d*(?:[eE][-+]?\d+)?|->|::|<<|>>|<=|>=|==|!=|&&|\|\||\+\+|--|[-+*/%=<>!&|^~?:;,.(){}\[\]]|\"(?:\\.|[^\"\\])*\"|'(?:\\.|[^'\\])*'")


def tokens(path):
    text = open(path, errors="replace").read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = "\n".join(l for l in text.split("\n") if not l.lstrip().startswith("#include"))
    return TOKEN.findall(text)


def shingles(tok):
    return {tuple(tok[k:k + N]) for k in range(len(tok) - N + 1)}


ref = set()
for p in glob.glob(REF + "/src/*") + glob.glob(REF + "/t/*") + glob.glob(REF + "/target/*"):
    if p.endswith((".cpp", ".h", ".c")):
        ref |= shingles(tokens(p))

files = sys.argv[1:] or sorted(glob.glob(ROOT + "/historian_amd/csrc/host/*.cpp") + glob.glob(ROOT + "/historian_amd/csrc/host/*.h") +
                               glob.glob(ROOT + "/historian_amd/csrc/host/t/*") + glob.glob(ROOT + "/historian_amd/csrc/*.hip") +
                               glob.glob(ROOT + "/historian_amd/csrc/*.h") + glob.glob(ROOT + "/include/*.h") + glob.glob(ROOT + "/oracle/*.c*"))
for p in files:
    tok = tokens(p)
    if len(tok) < N:
        continue
    hit = [tuple(tok[k:k + N]) in ref for k in range(len(tok) - N + 1)]
    longest, cur = 0, 0
    for h in hit:
        cur = cur + 1 if h else 0
        longest = max(longest, cur)
    print("%-58s %6d tokens  %5.1f %% of 10-token shingles shared   longest shared run %d tokens" %
          (os.path.relpath(p, ROOT), len(tok), 100.0 * sum(hit) / len(hit), longest + N - 1 if longest else 0))
