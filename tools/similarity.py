"""Share of a source file's lines that occur verbatim in the reference files it mirrors (comments and whitespace
normalised away, trivial lines such as lone braces ignored) - the measure VERDICT r01 applied to the host mirror.
    python tools/similarity.py            (needs /root/reference; not run on the GPU box)"""
import os, re, sys
REF = "/root/reference"
HOST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "historian_amd", "csrc", "host")
MIRRORS = {
    "hx_host_profile.cpp": ["src/profile.cpp", "src/pairhmm.cpp", "src/profile.h", "src/pairhmm.h"],
    "hx_host_forward.cpp": ["src/forward.cpp", "src/forward.h"],
    "hx_host_walk.cpp": ["src/forward.cpp", "src/forward.h"],
    "hx_host.h": ["src/forward.h", "src/profile.h", "src/pairhmm.h", "src/logsumexp.h", "src/alignpath.h", "src/model.h", "src/util.h",
                  "src/fastseq.h", "src/recon.h", "src/quickalign.h", "src/diagenv.h", "src/span.h"],
    "hx_host_base.cpp": ["src/util.cpp", "src/logsumexp.cpp", "src/alignpath.cpp", "src/model.cpp", "src/fastseq.cpp"],
    "hx_host_recon.cpp": ["src/recon.cpp"],
    "hx_host_span.cpp": ["src/span.cpp", "src/alignpath.cpp"],
    "hx_host_quickalign.cpp": ["src/quickalign.cpp", "src/diagenv.cpp"],
    "t/testforward.cpp": ["t/testforward.cpp"], "t/testbackward.cpp": ["t/testbackward.cpp"],
    "t/testnullforward.cpp": ["t/testnullforward.cpp"], "t/pair_setup.h": ["t/testforward.cpp", "t/testbackward.cpp", "t/testnullforward.cpp"],
}

def norm_lines(path):
    text = open(path, errors="replace").read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = []
    for line in text.split("\n"):
        line = re.sub(r"//.*$", "", line)
        line = re.sub(r"\s+", "", line)
        if len(line) >= 8 and not line.startswith("#include"):
            out.append(line)
    return out

for f, refs in MIRRORS.items():
    p = os.path.join(HOST, f)
    if not os.path.exists(p):
        continue
    ref_lines = set()
    for r in refs:
        rp = os.path.join(REF, r)
        if os.path.exists(rp):
            ref_lines.update(norm_lines(rp))
    mine = norm_lines(p)
    shared = [l in ref_lines for l in mine]
    runs, cur = 0, 0
    for s in shared + [False]:
        if s: cur += 1
        else:
            if cur >= 6: runs += cur
            cur = 0
    print("%-26s %4d lines  shared %5.1f %%   in runs >= 6: %d" % (f, len(mine), 100.0 * sum(shared) / max(1, len(mine)), runs))
