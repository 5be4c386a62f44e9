"""Instruction mix of the loops of one kernel in an AMDGPU assembly listing.
usage: python tools/isa_loop.py file.s <kernel-name-substring>
Prints, for every backward branch (a loop), the instruction counts by class between its target label and the branch."""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r"^[A-Za-z_]\w*:", l) and want in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end + 1]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\w+):", l)
    if m:
        labels[m.group(1)] = i


def klass(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


for i, l in enumerate(body):
    m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\w+)", l) or re.match(r"\s+s_branch\s+(\.LBB\w+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        seg = body[labels[m.group(1)]:i + 1]
        cnt = {}
        ops = {}
        for s in seg:
            mm = re.match(r"\s+([a-z_0-9]+)", s)
            if mm and not s.strip().startswith((".", ";")):
                k = klass(mm.group(1))
                cnt[k] = cnt.get(k, 0) + 1
                ops[mm.group(1)] = ops.get(mm.group(1), 0) + 1
        tot = sum(cnt.values())
        if tot > 200:
            print("loop %s .. line %d: %d instructions %s" % (m.group(1), i, tot, cnt))
            top = sorted(ops.items(), key=lambda kv: -kv[1])[:28]
            print("   " + ", ".join("%s %d" % kv for kv in top))
