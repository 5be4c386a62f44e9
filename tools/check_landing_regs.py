"""Check of the hand-issued row-record fetches in a compiled sweep kernel (hx_band2.hip, hx_band.hip).

The kernels fetch a row record with an inline-asm `global_load_dwordx2` into a "landing" register pair that the compiler
only ever sees read, and take the value out behind a counted `s_waitcnt vmcnt(N)`.  That is correct as long as the compiler
keeps the pair where it is and never touches it itself.  This script reads the assembly listing (hipcc -S) and checks, for
every kernel that announces its landing registers ("; landing registers of the row-record fetches: v[a:b]"):
  * every hand-issued fetch writes exactly that pair;
  * between the announcement and the first drain behind the last fetch (a stand-alone inline `s_waitcnt vmcnt(0)`) nothing
    else names either register, except the v_mov_b32 pair directly behind an s_waitcnt inside the same inline-asm block.
usage: python tools/check_landing_regs.py file.s   (exit code 1 and a message per violation)"""
import re
import sys


def regs_in(line):
    """vector registers named by an instruction line: set of ints"""
    code = line.split(";")[0]
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", code):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", code):
        out.add(int(m.group(1)))
    return out


def check(path):
    lines = open(path).read().split("\n")
    problems, kernels = [], 0
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for n, a in enumerate(starts):
        b = next((i for i in range(a, len(lines)) if lines[i].strip().startswith("s_endpgm")), len(lines) - 1)
        body = lines[a:b + 1]
        ann = [re.search(r"landing registers of the row-record fetches: v\[(\d+):(\d+)\]", l) for l in body]
        ann = [m for m in ann if m]
        if not ann:
            continue
        kernels += 1
        name = lines[a].rstrip(":")[:90]
        pairs = {(int(m.group(1)), int(m.group(2))) for m in ann}
        if len(pairs) != 1:
            problems.append("%s: several landing pairs %s" % (name, sorted(pairs)))
            continue
        lo, hi = pairs.pop()
        land = {lo, hi}
        in_asm, waited = False, False
        fetches = takes = 0
        # the pair is reserved from its announcement (in front of the first fetch) to the first drain behind the last fetch - a
        # stand-alone inline `s_waitcnt vmcnt(0)`; in front of and behind that stretch (staging, another wavefront's code, lpEnd)
        # the registers are anybody's
        first = next(k for k, l in enumerate(body) if "landing registers of the row-record fetches" in l)
        drains = [k for k in range(first, len(body) - 2) if body[k].strip().startswith(";;#ASMSTART")
                  and body[k + 1].strip() == "s_waitcnt vmcnt(0)" and body[k + 2].strip().startswith(";;#ASMEND")]
        last_fetch = max((k for k, l in enumerate(body) if l.strip().startswith("global_load_dwordx2 v[%d:%d]," % (lo, hi))), default=-1)
        drains = [k for k in drains if k > last_fetch]
        if not drains or last_fetch < 0:
            problems.append("%s: no fetch, or no drain behind the last one" % name)
            continue
        for k, l in enumerate(body):
            if k < first or k > drains[0]:
                continue
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm, waited = True, False
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t.startswith((";", ".")) or re.match(r"^[\w.$]+:", t):
                continue
            used = regs_in(t) & land
            if in_asm and t.startswith("s_waitcnt vmcnt("):
                waited = True
                continue
            if not used:
                continue
            if in_asm and t.startswith("global_load_dwordx2 v[%d:%d]," % (lo, hi)):
                fetches += 1
                continue
            if in_asm and waited and re.match(r"v_mov_b32 v\d+, v(%d|%d)$" % (lo, hi), t):
                takes += 1
                continue
            problems.append("%s: line %d touches the landing registers v[%d:%d]: %s" % (name, a + k + 1, lo, hi, t))
        if fetches == 0 or takes == 0:
            problems.append("%s: %d fetches, %d reads behind a wait" % (name, fetches, takes))
    return kernels, problems


if __name__ == "__main__":
    kernels, problems = check(sys.argv[1])
    for p in problems:
        print(p)
    print("%d kernel(s) with landing registers checked, %d problem(s)" % (kernels, len(problems)))
    sys.exit(1 if problems or kernels == 0 else 0)
