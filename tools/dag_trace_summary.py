"""Per-launch averages of the phase lines a -DHX_DAG_TRACE build of the general-profile Forward pipeline prints (cycles per step).
usage: python tools/dag_trace_summary.py file"""
import collections
import re
import sys

rows = []
for line in open(sys.argv[1]):
    m = re.search(r"trace job (\d+) strip (\d+) steps (\d+) wait (\d+) issue (\d+) loads (\d+) accumulate (\d+) sums (\d+) rotate (\d+)", line)
    if m:
        rows.append(tuple(int(x) for x in m.groups()))
by = collections.defaultdict(list)
for r in rows:
    by[r[2]].append(r)
for steps, rs in sorted(by.items()):
    n = len(rs)
    tot = [sum(r[k] for r in rs) / n for k in range(3, 9)]
    worst = max(rs, key=lambda r: sum(r[4:9]))
    print("steps %5d strips %3d  avg cycles: wait %5d issue %5d loads %5d accumulate %5d sums %5d rotate %5d | own work %5d (%.2f us), slowest strip %d: %s = %d" %
          (steps, n, *tot, sum(tot[1:]), sum(tot[1:]) / 2400, worst[1], worst[4:9], sum(worst[4:9])))
