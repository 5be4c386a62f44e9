#!/bin/bash
# usage: tools/pmc_quick.sh <tag> <kernel-name-substring> <counter sets, ';'-separated> <python script and its arguments>
# One rocprofv3 --pmc pass per counter set (never combined with trace domains); prints the per-launch mean of every counter
# over the launches of the matching kernel.
tag=$1; kern=$2; sets=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd "$R" || exit 1
mkdir -p gpurun_out
i=0
IFS=';' read -ra SETS <<< "$sets"
for set in "${SETS[@]}"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcq_${tag}_$i
  timeout -k 5 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmcq_${tag}_$i -- python "$@" > gpurun_out/pmcq_${tag}_$i.log 2>&1 || { echo "pass $i ($set) failed"; grep -m3 -i "error\|exceeds\|invalid" gpurun_out/pmcq_${tag}_$i.log; continue; }
  python - "$kern" gpurun_out/pmcq_${tag}_$i <<'PY'
import csv, glob, collections, sys
kern, d = sys.argv[1], sys.argv[2]
for f in glob.glob(d + "/*/*counter_collection.csv"):
    agg = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k, v in sorted(agg.items()):
        print("%-32s %.5g  (%d launches)" % (k, v / n[k], n[k]))
PY
done
