#!/bin/bash
# Round profile of the default bench: kernel-trace stats + HBM traffic counters (separate PMC passes,
# as /opt/skills/guides/MI355X_MICROARCH.md prescribes).  Run on the GPU box; copy summaries to profiles/.
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd "$R" || exit 1
mkdir -p gpurun_out/$tag
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/trace -- python bench.py --steps 3 --warmup 1 > gpurun_out/$tag/bench.log 2>&1
cp $(find gpurun_out/$tag/trace -name "*kernel_stats.csv" | head -1) gpurun_out/$tag/kernel_stats.csv
grep '^{' gpurun_out/$tag/bench.log > gpurun_out/$tag/bench.json
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d gpurun_out/$tag/pmc_$c -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --banded-pairs '' > gpurun_out/$tag/pmc_$c.log 2>&1
done
python - <<PY
import csv,glob,collections,json,re
res={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    for f in glob.glob("gpurun_out/$tag/pmc_%s/*/*counter_collection.csv"%c):
        agg=collections.defaultdict(lambda: [0.0,0])
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"]!=c: continue
            m=re.search(r"k_fill_leaf_linear<([^>]*)>",r["Kernel_Name"])
            if m: key="trunc" if m.group(1).replace(" ","").endswith("true") else "linear"
            elif "k_fill_chain<0" in r["Kernel_Name"]: key="fast" if "FastLse" in r["Kernel_Name"] else "exact"
            else: continue
            agg[key][0]+=float(r["Counter_Value"]); agg[key][1]+=1
        for k,(v,n) in agg.items(): res[c+":"+k]=v/n
print(json.dumps(res))
open("gpurun_out/$tag/pmc_summary.json","w").write(json.dumps(res,indent=1))
PY
# the box itself: achievable HBM write bandwidth for the fills' store pattern, clocks (the pool's boxes differ by up to 25 %
# on the write-bound linear kernel while compute-bound kernels agree to 2 %)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o /tmp/ubench_store tools/ubench_store.hip 2>/dev/null && timeout -k 5 100 /tmp/ubench_store > gpurun_out/$tag/box_store_bandwidth.txt 2>&1
(rocm-smi --showclocks --showmemuse 2>/dev/null | grep -v "^$" | head -30) >> gpurun_out/$tag/box_store_bandwidth.txt
cat gpurun_out/$tag/kernel_stats.csv; cat gpurun_out/$tag/bench.json
