#!/bin/bash
# memory-side counters of the exact-table Forward fill on the headline workload (what bounds it?)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd "$R" || exit 1
mkdir -p gpurun_out
i=0
for set in "TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES" \
           "TA_TA_BUSY TA_TOTAL_WAVEFRONTS GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmcx_$i -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --mode exact --single-mode > gpurun_out/pmcx_$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/pmcx_$i.log; exit 1; }
done
python - <<PY
import csv,glob,collections
for i in (1,2):
    for f in glob.glob("gpurun_out/pmcx_%d/*/*counter_collection.csv"%i):
        agg=collections.defaultdict(float); n=0
        for r in csv.DictReader(open(f)):
            if "k_fill_chain" in r["Kernel_Name"]:
                agg[r["Counter_Name"]]+=float(r["Counter_Value"])
        for k,v in sorted(agg.items()): print("%-36s %.4g"%(k,v))
PY
