#!/bin/bash
# usage: tools/pmc_step.sh <tag> <lx> <ly> <jobs>  -- memory-side counters of the general Forward pipeline
# (few counters per pass: the TCP block has 4 counter slots)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd "$R" || exit 1
mkdir -p gpurun_out
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY" \
           "TCP_PENDING_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES TCP_TCC_WRITE_REQ TCP_TCC_WRITE_REQ_LATENCY" \
           "TA_TA_BUSY TA_TOTAL_WAVEFRONTS GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 5 90 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmcs_${tag}_$i -- python tools/dag_step.py "$@" > gpurun_out/pmcs_${tag}_$i.log 2>&1 || { echo "pass $i failed"; grep -m2 -i "error\|exceeds" gpurun_out/pmcs_${tag}_$i.log; exit 1; }
done
python - <<PY
import csv,glob,collections
for i in range(1,4):
    for f in glob.glob("gpurun_out/pmcs_${tag}_%d/*/*counter_collection.csv"%i):
        agg=collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if "dag_pipe" in r["Kernel_Name"]:
                key=("fast" if "FastLse" in r["Kernel_Name"] else "exact")
                agg[key][r["Counter_Name"]]+=float(r["Counter_Value"])
        for k,v in sorted(agg["fast"].items()): print("fast %-40s %.4g"%(k,v))
PY
