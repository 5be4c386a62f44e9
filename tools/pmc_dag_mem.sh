#!/bin/bash
# usage: tools/pmc_dag_mem.sh <tag> <kernel-name-substring>  -- vector-memory pipeline counters of a general-profile Forward
# fill on tools/dag_bench.py 32 (unbanded batch).  One small counter set per pass.
tag=$1; kern=$2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd "$R" || exit 1
mkdir -p gpurun_out
i=0
for set in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 5 240 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmcm_${tag}_$i -- python tools/dag_bench.py 32 fwdonly unbanded > gpurun_out/pmcm_${tag}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/pmcm_${tag}_$i.log; exit 1; }
done
python - <<PY
import csv,glob,collections
for i in range(1,5):
    for f in glob.glob("gpurun_out/pmcm_${tag}_%d/*/*counter_collection.csv"%i):
        agg=collections.defaultdict(float); n=collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if "$kern" in r["Kernel_Name"]:
                agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
        for k,v in sorted(agg.items()): print("%-36s %.5g  (%d launches)"%(k,v/n[k],n[k]))
PY
