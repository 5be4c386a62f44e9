#!/bin/bash
# usage: tools/pmc_fast.sh <tag>  -- shader-side counters of the headline fill (k_fill_chain<..FastLse..>, python bench.py)
tag=$1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd "$R" || exit 1
mkdir -p gpurun_out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 5 240 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmcf_${tag}_$i -- python bench.py --no-cpu-baseline --single-mode --steps 1 --warmup 0 > gpurun_out/pmcf_${tag}_$i.log 2>&1 || { echo "pass $i failed"; tail -2 gpurun_out/pmcf_${tag}_$i.log; exit 1; }
done
python - <<PY
import csv,glob,collections
for i in range(1,5):
    for f in glob.glob("gpurun_out/pmcf_${tag}_%d/*/*counter_collection.csv"%i):
        agg=collections.defaultdict(float); n=collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if "k_fill_chain" in r["Kernel_Name"] and "FastLse" in r["Kernel_Name"]:
                agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
        for k,v in sorted(agg.items()): print("%-28s %.5g  (%d launches)"%(k,v/n[k],n[k]))
PY
