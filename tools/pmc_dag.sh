#!/bin/bash
# usage: tools/pmc_dag.sh <tag> <kernel-name-substring>  -- shader-side counters of a general-profile Forward fill on
# tools/dag_bench.py 32 (unbanded batch only).  Few counters per pass; PMC never combined with other trace domains.
tag=$1; kern=$2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd "$R" || exit 1
mkdir -p gpurun_out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout -k 5 240 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmcd_${tag}_$i -- python tools/dag_bench.py 32 ${DAG_BENCH_ARGS:-fwdonly unbanded} > gpurun_out/pmcd_${tag}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/pmcd_${tag}_$i.log; exit 1; }
done
python - <<PY
import csv,glob,collections
for i in range(1,5):
    for f in glob.glob("gpurun_out/pmcd_${tag}_%d/*/*counter_collection.csv"%i):
        agg=collections.defaultdict(float); n=collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if "$kern" in r["Kernel_Name"]:
                agg[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
        for k,v in sorted(agg.items()): print("%-28s %.5g  (%d launches)"%(k,v/n[k],n[k]))
PY
