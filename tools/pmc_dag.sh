#!/bin/bash
# usage: tools/pmc_dag.sh <tag>   -- PMC passes for the general-profile (DAG) Forward pipeline on tools/dag_bench.py
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --output-format csv -d gpurun_out/pmcd_${tag}_1 -- python tools/dag_bench.py 32 fwdonly > gpurun_out/pmcd_${tag}_1.log 2>&1
timeout -k 10 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmcd_${tag}_2 -- python tools/dag_bench.py 32 fwdonly > gpurun_out/pmcd_${tag}_2.log 2>&1
python - <<PY
import csv,glob,collections
for d in ("pmcd_${tag}_1","pmcd_${tag}_2"):
    for f in glob.glob("gpurun_out/%s/*/*counter_collection.csv"%d):
        agg=collections.defaultdict(lambda: collections.defaultdict(float))
        cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            if "dag_pipe" in r["Kernel_Name"]:
                key=("fast" if "FastLse" in r["Kernel_Name"] else "exact")
                agg[key][r["Counter_Name"]]+=float(r["Counter_Value"])
        for key in agg:
            for k,v in sorted(agg[key].items()): print("%-6s %-24s %.4g"%(key,k,v))
PY
