#!/bin/bash
# usage: tools/pmc.sh <tag> <bench args...>   -- collects two PMC passes for the forward kernel
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd "$R" || exit 1
mkdir -p gpurun_out
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --output-format csv -d gpurun_out/pmc_${tag}_1 -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > gpurun_out/pmc_${tag}_1.log 2>&1
timeout -k 10 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_${tag}_2 -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > gpurun_out/pmc_${tag}_2.log 2>&1
python - <<PY
import csv,glob,collections
for d in ("pmc_${tag}_1","pmc_${tag}_2"):
    for f in glob.glob("gpurun_out/%s/*/*counter_collection.csv"%d):
        agg=collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if "k_fill_chain" in r["Kernel_Name"]:
                agg[r["Counter_Name"]]+=float(r["Counter_Value"])
        for k,v in sorted(agg.items()): print("%-24s %.4g"%(k,v))
PY
grep '^{' gpurun_out/pmc_${tag}_1.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kernel_ms',d['roofline']['kernel_ms'],'cells',d['config']['cells_per_gpu_per_step'])"
