#!/bin/bash
# Kernel trace and shader counters of tools/sumprod_bench.py (counts-mode column kernels).  Run on the GPU box from the repo root:
#   bash tools/pmc_sumprod.sh [columns]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
mkdir -p $R/gpurun_out
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sp_trace -- python3 $R/tools/sumprod_bench.py "$@" > $R/gpurun_out/sp_trace.log 2>&1 || { echo "trace failed"; tail -3 $R/gpurun_out/sp_trace.log; exit 1; }
f=$(find $R/gpurun_out/sp_trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -c1-160 "$f" | head -8
i=0
for set in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_BUSY_avr TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 5 200 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/sp_pmc_$i -- python3 $R/tools/sumprod_bench.py "$@" > $R/gpurun_out/sp_pmc_$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/gpurun_out/sp_pmc_$i.log; exit 1; }
  f=$(find $R/gpurun_out/sp_pmc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0][-60:]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
for k, d in acc.items():
    if "sumprod" in k or "outer" in k or "row_sums" in k:
        print(k, {c: "%.4g" % v for c, v in d.items()})
PY
done
