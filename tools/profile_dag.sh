#!/bin/bash
# Profile of the general-profile (DAG) fills on the gp120 internal-node profiles (tools/dag_bench.py).
tag=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd "$R" || exit 1
mkdir -p gpurun_out
mkdir -p gpurun_out/$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/dagtrace -- python tools/dag_bench.py 32 > gpurun_out/$tag/dagpipe_bench.txt 2>&1
cp $(find gpurun_out/$tag/dagtrace -name "*kernel_stats.csv" | head -1) gpurun_out/$tag/dagpipe_kernel_stats.csv
cat gpurun_out/$tag/dagpipe_bench.txt | grep -v "^W2\|^E2\|^I2"; cat gpurun_out/$tag/dagpipe_kernel_stats.csv
