#!/bin/bash
# Profile of the guide-alignment Viterbi batch (tools/quickalign_bench.py): kernel-trace stats.
tag=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd "$R" || exit 1
mkdir -p gpurun_out
mkdir -p gpurun_out/$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/qatrace -- python tools/quickalign_bench.py 512 2000 3 > gpurun_out/$tag/quickalign_bench.log 2>&1
cp $(find gpurun_out/$tag/qatrace -name "*kernel_stats.csv" | head -1) gpurun_out/$tag/quickalign_kernel_stats.csv
grep '^{' gpurun_out/$tag/quickalign_bench.log > gpurun_out/$tag/quickalign_bench.json
cat gpurun_out/$tag/quickalign_kernel_stats.csv; cat gpurun_out/$tag/quickalign_bench.json
