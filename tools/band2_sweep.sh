#!/bin/bash
# banded fill, two pairs per wavefront (hx_band2.hip) against one (hx_band.hip): pairs x wavefronts per workgroup x policy
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R" || exit 1; mkdir -p gpurun_out
OUT=gpurun_out/band2_sweep.txt; : > $OUT
for pairs in ${PAIRS:-512 2048 2560 4096}; do
  for mode in ${MODES:-trunc linear}; do
    for cfg in "0 0" "1 1" "1 2" "1 4"; do
      set -- $cfg
      line=$(HX_BAND2=$1 HX_BAND2_NW=$2 python bench.py --band 20 --pairs $pairs --mode $mode --single-mode --no-cpu-baseline --steps 5 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('%.3f ms  %.1f Gcell/s  frac %.3f' % (d['roofline']['kernel_ms'], d['value']/1e9, d['roofline']['frac']))") || exit 1
      echo "pairs $pairs mode $mode band2 $1 nw $2: $line" | tee -a $OUT
    done
  done
done
