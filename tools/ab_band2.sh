#!/bin/bash
# quick same-box timings of the banded leaf fill (tools/band_quick.py): batch size x wavefronts per workgroup x policy
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R" || exit 1
for pairs in ${PAIRS:-4096 2048 1024 512}; do
  for mode in ${MODES:-trunc linear}; do
    for cfg in "0 0" "1 1" "1 2" "1 4"; do
      set -- $cfg
      HX_BENCH_NOCHECK=1 HX_BAND2=$1 HX_BAND2_NW=$2 timeout -k 5 120 python tools/band_quick.py $pairs $mode 20 2000 5 | cut -c1-150 || exit 1
    done
  done
done
