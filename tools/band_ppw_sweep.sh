#!/bin/bash
# usage: tools/band_ppw_sweep.sh <pairs>  -- bench.py --band 20 over the pairs-per-workgroup choices of the band kernel
pairs=$1
for m in ${MODES:-linear fast}; do
  for p in ${PPWS:-2 -3 -4 -5 -6}; do
    r=$(HX_BAND_PPW=$p timeout -k 10 120 python bench.py --band 20 --pairs $pairs --mode $m --no-cpu-baseline --single-mode --steps 3 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f Gcell/s kernel %.2f ms' % (d['value']/1e9, d['roofline']['kernel_ms']))" 2>&1 | tail -1)
    echo "$m pairs $pairs ppw $p: $r"
  done
done
