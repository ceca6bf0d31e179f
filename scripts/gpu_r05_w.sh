#!/bin/bash
mkdir -p gpurun_out/r05_w
for g in 1e5 3e5 1e6 3e6; do
  for shape in "367900 480 4" "15213 1595 6"; do set -- $shape
    echo "== grain $g, $1 x $2"
    FSNAP_HOST_GRAIN=$g FSNAP_ROWSPACE_TIMING=1 timeout 600 python scripts/rowspace_large_k.py $1 $2 $3 2>&1 | grep "call 2\|certified\|prepare (all)\|deflate  \|product" | awk '/call 1/{f=1} 1' | tail -7 | cut -c1-100
  done
done > gpurun_out/r05_w/grain.txt 2>&1
cat gpurun_out/r05_w/grain.txt
