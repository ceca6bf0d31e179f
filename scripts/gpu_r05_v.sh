#!/bin/bash
mkdir -p gpurun_out/r05_v
FSNAP_ROWSPACE_TIMING=1 timeout 900 python scripts/rowspace_large_k.py 367900 480 4 > gpurun_out/r05_v/rowspace_367900x480.txt 2>&1
grep "call 2\|lstsq on" gpurun_out/r05_v/rowspace_367900x480.txt | cut -c1-120
awk '/call 1:/{f=1;next} /call 2:/{f=0} f' gpurun_out/r05_v/rowspace_367900x480.txt | head -60
