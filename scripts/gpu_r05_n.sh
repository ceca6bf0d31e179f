#!/bin/bash
mkdir -p gpurun_out/r05_n
FSNAP_ROWSPACE_TIMING=1 timeout 900 python scripts/rowspace_large_k.py > gpurun_out/r05_n/rowspace_large_k.txt 2>&1; tail -70 gpurun_out/r05_n/rowspace_large_k.txt
