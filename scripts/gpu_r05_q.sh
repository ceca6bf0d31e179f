#!/bin/bash
mkdir -p gpurun_out/r05_q
FSNAP_ROWSPACE_TIMING=1 timeout 900 python scripts/rowspace_large_k.py > gpurun_out/r05_q/rowspace_large_k.txt 2>&1; grep "call\|lstsq on" gpurun_out/r05_q/rowspace_large_k.txt | cut -c1-110
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05_q/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_q/pytest.txt
