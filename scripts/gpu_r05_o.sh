#!/bin/bash
mkdir -p gpurun_out/r05_o
FSNAP_ROWSPACE_TIMING=1 timeout 900 python scripts/rowspace_large_k.py > gpurun_out/r05_o/rowspace_large_k.txt 2>&1; grep -v "^\[fsnap_lstsq_rows\] \(stat\|factor\)" gpurun_out/r05_o/rowspace_large_k.txt | tail -60
timeout 900 python -m pytest tests/test_gpu_rowspace.py tests/test_gpu_configs.py -x -q > gpurun_out/r05_o/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_o/pytest.txt
