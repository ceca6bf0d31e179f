#!/bin/bash
mkdir -p gpurun_out/r05_r
FSNAP_ROWSPACE_TIMING=1 timeout 900 python scripts/rowspace_large_k.py > gpurun_out/r05_r/rowspace_large_k.txt 2>&1; grep "call\|lstsq on\|certified\|estimators\|solve + ref" gpurun_out/r05_r/rowspace_large_k.txt | cut -c1-110 | tail -24
timeout 900 python -m pytest tests/test_gpu_rowspace.py tests/test_gpu_configs.py -x -q > gpurun_out/r05_r/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_r/pytest.txt
