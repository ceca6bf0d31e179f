#!/bin/bash
mkdir -p gpurun_out/r05_ad
FSNAP_ROWSPACE_TIMING=1 timeout 600 python scripts/rowspace_large_k.py 15213 1595 6 2>&1 | grep "call 2\|certified\|estimators" | tail -6 | cut -c1-100
timeout 900 python -m pytest tests/test_gpu_rowspace.py tests/test_gpu_configs.py -x -q > gpurun_out/r05_ad/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r05_ad/pytest.txt
