#!/bin/bash
mkdir -p gpurun_out/r05_ac
timeout 900 python -m pytest tests/test_gpu_rowspace.py -x -q -k "first_pass_from" > gpurun_out/r05_ac/pytest.txt 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r05_ac/pytest.txt
