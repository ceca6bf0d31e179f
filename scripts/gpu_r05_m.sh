#!/bin/bash
mkdir -p gpurun_out/r05_m
timeout 900 python -m pytest tests/test_gpu_rowspace.py -x -q > gpurun_out/r05_m/pytest_rowspace.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_m/pytest_rowspace.txt
timeout 300 python scripts/error_analysis_profile.py > gpurun_out/r05_m/error_analysis_profile.txt 2>&1; head -50 gpurun_out/r05_m/error_analysis_profile.txt
