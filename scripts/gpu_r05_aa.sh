#!/bin/bash
mkdir -p gpurun_out/r05_aa
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "factor_is_reused or large_k_device or forms" > gpurun_out/r05_aa/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r05_aa/pytest.txt
timeout 300 python scripts/refine_solve_timing.py > gpurun_out/r05_aa/refine_solve_timing.txt 2>&1; grep "K =" gpurun_out/r05_aa/refine_solve_timing.txt
timeout 300 python scripts/class_fit_survey.py 367900x480 15213x1595 2>&1 | grep " x " 
