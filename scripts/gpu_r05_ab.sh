#!/bin/bash
mkdir -p gpurun_out/r05_ab
timeout 300 python scripts/refine_solve_timing.py 2>&1 | grep "K =" | tee gpurun_out/r05_ab/refine_solve_timing.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_assembly.py tests/test_gpu_transpose.py -x -q -k "factor_is_reused or accumulate or transpose or fused" > gpurun_out/r05_ab/pytest.txt 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r05_ab/pytest.txt
