#!/bin/bash
mkdir -p gpurun_out/r05_u
for v in auto pageable staged; do echo "FSNAP_H2D_SMALL=$v"; if [ $v = auto ]; then timeout 300 python scripts/class_overhead.py 2>&1 | grep "ms$\| ms" | grep -v "function\|tottime" | head -5; else FSNAP_H2D_SMALL=$v timeout 300 python scripts/class_overhead.py 2>&1 | grep " ms" | grep -v "function\|tottime" | head -5; fi; done > gpurun_out/r05_u/class_overhead_h2d.txt 2>&1; cat gpurun_out/r05_u/class_overhead_h2d.txt
timeout 600 python scripts/class_fit_survey.py 1772880x110 1000000x128 367900x480 > gpurun_out/r05_u/class_fit_survey.txt 2>&1; grep -v amdgpu gpurun_out/r05_u/class_fit_survey.txt
timeout 300 python scripts/ga_loop_timing.py 2>&1 | grep "1000000 x 128" | cut -c1-200 > gpurun_out/r05_u/ga_loop.txt; cat gpurun_out/r05_u/ga_loop.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05_u/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_u/pytest.txt
