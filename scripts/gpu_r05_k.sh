#!/bin/bash
mkdir -p gpurun_out/r05_k
for n in 1 2 3 4; do echo "FSNAP_STAGE_THREADS=$n"; FSNAP_STAGE_THREADS=$n timeout 300 python scripts/class_overhead.py 2>&1 | grep -v "^ \|^$\|ncalls\|Ordered\|List\|function calls" | head -8; done > gpurun_out/r05_k/class_overhead_ab.txt 2>&1
cat gpurun_out/r05_k/class_overhead_ab.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r05_k/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_k/pytest.txt
timeout 300 python scripts/ga_loop_timing.py > gpurun_out/r05_k/ga_loop.txt 2>&1; cat gpurun_out/r05_k/ga_loop.txt
