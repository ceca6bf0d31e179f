#!/bin/bash
mkdir -p gpurun_out/r05_l
O=gpurun_out/r05_l
for n in 64 128 200 256 480 1024 1595; do for v in 1 0; do echo "== REDUNDANT=$v n=$n"; timeout 120 tools/bin/chol_pipeline_check_r$v $n 2>&1 | grep "form"; done; done > $O/pipeline_ab.txt 2>&1
cat $O/pipeline_ab.txt
for v in 1 0; do echo "== trace REDUNDANT=$v"; timeout 120 tools/bin/chol_pipeline_check_trace_r$v 1595 2>&1 | tail -8; done > $O/trace.txt 2>&1
cat $O/trace.txt
timeout 300 python scripts/chol_large_test.py 192 256 288 384 480 512 768 1024 1595 2048 > $O/chol_large_k_sweep.txt 2>&1; cat $O/chol_large_k_sweep.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "chol or cholesky or large_k or device_solve or forms" > $O/pytest_chol.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_chol.txt
