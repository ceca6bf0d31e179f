#!/bin/bash
O=gpurun_out/r04v8
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "upload_paths or mirror_packed or collective" > $O/pytest.log 2>&1
grep -n "passed\|failed\|error" $O/pytest.log | tail -3
for nbk in 0 4 8 12 16 24 32; do echo "FSNAP_CHOL_NBK=$nbk: $(FSNAP_CHOL_NBK=$nbk timeout 120 python scripts/host_solve_bench.py 2>&1 | tail -1)"; done | tee $O/host_solve_nbk.txt
echo "variant 1 (unblocked): $(FSNAP_CHOL_VARIANT=1 timeout 120 python scripts/host_solve_bench.py 2>&1 | tail -1)" | tee -a $O/host_solve_nbk.txt
lscpu | grep -i "model name" | tee -a $O/host_solve_nbk.txt
