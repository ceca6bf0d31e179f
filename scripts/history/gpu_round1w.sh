#!/bin/bash
# large-K device Cholesky rework + weighting-kernel variants
export TMPDIR=/tmp
mkdir -p gpurun_out/w
O=gpurun_out/w
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "large_k or solve_device or device_solve" > $O/pytest_chol.log 2>&1
tail -5 $O/pytest_chol.log
timeout 200 python scripts/chol_large_test.py > $O/chol_large.log 2>&1
cat $O/chol_large.log
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/weight_rows_variants.hip -o /tmp/wrv 2> $O/wrv_build.log && timeout 200 /tmp/wrv > $O/wrv.log 2>&1
cat $O/wrv.log
