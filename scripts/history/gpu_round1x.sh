#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/x
O=gpurun_out/x
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "large_k or solve_device or device_solve or weight or predict or residual or error" > $O/pytest_sub.log 2>&1
tail -3 $O/pytest_sub.log
for nb in 2048 8192 16384 62500; do FSNAP_GEMV_BLOCKS=$nb timeout 100 python scripts/gemv_grid_test.py 2>&1 | grep GEMV; done | tee $O/gemv_grid.log
timeout 200 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/x/bench.json'))
print('value',d['value'],'ms',d['ms_per_step'],'kernel',d['roofline']['kernel_ms_avg'],'weighting',d['weighting_kernel'])
PY
timeout 100 python bench.py --no-cpu-baseline --rows 15213 --cols 1595 --steps 20 --warmup 3 > $O/bench_k1595.json 2>> $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/x/bench_k1595.json'))
print('K1595 value',d['value'],'ms',d['ms_per_step'],'kernel',d['roofline']['kernel_ms_avg'],'frac',d['roofline']['frac'])
PY
