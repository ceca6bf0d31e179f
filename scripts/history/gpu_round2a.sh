#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/2a
O=$PWD/gpurun_out/2a
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tiled or general_k or large_k or k142 or svd_fit_large or streaming or accumul or full_size or additiv" > $O/pytest_tiled.log 2>&1
tail -4 $O/pytest_tiled.log
for shape in "15213 1595" "367900 480" "13035 142" "200000 256"; do
  set -- $shape
  timeout 100 python bench.py --no-cpu-baseline --rows $1 --cols $2 --steps 20 --warmup 3 --preheat 60 > $O/bench_$1x$2.json 2>> $O/bench.err
  python - $O/bench_$1x$2.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(d['config']['rows_per_gpu'],d['config']['K'],'step ms %.3f'%d['ms_per_step'],'kernel ms %.4f'%d['roofline']['kernel_ms_avg'],'frac %.3f'%d['roofline']['frac'], 'reduce %.3f'%d['roofline']['reduce_kernel_ms_avg'])
PY
done
