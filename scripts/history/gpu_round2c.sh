#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/2c
O=$PWD/gpurun_out/2c
run() {
  timeout 100 python bench.py --no-cpu-baseline --rows $1 --cols $2 --steps 30 --warmup 3 --preheat 150 > $O/b.json 2>> $O/bench.err
  python - $O/b.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
r=d['roofline']
print(d['config']['rows_per_gpu'],d['config']['K'],r['kernel'],'step ms %.3f'%d['ms_per_step'],'kernel ms %.4f'%r['kernel_ms_avg'],'frac %.3f'%r['frac'],'GB/s %.0f'%r['achieved_GBps_algorithmic'])
PY
}
run 1772880 110
run 1772880 112
run 1772880 104
run 1772880 96
run 1000000 128
run 1000000 126
run 1000000 56
run 1000000 64
run 1000000 31
