#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/2b
O=$PWD/gpurun_out/2b
run() {
  timeout 100 python bench.py --no-cpu-baseline --rows $1 --cols $2 --steps 20 --warmup 3 --preheat 60 --option nsplit=$3 > $O/b.json 2>> $O/bench.err
  python - $O/b.json $3 <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(d['config']['rows_per_gpu'],d['config']['K'],'nsplit',sys.argv[2],'->',d['config']['launch'].get('nsplit'),'step ms %.3f'%d['ms_per_step'],'kernel ms %.4f'%d['roofline']['kernel_ms_avg'],'frac %.3f'%d['roofline']['frac'], 'reduce %.3f'%d['roofline']['reduce_kernel_ms_avg'])
PY
}
for n in 0 14 28 29 43 57 86; do run 367900 480 $n; done
for n in 0 3 5 6 8 11 16; do run 15213 1595 $n; done
for n in 0 8 16 32 64; do run 200000 256 $n; done
