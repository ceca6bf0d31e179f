#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/2g
O=$PWD/gpurun_out/2g
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tiled or general_k or large_k or k142 or svd_fit_large or streaming or accumul" > $O/pytest_tiled.log 2>&1
tail -6 $O/pytest_tiled.log
run() {
  timeout 100 python bench.py --no-cpu-baseline --rows $1 --cols $2 --steps 20 --warmup 3 --preheat 60 $3 > $O/b.json 2>> $O/bench.err
  python - $O/b.json "$3" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(d['config']['rows_per_gpu'],d['config']['K'],sys.argv[2],'items',d['config']['launch']['workgroups'],'nsplit',d['config']['launch']['nsplit'],'step ms %.3f'%d['ms_per_step'],'kernel ms %.4f'%d['roofline']['kernel_ms_avg'],'frac %.3f'%d['roofline']['frac'], 'reduce %.3f'%d['roofline']['reduce_kernel_ms_avg'])
PY
}
for shape in "15213 1595" "367900 480" "13035 142" "200000 256" "100000 1000"; do
  set -- $shape
  run $1 $2 ""
  run $1 $2 "--option tiled2=0"
done
