#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/2d
O=$PWD/gpurun_out/2d
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
run() {
  timeout 100 python bench.py --no-cpu-baseline --rows $1 --cols $2 --steps 30 --warmup 3 --preheat 150 $3 > $O/b.json 2>> $O/bench.err
  python - $O/b.json "$3" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
r=d['roofline']
print(d['config']['rows_per_gpu'],d['config']['K'],r['kernel'],sys.argv[2],'wgs',d['config']['launch']['workgroups'],'step ms %.3f'%d['ms_per_step'],'kernel ms %.4f'%r['kernel_ms_avg'],'frac %.3f'%r['frac'],'GB/s %.0f'%r['achieved_GBps_algorithmic'])
PY
}
for k in 31 56 64 80; do
  run 1000000 $k ""
  run 1000000 $k "--option kernel=1"
done
run 1000000 31 "--option nblocks=1792"
run 1000000 31 "--option nblocks=2560"
run 1000000 64 "--option nblocks=1024"
run 1000000 64 "--option nblocks=1536"
run 1000000 16 ""
