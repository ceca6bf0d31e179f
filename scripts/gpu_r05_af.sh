#!/bin/bash
mkdir -p gpurun_out/r05_af
for shape in "13035 256" "100000 168" "100000 272" "40000 192"; do set -- $shape
 for g in 12 24 48 96; do
  timeout 200 python bench.py --rows $1 --cols $2 --steps 30 --warmup 3 --preheat 150 --no-cpu-baseline --svd-solver 0 --pipelined 0 --option quad_min_cpg=$g > gpurun_out/r05_af/b.json 2>> gpurun_out/r05_af/err.txt
  python - <<PY
import json
d=json.loads(open('gpurun_out/r05_af/b.json').read().strip().splitlines()[-1])
r=d['roofline']
print("$1 x $2 quad_min_cpg=$g: fit %.4f ms  kernel %.4f ms (%.2f)  reduce %.4f ms" % (d['ms_per_step'], r['kernel_ms_avg'], r['frac'], r.get('reduce_kernel_ms_avg',0)))
PY
 done
done | tee gpurun_out/r05_af/quad_min_cpg.txt
