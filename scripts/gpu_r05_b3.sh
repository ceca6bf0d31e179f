#!/bin/bash
mkdir -p gpurun_out/r05_b3
for i in 1 2 3 4; do
  timeout 300 python bench.py --no-cpu-baseline --svd-solver 0 > gpurun_out/r05_b3/bench_$i.json 2>> gpurun_out/r05_b3/bench.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r05_b3/bench_$i.json').read().strip().splitlines()[-1])
print($i, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_avg'], d.get('pipelined',{}).get('ms_per_step'))
PY
done
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_b3/bench_driver.json 2>> gpurun_out/r05_b3/bench.err
python -c "
import json
d=json.loads(open('gpurun_out/r05_b3/bench_driver.json').read().strip().splitlines()[-1])
print('driver flags', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_avg'])"
