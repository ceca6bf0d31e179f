#!/bin/bash
mkdir -p gpurun_out/r05_y
timeout 600 python bench.py > gpurun_out/r05_y/bench.json 2> gpurun_out/r05_y/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_y/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
print(json.dumps(d['cpu_baseline'])[:700])
PY
