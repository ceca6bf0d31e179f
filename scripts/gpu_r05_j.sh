#!/bin/bash
mkdir -p gpurun_out/r05_j
timeout 900 python -m pytest tests/test_gpu_rowspace.py -x -q > gpurun_out/r05_j/pytest_rowspace.txt 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_j/pytest_rowspace.txt
FSNAP_ROWSPACE_TIMING=1 timeout 300 python scripts/rowspace_calls.py > gpurun_out/r05_j/rowspace_phases.txt 2>&1; tail -24 gpurun_out/r05_j/rowspace_phases.txt
timeout 300 python scripts/class_overhead.py > gpurun_out/r05_j/class_overhead.txt 2>&1; cat gpurun_out/r05_j/class_overhead.txt | head -60
timeout 600 python bench.py > gpurun_out/r05_j/bench.json 2> gpurun_out/r05_j/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_j/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
s=d.get('svd_solver',{})
print({k:(v.get('ms_per_fit') if isinstance(v,dict) else v) for k,v in s.items()}, s.get('row_space',{}).get('info'))
PY
