#!/bin/bash
# full GPU suite + smoke + the default bench line on the final tree
mkdir -p gpurun_out/r05_i
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05_i/pytest.txt 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r05_i/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05_i/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r05_i/smoke.txt
timeout 600 python bench.py > gpurun_out/r05_i/bench.json 2> gpurun_out/r05_i/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_i/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('svd_solver'))
PY
