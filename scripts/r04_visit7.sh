#!/bin/bash
O=gpurun_out/r04v7
mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
grep -n "passed\|failed\|error" $O/pytest.log | tail -5
bash scripts/pmc_record_r04.sh 2>&1 | tail -40
timeout 300 python scripts/residual_probe.py 2>&1 | grep "fused_residual=1"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("default", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["traffic"], d["svd_solver"]["steps"]["ms_per_fit"], d["svd_solver"]["class_perform_fit"]["ms_per_fit"])
PY
