#!/bin/bash
O=gpurun_out/r04v6
mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
grep -n "passed\|failed\|error" $O/pytest.log | tail -5
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("default", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["traffic"], "h2d", d["h2d_upload_ms"], d["h2d_upload_path"], d["h2d_inclusive_rows_per_s"])
print(d["svd_solver"]["steps"]["ms_per_fit"], d["svd_solver"]["steps"]["residual_rhs_ms_per_call"], d["svd_solver"]["class_perform_fit"]["ms_per_fit"], d["svd_solver"]["row_space"]["ms_per_fit"])
PY
for nb in 512 1024; do
FSNAP_RESIDUAL_BLOCKS=$nb timeout 300 python scripts/residual_probe.py 2>&1 | grep "fused_residual=1" | sed "s/^/blocks=$nb /"
done
