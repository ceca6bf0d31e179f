#!/bin/bash
# round 4, visit 17: full GPU suite + smoke + default bench line after kernel 1Q's barrier went in
O=gpurun_out/r04v17
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; tail -3 $O/gpu_suite.log
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("default: ms/step %.4f value %.4g kernel %s %.4f frac %.3f traffic %s cpu_baseline %s" % (d["ms_per_step"], d["value"], r["kernel"], r["kernel_ms_avg"], r["frac"], r["traffic"], d["cpu_baseline"]["value"]))
PY
