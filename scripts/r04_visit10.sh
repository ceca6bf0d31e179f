#!/bin/bash
O=gpurun_out/r04v10
mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
grep -n "passed\|failed\|error" $O/pytest.log | tail -5
for shape in "100000 300" "50000 352"; do
  set -- $shape
  for ds in 0 2; do
  timeout 300 python bench.py --steps 30 --warmup 3 --preheat 100 --no-cpu-baseline --svd-solver 0 --pipelined 0 --rows $1 --cols $2 --option device_solve=$ds > $O/bench_$1x$2_ds$ds.json 2> $O/bench.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$1x$2_ds$ds.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("%-12s device_solve=$ds ms/step %.4f kernel %.4f" % ("$1x$2", d["ms_per_step"], r["kernel_ms_avg"]))
PY
  done
done
