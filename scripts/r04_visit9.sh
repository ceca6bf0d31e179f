#!/bin/bash
O=gpurun_out/r04v9
mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
grep -n "passed\|failed\|error" $O/pytest.log | tail -5
for shape in "15213 31" "1000000 31" "4000000 31" "100000 64" "1000000 80"; do
  set -- $shape
  for fp in 1 0; do
  timeout 300 python bench.py --steps 50 --warmup 5 --preheat 150 --no-cpu-baseline --svd-solver 0 --pipelined 0 --rows $1 --cols $2 --option fused_pack=$fp > $O/bench_$1x$2_fp$fp.json 2> $O/bench.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$1x$2_fp$fp.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("%-12s fused_pack=$fp ms/step %.4f value %.4g kernel %s %.4f ms frac %.3f reduce %.4f launch %s" % ("$1x$2", d["ms_per_step"], d["value"], r["kernel"], r["kernel_ms_avg"], r["frac"], r["reduce_kernel_ms_avg"], d["config"]["launch"]))
PY
  done
done
bash scripts/pmc_record_r04.sh 2>&1 | tail -9
