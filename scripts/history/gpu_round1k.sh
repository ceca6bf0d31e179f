#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
echo "== pytest gpu" ; timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; tail -6 $O/pytest_gpu.log
show() { python - <<PY
import json
try:
    d=json.load(open("$1"))
    print("$2", "kernel %.4f ms  reduce %.4f  frac %.3f  ms/step %.4f value %.4g"%(d["roofline"]["kernel_ms_avg"], d["roofline"]["reduce_kernel_ms_avg"], d["roofline"]["frac"], d["ms_per_step"], d["value"]))
except Exception as e: print("$2 failed", e)
PY
}
for rep in 1 2 3; do
  i=0
  for opts in "--option kernel=2" "--option kernel=4" "--option kernel=2 --option ablate=5" "--option kernel=4 --option ablate=5"; do
    i=$((i+1))
    timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline $opts > $O/bench_v${i}_$rep.json 2>> $O/bench.err; show $O/bench_v${i}_$rep.json "[$opts] rep$rep"
  done
done
