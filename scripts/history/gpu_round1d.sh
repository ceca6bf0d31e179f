#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
echo "== mfma peak microbench"; timeout 300 tools/mfma_f64_peak 2>&1 | tee $O/mfma_f64_peak.txt
echo "== pytest gpu" ; timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; tail -6 $O/pytest_gpu.log
show() { python - <<PY
import json
try:
    d=json.load(open("$1"))
    print("$2", "value %.4g rows/s  ms/step %.4f  kernel %.4f ms  reduce %.4f ms  frac %.3f"%(d["value"],d["ms_per_step"],d["roofline"]["kernel_ms_avg"],d["roofline"]["reduce_kernel_ms_avg"],d["roofline"]["frac"]), {k:round(v,4) for k,v in d.items() if k.startswith("step_")})
except Exception as e: print("$2 failed", e)
PY
}
echo "== bench default" ; timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err ; echo "bench rc=$?"; show $O/bench.json default; tail -2 $O/bench.err
for opt in "kernel=2" "kernel=3" ; do
    timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --option $opt > $O/bench_${opt}.json 2>> $O/bench.err
    show $O/bench_${opt}.json "$opt"
done
