#!/bin/bash
# First GPU visit: smoke, GPU parity tests, bench (+ variants), rocprofv3 kernel trace.
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -8 > $O/rocminfo.txt
echo "== smoke" ; timeout 600 python __graft_entry__.py --smoke > $O/smoke.log 2>&1 ; echo "smoke rc=$?" ; tail -3 $O/smoke.log
echo "== pytest gpu" ; timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; tail -15 $O/pytest_gpu.log
echo "== bench" ; timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err ; echo "bench rc=$?" ; cat $O/bench.json | head -c 3000 ; tail -3 $O/bench.err
for opt in "nontemporal=0" "nblocks=512" "nblocks=128"; do
  echo "== bench variant $opt"
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --option $opt > $O/bench_$opt.json 2>> $O/bench.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$opt.json")); print("$opt", "value %.4g rows/s"%d["value"], "ms/step %.4f"%d["ms_per_step"], "kernel ms %.4f"%d["roofline"]["kernel_ms_avg"], "frac %.3f"%d["roofline"]["frac"])
except Exception as e: print("variant failed", e)
PY
done
echo "== rocprof"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_r01 -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/rocprof.log 2>&1
echo "rocprof rc=$?"
find $O/prof_r01 -name "*stats*" | head
for f in $(find $O/prof_r01 -name "*kernel_stats.csv"); do head -8 $f; done
# keep the scratch small: drop the raw trace, keep the stats
find $O/prof_r01 -name "*kernel_trace.csv" -size +20M -delete
