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
    print("$2", "value %.4g rows/s  ms/step %.4f  kernel %.4f ms  reduce %.4f ms  frac %.3f"%(d["value"],d["ms_per_step"],d["roofline"]["kernel_ms_avg"],d["roofline"]["reduce_kernel_ms_avg"],d["roofline"]["frac"]), {k:round(v,4) for k,v in d.items() if k.startswith("step_")})
except Exception as e: print("$2 failed", e)
PY
}
for rep in 1 2; do
timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --option device_solve=1 > $O/bench_dev_$rep.json 2>> $O/bench.err; show $O/bench_dev_$rep.json "device-solve rep$rep"
timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > $O/bench_lib_$rep.json 2>> $O/bench.err; show $O/bench_lib_$rep.json "library-host-solve rep$rep"
timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --host-solve > $O/bench_host_$rep.json 2>> $O/bench.err; show $O/bench_host_$rep.json "torch-copy+host-solve rep$rep"
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r01h -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --option device_solve=1 > $O/rocprof_h.log 2>&1
for f in $(find $O/prof_r01h -name "*kernel_stats.csv"); do head -5 $f; done
find $O -name "*.csv" -size +8M -delete
