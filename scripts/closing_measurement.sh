#!/bin/bash
# Closing measurement of a round on the GPU box: pytest -m gpu, smoke(), the default bench.py line, the driver's flags, and the
# rocprofv3 kernel statistics of the same command (outputs under gpurun_out/<tag>_closing/; copy what is to be judged to profiles/).
#   gpurun --timeout 2400 -- 'bash scripts/closing_measurement.sh r06'
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${TAG}_closing
mkdir -p $O
export TMPDIR=/tmp
cd $R
(time timeout 1800 python -m pytest tests -q -m gpu --timeout 900) > $O/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -4 $O/gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/${TAG}_bench_default.json 2> $O/bench.err; echo "bench rc=$?"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_driver_flags.json 2>> $O/bench.err; echo "bench (driver flags) rc=$?"
cd /tmp
# (without the two-fits-in-flight leg, whose overlapping kernels of two contexts run at half speed each and would sit in the average)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof -o bench -- python $R/bench.py --pipelined 0 --no-cpu-baseline > $O/${TAG}_bench_under_rocprof.json 2> $O/rocprof.err; echo "rocprofv3 rc=$?"
cd $R
find $O/rocprof -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_bench_kernel_stats.csv \;
head -12 $O/${TAG}_bench_kernel_stats.csv
python - <<PY
import json
for n in ("default", "driver_flags", "under_rocprof"):
    try:
        d = json.loads(open("$O/${TAG}_bench_%s.json" % n).read().strip().splitlines()[-1])
    except Exception as e:
        print(n, "no line", e); continue
    r = d["roofline"]; s = d.get("svd_solver", {})
    print(n, "value %.4g rows/s  %.4f ms/step  kernel %.4f ms frac %.3f traffic %s" % (d["value"], d["ms_per_step"], r["kernel_ms_avg"], r["frac"], r["traffic"] and round(r["traffic"] / r["algorithmic_bytes_per_launch"], 3)))
    if s:
        print("   svd steps %.4f ms (%s refinement steps, rcond %.3g) class %.4f ms (max %.3f) row space %.3f ms (max %.3f)" % (
            s["steps"]["ms_per_fit"], s["steps"]["refinement_steps"], s["steps"]["rcond_est"], s["class_perform_fit"]["ms_per_fit"],
            s["class_perform_fit"].get("ms_max", 0), s.get("row_space", {}).get("ms_per_fit", 0), s.get("row_space", {}).get("ms_max", 0)))
    print("   cpu_baseline", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"), "weighting", d.get("weighting_kernel", {}).get("frac"))
PY
find $O/rocprof -name "*.csv" -size +4M -delete; find $O/rocprof -name "*.db" -delete
