#!/bin/bash
# round 4, visit 3: full GPU suite, bench (svd_solver leg, staged upload), GA loop, rocprof kernel stats
O=gpurun_out/r04v3
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
tail -4 $O/pytest.log
timeout 600 python bench.py --steps 50 --warmup 5 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("default", d["ms_per_step"], d["value"], d["roofline"]["kernel_ms_avg"], d["roofline"]["reduce_kernel_ms_avg"], d["roofline"]["frac"], "h2d", d["h2d_upload_ms"], d["h2d_inclusive_rows_per_s"])
print(json.dumps(d.get("svd_solver"), indent=1)[:3000])
print(d["cpu_baseline"])
PY
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pipelined 0 --option fused_residual=0 --option staged_upload=0 > $O/bench_old.json 2> $O/bench_old.err
python - <<PY
import json
d=json.loads(open("$O/bench_old.json").read().strip().splitlines()[-1])
print("fused_residual=0 staged_upload=0: h2d", d["h2d_upload_ms"], d["svd_solver"]["steps"])
PY
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pipelined 0 --option fused_residual=2 > $O/bench_fr2.json 2> $O/bench_fr2.err
python - <<PY
import json
d=json.loads(open("$O/bench_fr2.json").read().strip().splitlines()[-1])
print("fused_residual=2:", d["svd_solver"]["steps"])
PY
timeout 300 python scripts/ga_loop_timing.py > $O/ga_loop.txt 2>&1; cat $O/ga_loop.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --pipelined 0 > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 $f | cut -c1-200
find $O/prof -name "*.db" -delete; find $O/prof -name "*trace.csv" -size +2M -delete
