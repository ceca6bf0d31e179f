#!/bin/bash
# round 4, visit 2: residual kernel + prologue change; bench with svd_solver; rocprof stats; host solve phases
O=gpurun_out/r04v2
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "residual or fused_packing or refinement or svd_solver or column_block_shapes" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 600 python bench.py --steps 50 --warmup 5 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("default", d["ms_per_step"], d["value"], d["roofline"]["kernel_ms_avg"], d["roofline"]["reduce_kernel_ms_avg"], d["roofline"]["frac"])
print(json.dumps(d.get("svd_solver"), indent=1)[:3000])
print(d["cpu_baseline"])
PY
for fr in 0 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pipelined 0 --option fused_residual=$fr > $O/bench_fr$fr.json 2> $O/bench_fr$fr.err
python - <<PY
import json
d=json.loads(open("$O/bench_fr$fr.json").read().strip().splitlines()[-1])
print("fused_residual=$fr", d["svd_solver"]["steps"])
PY
done
FSNAP_SOLVE_TIMING=1 timeout 300 python bench.py --steps 5 --warmup 2 --preheat 50 --no-cpu-baseline --pipelined 0 --svd-solver 0 2> $O/phases.err > $O/phases.json
grep "fsnap_solve\]" $O/phases.err | tail -24
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --pipelined 0 > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats*" | head
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 $f
for shape in "125000 128" "250000 128" "13035 142"; do
  set -- $shape
  timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --svd-solver 0 --rows $1 --cols $2 > $O/bench_$1x$2.json 2> $O/bench_$1x$2.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$1x$2.json").read().strip().splitlines()[-1])
print("$1x$2", d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["kernel_ms_avg"], d["roofline"]["reduce_kernel_ms_avg"], d["roofline"]["frac"])
PY
done
