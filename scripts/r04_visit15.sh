#!/bin/bash
# round 4, visit 15: kernel 1Q with pairs from HBM (PACK = false), full GPU suite, short systems (where does the tiled kernel
# still win?), rocprofv3 kernel trace of two 1Q shapes
O=gpurun_out/r04v15
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; tail -4 $O/gpu_suite.log
for shape in "2000 168" "4000 168" "8192 168" "4000 256" "8192 256" "13035 256" "3000000 168"; do
  set -- $shape
  for q in 1 0; do
  timeout 300 python bench.py --steps 30 --warmup 5 --preheat 60 --no-cpu-baseline --svd-solver 0 --pipelined 0 --rows $1 --cols $2 --option quad=$q --option quad_min_rows=0 > $O/bench_$1x$2_q$q.json 2> $O/bench.err || tail -3 $O/bench.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$1x$2_q$q.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("%-14s quad=%s ms/step %.4f kernel %.4f (frac %.3f) reduce %.4f launch %s" % ("$1x$2", "$q", d["ms_per_step"], r["kernel_ms_avg"], r["frac"], r.get("reduce_kernel_ms_avg", float("nan")), {k: d["config"]["launch"][k] for k in ("workgroups", "chunks_per_wave", "kernel_or_pairs", "fused_pack") if k in d["config"]["launch"]}))
except Exception as e:
    print("$1x$2 quad=$q: no result (%s)" % e)
PY
  done
done 2>&1 | tee $O/quad_small.txt
cd /tmp && export TMPDIR=/tmp
for shape in "1000000 256" "1772880 168"; do
  set -- $shape
  (cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$1x$2 -o quad -- python bench.py --steps 30 --warmup 5 --preheat 60 --no-cpu-baseline --svd-solver 0 --pipelined 0 --rows $1 --cols $2 > $O/bench_prof_$1x$2.json 2> $O/prof.err)
  f=$(find $GRAFT_REPO_ROOT/$O/prof_$1x$2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/$O/kernel_stats_$1x$2.csv && head -6 $f
done
