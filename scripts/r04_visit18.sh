#!/bin/bash
# round 4, visit 18: kernel 1Q at 17 / 18 column blocks (257 ... 288 columns, two raw sets in flight): parity, then fit times
# next to the tiled kernel
O=gpurun_out/r04v18
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "quad" > $O/quad_tests.log 2>&1; tail -4 $O/quad_tests.log
for shape in "1000000 275" "367900 288" "100000 272" "1000000 264" "30000 275" "13035 275"; do
  set -- $shape
  for q in 1 0; do
  timeout 300 python bench.py --steps 30 --warmup 5 --preheat 60 --no-cpu-baseline --svd-solver 0 --pipelined 0 --rows $1 --cols $2 --option quad=$q > $O/bench_$1x$2_q$q.json 2> $O/bench.err || tail -3 $O/bench.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$1x$2_q$q.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("%-14s quad=%s ms/step %.4f kernel %s %.4f (frac %.3f) reduce %.4f launch %s" % ("$1x$2", "$q", d["ms_per_step"], r["kernel"], r["kernel_ms_avg"], r["frac"], r.get("reduce_kernel_ms_avg", float("nan")), {k: d["config"]["launch"][k] for k in ("workgroups", "chunks_per_wave") if k in d["config"]["launch"]}))
except Exception as e:
    print("$1x$2 quad=$q: no result (%s)" % e)
PY
  done
done 2>&1 | tee $O/quad_17_18.txt
