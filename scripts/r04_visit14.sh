#!/bin/bash
# round 4, visit 14: kernel 1Q (144 < K <= 256: the tile triangle dealt to the four waves of a workgroup) -- parity tests,
# then fit times next to the tiled kernel (option quad = 0) over widths and row counts
O=gpurun_out/r04v14
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "quad" > $O/quad_tests.log 2>&1; tail -5 $O/quad_tests.log
for shape in "100000 192" "100000 168" "1000000 256" "500000 224" "367900 200" "1772880 168" "40000 168" "20000 168" "13035 200" "200000 150"; do
  set -- $shape
  for q in 1 0; do
  timeout 300 python bench.py --steps 30 --warmup 5 --preheat 60 --no-cpu-baseline --svd-solver 0 --pipelined 0 --rows $1 --cols $2 --option quad=$q --option quad_min_rows=0 > $O/bench_$1x$2_q$q.json 2> $O/bench.err || tail -3 $O/bench.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$1x$2_q$q.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("%-14s quad=%s ms/step %.4f kernel %.4f (frac %.3f) reduce %.4f launch %s" % ("$1x$2", "$q", d["ms_per_step"], r["kernel_ms_avg"], r["frac"], r.get("reduce_kernel_ms_avg", float("nan")), {k: d["config"]["launch"][k] for k in ("workgroups", "chunks_per_wave", "kernel_or_pairs")}))
except Exception as e:
    print("$1x$2 quad=$q: no result (%s)" % e)
PY
  done
done 2>&1 | tee $O/quad_ab.txt
