#!/bin/bash
# round 4, visit 16: kernel 1Q on short, wide systems (where the tiled kernel still wins; chunks per workgroup 24 / 12 / 8)
O=gpurun_out/r04v16
mkdir -p $O
for shape in "13035 256" "20000 256" "30000 256" "50000 256" "13035 208" "20000 208" "30000 208" "8192 224"; do
  set -- $shape
  for cfg in "quad=0" "quad=1" "quad_min_cpg=12" "quad_min_cpg=8"; do
  timeout 300 python bench.py --steps 30 --warmup 5 --preheat 60 --no-cpu-baseline --svd-solver 0 --pipelined 0 --rows $1 --cols $2 --option $cfg --option quad_min_rows=0 > $O/bench_$1x$2_$cfg.json 2> $O/bench.err || tail -3 $O/bench.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$1x$2_$cfg.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("%-14s %-16s ms/step %.4f kernel %.4f (frac %.3f) reduce %.4f launch %s" % ("$1x$2", "$cfg", d["ms_per_step"], r["kernel_ms_avg"], r["frac"], r.get("reduce_kernel_ms_avg", float("nan")), {k: d["config"]["launch"][k] for k in ("workgroups", "chunks_per_wave", "kernel_or_pairs") if k in d["config"]["launch"]}))
except Exception as e:
    print("$1x$2 $cfg: no result (%s)" % e)
PY
  done
done 2>&1 | tee $O/quad_short.txt
