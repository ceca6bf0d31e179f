#!/bin/bash
O=gpurun_out/r04v11
mkdir -p $O
for shape in "13035 142" "15213 128" "30000 110"; do
  set -- $shape
  for nb in 0 96 128 160 204 256; do
  timeout 300 python bench.py --steps 100 --warmup 5 --preheat 200 --no-cpu-baseline --svd-solver 0 --pipelined 0 --rows $1 --cols $2 --option nblocks=$nb > $O/bench_$1x$2_nb$nb.json 2> $O/bench.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$1x$2_nb$nb.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("%-10s nblocks=%-3s ms/step %.4f kernel %.4f reduce %.4f wg %s cpw %s" % ("$1x$2", "$nb", d["ms_per_step"], r["kernel_ms_avg"], r["reduce_kernel_ms_avg"], d["config"]["launch"]["workgroups"], d["config"]["launch"]["chunks_per_wave"]))
PY
  done
done
