#!/bin/bash
# Diagnostics visit: ablations of kernel 1L and cache-resident problem sizes (timing only).
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
show() { python - <<PY
import json
try:
    d=json.load(open("$1"))
    r=d["config"]["rows_per_gpu"]
    print("$2", "kernel %.4f ms  (%.3f ns/row)  frac %.3f  ms/step %.4f"%(d["roofline"]["kernel_ms_avg"], d["roofline"]["kernel_ms_avg"]*1e6/r, d["roofline"]["frac"], d["ms_per_step"]))
except Exception as e: print("$2 failed", e)
PY
}
for rep in 1 2; do
for abl in 0 1 2 3 4; do
  timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --option ablate=$abl > $O/bench_abl${abl}_$rep.json 2>> $O/bench.err; show $O/bench_abl${abl}_$rep.json "ablate=$abl rep$rep"
done
done
for rows in 50000 100000 200000 400000 2000000 4000000; do
  timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --rows $rows > $O/bench_rows$rows.json 2>> $O/bench.err; show $O/bench_rows$rows.json "rows=$rows"
done
for nb in 256 512 1024 2048; do
  timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --option nblocks=$nb > $O/bench_nb$nb.json 2>> $O/bench.err; show $O/bench_nb$nb.json "nblocks=$nb"
done
