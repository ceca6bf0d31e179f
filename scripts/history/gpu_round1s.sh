#!/bin/bash
# Kernel 1L: interleaved park (ablate=9) and + operand prefetch (ablate=10) on 8/4/2-wave workgroups.
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
show() { python - <<PY
import json
try:
    d=json.load(open("$1")); l=d["config"]["launch"]
    print("$2", "kernel %.4f ms reduce %.4f frac %.3f step %.3f ms wgs %d err %s"%(d["roofline"]["kernel_ms_avg"],d["roofline"]["reduce_kernel_ms_avg"],d["roofline"]["frac"],d["ms_per_step"],l["workgroups"], d.get("cpu_baseline",{}).get("gpu_vs_oracle_max_rel_err")))
except Exception as e: print("$2 failed", e)
PY
}
for k in 2 5 6; do for a in 9 10; do
  timeout 300 python bench.py --steps 10 --warmup 3 --option kernel=$k --option ablate=$a > $O/c.json 2>$O/s.err; show $O/c.json "kernel=$k ablate=$a (oracle check)"
done; done
for rep in 1 2; do
for k in 2 5 6; do for a in 0 9 10; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --option kernel=$k --option ablate=$a > $O/v.json 2>>$O/s.err; show $O/v.json "kernel=$k ablate=$a"
done; done; done
T=tools/syrk_trace
for cfg in "512 0 8 10" "768 0 4 10" "1024 0 2 10"; do
  echo "--- $cfg"; timeout 100 $T 1000000 $cfg | grep -v "^  cu"
done
tail -3 $O/s.err
