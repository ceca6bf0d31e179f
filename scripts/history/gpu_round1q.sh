#!/bin/bash
# Kernel 1L: wave-priority variants (ablate=7, 8).
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
T=tools/syrk_trace
for cfg in "512 0 8 0" "512 0 8 7" "512 0 8 8" "768 0 4 7" "768 0 4 8" "512 0 8 0" "512 0 8 7" "512 0 8 8"; do
  echo "--- $cfg"; timeout 100 $T 1000000 $cfg | grep -v "^  cu"
done
show() { python - <<PY
import json
try:
    d=json.load(open("$1")); l=d["config"]["launch"]
    print("$2", "kernel %.4f ms reduce %.4f frac %.3f step %.3f ms wgs %d err %s"%(d["roofline"]["kernel_ms_avg"],d["roofline"]["reduce_kernel_ms_avg"],d["roofline"]["frac"],d["ms_per_step"],l["workgroups"], d.get("cpu_baseline",{}).get("gpu_vs_oracle_max_rel_err")))
except Exception as e: print("$2 failed", e)
PY
}
timeout 300 python bench.py --steps 20 --warmup 3 --option ablate=8 > $O/k8.json 2>$O/k8.err; show $O/k8.json "ablate=8 (with oracle check)"
for o in "kernel=2" "ablate=7" "ablate=8" "kernel=2" "ablate=7" "ablate=8" "kernel=2" "ablate=7" "ablate=8"; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --option $o > $O/v.json 2>>$O/k8.err; show $O/v.json "$o"
done
tail -3 $O/k8.err
