#!/bin/bash
# Kernel 1L variants: early park (ablate=6), 4-wave workgroups (kernel=5); traces + bench A/B.
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
T=tools/syrk_trace
for cfg in "512 0 8 0" "512 0 8 6" "768 0 4 0" "1536 0 4 0" "512 0 8 0" "512 0 8 6" "768 0 4 0"; do
  echo "--- $cfg"; timeout 100 $T 1000000 $cfg | grep -v "^  cu"
done
echo "=== phase clocks"; timeout 100 tools/syrk_trace2 1000000 768 0 4 0 | grep -v "^  cu"
timeout 100 tools/syrk_trace2 1000000 512 0 8 6 | grep -v "^  cu"
show() { python - <<PY
import json
try:
    d=json.load(open("$1")); l=d["config"]["launch"]
    print("$2", "kernel %.4f ms reduce %.4f frac %.3f step %.3f ms wgs %d err %s"%(d["roofline"]["kernel_ms_avg"],d["roofline"]["reduce_kernel_ms_avg"],d["roofline"]["frac"],d["ms_per_step"],l["workgroups"], d.get("cpu_baseline",{}).get("gpu_vs_oracle_max_rel_err")))
except Exception as e: print("$2 failed", e)
PY
}
timeout 300 python bench.py --steps 20 --warmup 3 --option kernel=5 > $O/k5.json 2>$O/k5.err; show $O/k5.json "kernel=5 (with oracle check)"
for o in "kernel=2" "ablate=6" "kernel=5" "kernel=2" "ablate=6" "kernel=5"; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --option $o > $O/v.json 2>>$O/k5.err; show $O/v.json "$o"
done
for nb in 512 768 1024 1536; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --option kernel=5 --option nblocks=$nb > $O/v.json 2>>$O/k5.err; show $O/v.json "kernel=5 nblocks=$nb"
done
tail -3 $O/k5.err
