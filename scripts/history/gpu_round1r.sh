#!/bin/bash
# Kernel 1L with 2-wave workgroups (18 tiles per wave): kernel=6
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
T=tools/syrk_trace
for cfg in "512 0 8 0" "1024 0 2 0" "2048 0 2 0" "512 0 8 0" "1024 0 2 0"; do
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
timeout 300 python bench.py --steps 20 --warmup 3 --option kernel=6 > $O/k6.json 2>$O/k6.err; show $O/k6.json "kernel=6 (with oracle check)"
for o in "kernel=2" "kernel=6" "kernel=2" "kernel=6" "kernel=2" "kernel=6"; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --option $o > $O/v.json 2>>$O/k6.err; show $O/v.json "$o"
done
for nb in 512 768 1024 1536 2048; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --option kernel=6 --option nblocks=$nb > $O/v.json 2>>$O/k6.err; show $O/v.json "kernel=6 nblocks=$nb"
done
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --option kernel=6 --rows 1772880 --cols 110 > $O/v.json 2>>$O/k6.err; show $O/v.json "kernel=6 1772880x110"
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --option kernel=2 --rows 1772880 --cols 110 > $O/v.json 2>>$O/k6.err; show $O/v.json "kernel=2 1772880x110"
tail -3 $O/k6.err
