#!/bin/bash
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
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "statistics or kernel or one_wave or tiled or masked or strided or bit_identical" 2>&1 | tail -3
timeout 300 python bench.py --steps 10 --warmup 3 > $O/a.json 2>$O/a.err; show $O/a.json "default (1A) 1e6x128 (oracle check)"
for rep in 1 2 3; do
for k in 2 7; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --option kernel=$k > $O/v.json 2>>$O/a.err; show $O/v.json "kernel=$k"
done; done
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --rows 1772880 --cols 110 > $O/v.json 2>>$O/a.err; show $O/v.json "default 1772880x110"
for c in "256 0 1 0" "256 0 1 0"; do timeout 60 tools/syrk_trace 1000000 $c | grep -v "^  cu"; done
