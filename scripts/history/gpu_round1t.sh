#!/bin/bash
# Kernel 1A (whole triangle in one wave, AGPR accumulators): correctness + timing.
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
timeout 300 python bench.py --steps 10 --warmup 3 --option kernel=7 > $O/a.json 2>$O/a.err; show $O/a.json "kernel=7 1e6x128 (oracle check)"
timeout 300 python bench.py --steps 5 --warmup 2 --option kernel=7 --rows 200003 --cols 110 > $O/a2.json 2>>$O/a.err; show $O/a2.json "kernel=7 200003x110 (oracle check)"
timeout 300 python bench.py --steps 5 --warmup 2 --option kernel=7 --rows 100001 --cols 96 > $O/a3.json 2>>$O/a.err; show $O/a3.json "kernel=7 100001x96 (oracle check)"
timeout 300 python bench.py --steps 5 --warmup 2 --option kernel=7 --rows 5003 --cols 127 > $O/a4.json 2>>$O/a.err; show $O/a4.json "kernel=7 5003x127 (oracle check)"
for rep in 1 2 3; do
for k in 2 7; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --option kernel=$k > $O/v.json 2>>$O/a.err; show $O/v.json "kernel=$k"
done; done
for nb in 128 256 512; do
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --option kernel=7 --option nblocks=$nb > $O/v.json 2>>$O/a.err; show $O/v.json "kernel=7 nblocks=$nb"
done
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --option kernel=7 --option nontemporal=0 > $O/v.json 2>>$O/a.err; show $O/v.json "kernel=7 nt=0"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --option kernel=7 --rows 1772880 --cols 110 > $O/v.json 2>>$O/a.err; show $O/v.json "kernel=7 1772880x110"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --option kernel=2 --rows 1772880 --cols 110 > $O/v.json 2>>$O/a.err; show $O/v.json "kernel=2 1772880x110"
tail -5 $O/a.err
