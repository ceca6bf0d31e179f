#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_d
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cluster_quad" > $O/pytest_cluster.txt 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_cluster.txt
B="timeout 300 python bench.py --no-cpu-baseline --svd-solver 0 --pipelined 0"
for shape in "367900 480" "200000 320" "500000 368" "100000 512"; do set -- $shape
  $B --rows $1 --cols $2 --steps 20 --warmup 3 --preheat 100 > $O/bench_$1x$2.json 2>> $O/bench.err
  $B --rows $1 --cols $2 --steps 20 --warmup 3 --preheat 100 --option quad_cluster=0 > $O/bench_$1x$2_tiled.json 2>> $O/bench.err
  python - <<PY
import json
for tag in ("", "_tiled"):
    try:
        d=json.loads(open("$O/bench_$1x$2%s.json" % tag).read()); r=d["roofline"]
        print("$1x$2", tag or "_1QC", "ms/step %.4f" % d["ms_per_step"], "kernel", r.get("kernel"), "%.4f ms" % r.get("kernel_ms_avg", 0), "frac %.3f" % r["frac"], "reduce", r.get("reduce_kernel_ms_avg"))
    except Exception as e: print("$1x$2", tag, "failed", e)
PY
done
tail -5 $O/bench.err
