#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_h
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.txt
timeout 300 python scripts/ga_loop_timing.py > $O/ga_loop.txt 2>&1; cat $O/ga_loop.txt | grep "1000000"
B="timeout 300 python bench.py --no-cpu-baseline --svd-solver 0 --pipelined 0"
for fl in 10 6 2; do
  $B --rows 500000 --cols 368 --steps 20 --warmup 3 --preheat 100 --option quad_flow=$fl > $O/bench_368_flow$fl.json 2>> $O/bench.err
  python - <<PY
import json
d=json.loads(open("$O/bench_368_flow$fl.json").read()); r=d["roofline"]
print("500000x368 flow $fl (lead %d)" % (($fl)>>2), "ms/step %.4f" % d["ms_per_step"], "%.4f ms" % r.get("kernel_ms_avg", 0), "frac %.3f" % r["frac"])
PY
done
cd /tmp
SB="python $R/bench.py --rows 500000 --cols 368 --steps 4 --warmup 1 --preheat 20 --no-cpu-baseline --svd-solver 0 --pipelined 0"
for fl in 6 2; do
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_flow$fl -o pmc -- $SB --option quad_flow=$fl > $O/pmc_flow$fl.log 2>&1
python - <<PY
import csv,glob
for f in glob.glob("$O/pmc_flow$fl/**/*counter_collection.csv", recursive=True):
    tot={}; cnt={}
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"][:40]
        if row["Counter_Name"]=="FETCH_SIZE":
            tot[k]=tot.get(k,0)+float(row["Counter_Value"]); cnt[k]=cnt.get(k,0)+1
    for k in tot:
        if "syrk" in k: print("flow $fl", k, "-> x2 x1024 = %.3f GB" % (tot[k]/cnt[k]*2*1024/1e9))
PY
done
find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
