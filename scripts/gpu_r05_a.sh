#!/bin/bash
# Round 5, first GPU trip: device-Cholesky forms (sweep per form), the whole -m gpu suite, the bench line, row-space phases.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_a
mkdir -p $O
export TMPDIR=/tmp
cd $R
for f in 5 4 2; do
  timeout 120 python scripts/chol_large_test.py --form $f 256 288 384 480 512 768 1024 1595 2048 > $O/chol_form$f.txt 2>&1; echo "chol form $f rc=$?"
  cat $O/chol_form$f.txt | grep "K=" 
done
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench.json").read())
print(d["value"], d["ms_per_step"], d["roofline"])
print(json.dumps(d.get("svd_solver"), indent=1)[:3000])
PY
FSNAP_ROWSPACE_TIMING=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_rs.json 2> $O/rowspace_phases.txt; tail -60 $O/rowspace_phases.txt
