#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_c
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt
timeout 200 python scripts/chol_large_test.py 128 160 192 224 256 288 320 384 480 512 640 768 1024 1280 1595 2048 2>&1 | grep "K=" > $O/chol_large_k_sweep.txt; cat $O/chol_large_k_sweep.txt
FSNAP_ROWSPACE_TIMING=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_rs.json 2> $O/rowspace_phases.txt; tail -12 $O/rowspace_phases.txt
python - <<PY
import json
d=json.loads(open("$O/bench_rs.json").read())
print(json.dumps(d.get("svd_solver",{}).get("row_space"))[:600])
PY
timeout 300 python bench.py --no-cpu-baseline --svd-solver 0 --rows 15213 --cols 1595 --steps 20 --warmup 3 --preheat 100 > $O/bench_15213x1595.json 2>> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --svd-solver 0 --rows 367900 --cols 480 --steps 20 --warmup 3 --preheat 100 > $O/bench_367900x480.json 2>> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --svd-solver 0 --rows 100000 --cols 272 --steps 20 --warmup 3 --preheat 100 > $O/bench_100000x272.json 2>> $O/bench.err
for f in 15213x1595 367900x480 100000x272; do python -c "import json; d=json.loads(open('$O/bench_$f.json').read()); print('$f', d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel_ms_avg'))"; done
