#!/bin/bash
# Closing measurement of a round: tests, smoke, default bench, rocprofv3 stats, PMC passes, micro-benchmarks,
# large-K shapes (tiled kernel + device Cholesky).
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc=$?"; tail -3 $O/pytest_gpu.log
echo "== smoke"; timeout 600 python __graft_entry__.py --smoke 2>&1 | tail -2
echo "== bench default"; timeout 900 python bench.py > $O/final_bench.json 2> $O/final_bench.err; echo "rc=$?"; cat $O/final_bench.json
# K sweep of the device Cholesky BEFORE the profiler passes (their post-processing competes for the host cores that
# issue its ~100 launches per solve)
timeout 200 python scripts/chol_large_test.py 2>&1 | grep "K=" > $O/chol_large_k_sweep.txt; cat $O/chol_large_k_sweep.txt
tools/mfma_f64_peak > $O/microbench_mfma_f64_peak.txt 2>&1
{ for c in "256 0 1 0" "512 0 8 0" "768 0 4 0" "1024 0 2 0"; do timeout 60 tools/syrk_trace 1000000 $c; echo; done; } > $O/microbench_syrk_trace.txt 2>&1
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final -o bench -- python $R/bench.py > $O/final_rocprof_bench.json 2> $O/final_rocprof.log
echo "rocprof rc=$?"
for f in $(find $O/prof_final -name "*kernel_stats.csv"); do head -6 $f; done
BENCH="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rm -rf $O/pmc_final/pass$i
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_final/pass$i -o pmc -- $BENCH > $O/pmc_final_pass$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
python $R/scripts/pmc_summary.py $O/pmc_final fsnap_syrk > $O/final_pmc.md; cat $O/final_pmc.md
echo "== large-K shapes"
cd $R
timeout 200 python bench.py --no-cpu-baseline --rows 15213 --cols 1595 --steps 20 --warmup 3 --preheat 100 > $O/bench_15213x1595.json 2>> $O/final_bench.err
timeout 200 python bench.py --no-cpu-baseline --rows 367900 --cols 480 --steps 20 --warmup 3 --preheat 100 > $O/bench_367900x480.json 2>> $O/final_bench.err
timeout 200 python bench.py --no-cpu-baseline --rows 1772880 --cols 110 --steps 30 --warmup 3 --preheat 150 > $O/bench_1772880x110.json 2>> $O/final_bench.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k1595 -o k1595 -- python $R/bench.py --no-cpu-baseline --rows 15213 --cols 1595 --steps 20 --warmup 3 --preheat 50 > /dev/null 2>> $O/final_rocprof.log
cp $(find $O/prof_k1595 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_15213x1595.csv
cd $R
python - <<'PY'
import json
for f in ("bench_15213x1595", "bench_367900x480", "bench_1772880x110"):
    d = json.load(open(f"gpurun_out/{f}.json"))
    print(f, "step ms %.3f kernel ms %.4f frac %.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"]))
PY
find $O -name "*.csv" -size +8M -delete
