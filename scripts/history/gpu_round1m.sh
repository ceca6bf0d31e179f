#!/bin/bash
# Final round-1 measurement visit: default bench, rocprofv3 stats of the same command, PMC traffic passes.
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
echo "== example"; timeout 300 python examples/fit_ta_golden.py 2>&1 | tail -3
echo "== bench default"; timeout 900 python bench.py > $O/final_bench.json 2> $O/final_bench.err; echo "rc=$?"; cat $O/final_bench.json; tail -2 $O/final_bench.err
echo "== bench default again"; timeout 900 python bench.py --no-cpu-baseline > $O/final_bench2.json 2>> $O/final_bench.err; cat $O/final_bench2.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final -o bench -- python $R/bench.py > $O/final_rocprof_bench.json 2> $O/final_rocprof.log
echo "rocprof rc=$?"
for f in $(find $O/prof_final -name "*kernel_stats.csv"); do head -8 $f; done
BENCH="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVES"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_final/pass$i -o pmc -- $BENCH > $O/pmc_final_pass$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
python $R/scripts/pmc_summary.py $O/pmc_final fsnap_syrk > $O/final_pmc.md; cat $O/final_pmc.md
python $R/scripts/pmc_summary.py $O/pmc_final fsnap_reduce | tail -n +3
find $O -name "*.csv" -size +8M -delete
