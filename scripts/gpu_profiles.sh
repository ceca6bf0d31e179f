#!/bin/bash
# Profiles of a round (tag = r02 ...): rocprofv3 kernel statistics of the headline bench, the BASELINE shapes and the
# secondary kernels, the PMC passes that feed profiles/pmc_traffic.json, the K sweep of the device Cholesky and the
# re-weighting loop.  Usage: gpurun -- 'bash scripts/gpu_profiles.sh r02'
tag=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${tag}_profiles
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 300 python bench.py --steps 50 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 300 python bench.py --no-cpu-baseline --rows 1000000 --cols 31 --steps 30 --warmup 3 > $O/bench_1000000x31.json 2>> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --rows 15213 --cols 31 --steps 30 --warmup 3 > $O/bench_15213x31.json 2>> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --rows 15213 --cols 1595 --steps 20 --warmup 3 --preheat 100 > $O/bench_15213x1595.json 2>> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --rows 367900 --cols 480 --steps 20 --warmup 3 --preheat 100 > $O/bench_367900x480.json 2>> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --rows 1772880 --cols 110 --steps 30 --warmup 3 --preheat 150 > $O/bench_1772880x110.json 2>> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --rows 13035 --cols 142 --steps 30 --warmup 3 --preheat 150 > $O/bench_13035x142.json 2>> $O/bench.err
timeout 200 python scripts/chol_large_test.py 2>&1 | grep "K=" > $O/chol_large_k_sweep.txt; cat $O/chol_large_k_sweep.txt
timeout 300 python scripts/ga_loop_timing.py > $O/ga_loop.txt 2>&1; cat $O/ga_loop.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k1595 -o k1595 -- python $R/bench.py --no-cpu-baseline --rows 15213 --cols 1595 --steps 20 --warmup 3 --preheat 50 > /dev/null 2>> $O/rocprof.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k31 -o k31 -- python $R/bench.py --no-cpu-baseline --rows 1000000 --cols 31 --steps 30 --warmup 3 > /dev/null 2>> $O/rocprof.log
(cd $R && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_secondary -o sec -- python scripts/secondary_kernels.py > $O/secondary_kernels.json 2>> $O/rocprof.log)
for d in prof_bench prof_k1595 prof_k31 prof_secondary; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; done
BENCH="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc/pass$i -o pmc -- $BENCH > $O/pmc_pass$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
cd $R
python scripts/pmc_summary.py $O/pmc fsnap_syrk > $O/pmc_syrk.md; cat $O/pmc_syrk.md
python scripts/pmc_traffic.py $O/pmc $O/bench.json > $O/pmc_traffic.json; cp profiles/pmc_traffic.json $O/pmc_traffic_record.json
cat $O/secondary_kernels.json | head -40
head -30 $O/prof_secondary_kernel_stats.csv
find $O -name "*.csv" -size +4M -delete; find $O -name "*.db" -delete
