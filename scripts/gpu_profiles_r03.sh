#!/bin/bash
# Profiles of round 3 (what scripts/gpu_profiles.sh collects, plus the row-space path after kernel 13B / the device pass
# factors, the shard a rank of an 8-GPU strong-scaling run works on, and PMC passes over kernel 13B).
# Usage: gpurun -- 'bash scripts/gpu_profiles_r03.sh'
tag=r03
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${tag}_profiles
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 300 python bench.py --steps 50 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2>> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --rows 1000000 --cols 31 --steps 30 --warmup 3 > $O/bench_1000000x31.json 2>> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --rows 15213 --cols 31 --steps 30 --warmup 3 > $O/bench_15213x31.json 2>> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --rows 15213 --cols 1595 --steps 20 --warmup 3 --preheat 100 > $O/bench_15213x1595.json 2>> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --rows 367900 --cols 480 --steps 20 --warmup 3 --preheat 100 > $O/bench_367900x480.json 2>> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --rows 1772880 --cols 110 --steps 30 --warmup 3 --preheat 150 > $O/bench_1772880x110.json 2>> $O/bench.err
timeout 300 python bench.py --no-cpu-baseline --rows 13035 --cols 142 --steps 30 --warmup 3 --preheat 150 > $O/bench_13035x142.json 2>> $O/bench.err
# what ONE rank of an 8- / 4- / 2-GPU strong-scaling run does (10^6 / N rows), through the collective code path in a communicator of one rank
for n in 125000 250000 500000; do
  timeout 300 python bench.py --no-cpu-baseline --force-dist --rows $n --steps 50 --warmup 5 > $O/bench_shard_${n}x128.json 2>> $O/bench.err
done
timeout 200 python scripts/chol_large_test.py 2>&1 | grep "K=" > $O/chol_large_k_sweep.txt; cat $O/chol_large_k_sweep.txt
timeout 300 python scripts/ga_loop_timing.py > $O/ga_loop.txt 2>&1; cat $O/ga_loop.txt
# row-space pass kernels alone (tools/trsm_check: Q = X R^-1 against a host substitution), kernel 13 for comparison
(for s in "15213 1595" "367900 480" "100000 256" "1772880 256" "6001 208"; do tools/trsm_check $s 0 5 | head -1; tools/trsm_check $s 1 5 | head -1; FSNAP_TRSM_KERNEL=13 tools/trsm_check $s 1 3 | head -1; done) > $O/trsm_check.txt 2>&1; cat $O/trsm_check.txt
(for s in "15213 1595" "367900 480" "1000000 128" "100000 256"; do FSNAP_ROWSPACE_TIMING=1 python scripts/lstsq_rows_profile.py $s 2>&1 | grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" | tail -12; done) > $O/lstsq_rows_phases.txt 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/rocprof.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k1595 -o k1595 -- python $R/bench.py --no-cpu-baseline --rows 15213 --cols 1595 --steps 20 --warmup 3 --preheat 50 > /dev/null 2>> $O/rocprof.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k31 -o k31 -- python $R/bench.py --no-cpu-baseline --rows 1000000 --cols 31 --steps 30 --warmup 3 > /dev/null 2>> $O/rocprof.log
for s in "15213 1595" "367900 480" "1000000 128" "100000 256"; do
  set -- $s
  (cd $R && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lstsq_$1x$2 -o l -- python scripts/lstsq_rows_profile.py $1 $2 > /dev/null 2>> $O/rocprof.log)
  f=$(find $O/prof_lstsq_$1x$2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/lstsq_rows_kernel_stats_$1x$2.csv
done
(cd $R && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_secondary -o sec -- python scripts/secondary_kernels.py > $O/secondary_kernels.json 2>> $O/rocprof.log)
for d in prof_bench prof_k1595 prof_k31 prof_secondary; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; done
BENCH="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc/pass$i -o pmc -- $BENCH > $O/pmc_pass$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
# kernel 13B: matrix-pipe busy and wait shares at both large-K shapes, L2 hit rate, HBM-side bytes
for s in "15213 1595" "367900 480"; do
  set -- $s
  i=0
  for cs in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    (cd $R && timeout 200 rocprofv3 --kernel-trace --pmc $cs --output-format csv -d $O/pmc_trsm_$1x$2/pass$i -o pmc -- tools/trsm_check $1 $2 1 3 > $O/pmc_trsm_$1x$2.log$i 2>&1)
  done
  (cd $R && python scripts/pmc_summary.py $O/pmc_trsm_$1x$2 fsnap_trsm_panel > $O/pmc_trsm_panel_$1x$2.md; cat $O/pmc_trsm_panel_$1x$2.md)
done
cd $R
python scripts/pmc_summary.py $O/pmc fsnap_syrk > $O/pmc_syrk.md; cat $O/pmc_syrk.md
python scripts/pmc_traffic.py $O/pmc $O/bench.json > $O/pmc_traffic.json; cp profiles/pmc_traffic.json $O/pmc_traffic_record.json
find $O -name "*.csv" -size +4M -delete; find $O -name "*.db" -delete
