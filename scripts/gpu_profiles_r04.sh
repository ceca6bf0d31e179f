#!/bin/bash
# Profiles of round 4: bench lines (headline, driver flags, the other BASELINE shapes, the K = 142 ACE width on kernel 1A
# and on the tiled kernel, the shards of a strong-scaling run), Cholesky sweep, GA loop, rocprofv3 kernel statistics, and the
# PMC passes behind profiles/pmc_traffic.json (headline, K = 142, the shard geometries, the tiled shapes).
# Usage: gpurun -- 'bash scripts/gpu_profiles_r04.sh'; then copy what is wanted from gpurun_out/r04_profiles to profiles/.
tag=r04
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${tag}_profiles
mkdir -p $O
export TMPDIR=/tmp
cd $R
B="timeout 300 python bench.py"
$B --steps 50 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
$B --steps 20 --warmup 5 > $O/bench_driver_flags.json 2>> $O/bench.err
Q="--no-cpu-baseline --svd-solver 0"
$B $Q --rows 1000000 --cols 31 --steps 30 --warmup 3 > $O/bench_1000000x31.json 2>> $O/bench.err
$B $Q --rows 15213 --cols 31 --steps 30 --warmup 3 > $O/bench_15213x31.json 2>> $O/bench.err
$B $Q --rows 15213 --cols 1595 --steps 20 --warmup 3 --preheat 100 > $O/bench_15213x1595.json 2>> $O/bench.err
$B $Q --rows 367900 --cols 480 --steps 20 --warmup 3 --preheat 100 > $O/bench_367900x480.json 2>> $O/bench.err
$B $Q --rows 1772880 --cols 110 --steps 30 --warmup 3 --preheat 150 > $O/bench_1772880x110.json 2>> $O/bench.err
$B $Q --rows 13035 --cols 142 --steps 30 --warmup 3 --preheat 150 > $O/bench_13035x142.json 2>> $O/bench.err
$B $Q --rows 1772880 --cols 142 --steps 30 --warmup 3 --preheat 100 > $O/bench_1772880x142.json 2>> $O/bench.err
$B $Q --rows 1772880 --cols 142 --steps 30 --warmup 3 --preheat 100 --option acc_max_k=128 > $O/bench_1772880x142_tiled.json 2>> $O/bench.err
$B $Q --rows 100000 --cols 192 --steps 30 --warmup 3 --preheat 150 > $O/bench_100000x192.json 2>> $O/bench.err
for n in 125000 250000 500000; do
  $B $Q --force-dist --rows $n --steps 50 --warmup 5 > $O/bench_shard_${n}x128.json 2>> $O/bench.err
done
timeout 200 python scripts/chol_large_test.py 2>&1 | grep "K=" > $O/chol_large_k_sweep.txt; cat $O/chol_large_k_sweep.txt
timeout 300 python scripts/ga_loop_timing.py > $O/ga_loop.txt 2>&1; cat $O/ga_loop.txt
cd /tmp
RP="rocprofv3 --kernel-trace --stats --output-format csv"
timeout 600 $RP -d $O/prof_bench -o bench -- python $R/bench.py --no-cpu-baseline --pipelined 0 > $O/bench_under_rocprof.json 2> $O/rocprof.log
timeout 300 $RP -d $O/prof_k142 -o k142 -- python $R/bench.py $Q --pipelined 0 --rows 1772880 --cols 142 --steps 30 --warmup 3 --preheat 50 > /dev/null 2>> $O/rocprof.log
timeout 300 $RP -d $O/prof_k1595 -o k1595 -- python $R/bench.py $Q --pipelined 0 --rows 15213 --cols 1595 --steps 20 --warmup 3 --preheat 50 > /dev/null 2>> $O/rocprof.log
timeout 300 $RP -d $O/prof_shard -o shard -- python $R/bench.py $Q --pipelined 0 --force-dist --rows 125000 --steps 50 --warmup 5 > /dev/null 2>> $O/rocprof.log
(cd $R && timeout 600 $RP -d $O/prof_secondary -o sec -- python scripts/secondary_kernels.py > $O/secondary_kernels.json 2>> $O/rocprof.log)
for d in prof_bench prof_k142 prof_k1595 prof_shard prof_secondary; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; done
# PMC: headline first (starts profiles/pmc_traffic.json over), then the other shapes appended
pmc_shape () {   # rows cols extra-bench-args kernel-filter with_sq
  local rows=$1 cols=$2 extra=$3 filt=$4 sq=$5
  local D=$O/pmc_${rows}x${cols}
  local SB="python $R/bench.py --rows $rows --cols $cols --steps 4 --warmup 1 --preheat 20 --no-cpu-baseline --svd-solver 0 --pipelined 0 $extra"
  cd /tmp; local i=0
  for set in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $D/pass$i -o pmc -- $SB > $D.log$i 2>&1
  done
  if [ "$sq" = "1" ]; then
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $D/pass3 -o pmc -- $SB > $D.log3 2>&1
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $D/pass4 -o pmc -- $SB > $D.log4 2>&1
    (cd $R && python scripts/pmc_summary.py $D $filt > $O/pmc_${filt}_${rows}x${cols}.md; cat $O/pmc_${filt}_${rows}x${cols}.md)
  fi
  cd $R
  timeout 300 python bench.py --rows $rows --cols $cols --steps 20 --warmup 3 --preheat 100 --no-cpu-baseline --svd-solver 0 --pipelined 0 $extra > $O/pmc_bench_${rows}x${cols}.json 2>> $O/bench.err
}
pmc_shape 1000000 128 "" fsnap_syrk 1
python scripts/pmc_traffic.py $O/pmc_1000000x128 $O/pmc_bench_1000000x128.json > $O/pmc_traffic_1000000x128.json
pmc_shape 1772880 142 "" fsnap_syrk 1
python scripts/pmc_traffic.py $O/pmc_1772880x142 $O/pmc_bench_1772880x142.json --append > $O/pmc_traffic_1772880x142.json
for n in 500000 250000 125000; do
  pmc_shape $n 128 "--force-dist" fsnap_syrk 0
  python scripts/pmc_traffic.py $O/pmc_${n}x128 $O/pmc_bench_${n}x128.json --append > $O/pmc_traffic_${n}x128.json
done
pmc_shape 367900 480 "" fsnap_syrk_tiled 0
python scripts/pmc_traffic.py $O/pmc_367900x480 $O/pmc_bench_367900x480.json --append > $O/pmc_traffic_367900x480.json
pmc_shape 15213 1595 "" fsnap_syrk_tiled 0
python scripts/pmc_traffic.py $O/pmc_15213x1595 $O/pmc_bench_15213x1595.json --append > $O/pmc_traffic_15213x1595.json
cp profiles/pmc_traffic.json $O/pmc_traffic_record.json
$B --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_after.json 2>> $O/bench.err; python -c "import json; d=json.loads(open('$O/bench_after.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])"
find $O -name "*.csv" -size +4M -delete; find $O -name "*.db" -delete
