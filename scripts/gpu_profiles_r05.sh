#!/bin/bash
# Profiles of round 5: bench lines (headline, driver flags, the other BASELINE shapes, kernel 1QC's shapes, shards), Cholesky sweep,
# GA loop, rocprofv3 kernel statistics (bench, K = 1595, kernel 1QC at 367900 x 480, kernel 1Q re-taken), PMC traffic passes.
# Usage: gpurun -- 'bash scripts/gpu_profiles_r05.sh'; then copy what is wanted from gpurun_out/r05_profiles to profiles/.
tag=r05
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
$B $Q --rows 367900 --cols 480 --steps 20 --warmup 3 --preheat 100 --option quad_cluster=0 > $O/bench_367900x480_tiled.json 2>> $O/bench.err
$B $Q --rows 500000 --cols 368 --steps 20 --warmup 3 --preheat 100 > $O/bench_500000x368.json 2>> $O/bench.err
$B $Q --rows 1772880 --cols 110 --steps 30 --warmup 3 --preheat 150 > $O/bench_1772880x110.json 2>> $O/bench.err
$B $Q --rows 13035 --cols 142 --steps 30 --warmup 3 --preheat 150 > $O/bench_13035x142.json 2>> $O/bench.err
$B $Q --rows 1772880 --cols 142 --steps 30 --warmup 3 --preheat 100 > $O/bench_1772880x142.json 2>> $O/bench.err
$B $Q --rows 1772880 --cols 168 --steps 30 --warmup 3 --preheat 100 > $O/bench_1772880x168.json 2>> $O/bench.err
$B $Q --rows 1000000 --cols 256 --steps 30 --warmup 3 --preheat 100 > $O/bench_1000000x256.json 2>> $O/bench.err
$B $Q --rows 100000 --cols 168 --steps 30 --warmup 3 --preheat 150 > $O/bench_100000x168.json 2>> $O/bench.err
$B $Q --rows 100000 --cols 192 --steps 30 --warmup 3 --preheat 150 > $O/bench_100000x192.json 2>> $O/bench.err
$B $Q --rows 100000 --cols 272 --steps 30 --warmup 3 --preheat 150 > $O/bench_100000x272.json 2>> $O/bench.err
$B $Q --rows 13035 --cols 256 --steps 30 --warmup 3 --preheat 150 > $O/bench_13035x256.json 2>> $O/bench.err
$B $Q --rows 367900 --cols 288 --steps 30 --warmup 3 --preheat 100 > $O/bench_367900x288.json 2>> $O/bench.err
for n in 125000 250000 500000; do
  $B $Q --force-dist --rows $n --steps 50 --warmup 5 > $O/bench_shard_${n}x128.json 2>> $O/bench.err
done
timeout 200 python scripts/chol_large_test.py 192 224 256 288 320 384 480 512 640 768 1024 1280 1595 2048 2>&1 | grep "K=" > $O/chol_large_k_sweep.txt; cat $O/chol_large_k_sweep.txt
timeout 300 python scripts/ga_loop_timing.py > $O/ga_loop.txt 2>&1; cat $O/ga_loop.txt
cd /tmp
RP="rocprofv3 --kernel-trace --stats --output-format csv"
timeout 600 $RP -d $O/prof_bench -o bench -- python $R/bench.py --no-cpu-baseline --pipelined 0 > $O/bench_under_rocprof.json 2> $O/rocprof.log
timeout 300 $RP -d $O/prof_k1595 -o k1595 -- python $R/bench.py $Q --pipelined 0 --rows 15213 --cols 1595 --steps 20 --warmup 3 --preheat 50 > /dev/null 2>> $O/rocprof.log
timeout 300 $RP -d $O/prof_k480 -o k480 -- python $R/bench.py $Q --pipelined 0 --rows 367900 --cols 480 --steps 20 --warmup 3 --preheat 50 > /dev/null 2>> $O/rocprof.log
timeout 300 $RP -d $O/prof_k256 -o k256 -- python $R/bench.py $Q --pipelined 0 --rows 1000000 --cols 256 --steps 20 --warmup 3 --preheat 50 > /dev/null 2>> $O/rocprof.log
timeout 300 $RP -d $O/prof_k168 -o k168 -- python $R/bench.py $Q --pipelined 0 --rows 1772880 --cols 168 --steps 20 --warmup 3 --preheat 50 > /dev/null 2>> $O/rocprof.log
timeout 300 $RP -d $O/prof_shard -o shard -- python $R/bench.py $Q --pipelined 0 --force-dist --rows 125000 --steps 50 --warmup 5 > /dev/null 2>> $O/rocprof.log
(cd $R && FSNAP_ROWSPACE_TIMING=0 timeout 300 $RP -d $O/prof_rowspace -o rs -- python scripts/rowspace_calls.py 400000 > /dev/null 2>> $O/rocprof.log)
for d in prof_bench prof_k1595 prof_k480 prof_k256 prof_k168 prof_shard prof_rowspace; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; done
pmc_shape () {   # rows cols extra-bench-args
  local rows=$1 cols=$2 extra=$3
  local D=$O/pmc_${rows}x${cols}
  local SB="python $R/bench.py --rows $rows --cols $cols --steps 4 --warmup 1 --preheat 20 --no-cpu-baseline --svd-solver 0 --pipelined 0 $extra"
  cd /tmp; local i=0
  for set in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $D/pass$i -o pmc -- $SB > $D.log$i 2>&1
  done
  cd $R
  timeout 300 python bench.py --rows $rows --cols $cols --steps 20 --warmup 3 --preheat 100 --no-cpu-baseline --svd-solver 0 --pipelined 0 $extra > $O/pmc_bench_${rows}x${cols}.json 2>> $O/bench.err
}
pmc_shape 1000000 128 ""
python scripts/pmc_traffic.py $O/pmc_1000000x128 $O/pmc_bench_1000000x128.json > $O/pmc_traffic_1000000x128.json
for s in "1772880 110" "1772880 142" "367900 480" "500000 368" "1000000 256" "15213 1595"; do set -- $s
  pmc_shape $1 $2 ""
  python scripts/pmc_traffic.py $O/pmc_$1x$2 $O/pmc_bench_$1x$2.json --append > $O/pmc_traffic_$1x$2.json
done
for n in 500000 250000 125000; do
  pmc_shape $n 128 "--force-dist"
  python scripts/pmc_traffic.py $O/pmc_${n}x128 $O/pmc_bench_${n}x128.json --append > $O/pmc_traffic_${n}x128.json
done
cp profiles/pmc_traffic.json $O/pmc_traffic_record.json
$B --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_after.json 2>> $O/bench.err; python -c "import json; d=json.loads(open('$O/bench_after.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])"
find $O -name "*.csv" -size +4M -delete; find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +1M -delete
du -sh $O
