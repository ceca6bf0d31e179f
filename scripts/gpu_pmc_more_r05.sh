#!/bin/bash
# PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate runs) for the shapes of the README table that had no record; appends to
# profiles/pmc_traffic.json on the box and leaves the merged record in gpurun_out/r05_pmc_more/pmc_traffic_record.json
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_pmc_more
mkdir -p $O
export TMPDIR=/tmp
pmc_shape () {
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
cd $R
# PMC_SHAPES="100000x192 13035x256" overrides the list
for s in ${PMC_SHAPES:-15213x31 1000000x31 13035x142 1772880x168 367900x288 100000x168}; do set -- ${s%x*} ${s#*x}
  pmc_shape $1 $2 ""
  python scripts/pmc_traffic.py $O/pmc_$1x$2 $O/pmc_bench_$1x$2.json --append > $O/pmc_traffic_$1x$2.json 2> $O/pmc_traffic_$1x$2.err
  tail -c 400 $O/pmc_traffic_$1x$2.json; echo
done
cp profiles/pmc_traffic.json $O/pmc_traffic_record.json
find $O -name "*.csv" -size +4M -delete; find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +1M -delete
du -sh $O
