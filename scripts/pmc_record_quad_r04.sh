#!/bin/bash
# Add the kernel-1Q shapes (10^6 x 256, 1 772 880 x 168, 367 900 x 200, 10^6 x 275) to profiles/pmc_traffic.json (records are tied to a digest of fsnap_syrk_quad.hip +
# fsnap_device_common.h): FETCH_SIZE / WRITE_SIZE passes + SQ / GRBM passes, bench lines of the same shapes.
# Usage: gpurun -- 'bash scripts/pmc_record_quad_r04.sh'; then copy gpurun_out/r04_pmcq/pmc_traffic_record.json to profiles/pmc_traffic.json.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_pmcq
mkdir -p $O
export TMPDIR=/tmp
cd $R
pmc_shape () {   # rows cols
  local rows=$1 cols=$2
  local D=$O/pmc_${rows}x${cols}
  local SB="python $R/bench.py --rows $rows --cols $cols --steps 4 --warmup 1 --preheat 20 --no-cpu-baseline --svd-solver 0 --pipelined 0"
  cd /tmp; local i=0
  for set in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $D/pass$i -o pmc -- $SB > $D.log$i 2>&1
  done
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $D/pass3 -o pmc -- $SB > $D.log3 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $D/pass4 -o pmc -- $SB > $D.log4 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $D/pass5 -o pmc -- $SB > $D.log5 2>&1
  (cd $R && python scripts/pmc_summary.py $D fsnap_syrk_quad > $O/pmc_syrk_quad_${rows}x${cols}.md; cat $O/pmc_syrk_quad_${rows}x${cols}.md)
  cd $R
  timeout 300 python bench.py --rows $rows --cols $cols --steps 50 --warmup 5 --preheat 150 --no-cpu-baseline --svd-solver 0 --pipelined 0 > $O/bench_${rows}x${cols}.json 2>> $O/bench.err
  python scripts/pmc_traffic.py $D $O/bench_${rows}x${cols}.json --append > $O/pmc_traffic_${rows}x${cols}.json
}
pmc_shape 1000000 256
pmc_shape 1772880 168
pmc_shape 367900 200
pmc_shape 1000000 275
cp profiles/pmc_traffic.json $O/pmc_traffic_record.json
# bench lines with the record in place (traffic filled in), and two more widths
for shape in "1000000 256" "1772880 168" "367900 200" "1000000 275" "367900 288" "1000000 264" "100000 272" "500000 224" "100000 192" "100000 168"; do
  set -- $shape
  timeout 300 python bench.py --rows $1 --cols $2 --steps 50 --warmup 5 --preheat 150 --no-cpu-baseline --svd-solver 0 --pipelined 0 > $O/bench_$1x$2.json 2>> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench_$1x$2.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("%-14s ms/step %.4f value %.4g kernel %s %.4f ms frac %.3f reduce %.4f traffic %s" % ("$1x$2", d["ms_per_step"], d["value"], r["kernel"], r["kernel_ms_avg"], r["frac"], r["reduce_kernel_ms_avg"], r["traffic"]))
PY
done
find $O -name "*.csv" -size +4M -delete; find $O -name "*.db" -delete
