#!/bin/bash
# Re-take profiles/pmc_traffic.json after a change of fsnap_syrk.hip / fsnap_device_common.h (the record is tied to a digest of
# those sources): FETCH_SIZE / WRITE_SIZE passes for the headline shape, K = 142, the shard geometries and the tiled shapes
# (+ SQ / GRBM passes for the first two), bench lines of the same shapes.
# Usage: gpurun -- 'bash scripts/pmc_record_r04.sh'; then copy gpurun_out/r04_pmc/pmc_traffic_record.json to profiles/pmc_traffic.json.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_pmc
mkdir -p $O
export TMPDIR=/tmp
cd $R
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
  timeout 300 python bench.py --rows $rows --cols $cols --steps 50 --warmup 5 --preheat 150 --no-cpu-baseline --svd-solver 0 --pipelined 0 $extra > $O/bench_${rows}x${cols}.json 2>> $O/bench.err
}
pmc_shape 1000000 128 "" fsnap_syrk 1
python scripts/pmc_traffic.py $O/pmc_1000000x128 $O/bench_1000000x128.json > $O/pmc_traffic_1000000x128.json
pmc_shape 1772880 142 "" fsnap_syrk 1
python scripts/pmc_traffic.py $O/pmc_1772880x142 $O/bench_1772880x142.json --append > $O/pmc_traffic_1772880x142.json
for n in 500000 250000 125000; do
  pmc_shape $n 128 "--force-dist" fsnap_syrk 0
  python scripts/pmc_traffic.py $O/pmc_${n}x128 $O/bench_${n}x128.json --append > $O/pmc_traffic_${n}x128.json
done
pmc_shape 367900 480 "" fsnap_syrk_tiled 0
python scripts/pmc_traffic.py $O/pmc_367900x480 $O/bench_367900x480.json --append > $O/pmc_traffic_367900x480.json
pmc_shape 15213 1595 "" fsnap_syrk_tiled 0
python scripts/pmc_traffic.py $O/pmc_15213x1595 $O/bench_15213x1595.json --append > $O/pmc_traffic_15213x1595.json
cp profiles/pmc_traffic.json $O/pmc_traffic_record.json
for f in 1000000x128 1772880x142 500000x128 250000x128 125000x128 367900x480 15213x1595; do
python - <<PY
import json
d=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("%-14s ms/step %.4f value %.4g kernel %s %.4f ms frac %.3f reduce %.4f" % ("$f", d["ms_per_step"], d["value"], r["kernel"], r["kernel_ms_avg"], r["frac"], r["reduce_kernel_ms_avg"]))
PY
done
find $O -name "*.csv" -size +4M -delete; find $O -name "*.db" -delete
