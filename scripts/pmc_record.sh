#!/bin/bash
# Re-take the HBM-traffic record of the headline SYRK kernel (profiles/pmc_traffic.json) after a change of fsnap_syrk.hip /
# fsnap_device_common.h: the record is tied to a digest of those sources and bench.py reports `traffic: null` without it.
# Separate --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ / GRBM counters) over a short bench run, then scripts/pmc_traffic.py.
# Usage: gpurun -- 'bash scripts/pmc_record.sh'; afterwards copy gpurun_out/pmc_record/pmc_traffic_record.json to
# profiles/pmc_traffic.json and gpurun_out/pmc_record/pmc_syrk.md to profiles/rNN_pmc_syrk_acc.md.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_record
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 300 python bench.py --steps 50 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
BENCH="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline"
cd /tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc/pass$i -o pmc -- $BENCH > $O/pmc_pass$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
cd $R
python scripts/pmc_summary.py $O/pmc fsnap_syrk > $O/pmc_syrk.md; cat $O/pmc_syrk.md
python scripts/pmc_traffic.py $O/pmc $O/bench.json > $O/pmc_traffic.json
# the tiled kernel's shapes (K > 128): their own records, so that bench.py reports `traffic` there too
for s in "367900 480 100" "15213 1595 100"; do
  set -- $s
  SB="python $R/bench.py --rows $1 --cols $2 --steps 4 --warmup 1 --preheat 20 --no-cpu-baseline --scaling strong"
  timeout 300 python bench.py --rows $1 --cols $2 --steps 20 --warmup 3 --preheat $3 --no-cpu-baseline --scaling strong > $O/bench_$1x$2.json 2>> $O/bench.err
  cd /tmp; i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVES SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$1x$2/pass$i -o pmc -- $SB > $O/pmc_$1x$2_pass$i.log 2>&1
  done
  cd $R
  python scripts/pmc_summary.py $O/pmc_$1x$2 fsnap_syrk_tiled > $O/pmc_syrk_tiled_$1x$2.md; cat $O/pmc_syrk_tiled_$1x$2.md
  python scripts/pmc_traffic.py $O/pmc_$1x$2 $O/bench_$1x$2.json --append > $O/pmc_traffic_$1x$2.json
  timeout 300 python bench.py --rows $1 --cols $2 --steps 20 --warmup 3 --preheat $3 --no-cpu-baseline --scaling strong > $O/bench_$1x$2.json 2>> $O/bench.err
  python -c "import json; d=json.loads(open('$O/bench_$1x$2.json').read()); print('$1 x $2', d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])"
done
cp profiles/pmc_traffic.json $O/pmc_traffic_record.json
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_after.json 2>> $O/bench.err; python -c "import json; d=json.loads(open('$O/bench_after.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])"
find $O -name "*.csv" -size +4M -delete; find $O -name "*.db" -delete
