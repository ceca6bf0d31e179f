#!/bin/bash
# Re-take every HBM-traffic record of profiles/pmc_traffic.json (after a change of fsnap_syrk.hip / fsnap_syrk_quad.hip /
# fsnap_device_common.h: a record is tied to a digest of those sources and bench.py reports `traffic: null` without one) and the
# bench lines profiles/<tag>_bench_<rows>x<K>.json the README table is generated from.
#   gpurun --timeout 3000 -- 'bash scripts/pmc_record_all.sh r06'
# Per shape: one bench run (the line + the launch geometry), two --pmc passes (FETCH_SIZE; WRITE_SIZE -- separate, as
# MI355X_MICROARCH.md prescribes) over a short bench run, scripts/pmc_traffic.py --append, the bench run again (its line then
# carries the record).  Outputs under gpurun_out/<tag>_record/; copy pmc_traffic.json and the bench lines to profiles/.
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${TAG}_record
mkdir -p $O
export TMPDIR=/tmp
cd $R
# FSNAP_PMC_SHAPES="13035 142;28672 142": only these shapes, appended to (or replacing their records in) the existing file --
# after a change of ONE kernel's source file (kernel 1S lives in fsnap_syrk_short.hip with a digest of its own)
if [ -z "$FSNAP_PMC_SHAPES" ]; then
  rm -f profiles/pmc_traffic.json          # the headline record comes first; everything is re-taken
fi
# rows K [extra bench flags]
SHAPES=("1000000 128" "1772880 110" "1772880 142" "367900 480" "500000 368" "1000000 256" "15213 1595" "500000 128 --force-dist"
        "250000 128 --force-dist" "125000 128 --force-dist" "15213 31" "1000000 31" "13035 142" "1772880 168" "367900 288" "100000 168"
        "100000 192" "13035 256" "100000 272")
if [ -n "$FSNAP_PMC_SHAPES" ]; then IFS=';' read -ra SHAPES <<< "$FSNAP_PMC_SHAPES"; fi
for s in "${SHAPES[@]}"; do
  read -r rows K extra <<< "$s"
  name=${rows}x${K}
  [ -n "$extra" ] && name=shard_${name}
  pre=100; [ "$rows" -lt 200000 ] && pre=300
  FL="--rows $rows --cols $K --warmup 3 --preheat $pre --no-cpu-baseline --scaling strong $extra"
  [ "$name" = "1000000x128" ] && FL="--rows $rows --cols $K --warmup 5"
  env="env"; [ -n "$extra" ] && env="env RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 FSNAP_COMM_FILE=$O/id_$name"
  timeout 300 $env python bench.py $FL --steps 20 > $O/bench_$name.json 2>> $O/bench.err || { echo "$name: bench failed"; continue; }
  cd /tmp; i=0
  for set in FETCH_SIZE WRITE_SIZE; do
    i=$((i+1))
    timeout 600 $env rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$name/pass$i -o pmc -- python $R/bench.py $FL --steps 4 --svd-solver 0 --pipelined 0 > $O/pmc_${name}_pass$i.log 2>&1
  done
  cd $R
  python scripts/pmc_traffic.py $O/pmc_$name $O/bench_$name.json --append > $O/traffic_$name.json 2>> $O/bench.err || echo "$name: no record"
  timeout 300 $env python bench.py $FL --steps 20 > $O/${TAG}_bench_$name.json 2>> $O/bench.err
  python -c "import json; d=json.loads(open('$O/${TAG}_bench_$name.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$name', round(d['ms_per_step'],4), r['kernel'], round(r['kernel_ms_avg'],4), round(r['frac'],3), r['traffic'] and round(r['traffic']/r['algorithmic_bytes_per_launch'],3))"
  find $O/pmc_$name -name "*.csv" -size +2M -delete; find $O/pmc_$name -name "*.db" -delete
done
cp profiles/pmc_traffic.json $O/pmc_traffic.json
