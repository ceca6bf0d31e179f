#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/y
O=$PWD/gpurun_out/y
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o chol -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --rows 15213 --cols 1595 --steps 20 --warmup 3 --preheat 50 > $O/bench_prof.json 2> $O/prof.err
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(6), 'avg_us', '%.1f'%(float(r['AverageNs'])/1e3), 'tot_ms', '%.2f'%(float(r['TotalDurationNs'])/1e6))
PY
cp "$f" $O/chol_kernel_stats.csv
rm -rf $O/prof
