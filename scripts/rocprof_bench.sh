#!/bin/bash
# rocprofv3 kernel statistics of a bench.py run (without the two-fits-in-flight leg: the overlapping kernels of its two contexts run
# at half speed each and would sit in the average).   gpurun -- 'bash scripts/rocprof_bench.sh r06 [bench flags]'
TAG=${1:-r06}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${TAG}_rocprof
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof -o bench -- python $R/bench.py --pipelined 0 --no-cpu-baseline "$@" > $O/${TAG}_bench_under_rocprof.json 2> $O/rocprof.err
echo "rocprofv3 rc=$?"
cd $R
for f in $(find $O/rocprof -name "*kernel_stats.csv"); do cp $f $O/${TAG}_bench_kernel_stats.csv; done
head -6 $O/${TAG}_bench_kernel_stats.csv | cut -c1-220
python -c "
import json; d=json.loads(open('$O/${TAG}_bench_under_rocprof.json').read().strip().splitlines()[-1]); print('bench line under rocprof: ms/step', d['ms_per_step'], 'kernel_ms_avg', d['roofline']['kernel_ms_avg'], 'frac', d['roofline']['frac'])"
rm -rf $O/rocprof
