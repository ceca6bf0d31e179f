#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_z
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
RP="rocprofv3 --kernel-trace --stats --output-format csv"
(cd $R && timeout 600 $RP -d $O/prof_rs1595 -o rs -- python scripts/rowspace_large_k.py 15213 1595 6 > $O/rs1595.log 2>&1)
(cd $R && timeout 600 $RP -d $O/prof_rs480 -o rs -- python scripts/rowspace_large_k.py 367900 480 4 > $O/rs480.log 2>&1)
for d in prof_rs1595 prof_rs480; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
head -8 $O/prof_rs480_kernel_stats.csv | cut -c1-200
head -8 $O/prof_rs1595_kernel_stats.csv | cut -c1-200
