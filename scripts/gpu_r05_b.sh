#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_b
mkdir -p $O
export TMPDIR=/tmp
cd $R
for f in 5 4 2; do
  timeout 120 python scripts/chol_large_test.py --form $f 256 288 384 480 512 768 1024 1595 2048 > $O/chol_form$f.txt 2>&1; echo "chol form $f rc=$?"
  cat $O/chol_form$f.txt | grep "K=" 
done
cd /tmp
for f in 5 4; do
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_chol$f -o chol -- python $R/scripts/chol_large_test.py --form $f 1595 > $O/prof_chol$f.log 2>&1
f2=$(find $O/prof_chol$f -name "*kernel_stats.csv" | head -1); [ -n "$f2" ] && cp $f2 $O/chol${f}_k1595_kernel_stats.csv && head -12 $O/chol${f}_k1595_kernel_stats.csv
done
find $O -name "*.db" -delete; find $O -name "*trace.csv" -size +2M -delete
