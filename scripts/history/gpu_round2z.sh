#!/bin/bash
# rocprofv3 kernel stats of the 367900 x 480 (tiled kernel) and 1772880 x 110 (kernel 1A, NB = 7) shapes
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/2z
mkdir -p $O
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p480 -o k480 -- python $R/bench.py --no-cpu-baseline --rows 367900 --cols 480 --steps 20 --warmup 3 --preheat 60 > $O/bench480.json 2> $O/err480.log
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p110 -o k110 -- python $R/bench.py --no-cpu-baseline --rows 1772880 --cols 110 --steps 20 --warmup 3 --preheat 60 > $O/bench110.json 2> $O/err110.log
cp $(find $O/p480 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_367900x480.csv
cp $(find $O/p110 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_1772880x110.csv
head -5 $O/kernel_stats_367900x480.csv | cut -c1-140; head -4 $O/kernel_stats_1772880x110.csv | cut -c1-140
rm -rf $O/p480 $O/p110
