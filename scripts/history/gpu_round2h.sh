#!/bin/bash
export TMPDIR=/tmp
R=$PWD
O=$PWD/gpurun_out/2h
mkdir -p $O
cd /tmp
for t2 in 1 0; do
  BENCH="python $R/bench.py --rows 15213 --cols 1595 --steps 4 --warmup 1 --preheat 20 --no-cpu-baseline --option tiled2=$t2"
  i=0
  for set in "FETCH_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS"; do
    i=$((i+1))
    rm -rf $O/t$t2/pass$i
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/t$t2/pass$i -o pmc -- $BENCH > $O/t${t2}_pass$i.log 2>&1
    echo "t2=$t2 pass $i rc=$?"
  done
  python $R/scripts/pmc_summary.py $O/t$t2 fsnap_syrk_tiled > $O/pmc_t$t2.md; cat $O/pmc_t$t2.md
done
find $O -name "*.csv" -size +4M -delete
