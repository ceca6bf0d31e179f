#!/bin/bash
# PMC passes over ONE kernel of a bench.py run at a given shape (separate passes: matrix-pipe busy / wave cycles, waits,
# L2 hit rate, FETCH_SIZE, WRITE_SIZE), summarised per dispatch.
# Usage (on the GPU box): bash scripts/pmc_kernel.sh <rows> <cols> <kernel name prefix> <tag> [extra bench.py arguments]
rows=$1; cols=$2; kern=$3; tag=$4; shift 4
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_$tag; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
BENCH="python $R/bench.py --no-cpu-baseline --scaling strong --pipelined 0 --rows $rows --cols $cols --steps 4 --warmup 1 --preheat 20 $*"
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pass$i -o pmc -- $BENCH > $O/log$i.txt 2>&1
  echo "pass $i rc=$?"
done
cd $R; python scripts/pmc_summary.py $O $kern
