#!/bin/bash
# PMC passes over the tiled SYRK kernel (K > 128) at a given shape: matrix-pipe busy cycles, L2 hit rate, HBM-side fetch.
# Usage (on the GPU box): bash scripts/pmc_tiled.sh <rows> <cols> <tag>
rows=${1:-15213}; cols=${2:-1595}; tag=${3:-tiled}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_$tag; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
BENCH="python $R/bench.py --no-cpu-baseline --rows $rows --cols $cols --steps 4 --warmup 1 --preheat 20"
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVES SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pass$i -o pmc -- $BENCH > $O/log$i.txt 2>&1
  echo "pass $i rc=$?"
done
cd $R; python scripts/pmc_summary.py $O fsnap_syrk_tiled
