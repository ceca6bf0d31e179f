#!/bin/bash
# PMC profile of the tiled kernel (K > 128) on the ACE-like and quadratic-SNAP-like shapes.
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
for shape in "15213 1595" "367900 480"; do
  set -- $shape
  BENCH="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --rows $1 --cols $2"
  $BENCH > $O/tiled_$1x$2.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$O/tiled_$1x$2.json")); print("$1x$2", d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"], d["config"]["launch"])
PY
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_tiled_$2/pass$i -o pmc -- $BENCH > $O/pmc_tiled_pass$i.log 2>&1
    echo "pmc pass $i rc=$?"
  done
  python $R/scripts/pmc_summary.py $O/pmc_tiled_$2 fsnap_syrk_tiled
done
find $O -name "*.csv" -size +8M -delete
