#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
echo "== pytest gpu" ; timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; tail -4 $O/pytest_gpu.log
show() { python - <<PY
import json
try:
    d=json.load(open("$1"))
    print("$2", "value %.4g rows/s  ms/step %.4f  kernel %.4f ms  reduce %.4f ms  frac %.3f"%(d["value"],d["ms_per_step"],d["roofline"]["kernel_ms_avg"],d["roofline"]["reduce_kernel_ms_avg"],d["roofline"]["frac"]), {k:round(v,4) for k,v in d.items() if k.startswith("step_")})
except Exception as e: print("$2 failed", e)
PY
}
for rep in 1 2; do
for opt in "kernel=2" "kernel=3" "kernel=1"; do
    timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --option $opt > $O/bench_${opt}_$rep.json 2>> $O/bench.err
    show $O/bench_${opt}_$rep.json "$opt rep$rep"
done
done
echo "== bench default" ; timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err ; echo "bench rc=$?"; show $O/bench.json default
cd /tmp
BENCH="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_r01e/pass$i -o pmc -- $BENCH > $O/pmc_pass$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r01e -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/rocprof_e.log 2>&1
for f in $(find $O/prof_r01e -name "*kernel_stats.csv"); do head -4 $f; done
find $O -name "*.csv" -size +8M -delete
