#!/bin/bash
# GPU visit 2: full gpu test suite, bench with breakdown, counter list, PMC passes.
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
echo "== pytest gpu" ; timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; tail -5 $O/pytest_gpu.log
echo "== bench" ; timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err ; echo "bench rc=$?" ; python - <<PY
import json
d=json.load(open("$O/bench.json"))
print("value %.4g rows/s  ms/step %.4f  kernel %.4f ms  reduce %.4f ms  frac %.3f"%(d["value"],d["ms_per_step"],d["roofline"]["kernel_ms_avg"],d["roofline"]["reduce_kernel_ms_avg"],d["roofline"]["frac"]))
print({k:v for k,v in d.items() if k.startswith("step_")})
print("cpu_baseline", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("gpu_vs_oracle_max_rel_err"))
PY
cd /tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
grep -c . $O/counters_list.txt
BENCH="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_r01/pass$i -o pmc -- $BENCH > $O/pmc_pass$i.log 2>&1
  echo "pmc pass $i ($set) rc=$?"
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r01b -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/rocprof_b.log 2>&1
echo "stats rc=$?"
find $O/pmc_r01 $O/prof_r01b -type f | head -40
for f in $(find $O/prof_r01b -name "*kernel_stats.csv"); do head -6 $f; done
find $O -name "*.csv" -size +8M -delete
