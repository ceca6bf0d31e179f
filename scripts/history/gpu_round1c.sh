#!/bin/bash
# GPU visit 3: LDS-shared kernel + tiled kernel parity, kernel A/B, PMC on the new kernel.
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
echo "== pytest gpu" ; timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; tail -6 $O/pytest_gpu.log
show() { python - <<PY
import json
try:
    d=json.load(open("$1"))
    print("$2", "value %.4g rows/s  ms/step %.4f  kernel %.4f ms  reduce %.4f ms  frac %.3f"%(d["value"],d["ms_per_step"],d["roofline"]["kernel_ms_avg"],d["roofline"]["reduce_kernel_ms_avg"],d["roofline"]["frac"]), {k:round(v,4) for k,v in d.items() if k.startswith("step_")})
except Exception as e: print("$2 failed", e)
PY
}
echo "== bench default" ; timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err ; echo "bench rc=$?"; show $O/bench.json default; tail -2 $O/bench.err
for opt in "kernel=1" "kernel=2" "kernel=3" ; do
  for nt in 0 1; do
    timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --option $opt --option nontemporal=$nt > $O/bench_${opt}_nt$nt.json 2>> $O/bench.err
    show $O/bench_${opt}_nt$nt.json "$opt nt=$nt"
  done
done
for nb in 256 384 768 1024; do
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --option kernel=2 --option nblocks=$nb > $O/bench_k2_nb$nb.json 2>> $O/bench.err
  show $O/bench_k2_nb$nb.json "kernel=2 nblocks=$nb"
done
echo "== other shapes (rows cols)"
for shape in "1772880 110" "13035 142" "15213 31" "15213 1595" "367900 480"; do
  set -- $shape
  timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --rows $1 --cols $2 > $O/bench_$1x$2.json 2>> $O/bench.err
  show $O/bench_$1x$2.json "$1x$2"
done
cd /tmp
BENCH="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_r01c/pass$i -o pmc -- $BENCH > $O/pmc_pass$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r01c -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/rocprof_c.log 2>&1
for f in $(find $O/prof_r01c -name "*kernel_stats.csv"); do head -4 $f; done
find $O -name "*.csv" -size +8M -delete
