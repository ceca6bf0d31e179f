#!/bin/bash
# Tiled kernel: XCD-aware item mapping and round-filling split count, A/B + L2 counters.
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
echo "== pytest gpu (tiled + parity)"; timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
show() { python - <<PY
import json
try:
    d=json.load(open("$1")); l=d["config"]["launch"]
    print("$2", "kernel %.4f ms reduce %.4f frac %.3f step %.3f ms wgs %d nsplit %d"%(d["roofline"]["kernel_ms_avg"],d["roofline"]["reduce_kernel_ms_avg"],d["roofline"]["frac"],d["ms_per_step"],l["workgroups"],l["nsplit"]))
except Exception as e: print("$2 failed", e)
PY
}
for shape in "15213 1595" "367900 480" "13035 142" "200000 256" "1000000 200"; do
  set -- $shape
  for opts in "xcd=0" "xcd=1" ; do
    timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --rows $1 --cols $2 --option $opts > $O/t_$1x$2_$opts.json 2>>$O/t.err
    show $O/t_$1x$2_$opts.json "$1x$2 $opts"
  done
done
for ns in 2 3 6 11; do
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --rows 15213 --cols 1595 --option nsplit=$ns > $O/t_ns.json 2>>$O/t.err; show $O/t_ns.json "1595 xcd=1 nsplit=$ns"
done
for ns in 14 15 28 42 56; do
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --rows 367900 --cols 480 --option nsplit=$ns > $O/t_ns.json 2>>$O/t.err; show $O/t_ns.json "480 xcd=1 nsplit=$ns"
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --rows 367900 --cols 480 --option nsplit=$ns --option xcd=0 > $O/t_ns.json 2>>$O/t.err; show $O/t_ns.json "480 xcd=0 nsplit=$ns"
done
cd /tmp
for shape in "15213 1595" "367900 480"; do
  set -- $shape
  BENCH="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --rows $1 --cols $2"
  i=0
  for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAVES"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_tiled2_$2/pass$i -o pmc -- $BENCH > $O/pmc_tiled_pass$i.log 2>&1
  done
  python $R/scripts/pmc_summary.py $O/pmc_tiled2_$2 fsnap_syrk_tiled
done
tail -3 $O/t.err
find $O -name "*.csv" -size +8M -delete
