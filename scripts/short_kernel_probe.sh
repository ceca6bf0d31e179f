#!/bin/bash
# Kernel 1S against kernel 1A / the tiled kernel on short systems of 81 ... 144 columns: SYRK kernel, reduction and complete-fit
# time (profiles/r06_short_kernel.txt).   gpurun -- 'bash scripts/short_kernel_probe.sh'
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { # rows K opts...
  rows=$1; K=$2; shift 2
  out=$(timeout 200 python bench.py --rows $rows --cols $K --steps 30 --warmup 5 --preheat 300 --no-cpu-baseline --scaling strong --svd-solver 0 --pipelined 0 "$@" 2>/dev/null | tail -1)
  python -c "
import json,sys
d=json.loads('''$out'''); r=d['roofline']
print('$rows x $K', ' '.join('$*'.split()), '| fit', round(d['ms_per_step']*1e3,1), 'us kernel', round(r['kernel_ms_avg']*1e3,1), 'us reduce', round(r['reduce_kernel_ms_avg']*1e3,1), 'us', r['kernel'], 'wg', d['config']['launch']['workgroups'], 'rows/chunk', d['config']['launch']['chunks_per_wave'])"
}
[ "$1" = "long" ] || for shape in "13035 142" "13035 128" "13035 110" "13035 96" "8000 142" "4000 142" "14336 142" "16384 128"; do
  run $shape
  run $shape --option short=0
done
# longer systems: several phases per workgroup (the next phase's rows are loaded while this one is multiplied) against kernel 1A
if [ "$1" = "long" ]; then
for rows in 20000 28672 40000 60000 100000 125000 250000; do
  for K in 142 128 110 96; do
    run $rows $K --option short=1
    run $rows $K --option short=0
  done
done
fi
