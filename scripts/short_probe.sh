#!/bin/bash
# Short systems: SYRK kernel and complete-fit time by kernel / grid (profiles/r06_short_systems.txt), and the crossover between
# kernel 1Q and the tiled kernel at 145 ... 288 columns (QUAD_MIN_ROWS).   gpurun -- 'bash scripts/short_probe.sh [crossover]'
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { # rows K opts...
  rows=$1; K=$2; shift 2
  out=$(timeout 200 python bench.py --rows $rows --cols $K --steps 30 --warmup 5 --preheat 300 --no-cpu-baseline --scaling strong --svd-solver 0 --pipelined 0 "$@" 2>/dev/null | tail -1)
  python -c "
import json,sys
d=json.loads('''$out'''); r=d['roofline']
print('$rows x $K', ' '.join('$*'.split()), '| fit', round(d['ms_per_step']*1e3,1), 'us kernel', round(r['kernel_ms_avg']*1e3,1), 'us', r['kernel'], 'wg', d['config']['launch']['workgroups'], 'nsplit', d['config']['launch'].get('nsplit'))"
}
if [ "$1" = "crossover" ]; then
  for K in 160 192 256 288; do for rows in 8192 13035 20000 30000 45000; do run $rows $K; run $rows $K --option tiled=1; done; done
  exit 0
fi
for ns in 0 25 42 64 85; do run 13035 142 --option tiled=1 --option nsplit=$ns; done
run 13035 142
for nb in 100 136 200 256; do run 13035 142 --option nblocks=$nb; done
run 13035 256
for ns in 0 13 26 40 51; do run 13035 256 --option tiled=1 --option nsplit=$ns; done
run 100000 168
for ns in 0 32 64 96; do run 100000 168 --option tiled=1 --option nsplit=$ns; done
run 15213 31
