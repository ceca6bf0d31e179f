cd $GRAFT_REPO_ROOT
run() { # rows K opts...
  rows=$1; K=$2; shift 2
  out=$(timeout 200 python bench.py --rows $rows --cols $K --steps 30 --warmup 5 --preheat 300 --no-cpu-baseline --scaling strong --svd-solver 0 --pipelined 0 "$@" 2>/dev/null | tail -1)
  python -c "
import json,sys
d=json.loads('''$out'''); r=d['roofline']
print('$rows x $K', ' '.join('$*'.split()), '| fit', round(d['ms_per_step']*1e3,1), 'us kernel', round(r['kernel_ms_avg']*1e3,1), 'us', r['kernel'], 'wg', d['config']['launch']['workgroups'], 'nsplit', d['config']['launch'].get('nsplit'))"
}
for ns in 0 25 42 64 85; do run 13035 142 --option tiled=1 --option nsplit=$ns; done
run 13035 142
for nb in 100 136 200 256; do run 13035 142 --option nblocks=$nb; done
run 13035 256
for ns in 0 13 26 40 51; do run 13035 256 --option tiled=1 --option nsplit=$ns; done
run 100000 168
for ns in 0 32 64 96; do run 100000 168 --option tiled=1 --option nsplit=$ns; done
run 15213 31
for nb in 119 200 400 800; do run 15213 31 --option nblocks=$nb; done
