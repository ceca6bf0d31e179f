#!/bin/bash
# N > 1 smoke matrix on ONE GPU through the peer-to-peer transport: bench.py with both scalings over rank counts and widths
# (what the driver's scaling run executes, minus the second device).   gpurun -- 'bash scripts/multi_rank_matrix.sh'
cd ${GRAFT_REPO_ROOT:-/root/repo}
t() { timeout 400 python bench.py --transport p2p --no-cpu-baseline "$@" > /tmp/o.out 2> /tmp/o.err; rc=$?
      python - "$rc" "$*" <<'PY'
import json, sys
rc, args = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open("/tmp/o.out").read().strip().splitlines()[-1])
    print(f"{args}: rc={rc} value {d['value']:.3e} strong {d.get('strong_ms_per_step', 0):.3f} ms weak {d.get('weak_ms_per_step', 0):.3f} ms ranks {d['n_ranks_seen']} allreduce_ms {[round(x, 3) for x in d['per_rank']['allreduce_ms']]}")
except Exception as e:
    print(f"{args}: rc={rc} NO LINE ({e}); stderr tail: {open('/tmp/o.err').read()[-300:]}")
PY
}
t --gpus 4 --steps 10 --warmup 3 --preheat 50
t --gpus 3 --steps 10 --warmup 3 --preheat 50 --rows 300000
t --gpus 2 --steps 10 --warmup 3 --preheat 50 --rows 367900 --cols 480
t --gpus 2 --steps 10 --warmup 3 --preheat 20 --rows 15213 --cols 1595
t --gpus 2 --steps 10 --warmup 3 --preheat 50 --rows 15213 --cols 31
t --gpus 2 --steps 10 --warmup 3 --preheat 50 --rows 100000 --cols 272
t --gpus 2 --steps 10 --warmup 3 --preheat 50 --rows 13035 --cols 142
