#!/bin/bash
set +e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
echo "== pytest gpu" ; timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; tail -8 $O/pytest_gpu.log
echo "== 5.1 GB problem (row offsets beyond 2^32 bytes)"
timeout 900 python - <<'PY'
import sys, time, numpy as np
sys.path.insert(0, '.')
from fitsnap_amd import _capi
from oracle import fitsnap_oracle as orc
m, K = 5_000_000, 128
t0 = time.time()
rng = np.random.default_rng(1)
A = rng.standard_normal((m, K), dtype=np.float64)
b = rng.standard_normal(m); w = rng.uniform(0.5, 2.0, m)
print("generated %.1f s" % (time.time() - t0))
ctx = _capi.HipContext(0)
ctx.upload_rows(A, b); ctx.set_weights(w)
G, c, s = ctx.normal_eq(); print("kernel ms", ctx.timing()["syrk_ms"], ctx.launch_info())
# additivity over a split that straddles the 4 GiB byte offset
h = 4_300_000
ctx.upload_rows(A[:h], b[:h]); ctx.set_weights(w[:h]); G1, c1, s1 = ctx.normal_eq()
ctx.upload_rows(A[h:], b[h:]); ctx.set_weights(w[h:]); G2, c2, s2 = ctx.normal_eq()
d = np.sqrt(np.diag(G))
print("additivity err", np.max(np.abs(G1 + G2 - G) / (d[:, None] * d[None, :])), s1[2] + s2[2] == s[2] == m)
# last rows vs oracle (catches a wrapped offset)
Gt, ct, st = orc.normal_eq(A[h:], b[h:], w[h:])
print("tail block vs oracle", np.max(np.abs(G2 - Gt) / (d[:, None] * d[None, :])))
PY
echo "== bench default" ; timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err ; echo "bench rc=$?"; python - <<PY
import json
d=json.load(open("$O/bench.json"))
print("value %.4g rows/s  ms/step %.4f  kernel %.4f ms  frac %.3f traffic %s"%(d["value"],d["ms_per_step"],d["roofline"]["kernel_ms_avg"],d["roofline"]["frac"],d["roofline"]["traffic"]), {k:round(v,4) for k,v in d.items() if k.startswith("step_")})
print(d["cpu_baseline"])
PY
for shape in "1772880 110" "13035 142" "15213 31" "15213 1595" "367900 480" "125464 30"; do
  set -- $shape
  timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --rows $1 --cols $2 > $O/bench_$1x$2.json 2>> $O/bench.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$1x$2.json")); print("$1x$2", "value %.4g rows/s ms/step %.4f kernel %.4f ms frac %.3f solve %.4f"%(d["value"],d["ms_per_step"],d["roofline"]["kernel_ms_avg"],d["roofline"]["frac"],d["step_host_solve_ms_avg"]))
except Exception as e: print("failed", e)
PY
done
