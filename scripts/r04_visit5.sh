#!/bin/bash
O=gpurun_out/r04v5
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rowspace.py tests/test_gpu_configs.py -x -q -m gpu -k "large_k or cholesky or alternates or rowspace or quadratic or 1595 or 480" > $O/pytest.log 2>&1
grep -n "passed\|failed" $O/pytest.log | tail -3
for f in 1 0; do
  echo "FSNAP_CHOL_FUSED=$f"
  FSNAP_CHOL_FUSED=$f timeout 200 python scripts/chol_large_test.py 257 384 480 768 1024 1595 2048 2>&1 | grep "K="
done | tee $O/chol_fused.txt
for f in 1 0; do
FSNAP_CHOL_FUSED=$f timeout 300 python bench.py --steps 20 --warmup 3 --preheat 100 --no-cpu-baseline --pipelined 0 --svd-solver 0 --rows 15213 --cols 1595 > $O/bench_15213x1595_f$f.json 2> $O/bench2.err
python - <<PY
import json
d=json.loads(open("$O/bench_15213x1595_f$f.json").read().strip().splitlines()[-1])
print("15213x1595 fused=$f", d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"])
PY
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pipelined 0 --svd-solver 0 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("default", d["ms_per_step"], d["value"], "h2d", d["h2d_upload_ms"], d["h2d_upload_path"])
PY
