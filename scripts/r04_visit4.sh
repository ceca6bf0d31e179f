#!/bin/bash
O=gpurun_out/r04v4
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_native_comm.py tests/test_gpu_rowspace.py tests/test_gpu_configs.py -x -q -m gpu -k "large_k or cholesky or alternates or triangle or rowspace or quadratic or 1595 or collective or one_rank" > $O/pytest.log 2>&1
tail -4 $O/pytest.log
for v in 0 1 2; do
  echo "FSNAP_CHOL_DIAG=$v"
  FSNAP_CHOL_DIAG=$v timeout 200 python scripts/chol_large_test.py 480 1024 1595 2>&1 | grep "K="
done | tee $O/chol_variants.txt
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --pipelined 0 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("default", d["ms_per_step"], d["value"], d["roofline"]["kernel_ms_avg"], d["roofline"]["reduce_kernel_ms_avg"], d["roofline"]["frac"], "h2d", d["h2d_upload_ms"], d["h2d_upload_path"])
print(d["svd_solver"]["steps"]); print(d["svd_solver"]["class_perform_fit"]); print(d["svd_solver"]["row_space"]["ms_per_fit"])
PY
timeout 300 python bench.py --steps 20 --warmup 3 --preheat 100 --no-cpu-baseline --pipelined 0 --svd-solver 0 --rows 15213 --cols 1595 > $O/bench_15213x1595.json 2> $O/bench2.err
python - <<PY
import json
d=json.loads(open("$O/bench_15213x1595.json").read().strip().splitlines()[-1])
print("15213x1595", d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"])
PY
