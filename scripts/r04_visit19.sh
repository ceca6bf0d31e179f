#!/bin/bash
# round 4, visit 19: one-pass residual at 257 ... 288 columns (NJ = 9): parity, time next to the two-kernel form
O=gpurun_out/r04v19
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "one_pass_residual" > $O/residual_tests.log 2>&1; tail -3 $O/residual_tests.log
python - > $O/residual_275.txt 2>&1 <<'PY'
import time, numpy as np
from fitsnap_amd import _capi
rng = np.random.default_rng(1)
for m, K in ((1000000, 275), (367900, 288), (1000000, 256)):
    A = rng.standard_normal((m, K)); b = rng.standard_normal(m); w = rng.uniform(0.5, 2.0, m); beta = rng.standard_normal(K) * 0.1
    c = _capi.HipContext(0); c.upload_rows(A, b); c.set_weights(w)
    for mode in (1, 0):
        c.set_option("fused_residual", mode)
        for _ in range(5): c.residual_rhs(beta, want_sse=True)
        c.sync(); t0 = time.perf_counter()
        for _ in range(20): c.residual_rhs(beta, want_sse=True)
        c.sync(); dt = (time.perf_counter() - t0) / 20
        print(f"{m} x {K} fused_residual={mode}: {dt*1e3:.3f} ms per call ({m*K*8/dt/1e12:.2f} TB/s of rows)")
    c.close()
PY
cat $O/residual_275.txt
