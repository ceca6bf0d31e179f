"""A few row-space solves of the headline shape, meant to be wrapped in rocprofv3 --kernel-trace --stats."""
import sys
import time

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fitsnap_amd import _capi                      # noqa: E402
from fitsnap_amd.synthetic import synth_problem    # noqa: E402

m, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1_000_000, 128)
A, b, w = synth_problem(m, K)
ctx = _capi.HipContext(0)
ctx.upload_rows(A, b)
ctx.set_weights(w)
ctx.lstsq_rows(1e-13)
ctx.sync()
t0 = time.perf_counter()
for _ in range(8):
    beta, rank, info = ctx.lstsq_rows(1e-13)
ctx.sync()
print(m, K, "lstsq_rows ms", (time.perf_counter() - t0) / 8 * 1e3, info)
