"""Per-call wall times of fsnap_residual_rhs (one pass / two passes) at a few shapes: looks for outliers."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from fitsnap_amd import _capi
from fitsnap_amd.synthetic import synth_problem

for m, K in ((1_000_000, 31), (1_000_000, 64), (1_000_000, 128), (4_000_000, 31), (200_000, 200)):
    A, b, w = synth_problem(m, K)
    ctx = _capi.HipContext(0)
    ctx.upload_rows(A, b)
    ctx.set_weights(w)
    beta = ctx.fit_resident(_capi.SOLVE_RIDGE, 1e-8)[0]
    for mode in (1, 2, 0):
        ctx.set_option("fused_residual", mode)
        ts = []
        for i in range(14):
            t0 = time.perf_counter()
            s = ctx.residual_rhs(beta)[0]
            ts.append((time.perf_counter() - t0) * 1e3)
        print(f"{m} x {K} fused_residual={mode}: first {ts[0]:.3f} ms, then min {min(ts[2:]):.3f} median {np.median(ts[2:]):.3f} max {max(ts[2:]):.3f} ms; "
              f"|s| {np.linalg.norm(s):.6e}", flush=True)
    ctx.close()
