"""A further right-hand side for statistics just factorised on the GPU (fsnap_solve_device_rhs: what a refinement step of a fit
pays for its K x K solve): sweeps with the factor left on the device (option chol_reuse = 1) against a second factorisation (0)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fitsnap_amd import _capi

ctx = _capi.HipContext(0)
rng = np.random.default_rng(1)
for K in (256, 480, 768, 1024, 1595, 2048):
    m = 4 * K
    A = rng.standard_normal((m, K)); b = rng.standard_normal(m)
    ctx.upload_rows(A, b); ctx.set_weights(np.ones(m))
    ptr = ctx.normal_eq_resident()
    rhs = rng.standard_normal(K)
    out = []
    for reuse in (1, 0):
        ctx.set_option("chol_reuse", reuse)
        ctx.solve_device(_capi.SOLVE_LSTSQ, 1e-13, K, ptr)
        ts = []
        for _ in range(25):
            t0 = time.perf_counter()
            x = ctx.solve_device(_capi.SOLVE_LSTSQ, 1e-13, K, ptr, rhs=rhs)[0]
            ts.append(time.perf_counter() - t0)
        out.append((float(np.median(ts[5:])) * 1e3, x))
    print(f"K = {K:5d}: sweeps with the kept factor {out[0][0]:7.3f} ms   factorise again {out[1][0]:7.3f} ms   max rel diff {np.abs(out[0][1]-out[1][1]).max()/np.abs(out[1][1]).max():.1e}", flush=True)
ctx.close()
