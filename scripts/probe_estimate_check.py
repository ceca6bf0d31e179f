"""Condition estimate of the DEVICE factor (K >= 232: Rayleigh-Ritz on the 31 probe vectors of the factorisation's strip) against the
singular values of the scaled rows, over families and widths; prints estimate / lambda_min."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fitsnap_amd import _capi

rng = np.random.default_rng(0)
ctx = _capi.HipContext(0)


def check(name, A):
    m, K = A.shape
    d = 1.0 / np.sqrt(np.einsum("ij,ij->j", A, A))
    lam = float(np.linalg.svd(A * d, compute_uv=False)[-1] ** 2)
    ctx.upload_rows(A, np.ones(m))
    ctx.set_weights(np.ones(m))
    beta, rank, rcond, ptr = ctx.fit_resident(_capi.SOLVE_LSTSQ_PROBE, 1e-13)
    piv, est, steps, where = _capi.cond_info()
    print(f"{name:34s} K={K:5d} lambda_min {lam:9.2e} pivot {piv:9.2e} estimate {est:9.2e} ratio {est / lam if lam > 0 else float('nan'):8.2f} rank {rank} steps {steps} where {where}", flush=True)


for K in (232, 256, 480, 1024, 1595):
    m = 3 * K
    check("gaussian", rng.standard_normal((m, K)))
    for kap in (1e2, 1e4, 1e5):
        Q1, _ = np.linalg.qr(rng.standard_normal((m, K)))
        Q2, _ = np.linalg.qr(rng.standard_normal((K, K)))
        check(f"geometric kappa {kap:.0e}", (Q1 * np.logspace(0, -np.log10(kap), K)) @ Q2.T)
        s = np.ones(K); s[-1] = 1.0 / kap
        check(f"one small direction kappa {kap:.0e}", (Q1 * s) @ Q2.T)
    for blk in (14, 18, 22):
        M = np.eye(K); M[K - blk:, K - blk:] = np.eye(blk) - np.triu(np.ones((blk, blk)), 1)
        check(f"hidden block {blk}", rng.standard_normal((m, K)) @ M)
ctx.close()
