"""Blocked device Cholesky (kernels 8a-8e) vs the host solve: accuracy against numpy and time per solve, for a
range of K.  Run on the GPU box: python scripts/chol_large_test.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fitsnap_amd import _capi  # noqa: E402


def problem(K, seed=0):
    rng = np.random.default_rng(seed)
    m = 4 * K
    A = rng.standard_normal((m, K)) * (10.0 ** rng.uniform(-3, 3, K))    # badly scaled columns: Jacobi scaling matters
    G = A.T @ A
    c = A.T @ rng.standard_normal(m)
    return G, c


class DevBuf:
    """The packed statistics in HBM (raw allocation + upload through the C ABI: no HIP binding of our own)."""

    def __init__(self, ctx, host):
        self.ptr = ctx.dev_alloc(host.nbytes)
        ctx.dev_upload(self.ptr, host)


def main():
    ctx = _capi.HipContext(0)
    # dummy rows so that the context is usable
    ctx.upload_rows(np.ones((8, 4)), np.ones(8))
    ctx.set_weights(np.ones(8))
    alpha = 1e-8
    args = sys.argv[1:]
    sizes = [int(x) for x in args] or [256, 384, 512, 640, 768, 1024, 1280, 1595, 2048]
    for K in sizes:
        G, c = problem(K, K)
        ref = np.linalg.solve(G + alpha * np.eye(K), c)
        host = np.concatenate([G.ravel(), c, np.zeros(3)])
        packed = DevBuf(ctx, host)
        time.sleep(0.3)
        out = {}
        for mode, name in ((1, "gpu"), (2, "host")):
            ctx.set_option("device_solve", mode)
            beta, rank, rc = ctx.solve_device(_capi.SOLVE_RIDGE, alpha, K, packed.ptr)
            for _ in range(3):
                ctx.solve_device(_capi.SOLVE_RIDGE, alpha, K, packed.ptr)
            # median of single-call times: the boxes run under a CPU quota, and a process that has just burnt it (the
            # BLAS threads that built the problem) is descheduled for tens of milliseconds now and then
            ts = []
            for _ in range(15):
                t0 = time.perf_counter()
                ctx.solve_device(_capi.SOLVE_RIDGE, alpha, K, packed.ptr)
                ts.append(time.perf_counter() - t0)
            dt = float(np.median(ts))
            err = np.max(np.abs(beta - ref) / (np.abs(ref) + 1e-300))
            nerr = np.linalg.norm(beta - ref) / np.linalg.norm(ref)
            out[name] = (dt, err, nerr, rank, rc)
        rhs = np.random.default_rng(1).standard_normal(K)
        ctx.set_option("device_solve", 1)
        b2, _, _ = ctx.solve_device(_capi.SOLVE_RIDGE, alpha, K, packed.ptr, rhs=rhs)
        ref2 = np.linalg.solve(G + alpha * np.eye(K), rhs)
        e2 = np.linalg.norm(b2 - ref2) / np.linalg.norm(ref2)
        print(f"K={K:5d}  gpu {out['gpu'][0]*1e3:7.3f} ms (rel {out['gpu'][2]:.2e}, min pivot {out['gpu'][4]:.2e})   "
              f"host {out['host'][0]*1e3:7.3f} ms (rel {out['host'][2]:.2e})   rhs-variant rel {e2:.2e}", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
