#!/usr/bin/env python
"""HBM-bound row kernels at odd widths (K = 31: the Ta example, K = 1595: quadratic SNAP), where the rows are not 16-byte
aligned: predictions + SSE (fsnap_gemv_rows_k), refinement right-hand side (+ fsnap_gemvT_rows_k), stand-alone
weighting (fsnap_weight_rows_k).  Wall time per call and GB/s of the algorithmic bytes; run under rocprofv3 for kernel times.
usage: streaming_kernels_odd_k.py [rows cols]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from fitsnap_amd import _capi                      # noqa: E402
from fitsnap_amd.synthetic import synth_problem    # noqa: E402

m, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4_000_000, 31)
A, b, w = synth_problem(m, K)
ctx = _capi.HipContext(0)
ctx.upload_rows(A, b)
ctx.set_weights(w)
beta = ctx.fit_resident(_capi.SOLVE_RIDGE, 1e-8)[0]
REP = 20


def timed(name, fn, nbytes):
    fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(REP):
        fn()
    ctx.sync()
    ms = (time.perf_counter() - t0) / REP * 1e3
    print(f"{m} x {K} {name}: {ms:.3f} ms per call = {nbytes / ms / 1e6:.0f} GB/s of {nbytes / 1e6:.0f} MB (wall, host side included)")


preds, sse = ctx.predict(beta, want_preds=True, want_sse=True)
ref = A @ beta
print("max |preds - A beta| / max |A beta| =", float(np.max(np.abs(preds - ref)) / np.max(np.abs(ref))))
timed("predict + sse", lambda: ctx.predict(beta, want_preds=False, want_sse=True), (8 * K + 24) * m)
s = ctx.residual_rhs(beta)[0]
u = w * w * (b - ref)
print("max |A^T u - ref| / max |ref| =", float(np.max(np.abs(s - A.T @ u)) / np.max(np.abs(A.T @ u))))
timed("residual_rhs", lambda: ctx.residual_rhs(beta), 2 * (8 * K + 16) * m)
d_aw = ctx.dev_alloc(m * K * 8)
d_bw = ctx.dev_alloc(m * 8)
wms = []
for i in range(8):
    ctx.weight_rows_device(d_aw, K, d_bw)
    wms.append(ctx.timing()["weight_ms"])
wms = float(np.mean(wms[2:]))
nb = (16 * K + 24) * m
print(f"{m} x {K} weight_rows kernel: {wms * 1e3:.1f} us = {nb / wms / 1e6:.0f} GB/s (HIP events)")
aw, bw = ctx.weight_rows()
print("weighted rows bit-identical to numpy:", bool(np.array_equal(aw, w[:, None] * A) and np.array_equal(bw, w * b)))
ctx.close()
