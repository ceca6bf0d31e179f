#!/usr/bin/env python
"""Run every secondary kernel DESIGN.md quotes a number for, a few times each, on BASELINE-sized inputs -- meant to be
wrapped in `rocprofv3 --kernel-trace --stats` (profiles/r02_secondary_kernel_stats.csv); prints the algorithmic bytes
per launch of each kernel so that the table can be turned into GB/s."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from fitsnap_amd import _capi                      # noqa: E402
from fitsnap_amd.synthetic import synth_problem    # noqa: E402

m, K = 1_000_000, 128
A, b, w = synth_problem(m, K)
ctx = _capi.HipContext(0)
ctx.upload_rows(A, b)
ctx.set_weights(w)
beta = ctx.fit_resident(_capi.SOLVE_RIDGE, 1e-8)[0]
REP = 12
walls = {}


def timed(name, fn):
    fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(REP):
        fn()
    ctx.sync()
    walls[name] = (time.perf_counter() - t0) / REP * 1e3


timed("predict+sse (fsnap_gemv_rows_k)", lambda: ctx.predict(beta, want_preds=False, want_sse=True))
timed("residual_rhs, one pass (fsnap_residual_rows_k + fold)", lambda: ctx.residual_rhs(beta))
ctx.set_option("fused_residual", 0)
timed("residual_rhs, two passes (fsnap_gemv_rows_k + fsnap_gemvT_rows_k)", lambda: ctx.residual_rhs(beta))
ctx.set_option("fused_residual", 1)
cat = (np.arange(m) // 250) % 120
ctx.error_stats(beta, cat.astype(np.int32), 120)
timed("error_stats (gemv + 2 x fsnap_error_stats_k)", lambda: ctx.error_stats(beta, None, 120))
ctx.set_option("repack", 1)
timed("fit with repack (fsnap_pack_weights_k + syrk + reduce)", lambda: ctx.fit_resident(_capi.SOLVE_RIDGE, 1e-8))
ctx.set_option("repack", 0)
timed("set_weights (staged H2D of 8 MB)", lambda: ctx.set_weights(w))
tr = np.random.default_rng(0).random(m) >= 0.1
mask = tr.astype(np.uint8)
rank = (np.cumsum(mask, dtype=np.int64) - mask).astype(np.int32)
wt = np.ascontiguousarray(w[tr])
ctx.set_weights_train(wt, mask, rank)
timed("set_weights_train (staged H2D of 7.2 MB + fsnap_expand_weights_k)", lambda: ctx.set_weights_train(wt))
ctx.set_weights(w)
timed("lstsq_rows (3 x fsnap_trsm_rows_k + 4 x syrk + gemv + gemvT)", lambda: ctx.lstsq_rows(1e-13))

# K = 31 (kernel 1P, HBM-bound) and the assembly kernel on a 32 MB batch
A31, b31, w31 = synth_problem(m, 31)
c31 = _capi.HipContext(0)
c31.upload_rows(A31, b31)
c31.set_weights(w31)
timed("K=31 statistics (fsnap_syrk_wave_p<2>)", lambda: c31.normal_eq())
beta31 = c31.fit_resident(_capi.SOLVE_RIDGE, 1e-8)[0]
timed("K=31 residual_rhs, one pass (fsnap_residual_rows_k<1>)", lambda: c31.residual_rhs(beta31))
c31.set_option("fused_residual", 0)
timed("K=31 residual_rhs, two passes", lambda: c31.residual_rhs(beta31))
c31.set_option("fused_residual", 1)
nrows, ncoef = 32768, 110
raw = np.random.default_rng(1).standard_normal((nrows, ncoef + 1))
ca = _capi.HipContext(0)
ca.rows_alloc(nrows, ncoef)
plan = dict(src_row=np.arange(nrows), kind=(np.arange(nrows) % 3 == 0).astype(np.int32) * 0 + 1, frac=-np.ones(nrows, dtype=np.int32),
            d=np.ones(nrows), truth=np.zeros(nrows), weight=np.ones(nrows))
timed("assemble 32768 x 110 (fsnap_assemble_k, incl. H2D of the 29 MB batch)",
      lambda: ca.assemble(raw, 0, plan["src_row"], plan["kind"], plan["frac"], plan["d"], plan["truth"], plan["weight"],
                          np.zeros((0, 1)), np.ones(ncoef), 1, ncoef, 0))
bytes_per_launch = {
    "fsnap_gemv_rows_k": (8 * K + 8) * m, "fsnap_gemvT_rows_k": (8 * K + 8) * m, "fsnap_residual_rows_k<4>": (8 * K + 17) * m,
    "fsnap_residual_rows_k<1>(K=31)": (8 * 31 + 17) * m, "fsnap_error_stats_k": 28 * m,
    "fsnap_pack_weights_k": 33 * m, "fsnap_expand_weights_k": 21 * m, "fsnap_trsm_rows_k": 16 * K * m,
    "fsnap_syrk_wave_p(K=31)": (8 * 31 + 16) * m, "fsnap_assemble_k": 16 * ncoef * nrows, "fsnap_qpack_k": 32 * m,
}
print(json.dumps({"rows": m, "K": K, "wall_ms_per_call": walls, "algorithmic_bytes_per_launch": bytes_per_launch}, indent=1))
