"""One-off check beyond 2^31 elements / 2^32 bytes of rows (default 2e7 x 128 fp64 = 20.5 GB): additivity of the statistics
over a row split expressed through the mask, predictions and grouped error statistics at the far end of the buffer,
the row-space orthogonalisation pass.  python scripts/big_rows_check.py [rows] [cols]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fitsnap_amd import _capi                      # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rng = np.random.default_rng(0)
t0 = time.time()
A = np.empty((m, K))
step = 1_000_000
for i in range(0, m, step):                       # block-wise: standard_normal((2e7, 128)) would need a second 20 GB
    A[i:i + step] = rng.standard_normal((min(step, m - i), K))
beta_true = rng.standard_normal(K)
b = A @ beta_true + 1e-3 * rng.standard_normal(m)
w = rng.uniform(0.5, 2.0, m)
print(f"{m} x {K}: {A.nbytes / 2**30:.1f} GiB generated in {time.time() - t0:.1f} s", flush=True)
ctx = _capi.HipContext(0)
t0 = time.time()
ctx.upload_rows(A, b)
print(f"upload {time.time() - t0:.2f} s", flush=True)
ones = np.ones(m, dtype=np.uint8)
ctx.set_weights(w, ones)
G, c, s = ctx.normal_eq()
cut = m // 2 + 12345
first = (np.arange(m) < cut).astype(np.uint8)
ctx.set_weights(w, first)
G1, c1, s1 = ctx.normal_eq()
ctx.set_weights(w, (1 - first).astype(np.uint8))
G2, c2, s2 = ctx.normal_eq()
d = np.sqrt(np.diag(G))
print("additivity  G: %.2e  c: %.2e  scalars: %.2e" % (np.max(np.abs(G1 + G2 - G) / np.outer(d, d)),
                                                       np.max(np.abs(c1 + c2 - c) / (d * np.sqrt(s[0]))),
                                                       np.max(np.abs(s1 + s2 - s) / np.abs(s))))
# the second half on its own against numpy (rows past 2^31 elements)
tail = slice(m - 200_000, m)
last = np.zeros(m, dtype=np.uint8)
last[tail] = 1
ctx.set_weights(w, last)
Gt, ct, st = ctx.normal_eq()
Aw = w[tail, None] * A[tail]
print("last 200 000 rows vs numpy  G: %.2e  c: %.2e  n: %d" % (np.max(np.abs(Gt - Aw.T @ Aw) / np.outer(d, d)),
                                                              np.max(np.abs(ct - Aw.T @ (w[tail] * b[tail])) / (d * np.sqrt(s[0]))),
                                                              int(st[2])))
ctx.set_weights(w, ones)
beta, rank, rc, _ = ctx.fit_resident(_capi.SOLVE_RIDGE, 1e-8)
print("fit vs truth %.2e" % (np.max(np.abs(beta - beta_true)) / np.max(np.abs(beta_true))))
preds, sse = ctx.predict(beta, want_preds=True, want_sse=True)
idx = np.r_[0:5, m // 2:m // 2 + 5, m - 5:m]
print("predictions at the ends / middle: %.2e   sse vs numpy (last 10^6 rows only checked): %.6e" % (
    np.max(np.abs(preds[idx] - A[idx] @ beta)), sse))
r = w[-1_000_000:] * (b[-1_000_000:] - A[-1_000_000:] @ beta)
print("tail residual check %.2e" % abs(np.sum((w[-1_000_000:] * (b[-1_000_000:] - preds[-1_000_000:])) ** 2) / np.sum(r * r) - 1))
b2, rank2, info = ctx.lstsq_rows(1e-13)
print("row-space solve vs the fit %.2e (passes %d)" % (np.max(np.abs(b2 - beta)) / np.max(np.abs(beta)), info["passes"]))
ctx.close()
