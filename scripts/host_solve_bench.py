"""Host K x K solve (fsnap_solve: Jacobi scaling + register-blocked Cholesky + sweeps) on the box's CPU: median time per call
for a few K; FSNAP_CHOL_NBK / FSNAP_CHOL_VARIANT / FSNAP_CHOL_THREADS select the factorisation's panel width / variant / threads."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from fitsnap_amd import _capi  # noqa: E402

rng = np.random.default_rng(0)
out = []
for K in [int(x) for x in sys.argv[1:]] or [128, 142, 192, 256]:
    A = rng.standard_normal((4 * K, K)) * (10.0 ** rng.uniform(-2, 2, K))
    G = A.T @ A
    c = A.T @ rng.standard_normal(4 * K)
    ref = np.linalg.solve(G + 1e-8 * np.eye(K), c)
    b, rank, rc = _capi.solve(_capi.SOLVE_RIDGE, 1e-8, G, c)
    ts = []
    for _ in range(400):
        t0 = time.perf_counter()
        _capi.solve(_capi.SOLVE_RIDGE, 1e-8, G, c)
        ts.append(time.perf_counter() - t0)
    out.append(f"K={K}: {np.median(ts) * 1e6:.1f} us (min {min(ts) * 1e6:.1f}), rel {np.linalg.norm(b - ref) / np.linalg.norm(ref):.1e}")
print("; ".join(out))
