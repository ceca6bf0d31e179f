"""Host K x K end of the row-space solve at large K, without a GPU: the factor chain of a kappa = 1e9 system (R1 from a QR of the rows,
R2 near the identity -- what the later CholeskyQR passes leave) through `_capi.rowspace_chain`, phases printed by
FSNAP_ROWSPACE_TIMING=1.  usage: python scripts/chain_host_timing.py [K] [log10 kappa] [repeats] [dependent columns]"""
import os, sys, time
os.environ.setdefault("FSNAP_ROWSPACE_TIMING", "1")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fitsnap_amd import _capi

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1595
lk = float(sys.argv[2]) if len(sys.argv) > 2 else 9.0
rep = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ndep = int(sys.argv[4]) if len(sys.argv) > 4 else 0
rng = np.random.default_rng(5)
U, _ = np.linalg.qr(rng.standard_normal((2 * K, K)))
V, _ = np.linalg.qr(rng.standard_normal((K, K)))
A = (U * np.logspace(0, -lk, K)) @ V.T
if ndep:                      # columns that are combinations of three others: the chain cannot be certified, the factors are multiplied out
    dep = rng.choice(K, ndep, replace=False)
    others = np.setdiff1d(np.arange(K), dep)
    for d in dep:
        A[:, d] = A[:, rng.choice(others, 3, replace=False)] @ rng.standard_normal(3)
R1 = np.linalg.qr(A, mode="r")
R1 = R1 * np.sign(np.diag(R1))[:, None]
R2 = np.eye(K) + 1e-2 * np.triu(rng.standard_normal((K, K))) / K
z = rng.standard_normal(K)
ref = np.linalg.lstsq(R2 @ R1, z, rcond=1e-13)[0]
for i in range(rep):
    t0 = time.perf_counter()
    beta, rank, info = _capi.rowspace_chain([R1, R2], z, 1e-13)
    dt = (time.perf_counter() - t0) * 1e3
    print(f"call {i}: {dt:8.2f} ms  rank {rank}  chain {info['chain']}  cond_bound {info['cond_bound']:.3e}  true cond {10 ** lk:.1e}  "
          f"|beta - ref| / |ref| = {np.abs(beta - ref).max() / np.abs(ref).max():.2e}", flush=True)
