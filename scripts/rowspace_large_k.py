"""Phases of the row-space solve at the quadratic-SNAP shape (15 213 x 1 595) on an ill-conditioned system (kappa = 1e9, full rank) and
on one with dependent columns (kappa of the kept part 1e4, `ndep` columns combinations of others).  FSNAP_ROWSPACE_TIMING=1 for the
phases inside fsnap_lstsq_rows."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fitsnap_amd.config import Config
from fitsnap_amd.parallel_tools import ParallelTools
from fitsnap_amd.solvers import solver_factory

m, K = int(sys.argv[1]) if len(sys.argv) > 1 else 15213, int(sys.argv[2]) if len(sys.argv) > 2 else 1595
ndep = int(sys.argv[3]) if len(sys.argv) > 3 else 6
rng = np.random.default_rng(1595)
U, _ = np.linalg.qr(rng.standard_normal((m, K)))
V, _ = np.linalg.qr(rng.standard_normal((K, K)))
A1 = (U * np.logspace(0, -9, K)) @ V.T
A2 = (U * np.logspace(0, -4, K)) @ V.T
dep = rng.choice(K, ndep, replace=False)
others = np.setdiff1d(np.arange(K), dep)
for d in dep:
    pick = rng.choice(others, 3, replace=False)
    A2[:, d] = A2[:, pick] @ rng.standard_normal(3)
x = rng.standard_normal(K)
w = rng.uniform(0.5, 2.0, m)
pt = ParallelTools()
sv = solver_factory.solver("SVD", pt, Config(pt, {"SOLVER": {"solver": "SVD"}}))
sv.keep_resident = True
for A, name in ((A1, "kappa 1e9, full rank"), (A2, f"{ndep} dependent columns")):
    b = A @ x + 1e-4 * rng.standard_normal(m)
    for i in range(3):
        t0 = time.perf_counter()
        sv.fit = None
        sv.perform_fit(A, b, w, trainall=True)
        dt = time.perf_counter() - t0
        rs = sv.last_row_space
        print(f"{name} call {i}: {dt*1e3:9.2f} ms  rank {sv.last_rank}  row space {None if rs is None else {k: float(v) for k, v in rs.items()}}", file=sys.stderr, flush=True)
    if os.environ.get("FSNAP_PROFILE_CALL"):
        import cProfile, pstats, io
        pr = cProfile.Profile()
        pr.enable()
        sv.fit = None
        sv.perform_fit(A, b, w, trainall=True)
        pr.disable()
        out = io.StringIO()
        pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(22)
        print(out.getvalue(), file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    ref = np.linalg.lstsq(w[:, None] * A, w * b, rcond=1e-13)[0]
    print(f"   numpy lstsq on the host: {(time.perf_counter()-t0)*1e3:.0f} ms; |fit - ref| / |ref| = {np.linalg.norm(sv.fit - ref) / np.linalg.norm(ref):.2e}", file=sys.stderr, flush=True)
pt.free()
