import sys, time, numpy as np
sys.path.insert(0, ".")
from fitsnap_amd.config import Config
from fitsnap_amd.parallel_tools import ParallelTools
from fitsnap_amd.solvers import solver_factory
m, K = 4001, 1595
rng = np.random.default_rng(1595)
t=time.time(); U, _ = np.linalg.qr(rng.standard_normal((m, K))); V, _ = np.linalg.qr(rng.standard_normal((K, K))); print("qr", time.time()-t, flush=True)
A = (U * np.logspace(0, -9, K)) @ V.T
b = A @ rng.standard_normal(K) + 1e-4 * rng.standard_normal(m)
w = rng.uniform(0.5, 2.0, m)
pt = ParallelTools(); cfg = Config(pt, {"SOLVER": {"solver": "SVD"}}); s = solver_factory.solver("SVD", pt, cfg)
import os
os.environ["FSNAP_SOLVE_TIMING"]="1"; os.environ["FSNAP_ROWSPACE_TIMING"]="1"
t=time.time(); s.perform_fit(A, b, w, trainall=True); print("perform_fit", time.time()-t, s.last_row_space, flush=True)
t=time.time(); s.perform_fit(A, b, w, trainall=True); print("perform_fit again", time.time()-t, flush=True)
