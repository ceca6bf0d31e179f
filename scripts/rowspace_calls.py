"""Per-call wall time of the SVD solver class on an ill-conditioned copy of the benchmark problem (the `row_space` leg of
bench.py): which call pays what.  Run with FSNAP_ROWSPACE_TIMING=1 for the phases inside fsnap_lstsq_rows."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fitsnap_amd.config import Config
from fitsnap_amd.parallel_tools import ParallelTools
from fitsnap_amd.solvers import solver_factory
from fitsnap_amd.synthetic import synth_problem

m, K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000, 128
A, b, w = synth_problem(m, K)
Ai = A.copy()
Ai[:, K - 1] = Ai[:, 0] * (np.linalg.norm(A[:, K - 1]) / np.linalg.norm(A[:, 0])) + 1.0e-9 * A[:, K - 1]
pt = ParallelTools()
sv = solver_factory.solver("SVD", pt, Config(pt, {"SOLVER": {"solver": "SVD"}}))
sv.keep_resident = True
for X, name, n in ((A, "well conditioned", 4), (Ai, "ill conditioned", 8)):
    for i in range(n):
        t0 = time.perf_counter()
        sv.fit = None
        sv.perform_fit(X, b, w, trainall=True)
        dt = time.perf_counter() - t0
        rs = sv.last_row_space
        print(f"{name} call {i}: {dt*1e3:8.3f} ms  refine steps {sv.last_refine_steps}  row space {None if rs is None else {k: (float(v) if np.ndim(v) == 0 else v) for k, v in rs.items()}}", file=sys.stderr, flush=True)
pt.free()
