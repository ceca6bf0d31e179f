"""cProfile of the host side of a re-weighting loop's candidate (perform_fit + error_analysis on resident rows)."""
import cProfile
import pstats
import sys

import numpy as np

sys.path.insert(0, ".")
from fitsnap_amd.config import Config                      # noqa: E402
from fitsnap_amd.parallel_tools import ParallelTools       # noqa: E402
from fitsnap_amd.solvers import solver_factory             # noqa: E402
from fitsnap_amd import synthetic as orc                   # noqa: E402  (input data only)

m, K, ngroups = 1000000, 128, 40
A, b, w = orc.synth_problem(m, K)
rng = np.random.default_rng(3)
groups = [f"g{g:02d}" for g in np.sort(rng.integers(0, ngroups, size=m))]
testing = (rng.random(m) < 0.1).tolist()
row_type = [("Energy", "Force", "Stress")[i % 3] for i in range(m)]
fsd = {"Groups": groups, "Testing": testing, "Row_Type": row_type}
t = np.asarray(testing)
pt = ParallelTools()
cfg = Config(pt, {"SOLVER": {"solver": "RIDGE"}, "RIDGE": {"alpha": 1e-8}})
s = solver_factory.solver("RIDGE", pt, cfg)
s.keep_resident = True
cands = [w * rng.uniform(0.5, 2.0) for _ in range(24)]


def loop(n0, n1):
    for it in range(n0, n1):
        w_it = cands[it]
        s.fit = None
        s.perform_fit(A, b, w_it[~t], fs_dict=fsd)
        s.error_analysis(A, b, w_it, fsd)
        s.errors.iloc[:, 2].to_numpy()[:3]


loop(0, 4)
pr = cProfile.Profile()
pr.enable()
loop(4, 24)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
