"""cProfile of a re-weighting loop (perform_fit + error_analysis per candidate, labels in arrays / Categoricals, keep_resident):
where the host time of a candidate goes."""
import cProfile
import pstats
import sys
import time

import numpy as np
import pandas as pd

sys.path.insert(0, ".")
from fitsnap_amd.config import Config
from fitsnap_amd.parallel_tools import ParallelTools
from fitsnap_amd.solvers import solver_factory
from fitsnap_amd import synthetic as orc

m, K, ngroups = 1000000, 128, 40
A, b, w = orc.synth_problem(m, K)
rng = np.random.default_rng(3)
groups = [f"g{g:02d}" for g in np.sort(rng.integers(0, ngroups, size=m))]
testing = rng.random(m) < 0.1
row_type = [("Energy", "Force", "Stress")[i % 3] for i in range(m)]
fsd = {"Groups": pd.Categorical(groups), "Testing": testing, "Row_Type": pd.Categorical(row_type)}
pt = ParallelTools()
cfg = Config(pt, {"SOLVER": {"solver": sys.argv[1] if len(sys.argv) > 1 else "RIDGE"}, "RIDGE": {"alpha": 1e-8}})
s = solver_factory.solver(cfg.sections["SOLVER"].solver, pt, cfg)
s.keep_resident = True


def candidate():
    w_it = w * rng.uniform(0.5, 2.0)
    w_tr = w_it[~testing]
    s.fit = None
    s.perform_fit(A, b, w_tr, fs_dict=fsd)
    s.error_analysis(A, b, w_it, fsd)


for _ in range(3):
    candidate()
t0 = time.perf_counter()
for _ in range(10):
    candidate()
print("per candidate incl. the caller's numpy:", (time.perf_counter() - t0) / 10 * 1e3, "ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    candidate()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
