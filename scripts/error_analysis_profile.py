"""cProfile of Solver.error_analysis on resident rows (10^6 x 128, 40 groups, package-produced label lists): where the 1.2 ms per
candidate of a re-weighting loop go."""
import os, sys, time, cProfile, pstats, io
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fitsnap_amd.config import Config
from fitsnap_amd.parallel_tools import ParallelTools, LabelList
from fitsnap_amd.solvers import solver_factory
from fitsnap_amd import synthetic as orc

m, K, ngroups = 1000000, 128, 40
A, b, w = orc.synth_problem(m, K)
rng = np.random.default_rng(3)
fsd = {"Groups": LabelList([f"g{g:02d}" for g in np.sort(rng.integers(0, ngroups, size=m))]),
       "Testing": LabelList((rng.random(m) < 0.1).tolist()),
       "Row_Type": LabelList([("Energy", "Force", "Stress")[i % 3] for i in range(m)])}
t = np.asarray(fsd["Testing"])
pt = ParallelTools()
s = solver_factory.solver("RIDGE", pt, Config(pt, {"SOLVER": {"solver": "RIDGE"}, "RIDGE": {"alpha": 1e-8}}))
s.keep_resident = True


def one(i):
    w_it = w * (1.0 + 0.01 * i)
    s.fit = None
    s.perform_fit(A, b, w_it[~t], fs_dict=fsd)
    t0 = time.perf_counter()
    s.error_analysis(A, b, w_it, fsd)
    return time.perf_counter() - t0


for i in range(4):
    one(i)
ts = [one(i) for i in range(10)]
print(f"error_analysis: {np.median(ts)*1e3:.3f} ms per call (median of 10)")
ws = [w * (1.0 + 0.01 * i) for i in range(20)]
pr = cProfile.Profile()
pr.enable()
for i in range(20):
    s.error_analysis(A, b, ws[i], fsd)
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(25)
print(out.getvalue())
pt.free()
