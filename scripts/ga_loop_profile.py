"""cProfile of one re-weighting candidate (perform_fit + error_analysis on resident rows)."""
import cProfile, pstats, sys, io, numpy as np
sys.path.insert(0, ".")
from fitsnap_amd.config import Config
from fitsnap_amd.parallel_tools import ParallelTools
from fitsnap_amd.solvers import solver_factory
from fitsnap_amd import synthetic as orc  # input data only

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
def cand():
    s.fit = None
    s.perform_fit(A, b, cand.w_tr, fs_dict=fsd)
    s.error_analysis(A, b, cand.w_it, fsd)
cand.w_it = w * 1.1          # the candidate's arrays are the caller's business: built once, outside the profile
cand.w_tr = cand.w_it[~t]
for _ in range(3):
    cand()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    cand()
pr.disable()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(28)
print(st.getvalue()[:6000])
