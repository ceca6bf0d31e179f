"""perform_fit of the solver classes that consume the same statistics (ARD, LASSO, ANL) next to SVD / RIDGE, on one GPU:
is anything hiding between the class and the library?  usage: python scripts/consumer_fit_survey.py [rows x K ...]"""
import os, sys, time, tempfile, cProfile, pstats, io
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fitsnap_amd.config import Config
from fitsnap_amd.parallel_tools import ParallelTools
from fitsnap_amd.solvers import solver_factory
from fitsnap_amd.synthetic import synth_problem

SOLVERS = (("SVD", {}), ("RIDGE", {"RIDGE": {"alpha": 1e-8}}), ("ARD", {}), ("ANL", {}), ("LASSO", {"LASSO": {"alpha": 1e-6, "max_iter": 5000}}))
shapes = [(15213, 31), (1000000, 128)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:] if "x" in a]
prof = "--profile" in sys.argv
tmp = tempfile.mkdtemp()
os.chdir(tmp)                                 # (ANL writes covariance.npy / mean.npy)
for m, K in shapes:
    A, b, w = synth_problem(m, K)
    for name, extra in SOLVERS:
        pt = ParallelTools()
        s = solver_factory.solver(name, pt, Config(pt, dict({"SOLVER": {"solver": name}}, **extra)))
        s.keep_resident = True
        if name in ("ARD", "LASSO"):          # the reference's signature takes no arrays (ard.py:15, lasso.py:15)
            pt.create_shared_array("a", m, K)
            pt.create_shared_array("b", m)
            pt.create_shared_array("w", m)
            pt.shared_arrays["a"].array[:] = A
            pt.shared_arrays["b"].array[:] = b
            pt.shared_arrays["w"].array[:] = w
            pt.fitsnap_dict["Testing"] = [False] * m
            call = lambda: s.perform_fit()
        else:
            call = lambda: s.perform_fit(A, b, w, trainall=True)
        ts = []
        for i in range(8):
            t0 = time.perf_counter()
            s.fit = None
            call()
            ts.append(time.perf_counter() - t0)
        print(f"{m:>8d} x {K:<4d} {name:6s} {np.median(ts[2:]) * 1e3:9.3f} ms per fit (first call {ts[0] * 1e3:8.1f} ms)", flush=True)
        if prof and name in ("ARD", "ANL", "LASSO"):
            pr = cProfile.Profile()
            pr.enable()
            s.fit = None
            call()
            pr.disable()
            out = io.StringIO()
            pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(14)
            print("\n".join(out.getvalue().splitlines()[4:26]), flush=True)
        pt.free()
    del A, b, w
