"""Where the time of SVD.perform_fit(a, b, w, trainall=True) with keep_resident goes at 10^6 x 128, next to the same steps through the
C ABI (bench.py's `svd_solver.steps` / `class_perform_fit`)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fitsnap_amd import _capi
from fitsnap_amd.config import Config
from fitsnap_amd.parallel_tools import ParallelTools
from fitsnap_amd.solvers import solver_factory
from fitsnap_amd.synthetic import synth_problem

m, K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000, 128
A, b, w = synth_problem(m, K)
pt = ParallelTools()
sv = solver_factory.solver("SVD", pt, Config(pt, {"SOLVER": {"solver": "SVD"}}))
sv.keep_resident = True


def med(f, n=30, warm=5):
    ts = []
    for i in range(n + warm):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts[warm:])) * 1e3


def fit():
    sv.fit = None
    sv.perform_fit(A, b, w, trainall=True)


print(f"perform_fit (class, weights handed in every call)   {med(fit):7.3f} ms")
ctx = pt.hip()


def up():
    ctx.set_weights(w)
    ctx.sync()


def up_nosync():
    ctx.set_weights(w)


def probe():
    ctx.fit_resident(_capi.SOLVE_LSTSQ_PROBE, 1e-13)


def both():
    ctx.set_weights(w)
    ctx.fit_resident(_capi.SOLVE_LSTSQ_PROBE, 1e-13)


print(f"set_weights + sync (8 MB staged upload)             {med(up):7.3f} ms")
print(f"set_weights, host side only                         {med(up_nosync):7.3f} ms")
ctx.sync()
print(f"fit_resident(LSTSQ_PROBE), weights resident         {med(probe):7.3f} ms")
print(f"set_weights + fit_resident                          {med(both):7.3f} ms")
import cProfile, pstats, io
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    fit()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18)
print(s.getvalue())
pt.free()
