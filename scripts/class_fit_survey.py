"""perform_fit of the SVD and RIDGE plugin classes (keep_resident, weights handed in every call) at the shapes of the README table,
next to the C-ABI fit times of bench.py: is anything hiding between the class and the library?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fitsnap_amd.config import Config
from fitsnap_amd.parallel_tools import ParallelTools
from fitsnap_amd.solvers import solver_factory
from fitsnap_amd.synthetic import synth_problem

shapes = [(15213, 31), (1772880, 110), (1000000, 128), (13035, 142), (100000, 272), (367900, 480), (15213, 1595)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for m, K in shapes:
    A, b, w = synth_problem(m, K)
    row = f"{m:>8d} x {K:<5d}"
    for name, extra in (("SVD", {}), ("RIDGE", {"RIDGE": {"alpha": 1e-8}})):
        pt = ParallelTools()
        s = solver_factory.solver(name, pt, Config(pt, dict({"SOLVER": {"solver": name}}, **extra)))
        s.keep_resident = True
        ts = []
        for i in range(12):
            t0 = time.perf_counter()
            s.fit = None
            s.perform_fit(A, b, w, trainall=True)
            ts.append(time.perf_counter() - t0)
        row += f"   {name} {np.median(ts[3:])*1e3:8.3f} ms (refine steps {getattr(s, 'last_refine_steps', 0)}, first call {ts[0]*1e3:7.1f})"
        pt.free()
    print(row, flush=True)
    del A, b, w
