#!/usr/bin/env python
"""Library-mode use of the MI355X linear-fit path (needs a gfx950 GPU).

Fits the Ta SNAP potential from the golden A / b / w matrices (the reference's committed
Descriptors.npy / Truth-Ref.npy / Weights.npy, kept as a test fixture) with the SVD and RIDGE
solvers through the same plugin API a FitSNAP user script uses
(cf. examples/library/*/example.py of the reference: `fs.solver.perform_fit(a, b, w, fs_dict)`).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from fitsnap_amd.config import Config                      # noqa: E402
from fitsnap_amd.parallel_tools import ParallelTools       # noqa: E402
from fitsnap_amd.solvers import solver_factory             # noqa: E402

d = np.load(os.path.join(ROOT, "tests", "golden", "ta_abw.npz"))
A, b, w = d["A"], d["b"], d["w"]
ref = np.load(os.path.join(ROOT, "tests", "golden", "ta_reference_fits.npz"))

pt = ParallelTools()                                        # one process, one GPU ("stubs" mode of the reference)
pt.hip().set_option("timing_every", 1)                      # bracket every kernel launch with events (off by default)
for name, extra, key in (("SVD", {}, "svd_all"), ("RIDGE", {"RIDGE": {"alpha": 1e-8}}, "ridge_sklearn_1e-8_all")):
    cfg = Config(pt, {"SOLVER": {"solver": name}, **extra})
    solver = solver_factory.solver(name, pt, cfg)
    solver.keep_resident = True                             # re-weighting loops: A, b stay in HBM
    solver.perform_fit(A, b, w, trainall=True)
    err = np.max(np.abs(solver.fit - ref[key]) / np.abs(ref[key]))
    t = pt.hip().timing()
    print(f"{name:5s}: {len(solver.fit)} coefficients, max rel. deviation from the reference's own fit {err:.2e}, "
          f"SYRK kernel {t['syrk_ms'] * 1e3:.1f} us")
pt.free()
