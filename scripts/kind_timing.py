"""fsnap_fit_resident per solve kind at a few shapes: what the LSTSQ kinds' condition estimate costs beside RIDGE (median ms of 30 fits)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fitsnap_amd import _capi
from fitsnap_amd.synthetic import synth_problem

shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(100000, 272), (367900, 480), (15213, 1595), (1000000, 128)]
ctx = _capi.HipContext(0)
ctx.set_option("timing_every", 0)
for m, K in shapes:
    A, b, w = synth_problem(m, K)
    ctx.upload_rows(A, b)
    ctx.set_weights(w)
    row = f"{m:>8d} x {K:<5d}"
    for name, kind, par in (("RIDGE", _capi.SOLVE_RIDGE, 1e-8), ("LSTSQ_PROBE", _capi.SOLVE_LSTSQ_PROBE, 1e-13), ("CHOL", _capi.SOLVE_CHOL, 0.0)):
        ts = []
        for i in range(40):
            t0 = time.perf_counter()
            ctx.fit_resident(kind, par)
            ts.append(time.perf_counter() - t0)
        row += f"   {name} {np.median(ts[10:]) * 1e3:7.3f} ms"
    print(row, _capi.cond_info(), flush=True)
ctx.close()
