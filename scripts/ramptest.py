"""How long does the GPU take to reach its steady clock?  Per-step time and kernel time of the first steps after idle."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from fitsnap_amd import _capi
from fitsnap_amd import synthetic as orc  # input data only
A, b, w = orc.synth_problem(1000000, 128)
ctx = _capi.HipContext(0); ctx.upload_rows(A, b); ctx.set_weights(w)
torch.cuda.synchronize()
time.sleep(2.0)
rows = []
t00 = time.perf_counter()
for i in range(400):
    t0 = time.perf_counter()
    ptr = ctx.normal_eq_resident()
    beta, _, _ = ctx.solve_device(_capi.SOLVE_RIDGE, 1e-8, 128, ptr)
    dt = time.perf_counter() - t0
    rows.append((time.perf_counter() - t00, dt, ctx.timing(2)["syrk_ms"]))
for lo, hi in ((0, 5), (5, 10), (10, 20), (20, 40), (40, 80), (80, 160), (160, 240), (240, 320), (320, 400)):
    seg = rows[lo:hi]
    print(f"steps {lo:3d}-{hi:3d}  t={seg[0][0]*1e3:7.1f} ms  step {np.mean([r[1] for r in seg])*1e3:.4f} ms  kernel {np.mean([r[2] for r in seg]):.4f} ms")
