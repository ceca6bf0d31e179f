"""Is the SYRK kernel time data dependent?  Same shape, kernels 1A (7) and 1L (2), three data sets."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from fitsnap_amd import _capi
from fitsnap_amd import synthetic as orc  # input data only
m, K = 1000000, 128
dev = torch.device("cuda", 0)
ctx = _capi.HipContext(0)
A, b, w = orc.synth_problem(m, K)
sets = {"synthetic (column scales 1..1e-4, w in {100,1,1e-8})": (A, b, w),
        "same A, w = 1": (A, b, np.ones(m)),
        "uniform(-0.5,0.5) A, b, w": (np.random.default_rng(0).random((m, K)) - 0.5, np.random.default_rng(1).random(m) - 0.5,
                                      np.random.default_rng(2).random(m) - 0.5),
        "same A scaled columns removed (A / s)": (A / (10.0 ** (-4.0 * np.arange(K) / (K - 1))), b, w)}
for name, (A_, b_, w_) in sets.items():
    ctx.upload_rows(np.ascontiguousarray(A_), b_)
    ctx.set_weights(w_)
    for k in (7, 2):
        ctx.set_option("kernel", k)
        for _ in range(300):
            ctx.normal_eq_resident()
        ks, _ = ctx.timing_history(100)
        print(f"{name:55s} kernel={k}  {np.mean(ks):.4f} ms")
    ctx.set_option("kernel", 0)
