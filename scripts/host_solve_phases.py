import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from fitsnap_amd import _capi
ctx = _capi.HipContext(0)
ctx.upload_rows(np.ones((8, 4)), np.ones(8)); ctx.set_weights(np.ones(8))
ctx.set_option("device_solve", 2)
for K in (512, 640):
    rng = np.random.default_rng(K)
    A = rng.standard_normal((4*K, K)) * (10.0 ** rng.uniform(-3, 3, K))
    G = A.T @ A; c = A.T @ rng.standard_normal(4*K)
    packed = torch.from_numpy(np.concatenate([G.ravel(), c, np.zeros(3)])).cuda()
    for rep in range(3):
        sys.stderr.write(f"--- K={K} rep {rep} (device buffer -> pinned staging -> host solve)\n")
        ctx.solve_device(_capi.SOLVE_RIDGE, 1e-8, K, packed.data_ptr())
    sys.stderr.write(f"--- K={K} plain host arrays\n")
    _capi.solve(_capi.SOLVE_RIDGE, 1e-8, G, c)
    _capi.solve(_capi.SOLVE_RIDGE, 1e-8, G, c)
