#!/usr/bin/env python
"""Streaming accumulation of the normal equations (needs a gfx950 GPU).

The reference's memory-lean driver (examples/library/transpose_trick/example.py:225-249) never builds A: per
configuration it forms (a, b, w), adds aw.T @ aw and aw.T @ bw to running K x K / K x 1 sums, all-reduces them and
solves.  Same flow here with batches of rows: upload a batch, `fsnap_normal_eq_accumulate` adds its statistics to a
packed buffer that stays in HBM, the batch is dropped; one solve at the end (`fsnap_solve_device`).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from fitsnap_amd import _capi                               # noqa: E402

d = np.load(os.path.join(ROOT, "tests", "golden", "ta_abw.npz"))
A, b, w = d["A"], d["b"], d["w"]
ref = np.load(os.path.join(ROOT, "tests", "golden", "ta_reference_fits.npz"))["ridge_sklearn_1e-8_all"]
K = A.shape[1]

ctx = _capi.HipContext(0)
dev = torch.device("cuda", 0)
total = torch.zeros(K * K + K + 3, dtype=torch.float64, device=dev)     # [G | c | b.b, sum b, n] running sums
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
batch = 1000                                                            # "configurations" of 1000 rows
for lo in range(0, A.shape[0], batch):
    hi = min(lo + batch, A.shape[0])
    ctx.upload_rows(A[lo:hi], b[lo:hi])
    ctx.set_weights(w[lo:hi])
    ctx.normal_eq_accumulate(total.data_ptr())
beta, rank, _ = ctx.solve_device(_capi.SOLVE_RIDGE, 1.0e-8, K, total.data_ptr())
print(f"{(A.shape[0] + batch - 1) // batch} batches, rank {rank}, "
      f"max rel. deviation from the reference's RIDGE fit {np.max(np.abs(beta - ref) / np.abs(ref)):.2e}")
ctx.close()
