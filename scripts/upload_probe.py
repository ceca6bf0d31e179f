"""How fast does a small (8-14 MB) host array reach HBM: the runtime's pageable hipMemcpyAsync (fsnap_dev_upload) against the staged
copy of fsnap_set_weights, for a buffer that is reused and for a fresh one every call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fitsnap_amd import _capi
from fitsnap_amd.synthetic import synth_problem

ctx = _capi.HipContext(0)
for m in (1000000, 1772880):
    A, b, w = synth_problem(m, 8)
    ctx.upload_rows(A, b)
    ctx.set_weights(w)
    ctx.sync()
    d = ctx.dev_alloc(m * 8)

    def med(f, n=30):
        ts = []
        for i in range(n + 5):
            t0 = time.perf_counter()
            f(i)
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts[5:])) * 1e3

    fresh = [w * (1 + 1e-3 * i) for i in range(40)]
    print(f"{m} weights ({m*8/1e6:.1f} MB):")
    print(f"   set_weights + sync, same array        {med(lambda i: (ctx.set_weights(w), ctx.sync())):7.3f} ms")
    print(f"   set_weights + sync, fresh array       {med(lambda i: (ctx.set_weights(fresh[i % 40]), ctx.sync())):7.3f} ms")
    print(f"   pageable hipMemcpy + sync, same array {med(lambda i: ctx.dev_upload(d, w)):7.3f} ms")
    print(f"   pageable hipMemcpy + sync, fresh      {med(lambda i: ctx.dev_upload(d, fresh[i % 40])):7.3f} ms")
    ctx.dev_free(d)
ctx.close()
