"""Tiled SYRK (K > 128): (split, pair) order inside an XCD (option xcd = 1) against the class-major order (xcd = 2):
bit-identical statistics, kernel time from HIP events."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from fitsnap_amd import _capi

shapes = [(367900, 480), (15213, 1595), (100000, 256), (200000, 1000), (50000, 200)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for m, K in shapes:
    rng = np.random.default_rng(m + K)
    A = rng.standard_normal((m, K))
    b = rng.standard_normal(m)
    w = rng.uniform(0.5, 2.0, m)
    ctx = _capi.HipContext(0)
    ctx.upload_rows(A, b)
    ctx.set_weights(w)
    ctx.set_option("timing_every", 1)
    res = {}
    for order in (1, 2, 1, 2):
        ctx.set_option("xcd", order)
        for _ in range(30):
            G, c, s = ctx.normal_eq()
        ts = []
        for _ in range(10):
            G, c, s = ctx.normal_eq()
            ts.append(ctx.timing(2)["syrk_ms"])
        res.setdefault(order, []).append((float(np.median(ts)), G, c))
    same = all(np.array_equal(res[1][0][1], r[1]) and np.array_equal(res[1][0][2], r[2]) for o in res for r in res[o])
    info = ctx.launch_info()
    fl = (K * K + 3 * K) * m
    print(f"{m} x {K}: splits {info['nsplit']}, kernel ms order1 {[round(r[0], 4) for r in res[1]]} order2 {[round(r[0], 4) for r in res[2]]} "
          f"({fl / (min(r[0] for r in res[2]) * 1e-3) / 1e12:.1f} TF/s with order 2), bit-identical {same}")
    ctx.close()
