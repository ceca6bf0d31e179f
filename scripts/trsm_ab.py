"""Row-space pass A/B on the GPU: kernel 13 (left-looking, 16-column blocks; FSNAP_TRSM_KERNEL=13) against kernel 13B
(128-column panels, the default) through fsnap_lstsq_rows -- same coefficients, phase times from FSNAP_ROWSPACE_TIMING.
Each variant runs in a process of its own (the switch is read once per process)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import sys, time, json, numpy as np
sys.path.insert(0, %r)
from fitsnap_amd import _capi
m, K, kappa = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
rng = np.random.default_rng(m + K)
A = rng.standard_normal((m, K)) * np.logspace(0, -np.log10(kappa), K)       # graded columns: kappa(A) ~ kappa
b = A @ rng.standard_normal(K) + 1e-3 * rng.standard_normal(m)
w = rng.uniform(0.5, 2.0, m)
w[rng.random(m) < 0.1] = 0.0
ctx = _capi.HipContext(0)
ctx.upload_rows(A, b)
ctx.set_weights(w)
beta, rank, info = ctx.lstsq_rows(1e-13)
ctx.sync()
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    beta, rank, info = ctx.lstsq_rows(1e-13)
    ts.append(time.perf_counter() - t0)
print("RESULT " + json.dumps({"ms": min(ts) * 1e3, "rank": rank, "info": info, "beta": beta.tolist()}))
''' % ROOT

shapes = [(15213, 1595, 1e3), (367900, 480, 1e3), (100000, 256, 1e6), (5000, 200, 1e2), (40001, 300, 1e9), (3000, 1024, 1e2)]
if len(sys.argv) > 1:
    shapes = [tuple(float(x) if i == 2 else int(x) for i, x in enumerate(a.split("x"))) for a in sys.argv[1:]]
for m, K, kappa in shapes:
    res = {}
    for kern in ("13", "13B"):
        env = dict(os.environ, FSNAP_ROWSPACE_TIMING="1")
        if kern == "13":
            env["FSNAP_TRSM_KERNEL"] = "13"
        else:
            env.pop("FSNAP_TRSM_KERNEL", None)
        out = subprocess.run([sys.executable, "-c", WORKER, str(m), str(K), str(kappa)], env=env, capture_output=True, text=True,
                             timeout=900)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
        if out.returncode != 0 or not line:
            print(f"{m} x {K} kernel {kern}: FAILED rc={out.returncode}\n{out.stderr[-1500:]}")
            continue
        r = json.loads(line[0][7:])
        passes = [l for l in out.stderr.splitlines() if "TRSM pass" in l]
        r["trsm_ms"] = [float(l.split()[-2]) for l in passes][-int(r["info"]["passes"]):]
        res[kern] = r
        print(f"{m} x {K} kappa {kappa:g} kernel {kern}: call {r['ms']:.2f} ms, passes {r['info']['passes']:.0f}, rank {r['rank']}, "
              f"factor upload + TRSM per pass (ms) {r['trsm_ms']}, converged {r['info']['converged']}")
    if len(res) == 2:
        import numpy as np
        a, b = np.array(res["13"]["beta"]), np.array(res["13B"]["beta"])
        print(f"    max |beta_13B - beta_13| / max |beta| = {np.max(np.abs(a - b)) / np.max(np.abs(a)):.3e}")
