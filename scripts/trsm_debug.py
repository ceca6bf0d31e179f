"""Debug harness: the failing row-space test problem (K = 200, m = 6001) under kernel 13 and 13B, with variations."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
sys.path.insert(0, %r + "/tests")
from fitsnap_amd import _capi
K, m, kappa, nan, graded = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
r = np.random.default_rng(K)
def conditioned(m, K, kappa, seed):
    rr = np.random.default_rng(seed)
    U, _ = np.linalg.qr(rr.standard_normal((m, K)))
    V, _ = np.linalg.qr(rr.standard_normal((K, K)))
    s = np.ones(K); s[-1] = 1.0 / kappa
    return (U * s) @ V.T
A = conditioned(m, K, kappa, K + 1)
if graded:
    A = A * (10.0 ** r.uniform(-2, 2, size=K))
b = r.standard_normal(m)
w = r.uniform(0.5, 2.0, m)
t = r.random(m) < 0.15
if nan:
    A[t] = np.nan
ctx = _capi.HipContext(0)
ctx.upload_rows(A, b)
ctx.set_weights(w, (~t).astype(np.uint8))
beta, rank, info = ctx.lstsq_rows(1e-13)
print("RESULT " + json.dumps({"rank": rank, "info": info, "norm": float(np.linalg.norm(beta)), "beta": beta[:4].tolist()}))
''' % (ROOT, ROOT)
cases = [(208, 6001, 1e3, 0, 0), (176, 6001, 1e3, 0, 0), (192, 6001, 1e3, 0, 0), (224, 6001, 1e3, 0, 0), (240, 6001, 1e3, 0, 0), (256, 6001, 1e3, 0, 0),
         (336, 6001, 1e3, 0, 0), (352, 6001, 1e3, 0, 0), (208, 64, 1e3, 0, 0), (208, 40000, 1e3, 0, 0)]
for c in cases:
    for kern in ("13B", "13B-TR2", "13B-TR1"):
        env = dict(os.environ)
        if kern == "13B-TR2":
            env["FSNAP_TRSM_TR"] = "2"
        if kern == "13B-TR1":
            env["FSNAP_TRSM_TR"] = "1"
        if kern == "13":
            env["FSNAP_TRSM_KERNEL"] = "13"
        out = subprocess.run([sys.executable, "-c", WORKER] + [str(x) for x in c], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(c, kern, "FAILED", out.stderr[-800:])
            continue
        r = json.loads(line[0][7:])
        print(c, kern, "rank", r["rank"], "norm %.6e" % r["norm"], "passes", r["info"]["passes"], "dev %.2e" % r["info"]["deviation"],
              "conv", r["info"]["converged"], "svd", r["info"]["svd"], "smin %.3e" % r["info"]["sigma_min"], "refine %.2e" % r["info"]["refine_step"])
