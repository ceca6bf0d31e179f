#!/usr/bin/env python
"""Flows that only a job with N > 1 ranks exercises, run with TWO ranks on one GPU (peer-to-peer transport) and checked against the
same flow in one process / against numpy on all rows:
  (a) the collective row-space solve of a wide ill-conditioned system (K = 480, kappa 1e9: device pass factors, factor chain);
  (b) a re-weighting loop on resident rows (keep_resident): perform_fit + error_analysis per candidate;
  (c) ParallelTools.free() in the middle of a job: the next fit joins a new communicator.
python scripts/multi_rank_flows.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def ill_system():
    r = np.random.default_rng(480)
    m, K = 24000, 480
    U, _ = np.linalg.qr(r.standard_normal((m, K)))
    V, _ = np.linalg.qr(r.standard_normal((K, K)))
    X = (U * np.logspace(0, -9, K)) @ V.T
    y = X @ r.standard_normal(K) + 1e-3 * r.standard_normal(m)
    return X, y


def ta():
    d = np.load(os.path.join(ROOT, "tests", "golden", "ta_abw.npz"))
    return d["A"], d["b"], d["w"]


def flows(pt, rank, world):
    from fitsnap_amd.config import Config
    from fitsnap_amd.solvers import solver_factory

    out = {}
    # (a)
    X, y = ill_system()
    sel = (np.arange(len(y)) // 50 % world) == rank
    s = solver_factory.solver("SVD", pt, Config(pt, {"SOLVER": {"solver": "SVD"}}))
    s.perform_fit(X[sel], y[sel], np.ones(int(sel.sum())), trainall=True)
    out["ill_row_space"] = np.array(s.last_row_space is not None)
    if rank == 0:
        out["ill_fit"] = s.fit.copy()
    # (b)
    A, b, w = ta()
    m = len(b)
    mine = (np.arange(m) // 43 % world) == rank
    rt = np.array(["Energy"] * 363 + ["Force"] * 12672 + ["Stress"] * 2178)
    fsd = {"Groups": (np.arange(m) // 43 % 3).astype(str)[mine].tolist(), "Testing": (np.arange(m) % 10 == 9)[mine].tolist(),
           "Row_Type": rt[mine].tolist()}
    train = ~np.asarray(fsd["Testing"])
    s = solver_factory.solver("RIDGE", pt, Config(pt, {"SOLVER": {"solver": "RIDGE"}, "RIDGE": {"alpha": 1e-8}}))
    s.keep_resident = True
    Am, bm = np.ascontiguousarray(A[mine]), np.ascontiguousarray(b[mine])
    for cand in range(4):
        wc = w[mine] * (1.0 + 0.5 * cand * (rt[mine] == "Force"))
        s.perform_fit(Am, bm, wc[train], fs_dict=fsd)
        s.error_analysis(Am, bm, wc, fsd)
        if rank == 0:
            out[f"ga_fit_{cand}"] = s.fit.copy()
            out[f"ga_err_{cand}"] = s.errors[["ncount", "mae", "rmse", "rsq"]].to_numpy(dtype=np.float64)
    # (c)
    pt.free()
    s = solver_factory.solver("RIDGE", pt, Config(pt, {"SOLVER": {"solver": "RIDGE"}, "RIDGE": {"alpha": 1e-8}}))
    s.perform_fit(A[mine], b[mine], w[mine], trainall=True)
    if rank == 0:
        out["after_free_fit"] = s.fit.copy()
    return out


def main():
    from fitsnap_amd.parallel_tools import ParallelTools

    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        pt = ParallelTools(comm="rccl", transport="p2p")
        out = flows(pt, pt.get_rank(), pt.get_size())
        if pt.get_rank() == 0:
            np.savez(os.path.join(sys.argv[2], "multi.npz"), **out)
        pt.all_barrier()
        pt.free()
        return
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for rank in range(2):
            env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", FSNAP_COMM_FILE=os.path.join(tmp, "id"),
                       FSNAP_COMM_TOKEN="flows", HSA_ENABLE_IPC_MODE_LEGACY="0", FSNAP_COMM_TIMEOUT="120")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", tmp], env=env))
        rcs = [p.wait(timeout=900) for p in procs]
        assert rcs == [0, 0], rcs
        multi = dict(np.load(os.path.join(tmp, "multi.npz")))
    pt = ParallelTools()
    single = flows(pt, 0, 1)
    X, y = ill_system()
    ref = np.linalg.lstsq(X, y, rcond=1e-13)[0]
    kappa = 1e9
    e_multi = np.linalg.norm(multi["ill_fit"] - ref) / np.linalg.norm(ref)
    e_single = np.linalg.norm(single["ill_fit"] - ref) / np.linalg.norm(ref)
    print(f"(a) ill-conditioned 24000 x 480, two ranks: row space {bool(multi['ill_row_space'])}, |fit - lstsq| / |lstsq| = {e_multi:.2e} "
          f"(one process: {e_single:.2e}; bar {50 * kappa * np.finfo(float).eps:.1e})")
    assert bool(multi["ill_row_space"]) and e_multi <= max(1e-6, 50 * kappa * np.finfo(float).eps)
    worst = 0.0
    for key in sorted(single):
        if key.startswith("ill_"):
            continue
        a, b = multi[key], single[key]
        rel = float(np.nanmax(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))
        worst = max(worst, rel)
        print(f"(b/c) {key:16s} two ranks vs one process: max relative difference {rel:.2e}")
    assert worst < 1e-6
    print("ok")


if __name__ == "__main__":
    main()
