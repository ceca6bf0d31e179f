#!/usr/bin/env python
"""Every linear solver class with TWO ranks on one GPU (peer-to-peer transport) against the same fit in one process: rows dealt by
configuration i % 2, fits compared on rank 0.   python scripts/multi_rank_solvers.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SOLVERS = (("SVD", {}), ("RIDGE", {"RIDGE": {"alpha": 1e-6}}), ("RIDGE", {"RIDGE": {"alpha": 1e-6, "local_solver": 1}}), ("ARD", {}),
           ("ANL", {}), ("LASSO", {"LASSO": {"alpha": 1e-6, "max_iter": 5000}}))


def data():
    d = np.load(os.path.join(ROOT, "tests", "golden", "ta_abw.npz"))
    return d["A"], d["b"], d["w"]


def fits(pt, rank, world, outdir):
    from fitsnap_amd.config import Config
    from fitsnap_amd.solvers import solver_factory

    A, b, w = data()
    m = len(b)
    mine = (np.arange(m) // 43 % world) == rank
    out = {}
    for i, (name, extra) in enumerate(SOLVERS):
        cfg = Config(pt, dict({"SOLVER": {"solver": name}}, **extra))
        s = solver_factory.solver(name, pt, cfg)
        cwd = os.getcwd()
        os.chdir(outdir)                      # (ANL writes covariance.npy / mean.npy)
        try:
            if name in ("SVD", "RIDGE"):
                s.perform_fit(A[mine], b[mine], w[mine], trainall=True)
            elif name == "ANL":
                s.perform_fit(A[mine], b[mine], w[mine], trainall=True)
            else:                             # ARD / LASSO: the reference's signature takes no arrays (ard.py:15, lasso.py:15)
                mm = int(mine.sum())
                pt.create_shared_array("a", mm, A.shape[1])
                pt.create_shared_array("b", mm)
                pt.create_shared_array("w", mm)
                pt.shared_arrays["a"].array[:] = A[mine]
                pt.shared_arrays["b"].array[:] = b[mine]
                pt.shared_arrays["w"].array[:] = w[mine]
                pt.fitsnap_dict["Testing"] = [False] * mm
                s.perform_fit()
        finally:
            os.chdir(cwd)
        if rank == 0:
            out[f"{i}_{name}"] = np.asarray(s.fit, dtype=float).ravel()
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        from fitsnap_amd.parallel_tools import ParallelTools

        pt = ParallelTools(comm="rccl", transport="p2p")
        out = fits(pt, pt.get_rank(), pt.get_size(), sys.argv[2])
        if pt.get_rank() == 0:
            np.savez(os.path.join(sys.argv[2], "multi.npz"), **out)
        pt.all_barrier()
        pt.free()
        return
    from fitsnap_amd.parallel_tools import ParallelTools

    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for rank in range(2):
            env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", FSNAP_COMM_FILE=os.path.join(tmp, "id"),
                       FSNAP_COMM_TOKEN="solvers", HSA_ENABLE_IPC_MODE_LEGACY="0", FSNAP_COMM_TIMEOUT="120")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "worker", tmp], env=env))
        rcs = [p.wait(timeout=900) for p in procs]
        assert rcs == [0, 0], rcs
        multi = dict(np.load(os.path.join(tmp, "multi.npz")))
        pt = ParallelTools()
        single = fits(pt, 0, 1, tmp)
        pt.free()
    worst = 0.0
    for key in single:
        a, b = multi[key], single[key]
        rel = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
        worst = max(worst, rel)
        print(f"{key:12s} two ranks vs one process: relative difference {rel:.2e}  (support {np.count_nonzero(a)} / {np.count_nonzero(b)})")
    print("worst", worst)
    assert worst < 1e-6


if __name__ == "__main__":
    main()
