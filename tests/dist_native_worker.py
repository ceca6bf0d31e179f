"""Worker of the two-process native test (tests/test_gpu_native_comm.py; RCCL: one process per GPU, peer-to-peer: the ranks may share one), launched with
RANK / WORLD_SIZE / LOCAL_RANK / FSNAP_COMM_FILE in the environment; no torch.  Every rank owns the "configurations"
(blocks of 43 rows) i with i % world == rank of the golden Ta set (the reference's row partition,
fitsnap3lib/parallel_tools.py:612-651), fits with SVD and RIDGE, runs the error analysis and an ill-conditioned
row-space solve, and writes what it saw to <outdir>/rank<r>.npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(outdir):
    from fitsnap_amd.config import Config
    from fitsnap_amd.parallel_tools import ParallelTools
    from fitsnap_amd.solvers import solver_factory

    pt = ParallelTools(comm="rccl")
    rank, world = pt._rank, pt._size
    d = np.load(os.path.join(ROOT, "tests", "golden", "ta_abw.npz"))
    A, b, w = d["A"], d["b"], d["w"]
    m = len(b)
    testing = np.random.default_rng(12345).random(m) < 0.1
    mine = (np.arange(m) // 43 % world) == rank
    out = {"torch_imported": np.array("torch" in sys.modules)}
    assert pt.get_ncpn(int(mine.sum())) == m
    assert pt.bcast_object({"hello": rank}, src=0) == {"hello": 0}
    assert pt.allgather_object(("r", rank)) == [("r", q) for q in range(world)]
    for name, extra in (("SVD", {}), ("RIDGE", {"RIDGE": {"alpha": 1e-8}})):
        cfg = Config(pt, dict({"SOLVER": {"solver": name}}, **extra))
        s = solver_factory.solver(name, pt, cfg)
        s.perform_fit(A[mine], b[mine], w[mine][~testing[mine]], fs_dict={"Testing": testing[mine].tolist()})
        assert (s.fit is not None) == (rank == 0)
        G, c, sc = s.last_statistics
        out[f"{name}_G"], out[f"{name}_c"], out[f"{name}_sc"] = G, c, sc
        if rank == 0:
            out[f"{name}_fit"] = s.fit.copy()
        rt = np.array(["Energy"] * 363 + ["Force"] * 12672 + ["Stress"] * 2178)
        fsd = {"Groups": (np.arange(m) // 43 % 3).astype(str)[mine].tolist(), "Testing": testing[mine].tolist(),
               "Row_Type": rt[mine].tolist()}
        s.error_analysis(A[mine], b[mine], w[mine], fsd)
        if rank == 0:
            out[f"{name}_errors"] = s.errors[["ncount", "mae", "rmse", "rsq"]].to_numpy(dtype=np.float64)
            out[f"{name}_errors_index"] = np.array(["|".join(str(x) for x in ix) for ix in s.errors.index])
    # ill-conditioned system, rows dealt to the ranks: collective row-space solve
    r = np.random.default_rng(77)
    mm, K = 16000, 40
    U, _ = np.linalg.qr(r.standard_normal((mm, K)))
    V, _ = np.linalg.qr(r.standard_normal((K, K)))
    X = (U * np.logspace(0, -10, K)) @ V.T
    y = X @ r.standard_normal(K) + 1e-3 * r.standard_normal(mm)
    sel = (np.arange(mm) % world) == rank
    cfg = Config(pt, {"SOLVER": {"solver": "SVD"}})
    s = solver_factory.solver("SVD", pt, cfg)
    s.perform_fit(X[sel], y[sel], np.ones(int(sel.sum())), trainall=True)
    assert s.last_row_space is not None and s.last_row_space["converged"] == 1.0
    if rank == 0:
        out["ill_fit"] = s.fit.copy()
    # a rank WITHOUT rows: everything on rank 0, the other ranks hold rows of the fits above and must drop them
    cfg = Config(pt, {"SOLVER": {"solver": "RIDGE"}, "RIDGE": {"alpha": 1e-8}})
    s = solver_factory.solver("RIDGE", pt, cfg)
    if rank == 0:
        s.perform_fit(A, b, w, trainall=True)
        out["zero_fit"] = s.fit.copy()
    else:
        s.perform_fit(A[:0], b[:0], w[:0], trainall=True)
    out["zero_G"] = s.last_statistics[0]
    out["zero_rows_resident"] = np.array(pt.hip().m)
    # K = 480 through the C ABI: all-reduce in HBM, blocked Cholesky on the GPU of every rank
    from fitsnap_amd import _capi

    r = np.random.default_rng(480)
    A4, b4, w4 = r.standard_normal((6000, 480)), r.standard_normal(6000), r.uniform(0.5, 2.0, 6000)
    sel = (np.arange(6000) // 100 % world) == rank
    ctx = pt.hip()
    ctx.upload_rows(A4[sel], b4[sel])
    ctx.set_weights(w4[sel])
    out["k480_beta"] = ctx.fit_dist(_capi.SOLVE_RIDGE, 1e-8, 480)[0]
    out["transport"] = np.array(ctx.comm_transport())
    # K = 1595 (the quadratic SNAP width): triangle payload (10 MB) through the collective, device Cholesky, and the
    # default solver's refinement with the factor kept on every rank
    r = np.random.default_rng(1595)
    A5, b5 = r.standard_normal((4000, 1595)), r.standard_normal(4000)
    sel = (np.arange(4000) // 100 % world) == rank
    cfg = Config(pt, {"SOLVER": {"solver": "SVD"}})
    s = solver_factory.solver("SVD", pt, cfg)
    s.perform_fit(A5[sel], b5[sel], np.ones(int(sel.sum())), trainall=True)
    if rank == 0:
        out["k1595_fit"] = s.fit.copy()
    out["k1595_rcond"] = np.array(s.last_rcond)
    out["k1595_steps"] = np.array(s.last_refine_steps)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
    pt.all_barrier()
    pt.free()


if __name__ == "__main__":
    main(sys.argv[1])
