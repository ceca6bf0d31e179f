"""CPU, world_size 2, gloo: the multi-rank path — every rank owns the rows of its own
configurations (config i -> rank i % size, parallel_tools.py:612-651), the packed K x K
statistics are all-reduced, rank 0 solves.  There is no GPU here, so each rank's LOCAL
statistics (the part the HIP kernel computes) are supplied by the oracle; everything else
(sharding, packing, all-reduce, solve, rank-0-only fit) is the product code."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from fitsnap_amd.config import Config
    from fitsnap_amd.parallel_tools import ParallelTools
    from fitsnap_amd.solvers import solver_factory
    from oracle import fitsnap_oracle as orc

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        d = np.load(os.path.join(ROOT, "tests", "golden", "ta_abw.npz"))
        A, b, w = d["A"], d["b"], d["w"]
        m = len(b)
        testing = np.random.default_rng(12345).random(m) < 0.1
        # "configurations" = blocks of 43 rows, dealt round-robin to ranks
        cfg_of_row = np.arange(m) // 43
        mine = (cfg_of_row % world) == rank
        pt = ParallelTools(comm="torch")
        assert pt._rank == rank and pt._size == world and pt.stubs == 0
        assert pt.get_ncpn(int(mine.sum())) == m
        cfg = Config(pt, {"SOLVER": {"solver": "RIDGE"}, "RIDGE": {"alpha": 1e-8}})
        s = solver_factory.solver("RIDGE", pt, cfg)
        # the GPU part of the path, replaced by the checker on this GPU-less box
        s._local_statistics = lambda a, bb, wf, mask, shared: orc.normal_eq(a, bb, wf, testing=(mask == 0))
        pt.create_shared_array("a", int(mine.sum()), A.shape[1])
        pt.create_shared_array("b", int(mine.sum()))
        pt.create_shared_array("w", int(mine.sum()))
        pt.shared_arrays["a"].array[:] = A[mine]
        pt.shared_arrays["b"].array[:] = b[mine]
        pt.shared_arrays["w"].array[:] = w[mine]
        pt.fitsnap_dict["Testing"] = testing[mine].tolist()
        s.perform_fit()
        G, c, sc = s.last_statistics
        fit = s.fit.copy() if s.fit is not None else np.zeros(0)
        # error analysis over rank-sharded rows: local predictions + gather on rank 0
        s.predict_rows = lambda a_=None, b_=None: pt.shared_arrays["a"].array @ s.fit
        rt = np.array(["Energy"] * 363 + ["Force"] * 12672 + ["Stress"] * 2178)
        pt.fitsnap_dict["Row_Type"] = rt[mine].tolist()
        pt.fitsnap_dict["Groups"] = (np.arange(m) // 43 % 3).astype(str)[mine].tolist()
        s.error_analysis()
        ea = np.zeros((0, 4))
        if rank == 0:
            ea = s.errors[["ncount", "mae", "rmse", "rsq"]].to_numpy(dtype=np.float64)
            with open(os.path.join(outdir, "ea_index.txt"), "w") as f:
                f.write("\n".join("|".join(str(x) for x in ix) for ix in s.errors.index))
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), fit=fit, G=G, c=c, sc=sc, ea=ea)
        pt.all_barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_row_sharded_fit_matches_single_process(tmp_path, ta, ta_fits):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    # fit lives on rank 0 only (svd.py:33 / ridge.py:26)
    assert r0["fit"].shape == (31,) and r1["fit"].shape == (0,)
    # every rank holds the same reduced statistics
    assert np.array_equal(r0["G"], r1["G"]) and np.array_equal(r0["c"], r1["c"])
    A, b, w = ta
    from oracle import fitsnap_oracle as orc
    t = ta_fits["testing_mask"]
    G, c, sc = orc.normal_eq(A, b, w, testing=t)
    assert np.max(np.abs(r0["G"] - G) / np.maximum(np.abs(G), 1e-300)) < 1e-9
    assert r0["sc"][2] == sc[2]
    ref = ta_fits["ridge_sklearn_1e-8_mask"]
    assert np.max(np.abs(r0["fit"] - ref) / np.abs(ref)) < 1e-6
    # gathered error table == single-process table built with pandas directly
    import pandas as pd
    m = len(b)
    df = pd.DataFrame({"truths": b, "preds": A @ r0["fit"], "weights": w, "Testing": t,
                       "Row_Type": ["Energy"] * 363 + ["Force"] * 12672 + ["Stress"] * 2178,
                       "Groups": (np.arange(m) // 43 % 3).astype(str)})
    idx = open(tmp_path / "ea_index.txt").read().split("\n")
    assert r1["ea"].shape == (0, 4) and len(idx) == r0["ea"].shape[0] == 2 * (2 * 3 + 3 * 2 * 3)
    for key, row in zip(idx, r0["ea"]):
        g, wt, tt, rt = key.split("|")
        sel = (df["Testing"] == (tt == "Testing")) & (df["Row_Type"] == rt)
        if g != "*ALL":
            sel &= df["Groups"] == g
        sub = df[sel]
        res = sub["truths"] - sub["preds"]
        if wt == "Unweighted":
            assert row[0] == len(sub) and row[1] == pytest.approx(np.mean(np.abs(res)), rel=1e-12)
        else:
            assert row[0] == np.count_nonzero(sub["weights"])
            assert row[1] == pytest.approx(np.mean(np.abs(sub["weights"] * res)), rel=1e-12)
