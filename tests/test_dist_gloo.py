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
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), fit=s.fit if s.fit is not None else np.zeros(0),
                 G=G, c=c, sc=sc)
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
