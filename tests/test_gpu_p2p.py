"""GPU (-m gpu): the peer-to-peer transport of the exchange step (csrc/fsnap_p2p.cpp) -- the reference's
comm.Allreduce(c), comm.Allreduce(d) (examples/library/transpose_trick/example.py:245-246) and the small collectives of
fitsnap3lib/parallel_tools.py:245-249, 426-441, 562-592 -- with SEVERAL ranks on the one GPU of the test box (hipIpc handles
open between processes that share a device).  The fits through this transport are in tests/test_gpu_native_comm.py and
tests/test_gpu_cli.py; here: the collectives themselves, bit for bit, and the bounded waits."""
import os
import subprocess
import sys

import numpy as np
import pytest

from p2p_worker import rank_data

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(tmp_path, world, scenario, **extra):
    procs = []
    for rank in range(world):
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
        env.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", FSNAP_COMM_FILE=str(tmp_path / "comm_id"),
                   FSNAP_COMM_TOKEN=f"p2p {scenario}", HSA_ENABLE_IPC_MODE_LEGACY="0", FSNAP_COMM_TIMEOUT="60")
        env.update(extra)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "p2p_worker.py"), str(tmp_path), scenario],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=300)[0])
        except subprocess.TimeoutExpired:
            p.kill()
            logs.append(p.communicate()[0] + "\n[killed after 300 s]")
    return procs, logs


@pytest.mark.parametrize("world", [2, 3])
def test_collectives_between_ranks_that_share_one_gpu(tmp_path, world):
    procs, logs = _launch(tmp_path, world, "collectives", FSNAP_P2P_SLOT_MB="1", FSNAP_P2P_MAILBOX_MB="1")
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-4000:]
    res = [dict(np.load(tmp_path / f"rank{r}.npz")) for r in range(world)]
    for n in (1, 2, 3, 16515, 131072, 300001):
        # three in-place all-reduces back to back: x -> N x -> N^2 x -> N^3 x, each the sum in RANK ORDER
        want = rank_data(0, n)
        for q in range(1, world):
            want = want + rank_data(q, n)
        for _ in range(2):
            acc = want.copy()
            for q in range(1, world):
                acc = acc + want
            want = acc
        for r in range(world):
            assert np.array_equal(res[r][f"dev_{n}"], want), (n, r)              # bit-exact: fixed association order
    ranks = np.arange(world)
    for r in range(world):
        assert np.array_equal(res[r]["host_sum"], [np.sum(ranks + 1.0), -np.sum(ranks), 0.5 * world])
        assert np.array_equal(res[r]["host_max"], [world, 0.0, 0.5]) and np.array_equal(res[r]["host_min"], [1.0, -(world - 1.0), 0.5])
        big = rank_data(0, 700001)
        for q in range(1, world):
            big = big + rank_data(q, 700001)
        assert np.array_equal(res[r]["host_big"], big)
        assert bytes(res[r]["gather"]) == b"".join(bytes([q]) * 5 + b"tail" for q in range(world))
        assert res[r]["gather_big_ok"].all()
        assert bytes(res[r]["bcast"]) == b"from the last rank"


def test_a_dead_peer_ends_the_wait_instead_of_hanging(tmp_path):
    procs, logs = _launch(tmp_path, 2, "dead_peer", FSNAP_COMM_TIMEOUT="4")
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-4000:]
    r0 = dict(np.load(tmp_path / "rank0.npz"))
    assert "did not" in str(r0["error"]) and "rank 0 of 2" in str(r0["error"]), str(r0["error"])
    assert 3.0 < float(r0["seconds"]) < 30.0


def test_larger_rows_on_the_same_contexts_between_two_series_of_fits(tmp_path):
    # strong then weak scaling in one job: every buffer of a context is re-allocated between two series of collective fits
    procs, logs = _launch(tmp_path, 2, "reupload")
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-4000:]
    r0, r1 = (dict(np.load(tmp_path / f"rank{r}.npz")) for r in range(2))
    for m in (120000, 260000):
        assert np.array_equal(r0[f"beta_{m}"], r1[f"beta_{m}"])                 # same bits on both ranks
        assert np.array_equal(r0[f"table_{m}"], [[m, 0.0], [m, 1.0]]) and np.array_equal(r0[f"table_{m}"], r1[f"table_{m}"])


def test_every_linear_solver_class_with_two_ranks_matches_one_process():
    # SVD, RIDGE (sklearn and local), ARD, ANL, LASSO on the golden Ta rows dealt to two ranks by configuration i % 2 (the
    # reference's partition, parallel_tools.py:612-651) against the same fits in one process: scripts/multi_rank_solvers.py
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "multi_rank_solvers.py")], capture_output=True, text=True,
                         timeout=900, env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert "worst" in out.stdout and out.stdout.count("two ranks vs one process") == 6


def test_flows_that_only_a_multi_rank_job_exercises():
    # scripts/multi_rank_flows.py: the collective row-space solve of a 24 000 x 480 system with kappa 1e9 (device pass factors,
    # factor chain) against numpy's lstsq on all rows; a re-weighting loop on resident rows (perform_fit + error_analysis per
    # candidate) and ParallelTools.free() in mid-job against the same flow in one process
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "multi_rank_flows.py")], capture_output=True, text=True,
                         timeout=1200, env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.stdout + out.stderr)[-3000:]
