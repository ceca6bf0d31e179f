"""CPU: the communicator-id hand-off of a native multi-GPU job (fitsnap_amd/rendezvous.py) between real processes --
file transport (single node) and TCP transport -- with a stand-in for fsnap_comm_id."""
import multiprocessing as mp
import os
import socket

import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, env, q):
    os.environ.update(env)
    from fitsnap_amd import rendezvous

    ident = rendezvous.exchange(rank, world, lambda: bytes([rank + 17]) * rendezvous.ID_BYTES)
    q.put((rank, ident))
    # every rank has the id before anybody may remove the file: stands in for the collective fsnap_comm_init
    import time
    time.sleep(0.3)
    rendezvous.done(rank)


@pytest.mark.parametrize("transport", ["file", "tcp"])
def test_every_rank_receives_rank_zeros_id(tmp_path, transport):
    world = 3
    if transport == "file":
        env = {"FSNAP_COMM_FILE": str(tmp_path / "id"), "FSNAP_COMM_TIMEOUT": "60"}
    else:
        env = {"LOCAL_WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()),
               "FSNAP_COMM_PORT_OFFSET": "0", "FSNAP_COMM_TIMEOUT": "60"}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, env, q)) for r in (2, 1, 0)]      # rank 0 starts last
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == {r: bytes([17]) * 128 for r in range(world)}
    if transport == "file":
        assert not (tmp_path / "id").exists()             # rank 0 cleaned up


def test_single_rank_needs_no_exchange():
    from fitsnap_amd import rendezvous

    assert rendezvous.exchange(0, 1, lambda: b"x" * 128) == b"x" * 128
