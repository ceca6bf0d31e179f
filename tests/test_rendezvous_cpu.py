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
        assert not (tmp_path / "id.g0").exists() and not (tmp_path / "id").exists()             # rank 0 cleaned up


def test_single_rank_needs_no_exchange():
    from fitsnap_amd import rendezvous

    assert rendezvous.exchange(0, 1, lambda: b"x" * 128) == b"x" * 128


def test_stale_and_foreign_files_are_not_accepted(tmp_path, monkeypatch):
    # a file of the right size left by a crashed job (other token), then the real one: the reader must wait for the real one
    import threading
    import time

    from fitsnap_amd import rendezvous

    monkeypatch.setenv("FSNAP_COMM_FILE", str(tmp_path / "id"))
    monkeypatch.setenv("FSNAP_COMM_TOKEN", "this job")
    monkeypatch.setenv("FSNAP_COMM_TIMEOUT", "30")
    monkeypatch.setitem(rendezvous._state, "generation", 0)
    stale = rendezvous._pack(b"\0" * 16, b"S" * 128)
    (tmp_path / "id.g0").write_bytes(stale)
    good = rendezvous._pack(rendezvous._token(0), b"G" * 128)

    def publish():
        time.sleep(0.5)
        tmp = tmp_path / "x.tmp"
        tmp.write_bytes(good)
        os.replace(tmp, tmp_path / "id.g0")

    t = threading.Thread(target=publish)
    t.start()
    assert rendezvous.exchange(1, 2, lambda: b"?" * 128) == b"G" * 128
    t.join()
    # a second communicator of the same process: another generation = another file name and another token
    assert rendezvous._state["generation"] == 1
    monkeypatch.setenv("FSNAP_COMM_TIMEOUT", "1")
    with pytest.raises(TimeoutError, match="rank 0 never published"):
        rendezvous.exchange(1, 2, lambda: b"?" * 128)             # id.g1 does not exist: the g0 file is NOT re-read
    assert rendezvous._token(0) != rendezvous._token(1)


def test_rank_zero_replaces_a_leftover_file_and_writes_it_private(tmp_path, monkeypatch):
    import stat

    from fitsnap_amd import rendezvous

    monkeypatch.setenv("FSNAP_COMM_FILE", str(tmp_path / "id"))
    monkeypatch.setitem(rendezvous._state, "generation", 0)
    (tmp_path / "id.g0").write_bytes(b"junk")
    ident = rendezvous.exchange(0, 2, lambda: b"N" * 128)
    raw = (tmp_path / "id.g0").read_bytes()
    assert ident == b"N" * 128 and rendezvous._unpack(raw, rendezvous._token(0)) == ident
    assert stat.S_IMODE(os.stat(tmp_path / "id.g0").st_mode) == 0o600
    rendezvous.done(0)
    assert not (tmp_path / "id.g0").exists()
    d = rendezvous._private_dir()
    assert stat.S_IMODE(os.stat(d).st_mode) == 0o700 and os.stat(d).st_uid == os.getuid()


def test_a_file_of_this_job_left_by_a_crashed_launch_is_not_accepted(tmp_path, monkeypatch):
    # same token (same parent, same port, no FSNAP_COMM_TOKEN: a re-launch after a crash between publish and done()), but
    # the file has not been touched since long before this reader started: ignored until the live rank 0 publishes
    import threading
    import time

    from fitsnap_amd import rendezvous

    monkeypatch.setenv("FSNAP_COMM_FILE", str(tmp_path / "id"))
    monkeypatch.setenv("FSNAP_COMM_TOKEN", "relaunched job")
    monkeypatch.setenv("FSNAP_COMM_TIMEOUT", "30")
    monkeypatch.setitem(rendezvous._state, "generation", 0)
    old = rendezvous._pack(rendezvous._token(0), b"O" * 128)          # the RIGHT token, the WRONG (dead) communicator
    (tmp_path / "id.g0").write_bytes(old)
    past = time.time() - 3600.0
    os.utime(tmp_path / "id.g0", (past, past))
    got = {}

    def rank0():
        time.sleep(0.5)
        got["id0"] = rendezvous._via_file(str(tmp_path / "id.g0"), 0, lambda: b"L" * 128, rendezvous._token(0))

    t = threading.Thread(target=rank0)
    t.start()
    assert rendezvous.exchange(1, 2, lambda: b"?" * 128) == b"L" * 128
    t.join()
    assert got["id0"] == b"L" * 128
    # the live rank 0 keeps its file fresh until done()
    m0 = os.stat(tmp_path / "id.g0").st_mtime
    time.sleep(3 * rendezvous._KEEPALIVE_S + 0.1)
    assert os.stat(tmp_path / "id.g0").st_mtime > m0
    rendezvous.done(0)
    time.sleep(2 * rendezvous._KEEPALIVE_S)
    assert rendezvous._state["keepalive"] is None
