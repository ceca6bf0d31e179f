"""Hand the 128-byte RCCL communicator id from rank 0 to every rank of a job, without MPI and without torch.

The reference bootstraps through MPI (``MPI.COMM_WORLD``, fitsnap3lib/parallel_tools.py:148-200) and would pass the id
with ``comm.bcast``.  This package is launched one process per GPU by ``torchrun`` (or by hand with RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT in the environment); the id travels

* through a file when all ranks share a node (``LOCAL_WORLD_SIZE == WORLD_SIZE``, the default; or ``FSNAP_COMM_FILE``
  names the file): rank 0 writes it atomically, the others poll for it, rank 0 removes it once everybody has joined;
* through a TCP socket on ``MASTER_ADDR : MASTER_PORT + FSNAP_COMM_PORT_OFFSET`` otherwise (``torchrun``'s own store
  occupies MASTER_PORT itself).

``exchange(rank, world, make_id)`` returns the id on every rank; ``done(rank)`` is called after the collective
``fsnap_comm_init`` and lets rank 0 clean up.
"""
from __future__ import annotations

import os
import socket
import tempfile
import time

ID_BYTES = 128
_TIMEOUT_S = float(os.environ.get("FSNAP_COMM_TIMEOUT", "300"))


def _parent_start_epoch():
    """Start time of the parent process (the launcher all local ranks share), seconds since the epoch; 0 if unknown."""
    try:
        with open(f"/proc/{os.getppid()}/stat") as f:
            ticks = int(f.read().rsplit(")", 1)[1].split()[19])
        with open("/proc/stat") as f:
            btime = next(int(line.split()[1]) for line in f if line.startswith("btime"))
        return btime + ticks / os.sysconf("SC_CLK_TCK")
    except Exception:
        return 0.0


def _default_file():
    port = os.environ.get("MASTER_PORT", "0")
    run = os.environ.get("TORCHELASTIC_RUN_ID", "none")
    restart = os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
    name = f"fsnap_comm_{os.getuid()}_{os.getppid()}_{port}_{run}_{restart}".replace("/", "_")
    return os.path.join(tempfile.gettempdir(), name)


def _single_node(world):
    return int(os.environ.get("LOCAL_WORLD_SIZE", world)) == world


def _via_file(path, rank, make_id, fresh_after):
    if rank == 0:
        ident = make_id()
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            f.write(ident)
        os.replace(tmp, path)                      # atomic: a reader sees nothing or all 128 bytes
        return ident
    deadline = time.monotonic() + _TIMEOUT_S
    while time.monotonic() < deadline:
        try:
            st = os.stat(path)
            if st.st_size == ID_BYTES and st.st_mtime >= fresh_after:
                with open(path, "rb") as f:
                    ident = f.read()
                if len(ident) == ID_BYTES:
                    return ident
        except FileNotFoundError:
            pass
        time.sleep(0.002)
    raise TimeoutError(f"rank {rank}: no communicator id at {path} after {_TIMEOUT_S:.0f} s")


def _via_tcp(rank, world, make_id):
    host = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("MASTER_PORT", "29500")) + int(os.environ.get("FSNAP_COMM_PORT_OFFSET", "17"))
    if rank == 0:
        ident = make_id()
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as srv:
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(("", port))
            srv.listen(world)
            srv.settimeout(_TIMEOUT_S)
            for _ in range(world - 1):
                conn, _ = srv.accept()
                with conn:
                    conn.sendall(ident)
        return ident
    deadline = time.monotonic() + _TIMEOUT_S
    while True:
        try:
            with socket.create_connection((host, port), timeout=5.0) as s:
                buf = b""
                while len(buf) < ID_BYTES:
                    part = s.recv(ID_BYTES - len(buf))
                    if not part:
                        break
                    buf += part
                if len(buf) == ID_BYTES:
                    return buf
        except OSError:
            pass
        if time.monotonic() > deadline:
            raise TimeoutError(f"rank {rank}: no communicator id from {host}:{port} after {_TIMEOUT_S:.0f} s")
        time.sleep(0.01)


_state = {"path": None}


def exchange(rank: int, world: int, make_id) -> bytes:
    """Collective: returns rank 0's ``make_id()`` on every rank."""
    if world == 1:
        return make_id()
    path = os.environ.get("FSNAP_COMM_FILE")
    if path or _single_node(world):
        path = path or _default_file()
        _state["path"] = path
        # a file left behind by a crashed job of an earlier launcher is older than our launcher
        return _via_file(path, rank, make_id, fresh_after=_parent_start_epoch() - 1.0 if not os.environ.get("FSNAP_COMM_FILE") else 0.0)
    return _via_tcp(rank, world, make_id)


def done(rank: int):
    """After every rank has joined the communicator: rank 0 removes the id file."""
    path, _state["path"] = _state["path"], None
    if rank == 0 and path:
        try:
            os.remove(path)
        except OSError:
            pass
