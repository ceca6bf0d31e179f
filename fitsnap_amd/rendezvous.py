"""Hand the 128-byte RCCL communicator id from rank 0 to every rank of a job, without MPI and without torch.

The reference bootstraps through MPI (``MPI.COMM_WORLD``, fitsnap3lib/parallel_tools.py:148-200) and would pass the id
with ``comm.bcast``.  This package is launched one process per GPU by ``torchrun``, by ``bench.py --gpus N`` itself, or by
hand with RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment; the id travels

* through a file when all ranks share a node (``LOCAL_WORLD_SIZE == WORLD_SIZE``, the default; or ``FSNAP_COMM_FILE``
  names the file): rank 0 writes it atomically, the others poll for it, rank 0 removes it once everybody has joined;
* through a TCP socket on ``MASTER_ADDR : MASTER_PORT + FSNAP_COMM_PORT_OFFSET`` otherwise (``torchrun``'s own store
  occupies MASTER_PORT itself).

What keeps a rank from picking up the WRONG id (``ncclCommInitRank`` with mismatched ids does not fail, it waits):

* every message is ``magic | job token | id``.  The token is a digest of something all ranks of THIS job know before
  they talk to each other -- ``FSNAP_COMM_TOKEN`` when set (``bench.py`` sets a random one for the ranks it spawns), else
  the identity of the launcher process they share (pid + start time of the parent) -- and of the GENERATION: the n-th
  communicator a process creates uses the n-th file name and the n-th token, so two communicators created back to back
  can never read each other's id.  A file left by a crashed job carries another token and is ignored (rank 0 also
  unlinks whatever it finds under its name before publishing), and a file that carries the RIGHT token but has not been
  touched since before the reader started is ignored too: a live rank 0 refreshes its file's mtime four times a second
  until ``done()``, a crashed job's file goes stale at once.  Ranks started by hand from different shells have no
  launcher in common: give them the same ``FSNAP_COMM_TOKEN``;
* the default file lives in a directory of this user's own (mode 0700, ownership checked), the file is created 0600.

``exchange(rank, world, make_id)`` returns the id on every rank; ``done(rank)`` is called after the collective
``fsnap_comm_init`` and lets rank 0 clean up.  Every wait is bounded by ``FSNAP_COMM_TIMEOUT`` seconds (default 300).
"""
from __future__ import annotations

import hashlib
import os
import socket
import stat
import tempfile
import threading
import time

_IMPORTED_AT = time.time()          # "this process is at least this old": a published id must not be older (see _via_file)
_FRESH_SLACK_S = 5.0
_KEEPALIVE_S = 0.25

ID_BYTES = 128
_MAGIC = b"FSNAPID1"
_TOKEN_BYTES = 16
_MSG_BYTES = len(_MAGIC) + _TOKEN_BYTES + ID_BYTES


def _timeout_s():
    try:
        t = float(os.environ.get("FSNAP_COMM_TIMEOUT", "300"))
    except ValueError:
        t = 300.0
    return t if t > 0 else 300.0


def _parent_identity():
    """pid and start time (clock ticks since boot) of the parent process: what all ranks of one launcher share."""
    ppid = os.getppid()
    try:
        with open(f"/proc/{ppid}/stat") as f:
            ticks = f.read().rsplit(")", 1)[1].split()[19]
    except Exception:
        ticks = "?"
    return f"{ppid}:{ticks}"


def _token(generation):
    secret = os.environ.get("FSNAP_COMM_TOKEN")
    if not secret:
        secret = "|".join([_parent_identity(), os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                           os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")])
    return hashlib.sha256(f"{secret}#{generation}".encode()).digest()[:_TOKEN_BYTES]


def _private_dir():
    """A directory only this user can write to (a predictable name in a world-writable /tmp invites a planted id)."""
    path = os.path.join(tempfile.gettempdir(), f"fsnap_comm_{os.getuid()}")
    try:
        os.mkdir(path, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(path)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise PermissionError(f"{path} must be a directory owned by uid {os.getuid()} with mode 0700")
    return path


def _default_file(generation):
    port = os.environ.get("MASTER_PORT", "0")
    run = os.environ.get("TORCHELASTIC_RUN_ID", "none")
    restart = os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
    name = f"id_{os.getppid()}_{port}_{run}_{restart}_g{generation}".replace("/", "_")
    return os.path.join(_private_dir(), name)


def _single_node(world):
    return int(os.environ.get("LOCAL_WORLD_SIZE", world)) == world


def _pack(token, ident):
    if len(ident) != ID_BYTES:
        raise ValueError(f"communicator id must be {ID_BYTES} bytes")
    return _MAGIC + token + ident


def _unpack(msg, token):
    """The id if ``msg`` is a complete message of THIS job and generation, else None."""
    if len(msg) != _MSG_BYTES or msg[:len(_MAGIC)] != _MAGIC or msg[len(_MAGIC):len(_MAGIC) + _TOKEN_BYTES] != token:
        return None
    return msg[len(_MAGIC) + _TOKEN_BYTES:]


def _via_file(path, rank, make_id, token):
    if rank == 0:
        try:
            os.unlink(path)                        # whatever an earlier (crashed) job left under this name
        except FileNotFoundError:
            pass
        ident = make_id()
        tmp = f"{path}.{os.getpid()}.tmp"
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(_pack(token, ident))
        os.replace(tmp, path)                      # atomic: a reader sees nothing or the whole message
        # keep the file's mtime current while this job is alive and still joining: a reader ignores a file that has not
        # been touched since before the reader itself started -- that is what a job leaves behind which crashed between
        # publishing and done() and is re-launched under the same token (same parent, same port, no FSNAP_COMM_TOKEN)
        stop = threading.Event()

        def keepalive():
            while not stop.wait(_KEEPALIVE_S):
                try:
                    os.utime(path, None)
                except OSError:
                    return

        threading.Thread(target=keepalive, name="fsnap-comm-id-keepalive", daemon=True).start()
        _state["keepalive"] = stop
        return ident
    timeout = _timeout_s()
    deadline = time.monotonic() + timeout
    foreign = False
    first_mtime = None
    while time.monotonic() < deadline:
        try:
            with open(path, "rb") as f:
                msg = f.read(_MSG_BYTES + 1)
                mtime = os.fstat(f.fileno()).st_mtime
            # "fresh" = written by a LIVE rank 0: either its mtime is not older than this process (same clock: one node, or a
            # file server in step with it, up to _FRESH_SLACK_S), or -- whatever the clocks say (a shared file system whose
            # server is minutes off) -- this reader has seen the mtime MOVE: only the keepalive of a live rank 0 touches it
            if first_mtime is None:
                first_mtime = mtime
            fresh = mtime >= _IMPORTED_AT - _FRESH_SLACK_S or mtime != first_mtime
            ident = _unpack(msg, token)
            if ident is not None and fresh:        # a live rank 0 touches its file every _KEEPALIVE_S seconds
                return ident
            foreign = foreign or len(msg) > 0
        except FileNotFoundError:
            pass
        time.sleep(0.002)
    why = ("a file is there but it was written for another job or generation (ranks that do not share a launcher "
           "process need the same FSNAP_COMM_TOKEN)") if foreign else "rank 0 never published it"
    raise TimeoutError(f"rank {rank}: no communicator id at {path} after {timeout:.0f} s: {why}")


def _via_tcp(rank, world, make_id, token):
    host = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("MASTER_PORT", "29500")) + int(os.environ.get("FSNAP_COMM_PORT_OFFSET", "17"))
    timeout = _timeout_s()
    if rank == 0:
        ident = make_id()
        msg = _pack(token, ident)
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as srv:
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(("", port))
            srv.listen(world)
            srv.settimeout(timeout)
            served = 0
            while served < world - 1:
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    raise TimeoutError(f"rank 0: only {served} of {world - 1} ranks asked for the communicator id on port "
                                       f"{port} within {timeout:.0f} s") from None
                with conn:
                    conn.settimeout(10.0)
                    try:
                        hello = b""
                        while len(hello) < _TOKEN_BYTES:
                            part = conn.recv(_TOKEN_BYTES - len(hello))
                            if not part:
                                break
                            hello += part
                        if hello != token:         # somebody else's rank (another job on this port): not counted
                            continue
                        conn.sendall(msg)
                        served += 1
                    except OSError:
                        continue
        return ident
    deadline = time.monotonic() + timeout
    while True:
        try:
            with socket.create_connection((host, port), timeout=5.0) as s:
                s.sendall(token)
                buf = b""
                while len(buf) < _MSG_BYTES:
                    part = s.recv(_MSG_BYTES - len(buf))
                    if not part:
                        break
                    buf += part
                ident = _unpack(buf, token)
                if ident is not None:
                    return ident
        except OSError:
            pass
        if time.monotonic() > deadline:
            raise TimeoutError(f"rank {rank}: no communicator id from {host}:{port} after {timeout:.0f} s")
        time.sleep(0.01)


_state = {"path": None, "generation": 0, "keepalive": None}


def exchange(rank: int, world: int, make_id) -> bytes:
    """Collective: returns rank 0's ``make_id()`` on every rank.  The n-th call of a process belongs to generation n
    (every rank of a job creates its communicators in the same order)."""
    if world == 1:
        return make_id()
    _stop_keepalive()                              # of an earlier exchange() that was never followed by done()
    generation = _state["generation"]
    _state["generation"] = generation + 1
    token = _token(generation)
    path = os.environ.get("FSNAP_COMM_FILE")
    if path or _single_node(world):
        path = f"{path}.g{generation}" if path else _default_file(generation)
        _state["path"] = path
        try:
            return _via_file(path, rank, make_id, token)
        except BaseException:
            _stop_keepalive()                      # rank 0 failed after publishing: nobody will call done()
            raise
    return _via_tcp(rank, world, make_id, token)


def _stop_keepalive():
    stop, _state["keepalive"] = _state.get("keepalive"), None
    if stop is not None:
        stop.set()


def done(rank: int):
    """After every rank has joined the communicator: rank 0 removes the id file."""
    path, _state["path"] = _state["path"], None
    _stop_keepalive()
    if rank == 0 and path:
        try:
            os.remove(path)
        except OSError:
            pass
