"""MI355X-side counterpart of the reference's runtime / shared-memory layer for the
linear-fit hot path (fitsnap3lib/parallel_tools.py:148-1077).

Same public surface as the reference for the pieces the hot path touches —
``ParallelTools(comm=None)``, ``create_shared_array(name, size1, size2=1, dtype='d', tm=0)``,
``shared_arrays[name].array``, ``fitsnap_dict``, ``add_2_fitsnap``, ``DistributedList``,
``SharedArray`` / ``StubsArray``, ``gather_fitsnap``, ``all_barrier``, ``single_print``,
``rank_zero`` / ``sub_rank_zero``, ``single_timeit``, ``free`` — with a different engine
underneath:

* the reference shares ONE host copy of A between the MPI ranks of a node through an
  MPI-3 shared-memory window (``Win.Allocate_shared``, parallel_tools.py:992-1009) and
  solves on rank 0.  Here there is one process per GPU (``torch.distributed``, RCCL over
  xGMI); every rank owns the rows of ITS configurations (the reference's per-proc row
  partition, config i -> proc i % size, parallel_tools.py:612-651) in a rank-local
  ``SharedArray`` whose rows are mirrored into that rank's HBM, and the ranks exchange
  only the K x K statistics (one all-reduce, the analogue of
  examples/library/transpose_trick/example.py:245-246);
* ``comm=None`` is the reference's "stubs" mode: one process, no collectives.

``comm`` selects the transport of a multi-rank job:

* ``None`` -- the reference's "stubs" mode: one process, no collectives;
* ``"rccl"`` -- one process per GPU, NATIVE RCCL over xGMI behind the C ABI (``fsnap_comm_*``): no torch anywhere on the
  linear path.  Rank / world size / device come from ``RANK`` / ``WORLD_SIZE`` / ``LOCAL_RANK`` (``torchrun`` sets
  them), the communicator id travels through ``fitsnap_amd.rendezvous`` -- or through ``exchange_id`` (a callable
  ``id_or_None -> id``, e.g. ``lambda b: mpi_comm.bcast(b, root=0)`` on the reference side);
* ``"torch"`` / ``True`` / a ``torch.distributed.ProcessGroup`` -- an already initialised ``torch.distributed`` group
  (gloo on CPUs in the world-size-2 tests; the NN solver's group): statistics are reduced through host tensors.
"""
from __future__ import annotations

import functools
from copy import deepcopy
from time import perf_counter, time

import numpy as np


def _printf(*args, **kw):
    kw.setdefault("flush", True)
    print(*args, **kw)


def _dummy_function(*args, **kwargs):
    return


class DistributedList:
    """This rank's fixed-length share of a row-metadata list (reference type:
    fitsnap3lib/parallel_tools.py:892-941).  The calculators fill it slice by slice while the rows are assembled;
    its length never changes, so that the shares of all ranks concatenate into one list with one entry per row
    (``gather_fitsnap``).  Single positions take a one-element sequence, slices a list of exactly the slice's
    length; anything that would change the length is refused with an AssertionError, an unsupported index type
    with NotImplementedError, as in the reference."""

    __slots__ = ("_items",)

    def __init__(self, proc_length: int):
        self._items = [" "] * int(proc_length)

    def __len__(self):
        return len(self._items)

    def __getitem__(self, item):
        return self._items[item]

    def __setitem__(self, key, value):
        size = len(self._items)
        if isinstance(key, slice):
            assert isinstance(value, list), "slice assignment needs a list"
            assert key.stop is not None and key.stop <= size, "slice runs past the end of the list"
            assert len(value) == len(range(*key.indices(size))), "slice assignment must keep the length"
        elif isinstance(key, int):
            assert key <= size and len(value) == 1, "single positions take a one-element sequence"
        else:
            raise NotImplementedError(f"DistributedList cannot be indexed with {type(key)}")
        self._items[key] = value

    def __repr__(self):
        return repr(self._items)

    def get_list(self):
        """Independent copy of the entries."""
        return deepcopy(self._items)


class LabelList(list):
    """A ``list`` of row labels (``Testing`` / ``Groups`` / ``Row_Type`` ...) that counts its edits.

    The reference keeps one Python list per label in ``pt.fitsnap_dict`` and its consumers index, slice, iterate and
    ``isinstance(x, list)``-test them (fitsnap3lib/solvers/solver.py:384-389) -- all of which a subclass of ``list``
    keeps.  What it adds is ``version``: every in-place edit (item / slice assignment, ``append``, ``sort`` ...) bumps
    it, so a solver that derived a training mask or category ids from 10^6 entries can tell in O(1) that they still
    hold (``Solver._labels_stamp``) instead of hashing the whole list on every call (9-26 ms per list at 10^6 rows,
    profiles/r04_ga_loop.txt).  This package's own producers (``gather_fitsnap``, ``FitSnap.load_descriptors``) hand out
    ``LabelList``s; a plain list handed in by a foreign caller keeps working through the content fingerprint."""

    __slots__ = ("version",)

    def __init__(self, *args):
        super().__init__(*args)
        self.version = 0

    def _edits(name):                                          # noqa: N805 - class-body helper
        base = getattr(list, name)

        def method(self, *args, **kw):
            self.version += 1
            return base(self, *args, **kw)

        method.__name__ = name
        method.__doc__ = base.__doc__
        return method

    for _name in ("__setitem__", "__delitem__", "__iadd__", "__imul__", "append", "extend", "insert", "pop", "remove",
                  "clear", "sort", "reverse"):
        locals()[_name] = _edits(_name)
    del _name, _edits

    def __reduce_ex__(self, protocol):
        return (LabelList, (list(self),))

    def copy(self):
        return LabelList(self)


class StubsArray:
    """Plain ndarray holder (fitsnap3lib/parallel_tools.py:1047-1077).  Unlike the
    reference (``np.ndarray(shape)`` = uninitialised memory, Appendix A of SURVEY.md) the
    buffer is zero-initialised: an MPI shared window is zero pages too, and NaN garbage in
    never-written rows would otherwise poison G."""

    _ITEMSIZE = {"d": 8, "i": 4}

    def __init__(self, size1, size2=1, dtype="d"):
        if dtype not in self._ITEMSIZE:
            raise TypeError("dtype {} has not been implemented yet".format(dtype))
        self.array = None
        self.sliced_array = None
        self.energies_index = None
        self.forces_index = None
        self.strain_index = None
        self._length = size1
        self._width = size2
        shape = (size1,) if size2 == 1 else (size1, size2)
        self.array = np.zeros(shape, dtype="float64" if dtype == "d" else "int32")

    def get_memory(self):
        return self.array.nbytes

    # length getters of the MPI variant (parallel_tools.py:1014-1028)
    def get_storage_length(self) -> int:
        return self._length

    def get_scraped_length(self) -> int:
        return self._length

    def get_node_length(self) -> int:
        return self._length

    def get_total_length(self) -> int:
        return self._length


class _Window:
    """Stand-in for the MPI window handle: ``shared_arrays[name].win.Free()``
    (parallel_tools.py:338-350, 372-376) drops the host buffer and any HBM mirror."""

    def __init__(self, owner):
        self._owner = owner

    def Free(self):  # noqa: N802 - reference spelling
        self._owner._free()


class SharedArray(StubsArray):
    """Rank-local array whose rows are mirrored in this rank's HBM on demand
    (reference: node-shared MPI window, parallel_tools.py:944-1044).

    ``array`` is the host view the calculators fill (same attribute as the reference);
    ``version`` must be bumped (``touch()``) by whoever writes into ``array`` after a fit
    so that the resident device copy is refreshed; the solvers of this package call
    ``device_rows`` which re-uploads when the version changed."""

    def __init__(self, size1, size2=1, dtype="d", multinode=0, comms=None):
        super().__init__(size1, size2, dtype)
        self._comms = comms
        self._nbytes = self.array.nbytes
        self._scraped_length = self._length
        self._total_length = self._length
        self._node_length = self._length
        self.win = _Window(self)
        self.version = 0
        self.device_version = -1
        if multinode and comms is not None:
            self.multinode_lengths()

    def touch(self):
        """The host view was written: any resident device copy is stale."""
        self.version += 1

    def mark_device_current(self):
        """The device rows and the host view hold the same data (after device-side assembly
        + download): solvers may skip the upload."""
        self.device_version = self.version

    def get_memory(self):
        return self._nbytes

    def get_scraped_length(self) -> int:
        return self._scraped_length

    def get_node_length(self) -> int:
        return self._node_length

    def get_total_length(self) -> int:
        return self._total_length

    def multinode_lengths(self) -> None:
        """Total row count over ranks (reference: ScaLAPACK bookkeeping,
        parallel_tools.py:1030-1044).  Rows never move between ranks here."""
        pt = self._comms
        self._total_length = int(pt.allreduce_scalar(self._length))
        self._node_length = self._length

    def _free(self):
        self.array = None
        self.sliced_array = None
        self._nbytes = 0


class _TorchTransport:
    """Collectives over an initialised ``torch.distributed`` group (CPU tests with gloo, or a caller-owned group)."""

    kind = "torch"

    def __init__(self, group):
        import torch.distributed as dist

        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("ParallelTools(comm=...) needs an initialised torch.distributed process group")
        self._dist = dist
        self._group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.on_gpu = dist.get_backend(group) == "nccl"

    def _device(self, pt):
        import torch

        return torch.device("cuda", pt.device_index()) if self.on_gpu else torch.device("cpu")

    def allreduce_host(self, pt, arr, op):
        import torch

        t = torch.from_numpy(arr).to(self._device(pt))
        red = {0: self._dist.ReduceOp.SUM, 1: self._dist.ReduceOp.MAX, 2: self._dist.ReduceOp.MIN}[op]
        self._dist.all_reduce(t, op=red, group=self._group)
        arr[...] = t.cpu().numpy()
        return arr

    def allgather_object(self, pt, obj):
        out = [None] * self.size
        self._dist.all_gather_object(out, obj, group=self._group)
        return out

    def bcast_object(self, pt, obj, src):
        box = [obj]
        self._dist.broadcast_object_list(box, src=src, group=self._group)
        return box[0]

    def barrier(self, pt):
        self._dist.barrier(group=self._group)

    def close(self, pt):
        pass


class _RcclTransport:
    """Native RCCL: every collective is a call into libfsnap_hip on this rank's context (``fsnap_comm_*``)."""

    kind = "rccl"

    def __init__(self, exchange_id=None, transport=None):
        import os

        self.rank = int(os.environ.get("RANK", "0"))
        self.size = int(os.environ.get("WORLD_SIZE", "1"))
        self._exchange_id = exchange_id
        # None = FSNAP_DIST_TRANSPORT, else RCCL; "p2p" = the one-shot all-reduce over hipIpc windows (csrc/fsnap_p2p.cpp:
        # one node, also N ranks on ONE GPU).  Only rank 0's choice matters: the transport travels with the id
        self._wire = transport
        self._joined = False
        self.on_gpu = True

    def join(self, pt, ctx):
        """Collective, once per context: enter the communicator with this rank's (new) context ``ctx``.  Runs again
        after ``ParallelTools.free()`` destroyed the context together with its communicator -- on every rank, because
        ``free()`` is called on every rank."""
        if self._joined:
            return
        from . import _capi, rendezvous

        make_id = lambda: _capi.comm_id(self._wire)        # noqa: E731
        if self._exchange_id is not None:
            ident = self._exchange_id(make_id() if self.rank == 0 else None)
        else:
            ident = rendezvous.exchange(self.rank, self.size, make_id)
        ctx.comm_init(self.size, self.rank, ident)
        if self._exchange_id is None:
            rendezvous.done(self.rank)
        self._joined = True

    def allreduce_host(self, pt, arr, op):
        return pt.hip().allreduce_host(arr, op)

    def _gather_blobs(self, pt, blob):
        ctx = pt.hip()
        sizes = np.zeros(self.size)
        sizes[self.rank] = len(blob)
        ctx.allreduce_host(sizes)
        width = max(int(sizes.max()), 1)
        parts = ctx.allgather_bytes(blob.ljust(width, b"\0"), self.size)
        return [parts[r][:int(sizes[r])] for r in range(self.size)]

    def allgather_object(self, pt, obj):
        import pickle

        return [pickle.loads(b) for b in self._gather_blobs(pt, pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL))]

    def bcast_object(self, pt, obj, src):
        import pickle

        ctx = pt.hip()
        blob = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL) if self.rank == src else b""
        n = np.array([float(len(blob))])
        ctx.allreduce_host(n, 1)                                   # max = the source's length
        return pickle.loads(ctx.bcast_bytes(blob if self.rank == src else None, int(n[0]), src))

    def barrier(self, pt):
        pt.hip().barrier()

    def close(self, pt):
        self._joined = False                                       # the context's destructor left the communicator


class ParallelTools:
    """See module docstring.  Attribute names follow fitsnap3lib/parallel_tools.py:157-200."""

    def __init__(self, comm=None, exchange_id=None, transport=None):
        self.check_fitsnap_exist = True
        self.create_shared_bool = True
        self.double_size = 8
        self._fp = None
        self._lmp = None
        self.logger = None
        self._hip = None
        self._device_index = None
        self._transport = None
        # FSNAP_FORCE_MULTI=1: run the collective code paths in a communicator of ONE rank (how a single-GPU box exercises
        # what a multi-GPU job does; tests and `python -m fitsnap3 --comm rccl` alike)
        import os

        self.force_multi = os.environ.get("FSNAP_FORCE_MULTI", "0") not in ("", "0")
        if comm is None or comm is False:
            self.stubs = 1
            self._comm = None
            self._rank = 0
            self._size = 1
        else:
            if isinstance(comm, str) and comm.lower() == "rccl":
                self._transport = _RcclTransport(exchange_id, transport)
            else:
                self._transport = _TorchTransport(None if comm in (True, "torch") else comm)
            self.stubs = 0
            self._comm = comm
            self._rank = self._transport.rank
            self._size = self._transport.size
        # one process per GPU: every rank is its own "node head" for its own rows
        self._sub_rank = 0
        self._sub_size = 1
        self._sub_comm = None
        self._sub_head_proc = self._rank
        self._node_index = self._rank
        self._number_of_nodes = self._size
        self._seed = 0.0
        self.shared_arrays = {}
        self.fitsnap_dict = {}
        self.local_lists = {}    # multi-rank: this rank's row-metadata lists after gather_fitsnap
        self.labels_version = 0  # bumped by touch_labels(); solvers with trust_label_version key their label caches on it
        if self.comm_kind == "rccl":
            self.hip()                                             # creates the context and joins the communicator
        self._set_seed()

    @property
    def comm_kind(self):
        """"stubs" | "rccl" (native, GPU) | "torch" (torch.distributed group)."""
        return "stubs" if self._transport is None else self._transport.kind

    @property
    def multi(self):
        """More than one rank takes part (``force_multi``: run the collective code paths in a communicator of one
        rank -- how the single-GPU test box exercises them)."""
        return self._transport is not None and (self._size > 1 or self.force_multi)

    # -- rank helpers (parallel_tools.py:245-336) ---------------------------------------
    def get_rank(self):
        return self._rank

    def get_size(self):
        return self._size

    def get_subrank(self):
        return self._sub_rank

    def get_subsize(self):
        return self._sub_size

    def get_node(self):
        return self._node_index

    def get_number_of_nodes(self) -> int:
        return self._number_of_nodes

    def _set_seed(self):
        seed = float(time()) if self._rank == 0 else 0.0
        self._seed = self.bcast_object(seed)

    def get_seed(self):
        return self._seed

    def single_print(self, *args, **kw) -> None:
        if self._rank == 0:
            _printf(*args, file=self._fp)

    def sub_print(self, *args, **kw) -> None:
        _printf("Node", self._node_index, ":", *args, file=self._fp)

    def all_print(self, *args, **kw) -> None:
        _printf("Rank", self._rank, ":", *args, file=self._fp)

    def _only_if(self, active, method):
        return method if active else _dummy_function

    def rank_zero(self, method):
        """Decorator: the call happens on rank 0 and is a no-op returning None elsewhere (parallel_tools.py:322-328)."""
        return self._only_if(self._rank == 0, method)

    def sub_rank_zero(self, method):
        """Same for the head of a node (parallel_tools.py:330-336); every rank is the head of its own GPU here."""
        return self._only_if(self._sub_rank == 0, method)

    def single_timeit(self, method):
        """Decorator: wall-clock time of a call in ms, stored under ``log_time[log_name]`` when the caller passes
        those keywords, else printed by rank 0 (parallel_tools.py:290-306)."""
        @functools.wraps(method)
        def timed(*args, **kw):
            start = perf_counter()
            try:
                return method(*args, **kw)
            finally:
                ms = (perf_counter() - start) * 1e3
                if "log_time" in kw:
                    kw["log_time"][kw.get("log_name", method.__name__.upper())] = int(ms)
                else:
                    self.single_print(f"'{method.__name__}' took {ms:.2f} ms on rank {self._rank}")
        return timed

    # -- collectives (reference: mpi4py) ---------------------------------
    def all_barrier(self):
        if self._transport is not None:
            self._transport.barrier(self)

    def sub_barrier(self):
        return

    def bcast_object(self, obj, src=0):
        if not self.multi:
            return obj
        return self._transport.bcast_object(self, obj, src)

    def allgather_object(self, obj):
        """One Python object per rank, in rank order (``comm.allgather``)."""
        if not self.multi:
            return [obj]
        return self._transport.allgather_object(self, obj)

    def allreduce_host(self, arr, op=0):
        """In-place reduction over the ranks of a C-contiguous float64 array on the host (op: 0 sum, 1 max, 2 min)."""
        if not self.multi:
            return arr
        return self._transport.allreduce_host(self, arr, op)

    def allreduce_scalar(self, value, op=0):
        if not self.multi:
            return value
        return float(self.allreduce_host(np.array([float(value)]), op)[0])

    def allreduce_statistics(self, packed):
        """Sum the packed K x K statistics [G | c | bTb, sum_bw, n_train] (host ndarray) over the ranks — the one
        data-path collective of a fit (reference form: examples/library/transpose_trick/example.py:245-246, two MPI
        Allreduce calls).  The native RCCL transport does not come through here for a fit: ``fsnap_fit_dist``
        all-reduces the statistics in HBM on the kernels' stream."""
        return self.allreduce_host(packed, 0)

    def gather_fitsnap(self, name, allgather=None):
        """All-gather a per-rank list held in ``fitsnap_dict`` (parallel_tools.py:426-441)."""
        if name not in self.fitsnap_dict:
            raise NameError("Dictionary element not yet in fitsnap_dictionary")
        if self.stubs:
            return
        held = self.fitsnap_dict[name]
        if isinstance(held, DistributedList):
            held = held._items                                 # ships the plain list, not the wrapper
        self.fitsnap_dict[name] = self.allgather_object(held)

    def get_ncpn(self, nconfigs: int):
        """Number of configurations over all ranks (parallel_tools.py:562-577)."""
        return int(round(self.allreduce_scalar(nconfigs))) if not self.stubs else nconfigs

    # -- device -------------------------------------------------------------------------
    def device_index(self):
        """HIP device of this rank: LOCAL_RANK when launched by torchrun, else rank % #devices."""
        if self._device_index is None:
            import os

            from . import _capi

            n = max(_capi.device_count(), 1)
            self._device_index = int(os.environ.get("LOCAL_RANK", self._rank)) % n
        return self._device_index

    def hip(self):
        """This rank's ``HipContext`` (created on first use; raises loudly without a GPU)."""
        if self._hip is None:
            from . import _capi

            self._hip = _capi.HipContext(self.device_index())
            # kernel timing events off in the fit loop (an event record between two dependent kernels idles the stream
            # for ~5.6 us: a third of a Ta-sized fit); FSNAP_KERNEL_TIMING=1 or ctx.set_option("timing_every", 1) turns
            # them back on for diagnostics (ctx.timing())
            import os

            self._hip.set_option("timing_every", 1 if os.environ.get("FSNAP_KERNEL_TIMING", "0") not in ("", "0") else 0)
            if self.comm_kind == "rccl":
                # a context without a communicator in a multi-rank job would fit this rank's shard alone: a context
                # (re-)created after free() joins again before anybody can use it (collective)
                try:
                    self._transport.join(self, self._hip)
                except BaseException:
                    self._hip.close()
                    self._hip = None
                    raise
        return self._hip

    # -- shared arrays (parallel_tools.py:338-424) ----------------------------------------
    def create_shared_array(self, name, size1, size2=1, dtype="d", tm=0):
        if not isinstance(name, str):
            raise TypeError("name must be a string")
        if name in self.shared_arrays and hasattr(self.shared_arrays[name], "win"):
            try:
                self.shared_arrays[name].win.Free()
            except Exception as e:  # pragma: no cover
                self.single_print(f"Trouble deallocating shared array with name {name}: {e}.")
        if self.stubs == 0 and self.create_shared_bool:
            self.shared_arrays[name] = SharedArray(size1, size2=size2, dtype=dtype, multinode=tm, comms=self)
        else:
            self.shared_arrays[name] = SharedArray(size1, size2=size2, dtype=dtype)

    def touch_labels(self) -> None:
        """A row-label list (``Testing``, ``Groups``, ``Row_Type``) was edited IN PLACE: solvers that were told to trust
        the version (``solver.trust_label_version = True`` with ``keep_resident``) drop what they derived from it.  Without
        that promise the solvers fingerprint the whole content of the lists on every call and need no notice."""
        self.labels_version += 1

    def add_2_fitsnap(self, name: str, an_object) -> None:
        if not isinstance(name, str):
            raise TypeError("name must be a string")
        if self.check_fitsnap_exist and name in self.fitsnap_dict:
            self.fitsnap_dict.pop(name)
        self.fitsnap_dict[name] = an_object

    def free(self):
        """Free all shared arrays (parallel_tools.py:338-350), the HBM mirror and (native transport) the communicator."""
        for name in list(self.shared_arrays):
            try:
                self.shared_arrays[name].win.Free()
            except Exception:
                pass
        if self._hip is not None:
            self._hip.close()                  # fsnap_ctx_destroy leaves the RCCL communicator first
            self._hip = None
        if self._transport is not None:
            self._transport.close(self)        # native transport: the next hip() joins a new communicator

    def slice_array(self, name: str) -> None:
        if name not in self.shared_arrays:
            raise IndexError("{} not found in shared objects".format(name))
        if name == "a":
            raise NotImplementedError("Slice A using new_slice_a")
        shared = self.shared_arrays[name]
        shared.sliced_array = shared.array[self._sub_rank::self._sub_size]

    def split_by_node(self, obj):
        """Round-robin split over ranks (parallel_tools.py:543-550): config i -> rank i % size."""
        mine = slice(self._node_index, None, self._number_of_nodes)
        if isinstance(obj, dict):
            obj.update({key: seq[mine] for key, seq in obj.items()})       # in place, like the reference
            return obj
        return obj[mine] if isinstance(obj, list) else obj

    def new_slice_a(self, a_len=None):
        """Row-offset table (parallel_tools.py:594-651).  One process per GPU: every rank owns
        all rows of its own arrays, so the table is the trivial [0, a_len - 1]."""
        if a_len is None:
            a_len = len(self.shared_arrays["a"].array)
        self.add_2_fitsnap("sub_a_size", int(a_len))
        self.add_2_fitsnap("sub_a_indices", np.array([0, int(a_len) - 1]))

    def get_ram(self):
        """Total host RAM in bytes (parallel_tools.py:862-876 uses psutil)."""
        try:
            import psutil

            return psutil.virtual_memory().total
        except Exception:
            return 0

    # -- LAMMPS handles (parallel_tools.py:519-560): only opened when a `lammps` module exists --
    def _lammps_class(self):
        try:
            from lammps import lammps
        except Exception as e:
            raise RuntimeError("the LAMMPS Python module is required to compute descriptors "
                               "(this repository replaces the POST-LAMMPS path only)") from e
        return lammps

    def check_lammps(self, lammps_noexceptions: int = 0) -> None:
        lmp = self._lammps_class()(cmdargs=["-screen", "none", "-log", "none"])
        if not (lmp.has_exceptions or lammps_noexceptions):
            raise Exception("Fitting interrupted! LAMMPS not compiled with C++ exceptions handling enabled")
        lmp.close()

    def initialize_lammps(self, lammpslog: int = 0, printlammps: int = 0):
        quiet = ("-screen", "none") + (() if lammpslog else ("-log", "none"))
        self._lmp = self._lammps_class()(cmdargs=list(quiet))
        return self._lmp

    def close_lammps(self):
        handle, self._lmp = self._lmp, None
        if handle is not None:
            handle.close()
        return None

    def exception(self, err):
        """Abort path (parallel_tools.py:840-860): no MPI.Abort here — re-raise."""
        raise err
