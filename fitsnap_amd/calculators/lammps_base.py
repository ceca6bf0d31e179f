"""LAMMPS-backed calculator base (fitsnap3lib/calculators/lammps_base.py:10-307): the
per-configuration driver ``process_configs`` / ``process_single`` and the zero-copy view of
the LAMMPS compute array.  Building LAMMPS commands (box, atoms, computes) is LAMMPS'
business and out of this repository's scope (OUT rows): ``_prepare_lammps``
must be provided by whoever supplies the ``lammps`` object (tests inject a fake)."""
from __future__ import annotations

import ctypes

import numpy as np

from .calculator import Calculator


def _extract_compute_np(lmp, name, compute_style, result_type, array_shape=None):
    """ndarray VIEW (no copy) of a LAMMPS compute; same call contract as the reference helper
    (lammps_base.py:280-307).  Without ``array_shape`` LAMMPS' own numpy wrapper decides the shape; with it the
    raw pointer is wrapped: a scalar (type 0) is returned as is, a vector (type 1) is a ``double*``, an array
    (type 2) a ``double**`` whose first row pointer addresses ONE contiguous row-major block -- the layout
    ``fsnap_assemble`` consumes."""
    if array_shape is None:
        return lmp.numpy.extract_compute(name, compute_style, result_type)
    handle = lmp.extract_compute(name, compute_style, result_type)
    if result_type == 0:
        return handle
    first = handle.contents if result_type == 2 else handle
    shape = tuple(int(n) for n in np.atleast_1d(array_shape))
    count = int(np.prod(shape))
    address = ctypes.addressof(first.contents) if hasattr(first, "contents") else ctypes.cast(first, ctypes.c_void_p).value
    block = (ctypes.c_double * count).from_address(address)
    return np.ctypeslib.as_array(block).reshape(shape)


class LammpsBase(Calculator):

    def __init__(self, name, pt, config):
        super().__init__(name, pt, config)
        self._data = {}
        self._i = 0
        self._lmp = None
        self._row_index = 0

    # -- per-configuration driver (lammps_base.py:52-125) --------------------------------
    def process_configs(self, data: dict, i: int):
        self._data = data
        self._i = i
        self._initialize_lammps()
        try:
            self._prepare_lammps()
            self._run_lammps()
            self._collect_lammps()
        finally:
            self._lmp = self.pt.close_lammps()

    def process_single(self, data: dict, i: int = 0):
        """(a, b, w) of ONE configuration without touching the shared arrays
        (lammps_base.py:101-125) — the transpose-trick feed."""
        self._data = data
        self._i = i
        self._initialize_lammps()
        try:
            self._prepare_lammps()
            self._run_lammps()
            a, b, w = self._collect_lammps_single()
        finally:
            self._lmp = self.pt.close_lammps()
        return a, b, w

    def accumulate_single(self, data: dict, i: int, d_packed_ptr: int):
        """``process_single`` fused with the accumulation that follows it in the reference's memory-lean driver
        (examples/library/transpose_trick/example.py:230-237: ``a, b, w = process_single(...)``, ``c += aw.T @ aw``,
        ``d += aw.T @ bw``): the packed statistics at ``d_packed_ptr`` (device memory, ``K*K + K + 3`` doubles, zeroed
        before the first configuration) += those of this configuration.  Its rows exist in registers only
        (``fsnap_assemble_accumulate``).  Returns the number of rows."""
        self._data = data
        self._i = i
        self._initialize_lammps()
        try:
            self._prepare_lammps()
            self._run_lammps()
            n = self._accumulate_lammps_single(d_packed_ptr)
        finally:
            self._lmp = self.pt.close_lammps()
        return n

    def _initialize_lammps(self, printlammps: int = 0):
        self._lmp = self.pt.initialize_lammps(getattr(self.config.args, "lammpslog", 0), printlammps)

    def _prepare_lammps(self):
        raise NotImplementedError("LAMMPS command generation is outside this repository's scope: "
                                  "override _prepare_lammps (see INTEGRATION.md)")

    def _run_lammps(self):
        self._lmp.command("run 0")

    def _collect_lammps(self):
        raise NotImplementedError

    def _collect_lammps_single(self):
        raise NotImplementedError

    def _accumulate_lammps_single(self, d_packed_ptr):
        raise NotImplementedError

    # -- atom extraction (lammps_base.py:233-253) -----------------------------------------
    def _per_atom_integers(self, field, num_atoms):
        """One integer per atom (``id`` / ``type``) as a flat array; LAMMPS Python modules differ in which numpy accessor
        serves integer fields, so the dedicated one is the second choice."""
        accessors = (self._lmp.numpy.extract_atom, getattr(self._lmp.numpy, "extract_atom_iarray", None))
        failure = None
        for accessor in accessors:
            if accessor is None:
                continue
            try:
                return accessor(name=field, nelem=num_atoms).ravel()
            except Exception as err:           # noqa: BLE001 - whatever the binding raises, try the other accessor
                failure = err
        raise failure

    def _extract_atom_ids(self, num_atoms: int):
        return self._per_atom_integers("id", num_atoms)

    def _extract_atom_types(self, num_atoms: int):
        return self._per_atom_integers("type", num_atoms)
