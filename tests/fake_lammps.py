"""A fake ``lammps`` module for tests: serves a synthetic ``compute snap|pace`` global array as
a ctypes ``double**`` (what ``lammps.extract_compute(name, 0, 2)`` returns), atom ids / types
/ positions and a cell volume.  ``CURRENT`` holds what the instance serves right now."""
import ctypes
import sys
import types

CURRENT = {}


class _FakeNumpy:
    def extract_atom(self, name, nelem=None, dim=1, **kw):
        if name == "id":
            return CURRENT["ids"]
        if name == "type":
            return CURRENT["types"]
        if name == "x":
            return CURRENT["pos"]
        raise KeyError(name)


class FakeLammps:
    has_exceptions = True
    installed_packages = []

    def __init__(self, *a, **k):
        self.numpy = _FakeNumpy()
        self._keep = None
        self.closed = False

    def command(self, s):
        pass

    def close(self):
        self.closed = True

    def version(self):
        return 20250722

    def get_natoms(self):
        return len(CURRENT["ids"])

    def get_thermo(self, key):
        assert key == "vol"
        return CURRENT["vol"]

    def create_atoms(self, **kw):
        pass

    def extract_compute(self, name, style, rtype):
        import numpy as np

        arr = CURRENT["raw"]
        assert arr.flags["C_CONTIGUOUS"] and arr.dtype == np.float64
        rows = (ctypes.POINTER(ctypes.c_double) * arr.shape[0])()
        for r in range(arr.shape[0]):
            rows[r] = ctypes.cast(arr.ctypes.data + r * arr.strides[0], ctypes.POINTER(ctypes.c_double))
        self._keep = rows
        return ctypes.cast(rows, ctypes.POINTER(ctypes.POINTER(ctypes.c_double)))


def install():
    stub = types.ModuleType("lammps")
    stub.lammps = FakeLammps
    sys.modules["lammps"] = stub
    return stub
