"""SNAP calculator: the post-LAMMPS assembly of ``compute snap`` output into rows of A, b, w
(fitsnap3lib/calculators/lammps_snap.py:15-23 get_width, :391-556 _collect_lammps, :224-389
_collect_lammps_single), executed by the device kernel ``fsnap_assemble``."""
from __future__ import annotations

import numpy as np

from .lammps_base import LammpsBase, _extract_compute_np
from .row_plan import config_row_plan, type_fractions


class _Batch:
    """Raw LAMMPS blocks + row plans staged on the host until one fsnap_assemble call."""

    def __init__(self, row0):
        self.row0 = row0
        self.raw = []
        self.plans = []
        self.fracs = []
        self.raw_rows = 0
        self.bytes = 0

    def add(self, raw, plan, frac):
        self.raw.append(np.array(raw, dtype=np.float64, copy=True))   # LAMMPS memory dies at close()
        self.plans.append(plan)
        if frac is not None:
            self.fracs.append(frac)
        self.raw_rows += raw.shape[0]
        self.bytes += raw.nbytes


class _LinearAssembly:
    """Implementation shared by LammpsSnap and LammpsPace (a mixin, so that both stay DIRECT
    children of LammpsBase, which is what the factory's grandchild rule needs)."""

    SECTION = "BISPECTRUM"
    COMPUTE = "snap"
    WITH_ATOM_TYPE = True

    def _sec(self):
        return self.config.sections[self.SECTION]

    def get_width(self):
        """lammps_snap.py:15-23."""
        sec = self._sec()
        if self.config.sections["CALCULATOR"].nonlinear:
            return sec.ncoeff
        a_width = sec.ncoeff * sec.numtypes
        if not sec.bzeroflag:
            a_width += sec.numtypes
        return int(a_width)

    # -- shared by the batched and the single-configuration paths ------------------------
    def _extract_config(self):
        """Raw compute array (view), 1-based atom types, volume of the current configuration,
        with the reference's sanity checks (lammps_snap.py:400-428)."""
        sec = self._sec()
        num_atoms = self._data["NumAtoms"]
        lmp_atom_ids = self._extract_atom_ids(num_atoms)
        lmp_types = self._extract_atom_types(num_atoms)
        assert np.all(lmp_atom_ids == 1 + np.arange(num_atoms)), \
            "LAMMPS seems to have lost atoms\nGroup and configuration: {} {}".format(self._data["Group"], self._data["File"])
        vol = self._lmp.get_thermo("vol")
        bik_rows = num_atoms if getattr(sec, "bikflag", False) else 1
        nrows = bik_rows + 3 * num_atoms + 6
        ncols = sec.ncoeff * sec.numtypes + 1
        raw = _extract_compute_np(self._lmp, self.COMPUTE, 0, 2, (nrows, ncols))
        raw = self._check_finite(raw)
        return raw, lmp_types, vol

    def _check_finite(self, raw):
        if (np.isinf(raw)).any() or (np.isnan(raw)).any():
            raise ValueError("Nan in computed data of file {} in group {}".format(self._data["File"], self._data["Group"]))
        return raw

    def _warn_no_neighbors(self, raw):
        """lammps_snap.py:437-453: B[0,0,0] sum check on the energy row."""
        sec = self._sec()
        if getattr(sec, "bikflag", False):
            return
        num_atoms = self._data["NumAtoms"]
        b000sum0 = 0.0 if sec.bzeroflag else 1.0
        nstride = sec.ncoeff
        if getattr(sec, "chemflag", False):
            nstride //= sec.numtypes ** 3
            if getattr(sec, "wselfallflag", False):
                b000sum0 *= sec.numtypes ** 3
        b000sum = float(np.sum(raw[0, :sec.ncoeff * sec.numtypes:nstride] / num_atoms))
        if abs(b000sum - b000sum0) < 1.0e-10:
            print("! WARNING: Configuration has no SNAP neighbors \nGroup and configuration: {} {}".format(
                self._data["Group"], self._data["File"]))

    def _plan(self, raw_row0, frac_index, lmp_types, vol):
        calc = self.config.sections["CALCULATOR"]
        sec = self._sec()
        bik = bool(getattr(sec, "bikflag", False))
        if calc.energy and bik and not sec.bzeroflag:
            raise NotImplementedError("per atom energy is not implemented without bzeroflag")
        d = self._data
        return config_row_plan(d["NumAtoms"], lmp_types, vol, d["Energy"], d["Forces"], d["Stress"], d["eweight"],
                               d["fweight"], d["vweight"], calc.energy, calc.force, calc.stress, bik, raw_row0, frac_index,
                               with_atom_type=self.WITH_ATOM_TYPE)

    # -- batched path: rows go straight into the resident HBM arrays ---------------------
    def _collect_lammps(self):
        sec = self._sec()
        raw, lmp_types, vol = self._extract_config()
        if self.config.sections["CALCULATOR"].energy:
            self._warn_no_neighbors(raw)
        if self._batch is None:
            self._batch = _Batch(self.shared_index)
        bt = self._batch
        frac = None
        frac_index = -1
        if not sec.bzeroflag:
            frac = type_fractions(self._data["AtomTypes"], sec.type_mapping, sec.numtypes)
            frac_index = len(bt.fracs)
        plan, meta = self._plan(bt.raw_rows, frac_index, lmp_types, vol)
        bt.add(raw, plan, frac)
        n = len(plan["src_row"])
        dindex = self.distributed_index
        fd = self.pt.fitsnap_dict
        for key, val in meta.items():
            fd[key][dindex:dindex + n] = val
        fd["Groups"][dindex:dindex + n] = ["{}".format(self._data["Group"])] * n
        fd["Configs"][dindex:dindex + n] = ["{}".format(self._data["File"])] * n
        fd["Testing"][dindex:dindex + n] = [bool(self._data["test_bool"])] * n
        self.shared_index += n
        self.distributed_index += n
        if bt.bytes >= self.BATCH_BYTES:
            self._flush_batch()

    def _flush_batch(self):
        bt = self._batch
        if bt is None or not bt.plans:
            self._batch = None
            return
        sec = self._sec()
        plan = {k: np.concatenate([p[k] for p in bt.plans]) for k in bt.plans[0]}
        fr = np.array(bt.fracs, dtype=np.float64).reshape(-1, sec.numtypes) if bt.fracs else np.zeros((0, sec.numtypes))
        self.pt.hip().assemble(np.concatenate(bt.raw, axis=0), bt.row0, plan["src_row"], plan["kind"], plan["frac"],
                               plan["d"], plan["truth"], plan["weight"], fr, np.asarray(sec.blank2J, dtype=np.float64),
                               sec.numtypes, sec.ncoeff, 0 if sec.bzeroflag else 1)
        self._batch = None

    # -- single configuration (transpose-trick feed): fresh (a, b, w), no shared arrays ---
    def _collect_lammps_single(self):
        from .. import _capi

        sec = self._sec()
        raw, lmp_types, vol = self._extract_config()
        frac = None if sec.bzeroflag else type_fractions(self._data["AtomTypes"], sec.type_mapping, sec.numtypes)
        plan, _ = self._plan(0, -1 if frac is None else 0, lmp_types, vol)
        n = len(plan["src_row"])
        if not hasattr(self, "_single_ctx") or self._single_ctx is None:
            self._single_ctx = _capi.HipContext(self.pt.device_index())
        ctx = self._single_ctx
        ctx.rows_alloc(n, self.get_width())
        fr = np.zeros((0, sec.numtypes)) if frac is None else frac.reshape(1, -1)
        ctx.assemble(np.array(raw, copy=True), 0, plan["src_row"], plan["kind"], plan["frac"], plan["d"], plan["truth"],
                     plan["weight"], fr, np.asarray(sec.blank2J, dtype=np.float64), sec.numtypes, sec.ncoeff,
                     0 if sec.bzeroflag else 1)
        return ctx.download_rows()

    # -- single configuration, fused with `c += aw.T @ aw; d += aw.T @ bw`: no rows anywhere ---
    def _accumulate_lammps_single(self, d_packed_ptr):
        sec = self._sec()
        raw, lmp_types, vol = self._extract_config()
        frac = None if sec.bzeroflag else type_fractions(self._data["AtomTypes"], sec.type_mapping, sec.numtypes)
        plan, _ = self._plan(0, -1 if frac is None else 0, lmp_types, vol)
        fr = np.zeros((0, sec.numtypes)) if frac is None else frac.reshape(1, -1)
        self.pt.hip().assemble_accumulate(raw, plan["src_row"], plan["kind"], plan["frac"], plan["d"], plan["truth"],
                                          plan["weight"], fr, np.asarray(sec.blank2J, dtype=np.float64), sec.numtypes,
                                          sec.ncoeff, 0 if sec.bzeroflag else 1, d_packed_ptr)
        return len(plan["src_row"])


class LammpsSnap(_LinearAssembly, LammpsBase):
    """[CALCULATOR] calculator = LAMMPSSNAP"""

    SECTION = "BISPECTRUM"
    COMPUTE = "snap"
    WITH_ATOM_TYPE = True

    def __init__(self, name, pt, config):
        super().__init__(name, pt, config)
        self.pt.check_lammps()
