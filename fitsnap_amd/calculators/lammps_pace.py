"""PACE (ACE) calculator: same post-LAMMPS transform as SNAP on ``compute pace`` output
(fitsnap3lib/calculators/lammps_pace.py:14-22 get_width, :369-509 _collect_lammps, :197-366
_collect_lammps_single).  Differences kept from the reference: settings come from the
``[ACE]`` section, non-finite descriptors are ``nan_to_num``-ed with a warning instead of
raising (:399-403), and no ``Atom_Type`` list is produced."""
from __future__ import annotations

import numpy as np

from .lammps_base import LammpsBase
from .lammps_snap import _LinearAssembly


class LammpsPace(_LinearAssembly, LammpsBase):
    """[CALCULATOR] calculator = LAMMPSPACE"""

    SECTION = "ACE"
    COMPUTE = "pace"
    WITH_ATOM_TYPE = False

    def __init__(self, name, pt, config):
        super().__init__(name, pt, config)
        self.pt.check_lammps()

    def _check_finite(self, raw):
        if (np.isinf(raw)).any() or (np.isnan(raw)).any():
            self.pt.single_print("! WARNING! applying np.nan_to_num()")
            raw = np.nan_to_num(raw)
        if (np.isinf(raw)).any() or (np.isnan(raw)).any():
            raise ValueError("NaN in computed data of file {} in group {}".format(self._data["File"], self._data["Group"]))
        return raw

    def _warn_no_neighbors(self, raw):
        sec = self._sec()
        if getattr(sec, "bikflag", False) or sec.bzeroflag:
            return
        b000sum = float(np.sum((raw[0, :sec.ncoeff * sec.numtypes] / self._data["NumAtoms"])[::sec.ncoeff]))
        if abs(b000sum - 1.0) < 1.0e-10:
            self.pt.single_print("! WARNING: Configuration has no PACE neighbors. \nGroup and configuration: {} {}".format(
                self._data["Group"], self._data["File"]))
