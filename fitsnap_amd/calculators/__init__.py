"""Calculators behind the reference's Calculator plugin API (fitsnap3lib/calculators/):
the post-LAMMPS half — A/b/w allocation and the `_collect_lammps` assembly — runs on the
GPU; LAMMPS driving itself stays with LAMMPS."""
from .calculator import Calculator  # noqa: F401
from .lammps_base import LammpsBase, _extract_compute_np  # noqa: F401
from .lammps_snap import LammpsSnap  # noqa: F401
from .lammps_pace import LammpsPace  # noqa: F401
from .calculator_factory import calculator, search  # noqa: F401
