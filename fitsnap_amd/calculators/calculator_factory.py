"""Calculator factory with the reference's discovery rule (fitsnap3lib/calculators/calculator_factory.py:11-38): a
calculator is a GRANDCHILD of ``Calculator`` (Calculator -> LammpsBase -> LammpsSnap / LammpsPace) whose class name
equals ``[CALCULATOR] calculator`` case-insensitively.  The imports below are the registration."""
from .._discovery import find_plugin
from .calculator import Calculator
from . import lammps_base, lammps_pace, lammps_snap  # noqa: F401


def search(calculator_name):
    return find_plugin(Calculator, calculator_name, 2, "calculators")


def calculator(calculator_name, pt, cfg):
    """Calculator Factory"""
    obj = search(calculator_name)
    obj.__init__(calculator_name, pt, cfg)
    return obj
