"""Calculator factory: the reference's discovery rule
(fitsnap3lib/calculators/calculator_factory.py:11-38) — a calculator must be a GRANDCHILD
of ``Calculator`` whose class name equals ``[CALCULATOR] calculator`` case-insensitively."""
from .calculator import Calculator
from .lammps_base import LammpsBase  # noqa: F401
from .lammps_pace import LammpsPace  # noqa: F401
from .lammps_snap import LammpsSnap  # noqa: F401


def calculator(calculator_name, pt, cfg):
    """Calculator Factory"""
    instance = search(calculator_name)
    instance.__init__(calculator_name, pt, cfg)
    return instance


def search(calculator_name):
    instance = None
    for cls in Calculator.__subclasses__():
        for cls2 in cls.__subclasses__():
            if cls2.__name__.lower() == calculator_name.lower():
                instance = Calculator.__new__(cls2)
    if instance is None:
        raise IndexError("{} was not found in fitsnap calculators".format(calculator_name))
    return instance
