"""Solver factory: same discovery rule as the reference
(fitsnap3lib/solvers/solver_factory.py:18-34): any imported subclass of ``Solver`` whose
class name equals ``[SOLVER] solver`` case-insensitively; created with ``Solver.__new__``
then ``__init__(name, pt, config)``; unknown name -> IndexError."""
from .solver import Solver
from .anl import ANL  # noqa: F401
from .ard import ARD  # noqa: F401  (import = registration, as in the reference)
from .ridge import RIDGE  # noqa: F401
from .svd import SVD  # noqa: F401


def solver(solver_name, pt, cfg):
    """Solver Factory"""
    instance = search(solver_name)
    instance.__init__(solver_name, pt, cfg)
    return instance


def search(solver_name):
    instance = None
    for cls in Solver.__subclasses__():
        if cls.__name__.lower() == solver_name.lower():
            instance = Solver.__new__(cls)
    if instance is None:
        raise IndexError("{} was not found in fitsnap solvers".format(solver_name))
    return instance
