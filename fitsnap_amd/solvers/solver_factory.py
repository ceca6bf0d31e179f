"""Solver factory with the reference's discovery rule (fitsnap3lib/solvers/solver_factory.py:18-34): any imported
direct subclass of ``Solver`` whose class name equals ``[SOLVER] solver`` case-insensitively; unknown name ->
IndexError.  The imports below are the registration."""
from .._discovery import find_plugin
from .solver import Solver
from . import anl, ard, lasso, ridge, svd  # noqa: F401  (ANL, ARD, LASSO, RIDGE, SVD)


def search(solver_name):
    return find_plugin(Solver, solver_name, 1, "solvers")


def solver(solver_name, pt, cfg):
    """Solver Factory"""
    obj = search(solver_name)
    obj.__init__(solver_name, pt, cfg)
    return obj
