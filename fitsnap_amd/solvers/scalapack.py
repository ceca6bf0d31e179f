"""``[SOLVER] solver = ScaLAPACK`` behind the reference's plugin API (fitsnap3lib/solvers/scalapack.py:9-45).

The reference's ScaLAPACK solver is its one multi-node linear solve: the node heads hold row blocks of ``aw`` and MKL's
``pdgels`` computes the least-squares solution of all rows (lib/scalapack_solver/scalapack.pyx:152-178); it needs Intel
MKL-ScaLAPACK + BLACS and is not built in the reference's own CI (out of scope as a port target).  Its ROLE is
what this package's multi-GPU path provides -- rows sharded over one process per GPU, the K x K statistics summed by one
RCCL all-reduce, the least-squares solution of all rows -- so an input file that names it keeps working: this class is the
SVD solver (same ``lstsq`` semantics, refinement with the row-space residual, row-space path when the rows are
ill-conditioned) under the reference's name, with the reference's restriction that no row may be marked for testing
(scalapack.py:16-18).  No ScaLAPACK is linked and none is emulated."""
from __future__ import annotations

from .solver import Solver
from .svd import SVD


class ScaLAPACK(Solver):
    # a direct child of Solver, like every class the factory can find (solver_factory.py:18-34); the fit is SVD's

    RCOND = SVD.RCOND

    def __init__(self, name, pt, config):
        super().__init__(name, pt, config)
        self.refine_steps = 2
        self.row_space = True

    def perform_fit(self):
        """scalapack.py:13-45 -- no arguments: rows from ``pt.shared_arrays``, all of them training rows."""
        testing = self.pt.fitsnap_dict.get("Testing")
        flagged = bool(testing is not None and any(testing))
        if self.pt.multi:                       # the label lists are rank-local: agree before anybody raises
            flagged = any(self.pt.allgather_object(flagged))
        if flagged:
            raise NotImplementedError("Testing w/ the ScaLAPACK solver is not implemented!")
        SVD.perform_fit(self)
