"""SVD solver behind the reference's plugin API (fitsnap3lib/solvers/svd.py:13-54)."""
from __future__ import annotations

from sys import float_info as fi

import numpy as np

from .. import _capi
from .solver import Solver


class SVD(Solver):
    """``perform_fit`` = reference semantics of ``lstsq(aw, bw, 1.0e-13)`` (svd.py:54),
    computed from the GPU normal-equation statistics: Jacobi-scaled Cholesky with one
    refinement step when the system is numerically full rank, truncated eigen
    pseudo-inverse (minimum-norm, gelsd-like) otherwise.  Exactly-zero columns get a zero
    coefficient, as lstsq's minimum-norm solution gives them."""

    RCOND = 1.0e-13  # svd.py:54

    def __init__(self, name, pt, config):
        super().__init__(name, pt, config)
        # lstsq works on A_w (error ~ kappa eps); the normal equations square kappa.  Two steps of
        # refinement with the row-space residual close that gap (Solver._refine).
        self.refine_steps = 2

    def perform_fit(self, a=None, b=None, w=None, fs_dict=None, trainall=False):
        """Weighted least-squares fit; same call contract as the reference (svd.py:18-54).

        With no arguments the rows come from ``pt.shared_arrays['a' | 'b' | 'w']`` and the training mask from
        ``pt.fitsnap_dict['Testing']``.  Otherwise ``a`` (m x K), ``b`` (m) and ``w`` are used, where ``w`` holds one
        weight per TRAINING row; the rows to leave out are taken from ``fs_dict['Testing']`` if a dictionary is
        given, and nothing is left out if ``trainall`` is set (precedence: fs_dict, trainall, pt.fitsnap_dict).
        Collective in a multi-rank job; the coefficients end up in ``self.fit`` on rank 0 only."""
        pt = self.pt
        # every rank contributes its rows' statistics; only rank 0 solves (svd.py:33)
        if not ("EXTRAS" in self.config.sections and self.config.sections["EXTRAS"].apply_transpose):
            fit = self._fit_and_solve(_capi.SOLVE_LSTSQ, self.RCOND, a, b, w, fs_dict, trainall)
            # refine only a full-rank solve; rank 0 decides, every rank follows (collective)
            do_refine = bool(self.refine_steps) and (pt._rank != 0 or self.last_rank == len(fit))
            if not (pt.stubs or pt._size == 1):
                do_refine = pt.bcast_object(do_refine if pt._rank == 0 else None, src=0)
            if do_refine:
                fit = self._refine(fit, _capi.SOLVE_LSTSQ, self.RCOND, self.refine_steps)
            if pt._rank == 0:
                self.fit = fit
            return
        G, c, _ = self._fit_statistics(a, b, w, fs_dict, trainall)
        if pt._rank == 0:
            rcond = self.RCOND
            if "EXTRAS" in self.config.sections and self.config.sections["EXTRAS"].apply_transpose:
                # svd.py:48-53: lstsq on (aw.T aw, aw.T bw) when cond(aw)^2 < 1/eps, i.e. the
                # 1e-13 cut then applies to the singular values of G itself (= eigenvalues)
                ev = np.linalg.eigvalsh(G)
                cond2 = abs(ev[-1]) / max(abs(ev[0]), np.finfo(float).tiny)
                if cond2 < 1 / fi.epsilon:
                    rcond = np.sqrt(self.RCOND)
                else:
                    print("The Matrix is ill-conditioned for the transpose trick")
            self.fit = self._solve(_capi.SOLVE_LSTSQ, rcond, G, c)
