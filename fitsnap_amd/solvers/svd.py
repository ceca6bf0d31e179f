"""SVD solver behind the reference's plugin API (fitsnap3lib/solvers/svd.py:13-54)."""
from __future__ import annotations

from sys import float_info as fi

import numpy as np

from .. import _capi
from .._hostblas import blas_threads
from .solver import Solver


class SVD(Solver):
    """``perform_fit`` = reference semantics of ``lstsq(aw, bw, 1.0e-13)`` (svd.py:54).

    Well-conditioned systems (the common case; Ta: kappa 2.7e5): Jacobi-scaled Cholesky of the GPU normal-equation
    statistics + two refinement steps with the row-space residual.  Ill-conditioned or rank-deficient systems (scaled
    pivot below 1e-11, or directions dropped that are not zero columns): ``fsnap_lstsq_rows`` -- CholeskyQR passes over
    the rows on the GPU and dgelsd's own K x K end (SVD of the triangular factor, singular values below
    1e-13 sigma_max dropped, minimum-norm solution).  Exactly-zero columns get a zero coefficient, as lstsq's
    minimum-norm solution gives them."""

    RCOND = 1.0e-13  # svd.py:54

    def __init__(self, name, pt, config):
        super().__init__(name, pt, config)
        # lstsq works on A_w (error ~ kappa eps); the normal equations square kappa.  Two steps of
        # refinement with the row-space residual close that gap (Solver._refine).
        self.refine_steps = 2
        self.row_space = True      # row-space (CholeskyQR + SVD) solve when the statistics are ill-conditioned

    def perform_fit(self, a=None, b=None, w=None, fs_dict=None, trainall=False):
        """Weighted least-squares fit; same call contract as the reference (svd.py:18-54).

        With no arguments the rows come from ``pt.shared_arrays['a' | 'b' | 'w']`` and the training mask from
        ``pt.fitsnap_dict['Testing']``.  Otherwise ``a`` (m x K), ``b`` (m) and ``w`` are used, where ``w`` holds one
        weight per TRAINING row; the rows to leave out are taken from ``fs_dict['Testing']`` if a dictionary is
        given, and nothing is left out if ``trainall`` is set (precedence: fs_dict, trainall, pt.fitsnap_dict).
        Collective in a multi-rank job; the coefficients end up in ``self.fit`` on rank 0 only."""
        pt = self.pt
        # every rank contributes its rows' statistics; the coefficients are published on rank 0 (svd.py:33)
        if not ("EXTRAS" in self.config.sections and self.config.sections["EXTRAS"].apply_transpose):
            self.last_row_space = None
            # PROBE: a system the Cholesky factorisations cannot resolve comes back at once (rank -1) instead of going
            # through the library's eigen-truncation of G -- that answer would be discarded for the row-space solve anyway,
            # and at K = 1595 the Jacobi sweeps behind it take a minute
            fit = self._fit_and_solve(_capi.SOLVE_LSTSQ_PROBE, self.RCOND, a, b, w, fs_dict, trainall)
            K = len(fit)
            on_gpu = (pt.comm_kind != "torch" or not pt.multi) and (self._rows_on_device() or pt.multi)
            if self.row_space and on_gpu and self._needs_row_space(K):
                # ill-conditioned or rank deficient: lstsq's answer lives in the rows, not in the K x K statistics
                fit = self._row_space_fit(K, self.RCOND)
            else:
                fit = self._resolve_probe(_capi.SOLVE_LSTSQ_PROBE, self.RCOND, fit)
                if self.refine_steps and self.last_rank == K:
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
                with blas_threads(len(c)):
                    ev = np.linalg.eigvalsh(G)
                cond2 = abs(ev[-1]) / max(abs(ev[0]), np.finfo(float).tiny)
                if cond2 < 1 / fi.epsilon:
                    rcond = np.sqrt(self.RCOND)
                else:
                    print("The Matrix is ill-conditioned for the transpose trick")
            self.fit = self._solve(_capi.SOLVE_LSTSQ, rcond, G, c)
