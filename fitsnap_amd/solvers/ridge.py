"""RIDGE solver behind the reference's plugin API (fitsnap3lib/solvers/ridge.py:6-60,
fitsnap3lib/lib/ridge_solver/regressor.py:4-21)."""
from __future__ import annotations

from .. import _capi
from .solver import Solver


class RIDGE(Solver):
    """``(aw.T aw + alpha I) beta = aw.T bw`` from the GPU statistics.

    ``[RIDGE] local_solver = 0`` mirrors sklearn ``Ridge(alpha, fit_intercept=False)``
    (ridge.py:47-53: Cholesky on the normal equations, eigen/SVD fallback);
    ``local_solver = 1`` mirrors ``Local_Ridge`` (regressor.py:10-16: explicit inverse,
    LinAlgError when singular).  With ``[EXTRAS] apply_transpose`` the reference feeds
    X = aw.T aw, y = aw.T bw to the regressor (ridge.py:41-43), i.e. solves
    (G G + alpha I) beta = G c; that K x K product is formed on the host."""

    def __init__(self, name, pt, config):
        super().__init__(name, pt, config)

    def perform_fit(self, a=None, b=None, w=None, fs_dict=None, trainall=False):
        """Weighted least-squares fit; same call contract as the reference (ridge.py:11-60).

        With no arguments the rows come from ``pt.shared_arrays['a' | 'b' | 'w']`` and the training mask from
        ``pt.fitsnap_dict['Testing']``.  Otherwise ``a`` (m x K), ``b`` (m) and ``w`` are used, where ``w`` holds one
        weight per TRAINING row; the rows to leave out are taken from ``fs_dict['Testing']`` if a dictionary is
        given, and nothing is left out if ``trainall`` is set (precedence: fs_dict, trainall, pt.fitsnap_dict).
        Collective in a multi-rank job; the coefficients end up in ``self.fit`` on rank 0 only."""
        pt = self.pt
        alval = self.config.sections["RIDGE"].alpha
        local = bool(self.config.sections["RIDGE"].local_solver)
        kind = _capi.SOLVE_RIDGE_INV if local else _capi.SOLVE_RIDGE
        if "EXTRAS" in self.config.sections and self.config.sections["EXTRAS"].apply_transpose:
            G, c, _ = self._fit_statistics(a, b, w, fs_dict, trainall)
            fit = self._solve(kind, alval, G.T @ G, G.T @ c)
            if pt._rank == 0:
                self.fit = fit
            return
        # what lies behind the Cholesky factorisations -- sklearn's SVD fallback (RIDGE), np.linalg.inv's LU (local solver)
        # -- runs in LAPACK for large K (Solver._solve); the library's own versions are single-core Jacobi / scalar LU
        probe = _capi.PROBE_OF[kind]
        fit = self._fit_and_solve(probe, alval, a, b, w, fs_dict, trainall)
        fit = self._resolve_probe(probe, alval, fit)
        if self.refine_steps:        # off by default: the reference's ridge is itself a normal-equation solve
            fit = self._refine(fit, kind, alval, self.refine_steps)
        if pt._rank == 0:
            self.fit = fit
