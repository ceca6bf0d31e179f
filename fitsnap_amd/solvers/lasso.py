"""LASSO solver behind the reference's plugin API (fitsnap3lib/solvers/lasso.py:9-29).

The reference hands the weighted m x K matrix to scikit-learn's ``Lasso(alpha, fit_intercept=False, max_iter)``, whose
coordinate descent touches the rows only through ``X^T X``, ``X^T y`` and ``|y|^2`` (scikit-learn's own Gram variant,
linear_model/_cd_fast.pyx ``enet_coordinate_descent_gram``, is the same iteration).  Here those three come from ONE pass
of the fused GPU statistics kernel (summed over the ranks) and the sweeps run on the K x K statistics inside the library
(``fsnap_lasso_gram``): the same iterates in exact arithmetic, 1e-11 from the reference class's coefficients on the Ta
golden rows with identical sweep counts, and K^2 instead of m K flops per sweep."""
from __future__ import annotations

from .. import _capi
from .solver import Solver


class LASSO(Solver):

    TOL = 1.0e-4        # sklearn Lasso's default ``tol`` (the reference does not set it)

    def __init__(self, name, pt, config):
        super().__init__(name, pt, config)
        self.n_iter_ = None
        self.dual_gap_ = None

    def perform_fit(self):
        """lasso.py:15-29 -- no arguments: data comes from ``pt.shared_arrays``."""
        pt = self.pt
        G, c, s = self._fit_statistics(None, None, None, None, False)
        y_norm2, n = float(s[0]), float(s[2])
        if self.config.sections["EXTRAS"].apply_transpose:
            # lasso.py:22-24: X = aw.T aw = G (K "samples"), y = aw.T bw = c
            X, y = G, c
            G, c = X.T @ X, X.T @ y
            y_norm2, n = float(y @ y), float(len(y))
        sec = self.config.sections["LASSO"]
        coef, self.n_iter_, gap = _capi.lasso_gram(G, c, y_norm2, sec.alpha * n, sec.max_iter, self.TOL)
        self.dual_gap_ = gap / n if n > 0 else gap          # sklearn reports the gap per sample
        if self.n_iter_ >= sec.max_iter and gap >= self.TOL * y_norm2:
            pt.single_print("LASSO: objective did not converge in %d sweeps (duality gap %.3e, tolerance %.3e); "
                            "increase max_iter or alpha" % (sec.max_iter, gap, self.TOL * y_norm2))
        if pt._rank == 0:
            self.fit = coef
