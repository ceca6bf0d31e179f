"""ANL (analytical Bayesian linear fit) behind the reference's plugin API
(fitsnap3lib/solvers/anl.py:7-68): posterior mean ``pinv(aw.T aw + nugget I) aw.T bw`` and
covariance ``sigma_hat * pinv(...)`` — another consumer of the same GPU statistics (G, c)
plus one streamed residual pass (``fsnap_predict``'s weighted SSE).  The K x K pseudo-inverse
is host algebra, as in the reference."""
from __future__ import annotations

import numpy as np

from .._hostblas import blas_threads
from .solver import Solver


class ANL(Solver):

    def __init__(self, name, pt, config):
        super().__init__(name, pt, config)
        self.save_files = True      # the reference writes covariance.npy / mean.npy into the cwd (anl.py:60-61)

    def perform_fit(self, a=None, b=None, w=None, trainall=False):
        pt, config = self.pt, self.config
        G, c, s = self._fit_statistics(a, b, w, None, trainall)
        nbas = len(c)
        npt = float(s[2])
        cov_nugget = config.sections["SOLVER"].cov_nugget
        transposed = False
        if config.sections["EXTRAS"].apply_transpose:
            # anl.py:31-36: the regression is run on (aw.T aw, aw.T bw) = (G, c) instead of the rows when
            # cond(aw)^2 = lambda_max(G) / lambda_min(G) < 1 / eps -- all of it K x K host algebra on the statistics
            with blas_threads(len(c)):
                ev = np.linalg.eigvalsh(G)
            if abs(ev[-1]) / max(abs(ev[0]), np.finfo(float).tiny) < 1.0 / np.finfo(float).eps:
                transposed = True
            else:
                print("The Matrix is ill-conditioned for the transpose trick")
        if transposed:
            with blas_threads(nbas):
                invptp = np.linalg.pinv(G.T @ G + cov_nugget * np.diag(np.ones((nbas,))))
            invptp = invptp * 0.5 + invptp.T * 0.5
            fit = np.dot(invptp, G.T @ c)
            res = c - G @ fit
            sse = float(res @ res)
            npt = float(nbas)                       # the "rows" of the transposed system
        else:
            with blas_threads(nbas):            # (an SVD of K x K: _hostblas.py)
                invptp = np.linalg.pinv(G + cov_nugget * np.diag(np.ones((nbas,))))       # anl.py:39
            invptp = invptp * 0.5 + invptp.T * 0.5                                     # anl.py:40
            fit = np.dot(invptp, c)
            # res = bw - aw @ fit; bp = res.res / 2  (anl.py:46-47): exact streamed residual on the GPU
            ctx = pt.hip()
            _, sse = ctx.predict(fit, want_preds=False, want_sse=True)
            sse = pt.allreduce_scalar(sse)
        bp = sse / 2.0
        ap = (npt - nbas) / 2.0
        sigmahat = bp / (ap - 1.0)
        if pt._rank != 0:
            return
        self.fit = fit
        self.cov = sigmahat * invptp                                                # anl.py:53
        if self.save_files:
            np.save("covariance.npy", self.cov)
            np.save("mean.npy", self.fit)
        nsam = config.sections["SOLVER"].nsam
        if nsam:
            self.fit_sam = np.random.multivariate_normal(self.fit, self.cov, size=(nsam,))   # anl.py:63-65
