"""CPU oracle for the FitSNAP linear-fit hot path.  TEST INFRASTRUCTURE ONLY.

This module restates, in plain numpy/scipy, the reference algorithm that the HIP path
replaces.  It is the *checker*: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under ``fitsnap_amd/``
imports it; the product path fails loudly when the HIP library is missing instead of
falling back to this code.

Pinning: every function below is checked in ``tests/test_oracle_golden.py`` against
golden vectors produced by importing the *reference itself* in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/ta_reference_fits.npz``) and against
the reference's committed ``Ta_pot.snapcoeff`` / ``Ta_metrics.md``.  ARD: the reference's class passes
``n_iter=`` to ARDRegression (fitsnap3lib/solvers/ard.py:40-45), a keyword scikit-learn renamed to ``max_iter`` (1.3)
and removed (1.5), so the golden generator runs the class with that one keyword forwarded under its new name and
``ard_fit`` is pinned bit-for-bit to those vectors (all rows / testing mask / directmethod / non-default scap, scai,
logcut / apply_transpose).  No test or fixture of the reference itself holds an ARD output.  LASSO (``lasso_fit``):
the reference class runs as it stands; pinned bit-for-bit to its vectors.

Reference citations are file:line into FitSNAP/FitSNAP (/root/reference at build time).
"""
from __future__ import annotations

import numpy as np
import scipy.linalg


# --------------------------------------------------------------------------------------
# mask + weighting
# --------------------------------------------------------------------------------------
def training_mask(m, testing=None):
    """fitsnap3lib/solvers/svd.py:35-40 (same in ridge.py:28-33, ard.py:18).

    ``training = [not elem for elem in fs_dict['Testing']]``; ``trainall`` -> all True.
    Returns a boolean ndarray (the reference builds a Python list and fancy-indexes).
    """
    if testing is None:
        return np.ones(m, dtype=bool)
    return ~np.asarray(testing, dtype=bool)


def weight_rows(a, b, w, testing=None):
    """fitsnap3lib/solvers/svd.py:44-46: ``aw, bw = w[:,None]*a[training], w*b[training]``."""
    tr = training_mask(len(b), testing)
    wt = np.asarray(w, dtype=np.float64)[tr]
    aw = wt[:, np.newaxis] * np.asarray(a, dtype=np.float64)[tr]
    bw = wt * np.asarray(b, dtype=np.float64)[tr]
    return aw, bw


def weight_rows_full(a, b, w, testing=None):
    """All m rows, masked rows zeroed — the layout ``fsnap_weight_rows`` writes (the
    reference compacts rows instead, svd.py:46; row order/zero rows do not change G, c)."""
    tr = training_mask(len(b), testing)
    wt = np.where(tr, np.asarray(w, dtype=np.float64), 0.0)
    aw = np.where(tr[:, None], wt[:, None] * np.asarray(a, dtype=np.float64), 0.0)
    bw = np.where(tr, wt * np.asarray(b, dtype=np.float64), 0.0)
    return aw, bw


# --------------------------------------------------------------------------------------
# normal equations (the "transpose trick")
# --------------------------------------------------------------------------------------
def normal_eq(a, b, w, testing=None):
    """G = aw.T @ aw, c = aw.T @ bw and the scalar statistics.

    fitsnap3lib/solvers/svd.py:50-51, ridge.py:42-43, ard.py:23-24,
    fitsnap3lib/lib/ridge_solver/regressor.py:11-12,
    examples/library/transpose_trick/example.py:234-240 (per-configuration
    ``c += aw.T@aw; d += aw.T@bw`` followed by Allreduce at :245-246).
    Returns (G, c, scalars) with scalars = [bw.bw, sum(bw), n_train].
    """
    aw, bw = weight_rows(a, b, w, testing)
    G = aw.T @ aw
    c = aw.T @ bw
    scal = np.array([bw @ bw, bw.sum(), float(len(bw))])
    return G, c, scal


# --------------------------------------------------------------------------------------
# solvers
# --------------------------------------------------------------------------------------
def svd_fit(a, b, w, testing=None, apply_transpose=False):
    """fitsnap3lib/solvers/svd.py:44-54.

    ``lstsq(aw, bw, 1.0e-13)`` (LAPACK gelsd); with EXTRAS.apply_transpose the system is
    replaced by (aw.T aw, aw.T bw) when ``cond(aw)**2 < 1/eps`` (svd.py:48-53).
    """
    aw, bw = weight_rows(a, b, w, testing)
    if apply_transpose:
        if np.linalg.cond(aw) ** 2 < 1 / np.finfo(float).eps:
            bw = aw.T @ bw
            aw = aw.T @ aw
    fit, _res, _rank, _s = scipy.linalg.lstsq(aw, bw, 1.0e-13)
    return fit


def local_ridge(X, y, alpha):
    """fitsnap3lib/lib/ridge_solver/regressor.py:10-16: ``inv(X.T X + alpha I) @ X.T y``."""
    xty = np.matmul(X.T, y)
    xtx = np.matmul(X.T, X)
    normal = xtx + alpha * np.eye(xtx.shape[0])
    return np.matmul(np.linalg.inv(normal), xty)


def sklearn_ridge_dense(X, y, alpha):
    """What ``sklearn.linear_model.Ridge(alpha, fit_intercept=False).fit(X, y)`` computes
    for a dense X with n_samples > n_features (third-party: scikit-learn 1.7.2,
    linear_model/_ridge.py ``_solve_cholesky``: ``A = X.T X; A.flat[::n+1] += alpha;
    scipy.linalg.solve(A, X.T y, assume_a='pos')``), as used at
    fitsnap3lib/solvers/ridge.py:47-57.  For n_samples <= n_features sklearn uses the
    kernel form; the call sites here always have m >> K except with apply_transpose,
    where X is K x K and both forms are the same linear system."""
    n = X.shape[1]
    if X.shape[0] > n:
        A = X.T @ X
        Xy = X.T @ y
        A.flat[:: n + 1] += alpha
        return scipy.linalg.solve(A, Xy, assume_a="pos", overwrite_a=True)
    Kmat = X @ X.T
    Kmat.flat[:: Kmat.shape[0] + 1] += alpha
    dual = scipy.linalg.solve(Kmat, y, assume_a="pos", overwrite_a=True)
    return X.T @ dual


def ridge_fit(a, b, w, alpha=1.0e-8, local_solver=False, testing=None, apply_transpose=False):
    """fitsnap3lib/solvers/ridge.py:37-59 (alpha default io/sections/solver_sections/ridge.py:13)."""
    aw, bw = weight_rows(a, b, w, testing)
    if apply_transpose:
        bw = aw.T @ bw
        aw = aw.T @ aw
    if local_solver:
        return local_ridge(aw, bw, alpha)
    return sklearn_ridge_dense(aw, bw, alpha)


def ard_hyper(bw, scap=1.0e-3, scai=1.0e-3, logcut=0.3):
    """Hyper-parameters of the non-direct ARD method, fitsnap3lib/solvers/ard.py:26-43."""
    ap = 1.0 / np.var(bw)
    return dict(alpha_1=scap * ap, alpha_2=scap * ap, lambda_1=ap * scai, lambda_2=ap * scai,
                threshold_lambda=10 ** (int(np.abs(np.log10(ap))) + logcut))


def ard_fit(a, b, w, testing=None, directmethod=False, alphabig=1.0e-12, lambdasmall=1.0e-6,
            threshold_lambda=100000, scap=1.0e-3, scai=1.0e-3, logcut=0.3, max_iter=1000, apply_transpose=False):
    """fitsnap3lib/solvers/ard.py:18-48 with ``n_iter`` spelled ``max_iter`` (scikit-learn renamed the keyword in 1.3
    and dropped the old name in 1.5).  Pinned bit-for-bit by tests/test_oracle_golden.py to vectors produced by the
    reference CLASS run with that one keyword forwarded (tests/golden/make_golden.py).  Third-party arithmetic:
    scikit-learn ARDRegression."""
    from sklearn.linear_model import ARDRegression

    aw, bw = weight_rows(a, b, w, testing)
    if apply_transpose:                                            # ard.py:22-24
        bw = aw.T @ bw
        aw = aw.T @ aw
    if directmethod:
        reg = ARDRegression(max_iter=max_iter, threshold_lambda=threshold_lambda, alpha_1=alphabig,
                            alpha_2=alphabig, lambda_1=lambdasmall, lambda_2=lambdasmall, fit_intercept=False)
    else:
        reg = ARDRegression(max_iter=max_iter, fit_intercept=False, **ard_hyper(bw, scap, scai, logcut))
    reg.fit(aw, bw)
    return reg.coef_


def ard_fit_extended(a, b, w, testing=None, directmethod=False, alphabig=1.0e-12, lambdasmall=1.0e-6,
                     threshold_lambda=100000, scap=1.0e-3, scai=1.0e-3, logcut=0.3, max_iter=1000, tol=1.0e-3):
    """The SAME iteration in extended precision (numpy longdouble: 64-bit mantissa on x86): scikit-learn 1.7.2
    ``ARDRegression.fit`` / ``_update_sigma`` (linear_model/_bayes.py) as called by fitsnap3lib/solvers/ard.py:18-48, with
    the Gram matrix, the residual and the K x K inverse (Cholesky of the column-equilibrated matrix) all formed in long
    double.  Test yardstick only: on the Ta rows the reference class's own float64 answer sits 3e-4 (element-wise) from
    this one -- ``pinvh(lambda I + alpha X^T X)`` on columns that span 15 decades loses kappa eps -- so "who is closer to
    the exact iteration" is what a 1e-6 bar can be held against, not the float64 golden itself.  Returns (coef, n_iter)."""
    LD = np.longdouble
    aw, bw = weight_rows(a, b, w, testing)
    if directmethod:
        hyper = dict(threshold_lambda=threshold_lambda, alpha_1=alphabig, alpha_2=alphabig, lambda_1=lambdasmall, lambda_2=lambdasmall)
    else:
        hyper = ard_hyper(bw, scap, scai, logcut)
    a1, a2, l1, l2, thr = (hyper[k] for k in ("alpha_1", "alpha_2", "lambda_1", "lambda_2", "threshold_lambda"))
    awl, bwl = aw.astype(LD), bw.astype(LD)
    G, c = awl.T @ awl, awl.T @ bwl
    K = len(c)
    n = LD(len(bw))
    d = np.sqrt(np.diag(G))
    d[d == 0] = 1
    Gh = G / np.outer(d, d)

    def spd_inverse(M):
        k = M.shape[0]
        L = np.zeros_like(M)
        for j in range(k):
            L[j, j] = np.sqrt(M[j, j] - np.dot(L[j, :j], L[j, :j]))
            for i in range(j + 1, k):
                L[i, j] = (M[i, j] - np.dot(L[i, :j], L[j, :j])) / L[j, j]
        Li = np.zeros_like(M)
        for j in range(k):
            Li[j, j] = 1 / L[j, j]
            for i in range(j + 1, k):
                Li[i, j] = -np.dot(L[i, j:i], Li[j:i, j]) / L[i, i]
        return Li.T @ Li

    def sigma_of(alpha_, lambda_, keep):
        dk = d[keep]
        return spd_inverse(np.diag(lambda_[keep] / dk ** 2) + alpha_ * Gh[np.ix_(keep, keep)]) / np.outer(dk, dk)

    eps = np.finfo(np.float64).eps
    coef_ = np.zeros(K, dtype=LD)
    keep = np.ones(K, dtype=bool)
    alpha_ = LD(1.0) / (LD(np.var(bw)) + eps)
    lambda_ = np.ones(K, dtype=LD)
    coef_old = None
    it = 0
    for it in range(max_iter):
        sigma_ = sigma_of(alpha_, lambda_, keep)
        coef_[keep] = alpha_ * (sigma_ @ c[keep])
        sse_ = np.sum((bwl - awl @ coef_) ** 2)
        gamma_ = 1 - lambda_[keep] * np.diag(sigma_)
        lambda_[keep] = (gamma_ + 2 * l1) / (coef_[keep] ** 2 + 2 * l2)
        alpha_ = (n - gamma_.sum() + 2 * a1) / (sse_ + 2 * a2)
        keep = lambda_ < thr
        coef_[~keep] = 0
        if it > 0 and np.sum(np.abs(coef_old - coef_)) < tol:
            break
        coef_old = coef_.copy()
        if not keep.any():
            break
    if keep.any():
        sigma_ = sigma_of(alpha_, lambda_, keep)
        coef_[keep] = alpha_ * (sigma_ @ c[keep])
    return np.asarray(coef_, dtype=np.float64), it + 1


def lasso_fit(a, b, w, testing=None, alpha=1.0e-8, max_iter=2000, apply_transpose=False):
    """fitsnap3lib/solvers/lasso.py:17-29: ``Lasso(alpha, fit_intercept=False, max_iter).fit(aw, bw).coef_`` (defaults:
    io/sections/solver_sections/lasso.py:13-14).  Pinned bit-for-bit to vectors produced by the reference class
    (tests/golden/make_golden.py).  Third-party arithmetic: scikit-learn's coordinate descent."""
    from sklearn.linear_model import Lasso

    aw, bw = weight_rows(a, b, w, testing)
    if apply_transpose:                                            # lasso.py:22-24
        bw = aw.T @ bw
        aw = aw.T @ aw
    reg = Lasso(alpha=alpha, fit_intercept=False, max_iter=max_iter)
    reg.fit(aw, bw)
    return reg.coef_


# --------------------------------------------------------------------------------------
# downstream of the fit
# --------------------------------------------------------------------------------------
def predict(a, fit):
    """fitsnap3lib/solvers/solver.py:377: ``preds = a @ fit``."""
    return np.asarray(a) @ np.asarray(fit)


def error_row(truths, preds, weights):
    """fitsnap3lib/solvers/solver.py:108-133 (``_ncount_mae_rmse_rsq_unweighted_and_weighted``)."""
    truths = np.asarray(truths, dtype=np.float64)
    preds = np.asarray(preds, dtype=np.float64)
    weights = np.asarray(weights, dtype=np.float64)
    res = truths - preds
    n = len(truths)
    ssr = np.square(res).sum()
    out = {"ncount": n, "mae": np.mean(np.abs(res)), "rmse": np.sqrt(ssr / n),
           "rsq": 1 - ssr / np.sum(np.square(truths - (truths / n).sum()))}
    w_res = weights * res
    w_n = np.count_nonzero(weights)
    w_ssr = np.square(w_res).sum()
    out.update({"w_ncount": w_n, "w_mae": np.mean(np.abs(w_res)), "w_rmse": np.sqrt(w_ssr / w_n),
                "w_rsq": 1 - w_ssr / np.sum(np.square(weights * truths - (weights * truths / w_n).sum()))})
    return out


# --------------------------------------------------------------------------------------
# synthetic workload of SURVEY.md 8(d) / BASELINE.md 2: input data, generated by fitsnap_amd.synthetic (the tests reach
# it through these names)
# --------------------------------------------------------------------------------------
from fitsnap_amd.synthetic import (SYNTH_CHUNK, SYNTH_SEED, synth_chunk, synth_params, synth_problem,  # noqa: E402,F401
                                   synth_testing_mask)


def anl_fit(a, b, w, testing=None, cov_nugget=0.0, apply_transpose=False):
    """fitsnap3lib/solvers/anl.py:31-53: posterior mean and covariance of the Bayesian linear fit; with
    EXTRAS.apply_transpose the regression runs on (aw.T aw, aw.T bw) when cond(aw)^2 < 1/eps (anl.py:31-36)."""
    aw, bw = weight_rows(a, b, w, testing)
    if apply_transpose and np.linalg.cond(aw) ** 2 < 1 / np.finfo(float).eps:
        bw = aw.T @ bw
        aw = aw.T @ aw
    npt, nbas = aw.shape
    invptp = np.linalg.pinv(np.dot(aw.T, aw) + cov_nugget * np.diag(np.ones((nbas,))))
    invptp = invptp * 0.5 + invptp.T * 0.5
    fit = np.dot(invptp, np.dot(aw.T, bw))
    res = bw - np.dot(aw, fit)
    bp = np.dot(res, res) / 2.0
    ap = (npt - nbas) / 2.0
    return fit, (bp / (ap - 1.0)) * invptp
