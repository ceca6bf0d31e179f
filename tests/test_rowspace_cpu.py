"""CPU: the host half of the row-space least-squares solve (fsnap_rowspace_factor / fsnap_rowspace_solve, no GPU) --
numpy stands in for the two kernels of a pass (Q <- Q Rp^-1 by substitution = scipy solve_triangular, G = Q^T Q).
Checked against the oracle's lstsq(aw, bw, 1e-13) (fitsnap3lib/solvers/svd.py:54): two backward-stable solvers differ
by ~kappa eps, so that is the bar beyond kappa ~ 1e9; below it the north-star tolerance 1e-6 applies."""
import os

import numpy as np
import pytest
import scipy.linalg as sl

from fitsnap_amd import _capi
from oracle import fitsnap_oracle as orc

from conftest import ROOT

EPS = np.finfo(float).eps


def trsm_substitution(R, X):
    """Q = X R^-1 by substitution: kernels 13 / 13A."""
    return sl.solve_triangular(R, X.T, trans="T", lower=False).T


def trsm_block_inverse(R, X, nb=16):
    """Q = X R^-1 the way kernel 13B does it (fsnap_trsm.hip): right-looking over 16-column blocks; a diagonal block is
    solved as q0 = x T^-1, r = x - q0 T, q = q0 + r T^-1 with the explicit inverse of the 16 x 16 block (one step of
    iterative refinement on top of the multiplication by the inverse), everything else is matrix products."""
    K = R.shape[0]
    S = X.copy()
    Q = np.empty_like(X)
    for j0 in range(0, K, nb):
        j1 = min(j0 + nb, K)
        T = R[j0:j1, j0:j1]
        Tinv = sl.solve_triangular(T, np.eye(j1 - j0))
        x = S[:, j0:j1]
        q = x @ Tinv
        q = q + (x - q @ T) @ Tinv
        Q[:, j0:j1] = q
        S[:, j1:] -= q @ R[j0:j1, j1:]
    return Q


def cholqr_lstsq(A, b, rcond=1.0e-13, maxpass=6, trsm=trsm_substitution):
    """The algorithm of fsnap_lstsq_rows with numpy in place of the GPU passes."""
    G = A.T @ A
    Q = A.copy()
    Rhat = None
    passes = 0
    for _ in range(maxpass):
        Rp, Rhat, info = _capi.rowspace_factor(G, Rhat)
        if Rp is None:
            break
        Q = trsm(Rp, Q)
        G = Q.T @ Q
        passes += 1
    beta, rank, sinfo = _capi.rowspace_solve(Rhat, Q.T @ b, rcond)
    step, _, _ = _capi.rowspace_solve(Rhat, Q.T @ (b - A @ beta), rcond)
    return beta + step, rank, passes, sinfo, Q, Rhat


def conditioned(m, K, kappa, mode, seed):
    r = np.random.default_rng(seed)
    U, _ = np.linalg.qr(r.standard_normal((m, K)))
    V, _ = np.linalg.qr(r.standard_normal((K, K)))
    if mode == "geometric":
        s = np.logspace(0, -np.log10(kappa), K)
    else:                                   # one weak direction
        s = np.ones(K)
        s[-1] = 1.0 / kappa
    return (U * s) @ V.T


@pytest.mark.parametrize("trsm", [trsm_substitution, trsm_block_inverse], ids=["substitution", "block_inverse_refined"])
@pytest.mark.parametrize("mode", ["geometric", "one"])
@pytest.mark.parametrize("kappa", [1e4, 1e8, 1e10, 1e12])
def test_matches_lstsq_up_to_kappa_eps(kappa, mode, trsm):
    # both forms of the pass: kernels 13 / 13A divide by R with a substitution, kernel 13B (K > 128) multiplies by the
    # inverses of the 16 x 16 diagonal blocks and refines once -- same orthogonality, same A = Q R_hat, same coefficients
    m, K = 6000, 48
    A = conditioned(m, K, kappa, mode, 3)
    r = np.random.default_rng(4)
    b = A @ r.standard_normal(K) + 1e-3 * r.standard_normal(m)
    ref = orc.svd_fit(A, b, np.ones(m))
    x, rank, passes, sinfo, Q, Rhat = cholqr_lstsq(A, b, trsm=trsm)
    assert rank == K and 2 <= passes <= 4
    assert np.abs(Q.T @ Q - np.eye(K)).max() < 1e-10                       # orthonormal columns
    assert np.linalg.norm(Q @ Rhat - A) <= 50 * K * EPS * np.linalg.norm(A)  # A = Q R_hat to working precision
    tol = max(1e-6 if kappa <= 1e8 else 0.0, 50 * kappa * EPS)
    assert np.linalg.norm(x - ref) <= tol * np.linalg.norm(ref)
    # what both solvers minimise agrees far below the coefficient tolerance
    res, res_ref = np.linalg.norm(A @ x - b), np.linalg.norm(A @ ref - b)
    assert res <= res_ref * (1 + max(1e-9, 50 * (kappa * EPS) ** 2)) and abs(res - res_ref) <= 1e-5 * res_ref


def test_exact_rank_deficiency_gives_the_minimum_norm_solution():
    r = np.random.default_rng(5)
    m = 5000
    A = r.standard_normal((m, 20))
    A = np.hstack([A, A[:, :3], A[:, 3:5] @ r.standard_normal((2, 2)), np.zeros((m, 2))])   # duplicates, combinations, zeros
    b = r.standard_normal(m)
    ref = orc.svd_fit(A, b, np.ones(m))
    x, rank, passes, sinfo, _, _ = cholqr_lstsq(A, b)
    assert rank == 20 and sinfo[0] == 1.0                                   # SVD path, 7 directions dropped
    assert np.linalg.norm(x - ref) <= 1e-9 * np.linalg.norm(ref)
    assert np.all(x[-2:] == 0.0)                                            # zero columns: coefficient 0 (lstsq: minimum norm)
    assert np.allclose(x[:3], x[20:23], rtol=1e-9)                          # duplicated columns share their coefficient


def test_near_collinear_columns_kept_or_dropped_like_gelsd():
    r = np.random.default_rng(6)
    m, K = 4000, 12
    base = r.standard_normal((m, K))
    b = r.standard_normal(m)
    for eps_col, expect_rank in ((1e-9, K + 1), (1e-15, K)):
        A = np.hstack([base, base[:, :1] + eps_col * r.standard_normal((m, 1))])
        _, _, rank_ref, _ = sl.lstsq(A, b, 1.0e-13)
        ref = orc.svd_fit(A, b, np.ones(m))
        x, rank, _, _, _, _ = cholqr_lstsq(A, b)
        assert rank == rank_ref == expect_rank
        if eps_col < 1e-13:     # direction dropped: a well-posed minimum-norm problem, tight agreement
            assert np.linalg.norm(x - ref) <= 1e-8 * np.linalg.norm(ref)
        else:                   # direction kept at kappa ~ 1e9 (coefficients +-5e6 that cancel): ~kappa eps each
            assert np.linalg.norm(A @ (x - ref)) <= 1e-7 * np.linalg.norm(b)
            assert np.linalg.norm(x - ref) <= 50 * 1e9 * EPS * np.linalg.norm(ref)


def test_graded_columns_truncate_on_the_singular_values_of_aw_not_of_the_equilibrated_matrix():
    # lstsq cuts on sigma(A_w): a column 1e-15 times smaller than the others is dropped although the column-scaled
    # matrix is perfectly conditioned -- the Jacobi scaling inside the passes must not change that decision
    r = np.random.default_rng(8)
    m, K = 3000, 10
    A = r.standard_normal((m, K)) * np.array([1.0] * (K - 2) + [1e-6, 1e-15])
    b = r.standard_normal(m)
    fit, _, rank_ref, _ = sl.lstsq(A, b, 1.0e-13)
    x, rank, _, _, _, _ = cholqr_lstsq(A, b)
    assert rank == rank_ref == K - 1
    assert np.linalg.norm(x[:-1] - fit[:-1]) <= 1e-8 * np.linalg.norm(fit[:-1])
    assert abs(x[-1]) <= 1e-9 * np.abs(fit[:-1]).max() and abs(fit[-1]) <= 1e-9 * np.abs(fit[:-1]).max()


def test_factor_reports_convergence_and_refuses_non_finite_input():
    Q, _ = np.linalg.qr(np.random.default_rng(1).standard_normal((200, 7)))
    Rp, Rhat, info = _capi.rowspace_factor(Q.T @ Q)             # first pass always factorises
    assert Rp is not None and info[0] < 1e-14
    Rp2, Rhat2, info2 = _capi.rowspace_factor(Q.T @ Q, Rhat)
    assert Rp2 is None and info2[1] == 1.0 and np.array_equal(Rhat2, Rhat)
    bad = np.eye(3)
    bad[1, 2] = np.nan
    with pytest.raises(ValueError):
        _capi.rowspace_factor(bad)
    with pytest.raises(ValueError):
        _capi.rowspace_solve(np.eye(3), np.array([1.0, np.inf, 0.0]))


def _cholqr_factors(A, passes=2):
    """Upper triangular factors of CholeskyQR passes on A (numpy), first pass first."""
    Q, out = A, []
    for _ in range(passes):
        G = Q.T @ Q
        d = np.sqrt(np.diag(G))
        R = np.linalg.cholesky(G / np.outer(d, d) + 1e-13 * np.eye(len(d))).T * d
        Q = sl.solve_triangular(R, Q.T, trans="T", lower=False).T
        out.append(R)
    return out


@pytest.mark.parametrize("K,kappa,expect_chain", [(300, 1e3, True), (400, 1e7, True), (320, 1e10, True), (320, 1e11, False),
                                                  (320, 3e12, False), (320, 1e13, False), (320, 1e14, False)])
def test_factor_chain_solves_without_the_product(K, kappa, expect_chain):
    """fsnap_rowspace_chain (what fsnap_lstsq_rows does for K > 256): the factors stay apart; the solve goes through them
    by back substitution when the condition bound allows, through the multiplied-out factor + the SVD end otherwise.  Either
    way the answer is pinv(R_hat) z, and the bound must really be an upper bound of cond_2(R_hat)."""
    rng = np.random.default_rng(K)
    m = 2 * K
    U, _ = np.linalg.qr(rng.standard_normal((m, K)))
    V, _ = np.linalg.qr(rng.standard_normal((K, K)))
    A = (U * np.logspace(0, -np.log10(kappa), K)) @ V.T
    factors = _cholqr_factors(A)
    Rhat = factors[1] @ factors[0]
    z = rng.standard_normal(K)
    beta, rank, info = _capi.rowspace_chain(factors, z, 1.0e-13)
    s = np.linalg.svd(Rhat, compute_uv=False)
    assert info["cond_bound"] >= s[0] / s[-1]
    assert bool(info["chain"]) == expect_chain
    # the estimate is not a bound, so it certifies the chain only with two orders of margin; in the band
    # kappa * rcond in [1e-2, 1] and beyond, the multiplied-out factor is judged exactly -- the answer is gelsd's there too
    ref, _, rank_ref, _ = sl.lstsq(Rhat, z, cond=1.0e-13, lapack_driver="gelsd")
    assert rank == rank_ref and (rank < K) == (kappa > 1e13)
    if rank == K:
        assert np.linalg.norm(beta - ref) <= 1e-6 * np.linalg.norm(ref) * max(1.0, kappa * 1e-10)
    else:
        # truncated: the kept singular values just above the cut (1.0x e-13 sigma_max) carry the solution of a random right-hand
        # side, and an eps-sized difference between two roundings of R_hat moves them by per cents -- same rank, same directions
        assert np.linalg.norm(beta - ref) <= 0.1 * np.linalg.norm(ref)


def test_factor_chain_inactive_columns_and_truncation():
    rng = np.random.default_rng(9)
    K = 280
    A = rng.standard_normal((700, K))
    A[:, 5] = 0.0                                                  # a zero column: inactive, coefficient 0
    act = np.ones(K, dtype=np.uint8)
    act[5] = 0
    A5 = A.copy()
    A5[:, 5] = rng.standard_normal(700)                            # any non-singular stand-in for the factorisation
    factors = _cholqr_factors(A5)
    for f in factors:                                              # unit row / column for the inactive column, as factor_pass does
        f[5, :] = 0.0
        f[:, 5] = 0.0
        f[5, 5] = 1.0
    z = rng.standard_normal(K)
    beta, rank, info = _capi.rowspace_chain(factors, z, 1.0e-13, active=act)
    assert info["chain"] == 1.0 and rank == K - 1 and beta[5] == 0.0
    keep = act.astype(bool)
    Rhat = (factors[1] @ factors[0])[np.ix_(keep, keep)]
    assert np.allclose(beta[keep], np.linalg.solve(Rhat, z[keep]), rtol=1e-9, atol=1e-12)
    # an exactly dependent column pair: the bound cannot certify the chain, the product + SVD end drops one direction
    B = rng.standard_normal((700, K))
    B[:, 7] = B[:, 3]
    G = B.T @ B
    d = np.sqrt(np.diag(G))
    R = np.linalg.cholesky(G / np.outer(d, d) + 1e-12 * np.eye(K)).T * d
    beta2, rank2, info2 = _capi.rowspace_chain([R], z, 1.0e-4)         # sigma_min ~ 1e-6 (the shift) is below the cut
    assert info2["chain"] == 0.0 and rank2 == K - 1 and np.all(np.isfinite(beta2))
    ref2, _, rank_ref, _ = sl.lstsq(R, z, cond=1.0e-4)
    assert rank_ref == K - 1 and np.allclose(beta2, ref2, rtol=1e-7, atol=1e-9)


def test_dense_pinv_hook_is_gelsd_on_the_factor():
    """The callback the Python layer hands to fsnap_set_dense_pinv (LAPACK SVD of the K x K factor), driven directly."""
    import ctypes
    rng = np.random.default_rng(4)
    n = 40
    T = np.triu(rng.standard_normal((n, n))) + 3.0 * np.eye(n)
    T[:, 9] = T[:, 2]                                              # exactly dependent columns: one direction to drop
    T = np.triu(T)
    T = np.ascontiguousarray(T)
    y = rng.standard_normal(n)
    x = np.zeros(n)
    rank = ctypes.c_int(-1)
    cb = _capi.make_dense_pinv()
    dp = ctypes.POINTER(ctypes.c_double)
    for token in (1, 1, 2):                                        # second call reuses the cached decomposition
        rc = cb(None, token, n, T.ctypes.data_as(dp), 1.0e-10, y.ctypes.data_as(dp), x.ctypes.data_as(dp), ctypes.byref(rank))
        ref, _, rank_ref, _ = sl.lstsq(T, y, cond=1.0e-10)
        assert rc == 0 and rank.value == rank_ref and np.allclose(x, ref, rtol=1e-9, atol=1e-11)


def _factor_with_dependent_columns(n, ndep, kept_cond, seed, noise=0.0):
    """Upper-triangular factor of a matrix whose `ndep` columns are combinations of three others (+ relative `noise`)."""
    r = np.random.default_rng(seed)
    A = r.standard_normal((4 * n, n)) * np.exp(r.uniform(np.log(1.0 / kept_cond), 0.0, n))[None, :]
    dep = r.choice(n, ndep, replace=False)
    others = np.setdiff1d(np.arange(n), dep)
    for d in dep:
        pick = r.choice(others, 3, replace=False)
        col = A[:, pick] @ r.standard_normal(3)
        A[:, d] = col * (np.linalg.norm(A[:, d]) / np.linalg.norm(col))
        if noise:
            A[:, d] += noise * np.linalg.norm(A[:, d]) / np.sqrt(4 * n) * r.standard_normal(4 * n)
    R = np.linalg.qr(A, mode="r")
    sg = np.sign(np.diag(R))
    sg[sg == 0] = 1.0
    return R * sg[:, None], r.standard_normal(n)


@pytest.mark.parametrize("n", [64, 128, 200])
@pytest.mark.parametrize("ndep", [1, 2, 3, 4])
@pytest.mark.parametrize("kept_cond,noise", [(1e2, 0.0), (1e4, 1e-15), (1e6, 0.0)])
def test_a_few_dropped_directions_are_projected_away_without_the_svd(n, ndep, kept_cond, noise, monkeypatch):
    # the K x K end with 1...4 singular values below the cut and a gap above them (duplicated / combined descriptor columns):
    # info[0] == 3 says the truncated solution came from back substitution between two projections (FactorSolver::deflate);
    # it agrees with the library's own Jacobi SVD (FSNAP_ROWSPACE_DEFLATE=0) to ~eps x the condition of the kept part and
    # with LAPACK's gelsd as closely as the Jacobi SVD does
    R, z = _factor_with_dependent_columns(n, ndep, kept_cond, seed=1000 * n + 10 * ndep + int(np.log10(kept_cond)), noise=noise)
    ref, _, rank_ref, sv = np.linalg.lstsq(R, z, rcond=1.0e-13)
    assert rank_ref == n - ndep
    beta, rank, info = _capi.rowspace_solve(R, z, 1.0e-13)
    assert rank == rank_ref and info[0] == 3.0
    assert info[1] <= sv[0] * (1 + 1e-12) and info[1] >= 0.5 * sv[0]            # sigma_max from below
    assert info[2] <= sv[rank - 1] * (1 + 1e-9) and info[2] >= sv[rank - 1] / (4 * np.sqrt(n))   # smallest kept one from below
    monkeypatch.setenv("FSNAP_ROWSPACE_DEFLATE", "0")
    beta_j, rank_j, info_j = _capi.rowspace_solve(R, z, 1.0e-13)
    monkeypatch.delenv("FSNAP_ROWSPACE_DEFLATE")
    assert rank_j == rank and info_j[0] == 1.0
    scale = np.abs(beta_j).max()
    assert np.abs(beta - beta_j).max() <= 200 * kept_cond * EPS * scale
    err, err_j = np.abs(beta - ref).max(), np.abs(beta_j - ref).max()
    assert err <= 2.0 * err_j + 200 * kept_cond * EPS * scale


def test_the_projection_path_steps_aside_when_it_cannot_call_the_rank():
    n = 128
    r = np.random.default_rng(77)
    U, _ = np.linalg.qr(r.standard_normal((n, n)))
    V, _ = np.linalg.qr(r.standard_normal((n, n)))
    z = r.standard_normal(n)

    def solve(s):
        R = np.linalg.qr((U * s) @ V.T, mode="r")
        sg = np.sign(np.diag(R))
        return _capi.rowspace_solve(R * sg[:, None], z, 1.0e-13), np.linalg.lstsq(R * sg[:, None], z, rcond=1.0e-13)

    base = np.sort(np.exp(r.uniform(np.log(1e-3), 0.0, n)))[::-1]
    base[0] = 1.0
    # (a) thirty values below the cut: more than the projection path takes on (4 with 8 vectors, 24 with 32)
    s = base.copy()
    s[-30:] = 1e-16 * np.arange(1, 31)
    (beta, rank, info), (ref, _, rk, _) = solve(s)
    assert rank == rk == n - 30 and info[0] == 1.0
    # (b) a value 1.5 x above the cut (kept, kappa ~ 7e12): the certificate of the deflated inverse cannot close
    s = base.copy()
    s[-1] = 1.5e-13
    (beta, rank, info), (ref, _, rk, _) = solve(s)
    assert rank == rk == n and info[0] == 1.0
    # (c) one below, one just above the cut
    s = base.copy()
    s[-2:] = [2.0e-13, 1e-15]
    (beta, rank, info), (ref, _, rk, _) = solve(s)
    assert rank == rk == n - 1 and info[0] == 1.0
    # (d) no gap at all: a geometric ladder down through the cut
    s = np.logspace(0, -16, n)
    (beta, rank, info), (ref, _, rk, _) = solve(s)
    assert rank == rk and info[0] == 1.0


@pytest.mark.parametrize("n,ndep", [(128, 5), (128, 12), (200, 8), (200, 24), (320, 17)])
def test_more_than_four_dropped_directions_take_the_wide_block(n, ndep, monkeypatch):
    # 5 ... 24 dependent columns: the subspace iteration runs again with 32 vectors (FactorSolver::deflate)
    R, z = _factor_with_dependent_columns(n, ndep, 1e4, seed=7 * n + ndep, noise=1e-15 if ndep % 2 else 0.0)
    ref, _, rank_ref, sv = np.linalg.lstsq(R, z, rcond=1.0e-13)
    assert rank_ref == n - ndep
    beta, rank, info = _capi.rowspace_solve(R, z, 1.0e-13)
    assert rank == rank_ref and info[0] == 3.0
    monkeypatch.setenv("FSNAP_ROWSPACE_DEFLATE", "0")
    beta_j, rank_j, info_j = _capi.rowspace_solve(R, z, 1.0e-13)
    monkeypatch.delenv("FSNAP_ROWSPACE_DEFLATE")
    assert rank_j == rank and info_j[0] == 1.0
    scale = np.abs(beta_j).max()
    assert np.abs(beta - beta_j).max() <= 200 * 1e4 * EPS * scale
    assert np.abs(beta - ref).max() <= 2.0 * np.abs(beta_j - ref).max() + 200 * 1e4 * EPS * scale


_THREAD_PROBE = r"""
import sys, hashlib, numpy as np
sys.path.insert(0, sys.argv[1])
from fitsnap_amd import _capi
K = 448
rng = np.random.default_rng(5)
U, _ = np.linalg.qr(rng.standard_normal((2 * K, K)))
V, _ = np.linalg.qr(rng.standard_normal((K, K)))
for spec, ndep in ((np.logspace(0, -9, K), 0), (np.logspace(0, -4, K), 5), (np.logspace(0, -4, K), 14)):
    A = (U * spec) @ V.T
    if ndep:
        dep = rng.choice(K, ndep, replace=False)
        others = np.setdiff1d(np.arange(K), dep)
        for d in dep:
            A[:, d] = A[:, rng.choice(others, 3, replace=False)] @ rng.standard_normal(3)
    R1 = np.linalg.qr(A, mode="r")
    R1 = R1 * np.sign(np.diag(R1))[:, None]
    R2 = np.eye(K) + 1e-2 * np.triu(rng.standard_normal((K, K))) / K
    z = rng.standard_normal(K)
    beta, rank, info = _capi.rowspace_chain([R1, R2], z, 1e-13)
    print(rank, info["chain"], repr(info["cond_bound"]), hashlib.sha256(beta.tobytes()).hexdigest())
"""


def test_large_k_host_phases_do_not_depend_on_the_thread_count(tmp_path):
    # from 384 columns on the K x K end runs on FSNAP_HOST_THREADS threads (estimators as tasks, cooperative substitutions,
    # product / inverse / projections by rows or columns): bit-identical coefficients and bounds with 1, 3 and 7 threads --
    # a chain that is certified, one with 5 dropped directions (16 vectors) and one with 14 (32 vectors)
    import subprocess
    import sys

    script = tmp_path / "probe.py"
    script.write_text(_THREAD_PROBE)
    outs = []
    for nt in ("1", "3", "7"):
        env = dict(os.environ, FSNAP_HOST_THREADS=nt)
        outs.append(subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=600, check=True).stdout)
    assert outs[0] == outs[1] == outs[2]
    lines = outs[0].strip().splitlines()
    assert len(lines) == 3 and lines[0].startswith("448 1.0") and lines[1].startswith("443 0.0") and lines[2].startswith("434 0.0")


def _upper_factor(A):
    R = np.linalg.qr(A, mode="r")
    return R * np.sign(np.diag(R))[:, None]


def _bound_families():
    rng = np.random.default_rng(11)
    for K in (64, 400):
        U, _ = np.linalg.qr(rng.standard_normal((2 * K, K)))
        V, _ = np.linalg.qr(rng.standard_normal((K, K)))
        for lk in (2.0, 6.0, 9.0, 10.5):
            yield f"graded-{K}-{lk}", _upper_factor((U * np.logspace(0, -lk, K)) @ V.T)
            one = np.ones(K)
            one[-1] = 10.0 ** -lk
            yield f"one-small-{K}-{lk}", _upper_factor((U * one) @ V.T)
            half = np.ones(K)
            half[K // 2:] = 10.0 ** -lk
            yield f"half-small-{K}-{lk}", _upper_factor((U * half) @ V.T)
    for theta, n in ((1.2, 64), (1.2, 90), (1.0, 64)):                 # Kahan's matrices: pivots say nothing about sigma_min
        yield f"kahan-{theta}-{n}", np.diag(np.sin(theta) ** np.arange(n)) @ (np.eye(n) - np.cos(theta) * np.triu(np.ones((n, n)), 1))
    for n in (18, 26, 40):                                              # the hidden family of the round-5 verdict, as a factor
        Z, _ = np.linalg.qr(rng.standard_normal((4000, n)))
        yield f"hidden-{n}", _upper_factor(Z @ (np.eye(n) - np.triu(np.ones((n, n)), 1)))


def test_condition_bound_of_a_factor_is_above_the_truth_and_sharp_where_it_decides():
    # The bound is built from LOWER estimates (Lanczos on R^T R and on its inverse -- round 6; 12 + 14 steps of power / inverse
    # iteration before --, Hager's 1-norm pair) times explicit margins: it must never fall below cond_2, and where the quick look
    # (margins of 10 per estimate) does not settle the question it must sit within the 2 x 1.25 of its margins
    rng = np.random.default_rng(3)
    for name, R in _bound_families():
        s = np.linalg.svd(R, compute_uv=False)
        cond = s[0] / s[-1]
        _, _, info = _capi.rowspace_chain([R], rng.standard_normal(len(R)), 1.0e-13)
        ratio = info["cond_bound"] / cond
        assert ratio >= 1.0, (name, ratio)
        assert ratio <= (5.0 if cond >= 1.0e9 else 2000.0), (name, ratio)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 31, 63, 64, 65, 67, 127, 129, 131, 197, 259])
def test_tiled_inverse_and_product_at_ragged_sizes(n):
    # The explicit inverse (64 x 64 tiles, strips of 32 columns, a 4 x 8 register block with scalar edges) only feeds certificates:
    # a wrong entry would not change a solution, it would change what the solver BELIEVES about sigma_min.  Pin it through the
    # numbers it reports -- smin = 1 / min(||X||_F, sqrt(||X||_1 ||X||_inf)) must be a lower bound of sigma_min within sqrt(n) --
    # at sizes that leave ragged rows (n % 4), ragged columns (n % 8), a ragged strip (n % 32) and a ragged tile (n % 64); and the
    # product of two factors (same tiles) through a chain that cannot be certified
    rng = np.random.default_rng(n)
    R = _upper_factor(rng.standard_normal((2 * n + 3, n)))
    z = rng.standard_normal(n)
    beta, rank, info = _capi.rowspace_solve(R, z, 1.0e-13)
    s = np.linalg.svd(R, compute_uv=False)
    assert rank == n and info[0] == 0.0                          # back substitution behind the Frobenius certificate
    assert np.allclose(beta, sl.solve_triangular(R, z), rtol=1e-9, atol=0)
    X = np.linalg.inv(R)
    expect = 1.0 / min(np.linalg.norm(X), np.sqrt(np.linalg.norm(X, 1) * np.linalg.norm(X, np.inf)))
    assert info[2] == pytest.approx(expect, rel=1e-10)
    assert info[2] <= s[-1] * (1 + 1e-12) and info[2] >= s[-1] / (2.0 * np.sqrt(n) + 1.0)
    if n >= 31:
        # two factors whose product has kappa = 1e12: no certificate for the chain, the factors are multiplied out (upper_product)
        U, _ = np.linalg.qr(rng.standard_normal((2 * n, n)))
        V, _ = np.linalg.qr(rng.standard_normal((n, n)))
        R1 = _upper_factor((U * np.logspace(0, -12, n)) @ V.T)
        R2 = np.eye(n) + np.triu(rng.standard_normal((n, n))) / (4.0 * n)
        beta2, rank2, info2 = _capi.rowspace_chain([R1, R2], z, 1.0e-13)
        assert not info2["chain"]
        ref, _, rank_ref, _ = sl.lstsq(R2 @ R1, z, cond=1.0e-13, lapack_driver="gelsd")
        assert rank2 == rank_ref
        assert np.linalg.norm(beta2 - ref) <= 1e-3 * np.linalg.norm(ref)


_TEAM_PROBE = r"""
import sys, hashlib, numpy as np
sys.path.insert(0, sys.argv[1])
from fitsnap_amd import _capi
K = 1100
rng = np.random.default_rng(11)
U, _ = np.linalg.qr(rng.standard_normal((K + 200, K)))
V, _ = np.linalg.qr(rng.standard_normal((K, K)))
R1 = np.linalg.qr((U * np.logspace(0, -8, K)) @ V.T, mode="r")
R1 = R1 * np.sign(np.diag(R1))[:, None]
R2 = np.eye(K) + 1e-2 * np.triu(rng.standard_normal((K, K))) / K
z = rng.standard_normal(K)
beta, rank, info = _capi.rowspace_chain([R1, R2], z, 1e-13)
print(rank, info["chain"], repr(info["cond_bound"]), hashlib.sha256(beta.tobytes()).hexdigest())
"""


def test_cooperative_substitutions_do_not_depend_on_the_team(tmp_path):
    # from 1 024 unknowns on a triangular substitution runs on a team (FSNAP_TRI_TEAM threads, blocks of 64 unknowns, dot products
    # on eight partial sums in a fixed order): the certified chain's solution and its condition bound carry the same bits with
    # one, two and five threads per substitution and with two or seven host threads around them
    import subprocess
    import sys

    script = tmp_path / "probe.py"
    script.write_text(_TEAM_PROBE)
    outs = []
    for team, nt in (("1", "2"), ("2", "7"), ("5", "7")):
        env = dict(os.environ, FSNAP_TRI_TEAM=team, FSNAP_HOST_THREADS=nt)
        outs.append(subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=600, check=True).stdout)
    assert outs[0] == outs[1] == outs[2]
    assert outs[0].startswith("1100 1.0")
