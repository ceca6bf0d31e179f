"""GPU (-m gpu): the row-space least-squares path (fsnap_lstsq_rows: CholeskyQR passes on the GPU + dgelsd's K x K
end) against the oracle's lstsq(aw, bw, 1e-13) (fitsnap3lib/solvers/svd.py:44-54) in the regime the normal equations
cannot reach: kappa(A_w) = 1e8 ... 1e12, duplicated / near-collinear columns, exact rank deficiency.

Tolerances.  north_star asks for 1e-6 relative; two backward-stable least-squares solvers agree to ~kappa eps (that is
the accuracy of lstsq itself), so the coefficient bar is max(1e-6, 50 kappa eps) norm-wise, tightened by three checks
that do not depend on kappa: (i) the residual norm equals lstsq's to 1e-6 relative and is never larger by more than
1e-9, (ii) against an extended-precision solution our error is no worse than 10 x lstsq's own, (iii) rank decisions
(which singular directions are dropped at 1e-13 sigma_max) are identical."""
import numpy as np
import pytest
import scipy.linalg as sl

from fitsnap_amd import _capi
from fitsnap_amd.config import Config
from fitsnap_amd.parallel_tools import ParallelTools
from fitsnap_amd.solvers import solver_factory
from oracle import fitsnap_oracle as orc

from conftest import maxrel

pytestmark = pytest.mark.gpu
EPS = np.finfo(float).eps


def make_svd():
    pt = ParallelTools()
    cfg = Config(pt, {"SOLVER": {"solver": "SVD"}})
    return pt, solver_factory.solver("SVD", pt, cfg)


def conditioned(m, K, kappa, mode, seed):
    r = np.random.default_rng(seed)
    U, _ = np.linalg.qr(r.standard_normal((m, K)))
    V, _ = np.linalg.qr(r.standard_normal((K, K)))
    if mode == "geometric":
        s = np.logspace(0, -np.log10(kappa), K)
    else:
        s = np.ones(K)
        s[-1] = 1.0 / kappa
    return (U * s) @ V.T


def extended_precision_solution(A, b, start):
    """Least-squares solution refined with residuals in long double (test infrastructure: the yardstick for 'who is
    closer to the truth' when the two fp64 answers differ by kappa eps)."""
    Al, bl = A.astype(np.longdouble), b.astype(np.longdouble)
    Q, R = np.linalg.qr(A)
    x = start.astype(np.longdouble)
    for _ in range(6):
        g = (Al.T @ (bl - Al @ x)).astype(np.float64)
        x = x + sl.solve_triangular(R, sl.solve_triangular(R, g, trans="T"))
    return x.astype(np.float64)


@pytest.mark.parametrize("mode", ["geometric", "one"])
@pytest.mark.parametrize("kappa", [1e8, 1e10, 1e12])
def test_svd_solver_ill_conditioned_matches_lstsq(kappa, mode):
    m, K = 20000, 64
    A = conditioned(m, K, kappa, mode, int(np.log10(kappa)))
    r = np.random.default_rng(11)
    b = A @ r.standard_normal(K) + 1e-3 * r.standard_normal(m)
    w = r.uniform(0.5, 2.0, m)
    ref = orc.svd_fit(A, b, w)
    pt, s = make_svd()
    s.perform_fit(A, b, w, trainall=True)
    info = s.last_row_space
    assert info is not None, "the row-space path must have been taken"
    assert info["converged"] == 1.0 and 2 <= info["passes"] <= 4 and info["deviation"] <= 1e-10
    assert s.last_rank == K
    x = s.fit
    aw, bw = w[:, None] * A, w * b
    kw = np.linalg.cond(aw)
    tol = max(1e-6, 50 * kw * EPS)
    assert np.linalg.norm(x - ref) <= tol * np.linalg.norm(ref)
    # with a non-zero residual r the minimiser's predictions move by ~kappa eps |r| under rounding, i.e. the residual
    # norm itself by ~(kappa eps)^2 |r| / 2 -- for lstsq just as for us
    res, res_ref = np.linalg.norm(aw @ x - bw), np.linalg.norm(aw @ ref - bw)
    assert res <= res_ref * (1 + max(1e-9, 50 * (kw * EPS) ** 2)) and abs(res - res_ref) <= 1e-5 * res_ref
    truth = extended_precision_solution(aw, bw, ref)
    err, err_ref = np.linalg.norm(x - truth), np.linalg.norm(ref - truth)
    assert err <= 10 * max(err_ref, kw * EPS * np.linalg.norm(truth))
    if kappa <= 1e8:
        assert np.linalg.norm(x - ref) <= 1e-6 * np.linalg.norm(ref)
    pt.free()


def test_duplicated_and_dependent_columns_minimum_norm():
    r = np.random.default_rng(5)
    m = 9001                                                    # not a multiple of 64: ragged last workgroup
    base = r.standard_normal((m, 40))
    A = np.hstack([base, base[:, :3], base[:, 3:5] @ r.standard_normal((2, 2)), np.zeros((m, 2))])
    b = r.standard_normal(m)
    w = r.choice([100.0, 1.0, 1e-3], size=m)
    _, _, rank_ref, _ = sl.lstsq(w[:, None] * A, w * b, 1.0e-13)
    ref = orc.svd_fit(A, b, w)
    pt, s = make_svd()
    s.perform_fit(A, b, w, trainall=True)
    assert s.last_row_space is not None and s.last_row_space["svd"] == 1.0
    assert s.last_rank == rank_ref == 40
    assert np.linalg.norm(s.fit - ref) <= 1e-8 * np.linalg.norm(ref)
    assert np.all(s.fit[-2:] == 0.0)
    assert np.allclose(s.fit[:3], s.fit[40:43], rtol=1e-8)      # a duplicated column shares its coefficient
    pt.free()


@pytest.mark.parametrize("eps_col,kept", [(1e-9, True), (1e-15, False)])
def test_near_collinear_column_kept_or_dropped_like_gelsd(eps_col, kept):
    r = np.random.default_rng(6)
    m, K = 6000, 24
    base = r.standard_normal((m, K))
    A = np.hstack([base, base[:, :1] + eps_col * r.standard_normal((m, 1))])
    b = r.standard_normal(m)
    w = np.ones(m)
    _, _, rank_ref, _ = sl.lstsq(A, b, 1.0e-13)
    ref = orc.svd_fit(A, b, w)
    pt, s = make_svd()
    s.perform_fit(A, b, w, trainall=True)
    assert s.last_row_space is not None
    assert s.last_rank == rank_ref == (K + 1 if kept else K)
    if kept:
        assert np.linalg.norm(A @ (s.fit - ref)) <= 1e-7 * np.linalg.norm(b)
        assert np.linalg.norm(s.fit - ref) <= 50 * 1e9 * EPS * np.linalg.norm(ref)
    else:
        assert np.linalg.norm(s.fit - ref) <= 1e-8 * np.linalg.norm(ref)
    pt.free()


@pytest.mark.parametrize("K,m", [(31, 5003), (110, 12000), (128, 8192), (200, 6001), (300, 5000), (64, 50)])
def test_row_space_all_kernel_families_and_ragged_shapes(K, m):
    # K = 31: kernel 1P on Q; 110: kernel 1A <7> with a partial last block; 128: 1A <8>; 200: tiled kernel; 300: beyond
    # the LDS-resident variant of the orthogonalisation kernel (K > 256: solved blocks re-read from global memory);
    # m = 50 < K: rank deficient by shape (the SVD end picks the minimum-norm solution)
    kappa = 1e9
    r = np.random.default_rng(K)
    if m > K:
        A = conditioned(m, K, kappa, "one", K + 1)
    else:
        A = r.standard_normal((m, K))
    A = A * (10.0 ** r.uniform(-2, 2, size=K))                   # graded columns on top
    b = r.standard_normal(m)
    w = r.uniform(0.5, 2.0, m)
    t = r.random(m) < 0.15
    A[t] = np.nan                                               # masked rows may hold garbage
    aw, bw = orc.weight_rows(np.nan_to_num(A), b, w, t)
    _, _, rank_ref, sv = sl.lstsq(aw, bw, 1.0e-13)
    ref = orc.svd_fit(np.nan_to_num(A), b, w, t)
    pt, s = make_svd()
    s.perform_fit(A, b, w[~t], fs_dict={"Testing": t.tolist()})
    assert s.last_row_space is not None and s.last_rank == rank_ref
    kw = sv[0] / sv[rank_ref - 1]
    tol = max(1e-6, 50 * kw * EPS)
    assert np.linalg.norm(s.fit - ref) <= tol * np.linalg.norm(ref)
    res, res_ref = np.linalg.norm(aw @ s.fit - bw), np.linalg.norm(aw @ ref - bw)
    assert res <= res_ref * (1 + 1e-9) + 1e-9 * np.linalg.norm(bw)
    pt.free()


def test_lstsq_rows_on_a_well_conditioned_system_equals_the_reference_fit(ta, ta_fits):
    # the golden Ta matrices never need the row-space path; called directly it must still give the reference's answer
    A, b, w = ta
    ctx = _capi.HipContext(0)
    ctx.upload_rows(A, b)
    ctx.set_weights(w)
    beta, rank, info = ctx.lstsq_rows(1.0e-13)
    assert rank == A.shape[1] and info["converged"] == 1.0 and info["svd"] == 0.0
    assert maxrel(beta, ta_fits["svd_all"]) < 1e-9
    t = ta_fits["testing_mask"]
    ctx.set_weights(w, (~t).astype(np.uint8))
    beta, rank, info = ctx.lstsq_rows(1.0e-13)
    assert maxrel(beta, ta_fits["svd_mask"]) < 1e-9
    # the fit loop is not disturbed: statistics after a row-space solve are those of the rows
    G, c, s3 = ctx.normal_eq()
    Gr, cr, sr = orc.normal_eq(A, b, w, t)
    d = np.sqrt(np.diag(Gr))
    assert np.max(np.abs(G - Gr) / (d[:, None] * d[None, :])) < 1e-12 and s3[2] == sr[2]
    ctx.close()


def test_golden_fit_does_not_take_the_row_space_path(ta, ta_fits):
    A, b, w = ta
    pt, s = make_svd()
    s.perform_fit(A, b, w, trainall=True)
    assert s.last_row_space is None and maxrel(s.fit, ta_fits["svd_all"]) < 1e-10
    pt.free()


def test_row_space_through_a_one_rank_communicator():
    # the collective form (statistics of every pass all-reduced, refinement right-hand side all-reduced) in a
    # communicator of one rank: same answer as without a communicator
    A = conditioned(12000, 48, 1e10, "geometric", 2)
    r = np.random.default_rng(3)
    b = A @ r.standard_normal(48) + 1e-3 * r.standard_normal(12000)
    ctx = _capi.HipContext(0)
    ctx.upload_rows(A, b)
    ctx.set_weights(np.ones(len(b)))
    plain, rank0, _ = ctx.lstsq_rows(1.0e-13)
    ctx.comm_init(1, 0, _capi.comm_id())
    coll, rank1, info = ctx.lstsq_rows(1.0e-13)
    assert rank0 == rank1 == 48 and info["converged"] == 1.0
    assert np.array_equal(plain, coll)
    ctx.close()


@pytest.mark.parametrize("m", [9000, 40001, 70003])
def test_lstsq_rows_general_k_kernel_every_row_tile_height(m):
    # kernel 13 (K > 128) runs 16-, 32- or 64-row tiles per wave depending on the number of rows (short matrices would
    # leave most SIMDs idle with 64-row tiles); K = 160 also takes the factor chain of the host end (K <= 256: product)
    K = 160
    r = np.random.default_rng(m)
    A = conditioned(m, K, 1e6, "geometric", 3) * (10.0 ** r.uniform(-1, 1, size=K))
    b = A @ r.standard_normal(K) + 1e-3 * r.standard_normal(m)
    w = r.uniform(0.5, 2.0, m)
    ctx = _capi.HipContext(0)
    ctx.upload_rows(A, b)
    ctx.set_weights(w)
    beta, rank, info = ctx.lstsq_rows(1.0e-13)
    ref = orc.svd_fit(A, b, w)
    assert rank == K and info["converged"] == 1.0
    kw = np.linalg.cond(w[:, None] * A)
    assert np.linalg.norm(beta - ref) <= max(1e-6, 50 * kw * EPS) * np.linalg.norm(ref)
    ctx.close()


def test_lstsq_rows_factor_chain_path_on_the_gpu():
    # K = 300 > 256: the factors of the passes stay apart on the host (FactorChain); ill-conditioned enough for two passes
    m, K = 20000, 300
    A = conditioned(m, K, 1e7, "geometric", 8)
    r = np.random.default_rng(1)
    b = A @ r.standard_normal(K) + 1e-3 * r.standard_normal(m)
    ctx = _capi.HipContext(0)
    ctx.upload_rows(A, b)
    ctx.set_weights(np.ones(m))
    beta, rank, info = ctx.lstsq_rows(1.0e-13)
    ref = orc.svd_fit(A, b, np.ones(m))
    assert rank == K and info["converged"] == 1.0 and info["svd"] == 0.0 and info["passes"] >= 2
    assert np.linalg.norm(beta - ref) <= max(1e-6, 50 * 1e7 * EPS) * np.linalg.norm(ref)
    ctx.close()


@pytest.mark.parametrize("case", ["duplicated", "fewer_rows_than_columns"])
def test_large_k_truncated_solve_runs_in_lapack(case):
    # K > 256 and a truncation is needed.  Two dependent columns: the two dropped directions are projected away around a back
    # substitution (FactorSolver::deflate, info["svd"] == 3), no SVD at all.  120 dropped directions: the K x K end goes
    # through the dense-pinv hook the Python layer installs (scipy's gesdd) instead of the library's Jacobi SVD (~20 s at
    # K = 1595); info["svd"] == 2 says so
    r = np.random.default_rng(17)
    K = 320
    if case == "duplicated":
        m = 2500
        A = r.standard_normal((m, K))
        A[:, 300] = A[:, 5]
        A[:, 310] = A[:, 6] - 2.0 * A[:, 7]
    else:
        m = 200
        A = r.standard_normal((m, K))
    b = r.standard_normal(m)
    w = r.uniform(0.5, 2.0, m)
    _, _, rank_ref, _ = sl.lstsq(w[:, None] * A, w * b, 1.0e-13)
    ref = orc.svd_fit(A, b, w)
    pt, s = make_svd()
    s.perform_fit(A, b, w, trainall=True)
    assert s.last_row_space is not None and s.last_row_space["svd"] == (3.0 if case == "duplicated" else 2.0)
    assert s.last_rank == rank_ref == (K - 2 if case == "duplicated" else m)
    assert np.linalg.norm(s.fit - ref) <= 1e-7 * np.linalg.norm(ref)
    pt.free()
    # without the hook the library's own SVD gives the same answer
    ctx = _capi.HipContext(0)
    ctx._lib.fsnap_set_dense_pinv(ctx._h, None, None)
    ctx.upload_rows(A, b)
    ctx.set_weights(w)
    beta, rank, info = ctx.lstsq_rows(1.0e-13)
    assert info["svd"] == (3.0 if case == "duplicated" else 1.0) and rank == rank_ref
    assert np.linalg.norm(beta - ref) <= 1e-7 * np.linalg.norm(ref)
    ctx.close()


@pytest.mark.parametrize("K,m,masked", [(128, 20011, False), (96, 9001, True), (200, 12000, False)])
def test_first_pass_from_the_statistics_of_the_fit_that_just_ran(K, m, masked):
    # option "rowspace_reuse_stats" (one-shot; Solver._row_space_fit sets it): the first CholeskyQR pass starts from the Gram
    # matrix that fsnap_fit_resident computed a moment ago on the same rows, weights and mask (still in the page-locked mirror)
    # instead of computing it again -- bit-identical coefficients, and the option is used up by ONE call
    r = np.random.default_rng(700 + K)
    A = r.standard_normal((m, K))
    A[:, K - 1] = A[:, 0] * 1.5 + 1e-9 * A[:, K - 1]                   # kappa ~ 1e9: what sends the SVD solver to the rows
    b = r.standard_normal(m)
    w = r.uniform(0.5, 2.0, m)
    mask = (r.random(m) > 0.1).astype(np.uint8) if masked else None
    ctx = _capi.HipContext(0)
    ctx.upload_rows(A, b)
    ctx.set_weights(w, mask)
    plain, rank0, info0 = ctx.lstsq_rows(1.0e-13)
    ctx.fit_resident(_capi.SOLVE_LSTSQ_PROBE, 1.0e-13)
    ctx.set_option("rowspace_reuse_stats", 1)
    kept, rank1, info1 = ctx.lstsq_rows(1.0e-13)
    assert rank1 == rank0 and np.array_equal(kept, plain) and info1["passes"] == info0["passes"]
    again, rank2, _ = ctx.lstsq_rows(1.0e-13)                          # the option is gone: statistics computed afresh, same answer
    assert rank2 == rank0 and np.array_equal(again, plain)
    # new weights after the fit: a caller that still sets the option gets the statistics of the OLD weights only if it lies
    # about them -- the host layer sets the option right behind its own fit, never across set_weights; here the honest order
    ctx.set_weights(2.0 * w, mask)
    ctx.fit_resident(_capi.SOLVE_LSTSQ_PROBE, 1.0e-13)
    ctx.set_option("rowspace_reuse_stats", 1)
    scaled, rank3, _ = ctx.lstsq_rows(1.0e-13)
    assert rank3 == rank0 and np.abs(scaled - plain).max() <= 1e-9 * np.abs(plain).max()      # lstsq is invariant under a uniform weight scale
    ctx.close()
