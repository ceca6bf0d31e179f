"""CPU: pin the oracle (oracle/fitsnap_oracle.py) against outputs of the reference
itself (tests/golden/ta_reference_fits.npz, produced by tests/golden/make_golden.py which
imports /root/reference) and against the reference's committed Ta_pot.snapcoeff /
Ta_metrics.md."""
import warnings

import numpy as np
import pytest

from oracle import fitsnap_oracle as orc

from conftest import maxrel


def test_svd_matches_reference_class_bitwise(ta, ta_fits):
    A, b, w = ta
    assert np.array_equal(orc.svd_fit(A, b, w), ta_fits["svd_all"])


def test_svd_matches_committed_snapcoeff(ta, ta_fits):
    # the reference's own acceptance bar: max(test - standard) < 1e-6 absolute
    # (tests/example_checker.py:62)
    A, b, w = ta
    assert np.max(np.abs(orc.svd_fit(A, b, w) - ta_fits["snapcoeff"])) < 1e-12


def test_svd_mask_matches_reference(ta, ta_fits):
    A, b, w = ta
    t = ta_fits["testing_mask"]
    assert np.array_equal(orc.svd_fit(A, b, w, testing=t), ta_fits["svd_mask"])
    # explicit-array path (w pre-masked) and shared-array path agree in the reference
    assert np.array_equal(ta_fits["svd_mask"], ta_fits["svd_mask_shared"])


def test_svd_transpose_matches_reference(ta, ta_fits):
    A, b, w = ta
    assert maxrel(orc.svd_fit(A, b, w, apply_transpose=True), ta_fits["svd_transpose_all"]) < 1e-9


@pytest.mark.parametrize("alpha,tag", [(1e-8, "1e-8"), (1e-4, "1e-4")])
def test_ridge_matches_reference(ta, ta_fits, alpha, tag):
    A, b, w = ta
    t = ta_fits["testing_mask"]
    assert maxrel(orc.ridge_fit(A, b, w, alpha), ta_fits[f"ridge_sklearn_{tag}_all"]) < 1e-12
    assert maxrel(orc.ridge_fit(A, b, w, alpha, local_solver=True), ta_fits[f"ridge_local_{tag}_all"]) < 1e-12
    assert maxrel(orc.ridge_fit(A, b, w, alpha, testing=t), ta_fits[f"ridge_sklearn_{tag}_mask"]) < 1e-12
    assert maxrel(orc.ridge_fit(A, b, w, alpha, local_solver=True, testing=t), ta_fits[f"ridge_local_{tag}_mask"]) < 1e-12


def test_normal_equations_reproduce_fit(ta, ta_fits):
    # (aw.T aw) beta = aw.T bw solved in the oracle reproduces the lstsq fit (7e-8)
    A, b, w = ta
    G, c, s = orc.normal_eq(A, b, w)
    d = 1 / np.sqrt(np.diag(G))
    beta = d * np.linalg.solve(G * d[:, None] * d[None, :], c * d)
    assert maxrel(beta, ta_fits["svd_all"]) < 1e-6
    assert s[2] == len(b)


def test_metrics_all_rows(ta, ta_fits):
    # '*ALL' rows of examples/Ta_Linear_JCP2014/20May21_Standard/Ta_metrics.md; golden row
    # blocks: [0:363] Energy, [363:13035] Force, [13035:15213] Stress
    A, b, w = ta
    preds = orc.predict(A, ta_fits["svd_all"])
    blocks = [slice(0, 363), slice(363, 13035), slice(13035, 15213)]
    for wi, wt in enumerate(("Unweighted", "Weighted")):
        for ri, sl in enumerate(blocks):
            n, mae, rmse, rsq = ta_fits["metrics_all"][wi * 3 + ri]
            row = orc.error_row(b[sl], preds[sl], w[sl])
            pre = "" if wt == "Unweighted" else "w_"
            assert row[pre + "ncount"] == n
            assert row[pre + "mae"] == pytest.approx(mae, rel=6e-6)
            assert row[pre + "rmse"] == pytest.approx(rmse, rel=6e-6)
            assert row[pre + "rsq"] == pytest.approx(rsq, abs=6e-6)


def test_second_golden_set_xyz(ta):
    import os
    from conftest import GOLDEN
    A, b, w = ta
    d = np.load(os.path.join(GOLDEN, "ta_xyz_delta.npz"))
    A2, b2, w2 = A + d["dA"], b + d["db"], w + d["dw"]
    assert np.array_equal(orc.svd_fit(A2, b2, w2), d["svd_all"])


def test_ard_captured_vector(ta, ta_fits):
    # the direct scikit-learn call with the reference's recipe (captured before the class itself could be run)
    A, b, w = ta
    fit = orc.ard_fit(A, b, w, scap=1e-3, scai=1e-3, logcut=0.3)
    assert np.array_equal(fit, ta_fits["ard_all"])
    assert np.array_equal(ta_fits["ard_all"], ta_fits["ard_class_all"])


@pytest.mark.parametrize("key,kw", [
    ("ard_class_all", {}),
    ("ard_class_mask", {"mask": True}),
    ("ard_class_direct", {"directmethod": True}),
    ("ard_class_scaled", {"mask": True, "scap": 1.0e-2, "scai": 1.0e-4, "logcut": 1.0}),
    ("ard_class_transpose", {"apply_transpose": True}),
])
def test_ard_matches_the_reference_class(ta, ta_fits, key, kw):
    # vectors from the reference's ARD class itself (ard.py:15-49), run by make_golden.py with ARDRegression's renamed
    # iteration keyword forwarded; same scikit-learn here, so the restatement must reproduce them bit for bit
    A, b, w = ta
    kw = dict(kw)
    testing = ta_fits["testing_mask"] if kw.pop("mask", False) else None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                             # the transposed Ta problem never converges
        fit = orc.ard_fit(A, b, w, testing, **kw)
    assert np.array_equal(fit, ta_fits[key])
    assert 20 <= np.count_nonzero(fit) <= 31


LASSO_CASES = [
    ("lasso_class_all", {}),
    ("lasso_class_mask", {"mask": True}),
    ("lasso_class_alpha1e-2_mask", {"mask": True, "alpha": 1.0e-2}),
    ("lasso_class_alpha1_all", {"alpha": 1.0}),
    ("lasso_class_alpha1_iter50_all", {"alpha": 1.0, "max_iter": 50}),
    ("lasso_class_transpose", {"alpha": 1.0e-2, "apply_transpose": True}),
]


@pytest.mark.parametrize("key,kw", LASSO_CASES)
def test_lasso_matches_the_reference_class(ta, ta_fits, key, kw):
    # vectors from the reference's LASSO class (lasso.py:15-29); same scikit-learn here: bit for bit
    A, b, w = ta
    kw = dict(kw)
    testing = ta_fits["testing_mask"] if kw.pop("mask", False) else None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                             # ConvergenceWarning at max_iter
        fit = orc.lasso_fit(A, b, w, testing, **kw)
    assert np.array_equal(fit, ta_fits[key])


def test_synthetic_generator_is_deterministic():
    A1, b1, w1 = orc.synth_problem(1000, 16)
    A2, b2, w2 = orc.synth_problem(1000, 16)
    assert np.array_equal(A1, A2) and np.array_equal(b1, b2) and np.array_equal(w1, w2)
    assert set(np.unique(w1)) <= {100.0, 1.0, 1e-8}


def test_anl_matches_reference_class(ta, ta_fits):
    A, b, w = ta
    fit, cov = orc.anl_fit(A, b, w, cov_nugget=1.0e-10)
    assert np.array_equal(fit, ta_fits["anl_fit"]) and np.array_equal(cov, ta_fits["anl_cov"])


def test_anl_transpose_matches_reference_class(ta, ta_fits):
    # EXTRAS.apply_transpose (anl.py:31-36) through the reference class itself
    A, b, w = ta
    fit, cov = orc.anl_fit(A, b, w, cov_nugget=1.0e-10, apply_transpose=True)
    assert np.array_equal(fit, ta_fits["anl_transpose_fit"]) and np.array_equal(cov, ta_fits["anl_transpose_cov"])
