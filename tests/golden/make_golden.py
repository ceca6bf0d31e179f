"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference; the GPU box has neither the
reference nor this need):

    python tests/golden/make_golden.py

What it does
  * imports fitsnap3lib from /root/reference with a stub ``lammps`` module
    (fitsnap3lib/parallel_tools.py:35 imports it unconditionally),
  * instantiates the reference's own SVD / RIDGE (/ ANL / ARD / LASSO) solver classes through its
    solver_factory and calls ``perform_fit(a, b, w, fs_dict|trainall)`` on the golden
    Ta matrices the reference commits under
    examples/Ta_Linear_JCP2014/20May21_Standard/{Descriptors,Truth-Ref,Weights}.npy,
  * stores inputs (A, b, w as data) and the reference's outputs (coefficient vectors),
    plus the committed Ta_pot.snapcoeff coefficients and the '*ALL' rows of
    Ta_metrics.md, as .npz fixtures.

Only data is written: no reference source text is copied.
"""
from __future__ import annotations

import os
import re
import sys
import types
import warnings

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
TA = os.path.join(REF, "examples/Ta_Linear_JCP2014/20May21_Standard")
TAXYZ = os.path.join(REF, "examples/Ta_XYZ/19Nov19_Standard")


def import_reference():
    if "lammps" not in sys.modules:
        stub = types.ModuleType("lammps")

        class lammps:  # noqa: N801 - mirrors the real class name
            def __init__(self, *a, **k):
                raise RuntimeError("stub lammps: not available")

        stub.lammps = lammps
        sys.modules["lammps"] = stub
    sys.path.insert(0, REF)
    from fitsnap3lib.parallel_tools import ParallelTools
    from fitsnap3lib.io.input import Config
    from fitsnap3lib.solvers import solver_factory
    return ParallelTools, Config, solver_factory


def settings(solver, extra=None):
    s = {
        "BISPECTRUM": {"numTypes": 1, "twojmax": 6, "rcutfac": 4.67637, "rfac0": 0.99363, "rmin0": 0.0,
                       "wj": 1.0, "radelem": 0.5, "type": "Ta", "wselfallflag": 0, "chemflag": 0,
                       "bzeroflag": 0, "quadraticflag": 0},
        "CALCULATOR": {"calculator": "LAMMPSSNAP", "energy": 1, "force": 1, "stress": 1},
        "SOLVER": {"solver": solver, "compute_testerrs": 1, "detailed_errors": 1},
        "OUTFILE": {"metrics": "Ta_metrics.md", "potential": "Ta_pot"},
        "REFERENCE": {"units": "metal", "atom_style": "atomic", "pair_style": "zero 10.0", "pair_coeff": "* *"},
    }
    if extra:
        for k, v in extra.items():
            s.setdefault(k, {}).update(v)
    return s


def parse_snapcoeff(path):
    with open(path) as f:
        lines = f.readlines()
    n = int(lines[2].split()[-1])
    return np.array([float(lines[4 + i].split()[0]) for i in range(n)])


def parse_metrics_all(path):
    rows = {}
    with open(path) as f:
        for line in f:
            m = re.match(r"\|\s*\('\*ALL', '(\w+)', '(\w+)'\)\s*\|\s*(\d+)\s*\|\s*([-\d.e+]+)\s*\|\s*([-\d.e+]+)\s*\|\s*([-\d.e+]+)", line)
            if m:
                rows[(m.group(1), m.group(2))] = (int(m.group(3)), float(m.group(4)), float(m.group(5)), float(m.group(6)))
    return rows


def main():
    ParallelTools, Config, solver_factory = import_reference()
    A = np.load(os.path.join(TA, "Descriptors.npy"))
    b = np.load(os.path.join(TA, "Truth-Ref.npy"))
    w = np.load(os.path.join(TA, "Weights.npy"))
    m, K = A.shape
    testing = (np.random.default_rng(12345).random(m) < 0.1)
    fs_dict = {"Testing": testing.tolist()}

    def run(solver, extra=None, use_mask=False):
        pt = ParallelTools()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            cfg = Config(pt, settings(solver, extra), arguments_lst=["--overwrite"])
        s = solver_factory.solver(solver, pt, cfg)
        if use_mask == "shared":
            # the no-argument path: pt.shared_arrays + pt.fitsnap_dict['Testing'] (svd.py:40-44)
            pt.create_shared_array('a', m, K)
            pt.create_shared_array('b', m)
            pt.create_shared_array('w', m)
            pt.shared_arrays['a'].array[:] = A
            pt.shared_arrays['b'].array[:] = b
            pt.shared_arrays['w'].array[:] = w
            pt.fitsnap_dict['Testing'] = testing.tolist()
            s.perform_fit()
        elif use_mask:
            # explicit-array path: the reference multiplies the UNMASKED w into a[training]
            # (svd.py:46), so w must already be restricted to the training rows
            s.perform_fit(A, b, w[~testing], fs_dict=fs_dict)
        else:
            s.perform_fit(A, b, w, trainall=True)
        return np.asarray(s.fit, dtype=np.float64).copy()

    out = {}
    out["svd_all"] = run("SVD")
    out["svd_mask"] = run("SVD", use_mask=True)
    out["svd_mask_shared"] = run("SVD", use_mask="shared")
    out["svd_transpose_all"] = run("SVD", {"EXTRAS": {"apply_transpose": 1}})
    for tag, alpha in (("1e-8", 1.0e-8), ("1e-4", 1.0e-4)):
        out[f"ridge_sklearn_{tag}_all"] = run("RIDGE", {"RIDGE": {"alpha": alpha, "local_solver": 0}})
        out[f"ridge_local_{tag}_all"] = run("RIDGE", {"RIDGE": {"alpha": alpha, "local_solver": 1}})
        out[f"ridge_sklearn_{tag}_mask"] = run("RIDGE", {"RIDGE": {"alpha": alpha, "local_solver": 0}}, use_mask=True)
        out[f"ridge_local_{tag}_mask"] = run("RIDGE", {"RIDGE": {"alpha": alpha, "local_solver": 1}}, use_mask=True)
    out["ridge_sklearn_1e-8_transpose_all"] = run("RIDGE", {"RIDGE": {"alpha": 1.0e-8, "local_solver": 0},
                                                             "EXTRAS": {"apply_transpose": 1}})
    # ANL (anl.py): posterior mean + covariance; the class writes covariance.npy / mean.npy into the cwd
    import tempfile
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            pt = ParallelTools()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                cfg = Config(pt, settings("ANL", {"SOLVER": {"nsam": 0, "cov_nugget": 1.0e-10}}), arguments_lst=["--overwrite"])
            sanl = solver_factory.solver("ANL", pt, cfg)
            sanl.perform_fit(A, b, w, trainall=True)
            out["anl_fit"] = np.asarray(sanl.fit).copy()
            out["anl_cov"] = np.asarray(sanl.cov).copy()
            # the same class with EXTRAS.apply_transpose (anl.py:31-36): the regression runs on (aw.T aw, aw.T bw)
            pt = ParallelTools()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                cfg = Config(pt, settings("ANL", {"SOLVER": {"nsam": 0, "cov_nugget": 1.0e-10}, "EXTRAS": {"apply_transpose": 1}}),
                             arguments_lst=["--overwrite"])
            sanl = solver_factory.solver("ANL", pt, cfg)
            sanl.perform_fit(A, b, w, trainall=True)
            out["anl_transpose_fit"] = np.asarray(sanl.fit).copy()
            out["anl_transpose_cov"] = np.asarray(sanl.cov).copy()
        finally:
            os.chdir(cwd)
    # error_analysis (solver.py:137-435) with synthetic group labels: per (group, weighting, train/test,
    # row type) ncount / mae / rmse / rsq — the rows the committed Ta_metrics.md cannot pin offline
    row_type = np.array(["Energy"] * 363 + ["Force"] * 12672 + ["Stress"] * 2178)
    groups = np.array(["g%d" % g for g in np.random.default_rng(7).integers(0, 5, m)])
    pt = ParallelTools()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cfg = Config(pt, settings("SVD"), arguments_lst=["--overwrite"])
    sea = solver_factory.solver("SVD", pt, cfg)
    fsd = {"Groups": groups.tolist(), "Testing": testing.tolist(), "Row_Type": row_type.tolist()}
    sea.perform_fit(A, b, w[~testing], fs_dict=fsd)
    sea.error_analysis(A, b, w, fsd)
    err = sea.errors
    out["ea_index"] = np.array(["|".join(str(x) for x in ix) for ix in err.index])
    out["ea_values"] = err[["ncount", "mae", "rmse", "rsq"]].to_numpy(dtype=np.float64)
    out["ea_groups"] = groups
    out["snapcoeff"] = parse_snapcoeff(os.path.join(TA, "Ta_pot.snapcoeff"))
    met = parse_metrics_all(os.path.join(TA, "Ta_metrics.md"))
    # order: (Unweighted|Weighted) x (Energy|Force|Stress) -> ncount, mae, rmse, rsq
    out["metrics_all"] = np.array([met[(wt, rt)] for wt in ("Unweighted", "Weighted")
                                   for rt in ("Energy", "Force", "Stress")], dtype=np.float64)
    out["testing_mask"] = testing
    # ARD: the reference class cannot run on sklearn >= 1.5 (n_iter kwarg); capture the
    # direct scikit-learn call with the reference's hyper-parameter recipe (ard.py:26-43)
    try:
        from sklearn.linear_model import ARDRegression
        aw, bw = w[:, None] * A, w * b
        ap = 1.0 / np.var(bw)
        scap, scai, logcut = 1.0e-3, 1.0e-3, 0.3   # io/sections/solver_sections/ard.py:19-21 defaults
        reg = ARDRegression(max_iter=1000, alpha_1=scap * ap, alpha_2=scap * ap, lambda_1=ap * scai,
                            lambda_2=ap * scai, fit_intercept=False,
                            threshold_lambda=10 ** (int(np.abs(np.log10(ap))) + logcut))
        reg.fit(aw, bw)
        out["ard_all"] = reg.coef_.copy()
    except Exception as e:  # pragma: no cover
        print("ARD capture failed:", e)
    # ARD through the REFERENCE CLASS (ard.py:15-49).  The class spells ARDRegression's iteration cap `n_iter`, a
    # keyword scikit-learn renamed to `max_iter` in 1.3 and dropped in 1.5; the constructor the reference module sees
    # is therefore the installed one with that single keyword forwarded under its new name.  Everything else -- the
    # training mask, the row weighting, apply_transpose, the inverse-variance recipe for alpha/lambda/threshold_lambda,
    # the directmethod branch -- is the reference's own code running.
    try:
        import fitsnap3lib.solvers.ard as ref_ard
        from sklearn.linear_model import ARDRegression as _ARDRegression

        def ard_with_renamed_keyword(n_iter=300, **kw):
            return _ARDRegression(max_iter=n_iter, **kw)

        ref_ard.ARDRegression = ard_with_renamed_keyword

        def run_ard(extra, mask):
            pt = ParallelTools()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                # the reference only builds config.sections['ARD'] when the input has an [ARD] section
                ext = {"ARD": {"directmethod": 0}}
                for k, v in (extra or {}).items():
                    ext.setdefault(k, {}).update(v)
                cfg = Config(pt, settings("ARD", ext), arguments_lst=["--overwrite"])
            s = solver_factory.solver("ARD", pt, cfg)
            pt.create_shared_array('a', m, K)
            pt.create_shared_array('b', m)
            pt.create_shared_array('w', m)
            pt.shared_arrays['a'].array[:] = A
            pt.shared_arrays['b'].array[:] = b
            pt.shared_arrays['w'].array[:] = w
            pt.fitsnap_dict['Testing'] = testing.tolist() if mask else [False] * m
            s.perform_fit()
            return np.asarray(s.fit, dtype=np.float64).copy()

        out["ard_class_all"] = run_ard(None, False)
        out["ard_class_mask"] = run_ard(None, True)
        out["ard_class_direct"] = run_ard({"ARD": {"directmethod": 1}}, False)
        out["ard_class_scaled"] = run_ard({"ARD": {"scap": 1.0e-2, "scai": 1.0e-4, "logcut": 1.0}}, True)
        out["ard_class_transpose"] = run_ard({"EXTRAS": {"apply_transpose": 1}}, False)
    except Exception as e:  # pragma: no cover
        print("ARD reference-class run failed:", repr(e))
    # LASSO through the reference class (lasso.py:15-29; runs on the installed scikit-learn as it stands)
    def run_lasso(extra, mask):
        pt = ParallelTools()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ext = {"LASSO": {}}
            for k, v in (extra or {}).items():
                ext.setdefault(k, {}).update(v)
            cfg = Config(pt, settings("LASSO", ext), arguments_lst=["--overwrite"])
            s = solver_factory.solver("LASSO", pt, cfg)
            pt.create_shared_array('a', m, K)
            pt.create_shared_array('b', m)
            pt.create_shared_array('w', m)
            pt.shared_arrays['a'].array[:] = A
            pt.shared_arrays['b'].array[:] = b
            pt.shared_arrays['w'].array[:] = w
            pt.fitsnap_dict['Testing'] = testing.tolist() if mask else [False] * m
            s.perform_fit()
        return np.asarray(s.fit, dtype=np.float64).copy()

    out["lasso_class_all"] = run_lasso(None, False)                                   # alpha = 1e-8, max_iter = 2000
    out["lasso_class_mask"] = run_lasso(None, True)
    out["lasso_class_alpha1e-2_mask"] = run_lasso({"LASSO": {"alpha": 1.0e-2}}, True)
    out["lasso_class_alpha1_all"] = run_lasso({"LASSO": {"alpha": 1.0}}, False)         # sparse, stops at max_iter
    out["lasso_class_alpha1_iter50_all"] = run_lasso({"LASSO": {"alpha": 1.0, "max_iter": 50}}, False)
    out["lasso_class_transpose"] = run_lasso({"LASSO": {"alpha": 1.0e-2}, "EXTRAS": {"apply_transpose": 1}}, False)
    np.savez_compressed(os.path.join(HERE, "ta_reference_fits.npz"), **out)
    np.savez_compressed(os.path.join(HERE, "ta_abw.npz"), A=A, b=b, w=w)

    # second, independent golden set (XYZ-scraped Ta): reference SVD fit on it
    if os.path.exists(os.path.join(TAXYZ, "Descriptors.npy")):
        A2 = np.load(os.path.join(TAXYZ, "Descriptors.npy"))
        b2 = np.load(os.path.join(TAXYZ, "Truth-Ref.npy"))
        w2 = np.load(os.path.join(TAXYZ, "Weights.npy"))
        pt = ParallelTools()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            cfg = Config(pt, settings("SVD"), arguments_lst=["--overwrite"])
        s = solver_factory.solver("SVD", pt, cfg)
        s.perform_fit(A2, b2, w2, trainall=True)
        # store only what differs from the first set to keep the fixture small
        np.savez_compressed(os.path.join(HERE, "ta_xyz_delta.npz"), dA=(A2 - A).astype(np.float64),
                            db=b2 - b, dw=w2 - w, svd_all=np.asarray(s.fit))
    for k, v in out.items():
        print(f"{k:40s} {np.asarray(v).shape}")
    print("svd vs snapcoeff max abs diff:", np.max(np.abs(out["svd_all"] - out["snapcoeff"])))


if __name__ == "__main__":
    main()
