"""GPU (-m gpu): EXTRAS.apply_transpose for the ANL and ARD solvers (fitsnap3lib/solvers/anl.py:31-36, ard.py:22-24):
the regression runs on (aw.T aw, aw.T bw) -- K x K host algebra on the GPU statistics."""
import numpy as np
import pytest

from fitsnap_amd.config import Config
from fitsnap_amd.parallel_tools import ParallelTools
from fitsnap_amd.solvers import solver_factory
from oracle import fitsnap_oracle as orc

pytestmark = pytest.mark.gpu


def test_anl_apply_transpose_against_the_reference_class(ta, ta_fits, tmp_path, monkeypatch):
    # The reference's transposed ANL inverts G.T G (cond ~5e21) with pinv: its OWN answer moves by percent under a
    # 1e-16 relative perturbation of G (oracle with G summed in another order: 3e-2 element-wise, 2e-3 norm-wise), so
    # bit-level parity with the reference class is pinned on the CPU (test_oracle_golden.py) and the GPU statistics can
    # only be held to that sensitivity.
    A, b, w = ta
    monkeypatch.chdir(tmp_path)
    pt = ParallelTools()
    cfg = Config(pt, {"SOLVER": {"solver": "ANL", "nsam": 0, "cov_nugget": 1.0e-10}, "EXTRAS": {"apply_transpose": 1}})
    s = solver_factory.solver("ANL", pt, cfg)
    s.perform_fit(A, b, w, trainall=True)
    ref = ta_fits["anl_transpose_fit"]
    assert s.fit.shape == ref.shape
    assert np.linalg.norm(s.fit - ref) <= 2e-2 * np.linalg.norm(ref)
    assert s.cov.shape == ta_fits["anl_transpose_cov"].shape and np.allclose(s.cov, s.cov.T)
    # same formulas on the same statistics: the host algebra itself is exact
    G, c, _ = s.last_statistics
    inv = np.linalg.pinv(G.T @ G + 1.0e-10 * np.eye(len(c)))
    inv = inv * 0.5 + inv.T * 0.5
    assert np.array_equal(s.fit, inv @ (G.T @ c))
    pt.free()


def test_ard_apply_transpose_matches_scikit_learn_on_the_statistics():
    rng = np.random.default_rng(4)
    m, K = 3000, 12
    A = rng.standard_normal((m, K))
    beta = np.zeros(K)
    beta[[1, 4, 7]] = [2.0, -1.0, 0.5]
    b = A @ beta + 0.05 * rng.standard_normal(m)
    w = np.ones(m)
    pt = ParallelTools()
    cfg = Config(pt, {"SOLVER": {"solver": "ARD"}, "EXTRAS": {"apply_transpose": 1}})
    s = solver_factory.solver("ARD", pt, cfg)
    for name, arr in (("a", A), ("b", b), ("w", w)):
        pt.create_shared_array(name, m, K if name == "a" else 1)
        pt.shared_arrays[name].array[:] = arr
    pt.fitsnap_dict["Testing"] = [False] * m
    s.perform_fit()
    G, c, _ = orc.normal_eq(A, b, w)
    ref = orc.ard_fit(G, c, np.ones(K))                       # ard.py:22-24: X = aw.T aw, y = aw.T bw
    assert np.array_equal(s.fit != 0, ref != 0)
    nz = ref != 0
    assert np.max(np.abs(s.fit[nz] - ref[nz]) / np.abs(ref[nz])) < 1e-3
    pt.free()
