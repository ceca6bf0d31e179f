"""CPU: the C-ABI library loads, exports every symbol include/fsnap_hip.h declares, and its
host-side K x K solve reproduces the reference's fits from oracle-computed statistics.
No GPU compute is called here."""
import os
import re

import numpy as np
import pytest

from fitsnap_amd import _capi, build
from oracle import fitsnap_oracle as orc

from conftest import ROOT, maxrel


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "fsnap_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fsnap_[a-z_0-9]+)\s*\(", text)))


def test_library_builds_and_loads():
    path = build.build_library()
    assert os.path.exists(path)
    lib = _capi.load_library()
    assert lib.fsnap_version() >= 100


def test_every_declared_symbol_is_exported_and_bound():
    lib = _capi.load_library()
    declared = _declared_functions()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in fsnap_hip.h but not exported"
        assert name in _capi.SIGNATURES, f"{name} has no ctypes prototype"
    assert sorted(_capi.SIGNATURES) == declared


def test_integration_document_names_every_entry_point():
    # INTEGRATION.md is the reference-side binding guide: an entry point that is not in it cannot be bound
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [name for name in _declared_functions() if name not in doc]
    assert not missing, f"INTEGRATION.md does not mention {missing}"


def test_every_documented_option_is_accepted_and_every_option_the_host_layer_sets_exists():
    # fsnap_set_option's keys live in three places: the header's comment, the strcmp chain of fsnap_capi.cpp and the
    # set_option() calls of the Python host layer.  (A key that only the host layer knows fails on the GPU box only.)
    header = open(os.path.join(ROOT, "include", "fsnap_hip.h")).read()
    end = header.index("int fsnap_set_option(")
    documented = set(re.findall(r'"([a-z][a-z0-9_]+)"', header[header.rindex("/*", 0, end):end]))
    source = open(os.path.join(ROOT, "fitsnap_amd", "csrc", "fsnap_capi.cpp")).read()
    accepted = set(re.findall(r'strcmp\(key, "([a-z0-9_]+)"\)', source))
    assert documented <= accepted, f"documented but not accepted: {sorted(documented - accepted)}"
    assert accepted - documented <= {"ablate"}, f"accepted but not documented: {sorted(accepted - documented)}"
    used = set()
    for base, _, files in os.walk(os.path.join(ROOT, "fitsnap_amd")):
        for name in files:
            if name.endswith(".py"):
                used |= set(re.findall(r'set_option\(\s*"([a-z0-9_]+)"', open(os.path.join(base, name)).read()))
    used |= set(re.findall(r'set_option\(\s*"([a-z0-9_]+)"', open(os.path.join(ROOT, "bench.py")).read()))
    assert used and used <= accepted, f"set by the host layer but unknown to the library: {sorted(used - accepted)}"


def test_no_gpu_means_loud_failure():
    if _capi.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_capi.FsnapError, match="no CPU fallback"):
        _capi.HipContext(0)


@pytest.mark.parametrize("kind,param,key,tol", [
    (_capi.SOLVE_LSTSQ, 1e-13, "svd_all", 1e-6),
    (_capi.SOLVE_CHOL, 0.0, "svd_all", 1e-6),
    (_capi.SOLVE_RIDGE, 1e-8, "ridge_sklearn_1e-8_all", 1e-6),
    (_capi.SOLVE_RIDGE_INV, 1e-8, "ridge_local_1e-8_all", 1e-6),
    (_capi.SOLVE_RIDGE, 1e-4, "ridge_sklearn_1e-4_all", 1e-6),
    (_capi.SOLVE_RIDGE_INV, 1e-4, "ridge_local_1e-4_all", 1e-6),
])
def test_solve_matches_reference_fits(ta, ta_fits, kind, param, key, tol):
    A, b, w = ta
    G, c, _ = orc.normal_eq(A, b, w)
    beta, rank, _ = _capi.solve(kind, param, G, c)
    assert rank == 31
    assert maxrel(beta, ta_fits[key]) < tol          # north_star: 1e-6 relative
    assert np.max(np.abs(beta - ta_fits[key])) < 1e-6  # reference's own bar (example_checker.py:62)


def test_solve_zero_column_gets_zero_coefficient():
    # lstsq's minimum-norm solution puts 0 on an identically-zero column (e.g. blank2J-masked
    # SNAP columns, lammps_snap.py:467-468)
    rng = np.random.default_rng(3)
    X = rng.standard_normal((200, 6))
    X[:, 2] = 0.0
    y = rng.standard_normal(200)
    ref = np.linalg.lstsq(X, y, rcond=1e-13)[0]
    beta, rank, _ = _capi.solve(_capi.SOLVE_LSTSQ, 1e-13, X.T @ X, X.T @ y)
    assert rank == 5 and beta[2] == 0.0
    assert maxrel(beta[[0, 1, 3, 4, 5]], ref[[0, 1, 3, 4, 5]]) < 1e-10


def test_solve_rank_deficient_is_minimum_norm():
    rng = np.random.default_rng(4)
    X = rng.standard_normal((300, 5))
    X = np.hstack([X, X[:, :1] + X[:, 1:2]])          # exactly dependent column
    y = rng.standard_normal(300)
    ref, _, rk, _ = np.linalg.lstsq(X, y, rcond=1e-13)
    beta, rank, _ = _capi.solve(_capi.SOLVE_LSTSQ, 1e-13, X.T @ X, X.T @ y)
    assert rank == rk == 5
    assert np.max(np.abs(beta - ref)) < 1e-8


def test_solve_error_mapping():
    G = np.array([[1.0, 2.0], [2.0, 1.0]])            # indefinite
    with pytest.raises(np.linalg.LinAlgError):
        _capi.solve(_capi.SOLVE_CHOL, 0.0, G, np.ones(2))
    with pytest.raises(np.linalg.LinAlgError):
        _capi.solve(_capi.SOLVE_RIDGE_INV, 0.0, np.zeros((2, 2)), np.ones(2))   # np.linalg.inv: Singular matrix
    with pytest.raises(ValueError):
        _capi.solve(_capi.SOLVE_RIDGE, 1e-8, np.array([[np.nan, 0.0], [0.0, 1.0]]), np.ones(2))
    with pytest.raises(ValueError):
        _capi.solve(_capi.SOLVE_RIDGE, 1e-8, np.eye(3), np.ones(2))


def test_solve_k128_synthetic():
    A, b, w = orc.synth_problem(20000, 128)
    G, c, _ = orc.normal_eq(A, b, w)
    beta, rank, _ = _capi.solve(_capi.SOLVE_RIDGE, 1e-8, G, c)
    assert rank == 128
    assert maxrel(beta, orc.ridge_fit(A, b, w, 1e-8)) < 1e-6


def test_single_hip_runtime_whatever_the_import_order():
    # libfsnap_hip.so first, torch second must not map a second libamdhip64 (torch's device
    # enumeration fails with two runtimes in one process)
    import subprocess
    import sys
    code = (
        "import re, sys; sys.path.insert(0, %r)\n"
        "from fitsnap_amd import _capi; _capi.load_library(build_if_missing=False)\n"
        "import torch\n"
        "maps = open('/proc/self/maps').read()\n"
        "print(len(set(re.findall(r'\\S*libamdhip64\\S*', maps))))\n" % ROOT
    )
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == "1"


@pytest.mark.parametrize("env", [{"FSNAP_CHOL_VARIANT": "1"}, {"FSNAP_CHOL_VARIANT": "2"}, {"FSNAP_CHOL_VARIANT": "3", "FSNAP_CHOL_NBK": "8"},
                                 {"FSNAP_CHOL_VARIANT": "3", "FSNAP_CHOL_NBK": "32", "FSNAP_CHOL_THREADS": "3"}])
def test_host_cholesky_variants_agree(env):
    # unblocked / 64-row panels + threads / register-blocked chunks (single- and multi-threaded): same solution for
    # sizes around the 32-column chunk and panel boundaries
    import subprocess
    import sys
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "from fitsnap_amd import _capi\n"
        "worst = 0.0\n"
        "for K in (5, 31, 32, 33, 63, 64, 65, 100, 128, 129, 200, 511, 700):\n"
        "    rng = np.random.default_rng(K)\n"
        "    X = rng.standard_normal((3 * K + 7, K)) * (10.0 ** rng.uniform(-3, 3, size=K))\n"
        "    G = X.T @ X; c = X.T @ rng.standard_normal(3 * K + 7)\n"
        "    beta, rank, rc = _capi.solve(_capi.SOLVE_RIDGE, 0.0, G, c)\n"
        "    d = np.sqrt(np.diag(G)); ref = np.linalg.solve(G / d[:, None] / d[None, :], c / d) / d\n"
        "    assert rank == K\n"
        "    worst = max(worst, float(np.max(np.abs(beta - ref) / np.max(np.abs(ref)))))\n"
        "print(worst)\n" % ROOT
    )
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=e)
    assert out.returncode == 0, out.stderr
    assert float(out.stdout.strip()) < 1e-9


@pytest.mark.parametrize("K", [1, 7, 31, 47, 48, 49, 63, 96, 110, 127, 128, 129, 142, 184, 200, 255, 256, 300, 383, 384, 400, 512, 520])
def test_host_solve_padding_is_transparent(K):
    # fsnap_solve pads the Jacobi-scaled matrix to a multiple of 32 columns (and past row strides that are multiples of
    # 1 KiB) with an identity block: every K must give the dense solution, for every kind of solve
    rng = np.random.default_rng(9000 + K)
    A = rng.standard_normal((3 * K + 5, K)) * (10.0 ** rng.uniform(-3, 3, K))
    G = A.T @ A
    c = A.T @ rng.standard_normal(3 * K + 5)
    for kind, param, alpha in ((_capi.SOLVE_RIDGE, 1e-8, 1e-8), (_capi.SOLVE_LSTSQ, 1e-13, 0.0), (_capi.SOLVE_CHOL, 0.0, 0.0),
                               (_capi.SOLVE_RIDGE_INV, 1e-6, 1e-6)):
        beta, rank, rcond = _capi.solve(kind, param, G, c)
        d = 1.0 / np.sqrt(np.diag(G) + alpha)                      # reference: dense solve of the equilibrated system
        ref = d * np.linalg.solve((G + alpha * np.eye(K)) * d[:, None] * d[None, :], c * d)
        assert rank == K
        assert np.linalg.norm(beta - ref) / np.linalg.norm(ref) < 1e-8, (K, kind)


def test_peer_to_peer_communicator_ids_carry_their_transport(monkeypatch):
    # fsnap_comm_id_p2p needs no GPU: 128 bytes, a magic prefix that fsnap_comm_init recognises (the transport travels with the
    # id), a random token behind it; FSNAP_DIST_TRANSPORT=p2p makes the plain fsnap_comm_id return one too
    a, b = _capi.comm_id("p2p"), _capi.comm_id("p2p")
    assert len(a) == len(b) == _capi.COMM_ID_BYTES and a[:8] == b[:8] == b"FSNP2P01" and a[8:32] != b[8:32]
    monkeypatch.setenv("FSNAP_DIST_TRANSPORT", "p2p")
    assert _capi.comm_id()[:8] == b"FSNP2P01"
    with pytest.raises(ValueError):
        _capi.comm_id("mpi")
