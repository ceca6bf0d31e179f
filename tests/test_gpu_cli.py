"""GPU (-m gpu): `python -m fitsnap3 <in> --descriptors DIR` — the fit stage as a drop-in:
ingest the reference's dump files, fit on the GPU, write .snapcoeff/.snapparam/metrics that
the reference's own checker accepts (tests/example_checker.py:54-62: max(test - standard) < 1e-6)."""
import os
import runpy
import sys

import numpy as np
import pandas as pd
import pytest

from fitsnap_amd.io.outputs.snap import parse_snapcoeff

from conftest import GOLDEN
from test_cli_cpu import TA_IN

pytestmark = pytest.mark.gpu


def test_python_m_fitsnap3_reproduces_committed_potential(tmp_path, ta, ta_fits, monkeypatch):
    A, b, w = ta
    np.save(tmp_path / "Descriptors.npy", A)
    np.save(tmp_path / "Truth-Ref.npy", b)
    np.save(tmp_path / "Weights.npy", w)
    m = len(b)
    # golden row blocks: 363 energy, 12672 force, 2178 stress rows
    df = pd.DataFrame({"Row_Type": ["Energy"] * 363 + ["Force"] * 12672 + ["Stress"] * 2178, "Groups": ["Ta"] * m,
                       "Configs": ["c"] * m, "Testing": [False] * m, "Atom_I": [0] * m, "Atom_Type": [0] * m})
    df.to_pickle(tmp_path / "FitSNAP.df")
    (tmp_path / "Ta.in").write_text(TA_IN.replace("dump_descriptors = 1", "dump_descriptors = 0"))
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(sys, "argv", ["fitsnap3", "Ta.in", "--descriptors", str(tmp_path), "--overwrite"])
    with pytest.raises(SystemExit) as e:
        runpy.run_module("fitsnap3", run_name="__main__", alter_sys=True)
    assert e.value.code == 0
    coeffs = parse_snapcoeff(tmp_path / "Ta_pot.snapcoeff")
    standard = parse_snapcoeff(os.path.join(GOLDEN, "Ta_pot.snapcoeff"))
    assert len(coeffs) == len(standard) == 31
    assert np.max(np.abs(coeffs - standard)) < 1e-6                 # the reference's acceptance bar
    assert np.max(np.abs(coeffs - standard) / np.abs(standard)) < 1e-6
    assert os.path.exists(tmp_path / "Ta_pot.snapparam")
    md = (tmp_path / "Ta_metrics.md").read_text()
    assert "('*ALL', 'Unweighted', 'Training', 'Energy')" in md
    row = [ln for ln in md.splitlines() if "('*ALL', 'Unweighted', 'Training', 'Energy')" in ln][0].split("|")
    assert int(row[2]) == 363 and float(row[3]) == pytest.approx(ta_fits["metrics_all"][0][1], rel=6e-6)


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_cli(tmp_path, world, rank, extra_env, extra_args=()):
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    env.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world),
               FSNAP_COMM_FILE=str(tmp_path / "comm_id"), FSNAP_COMM_TOKEN="cli test", HSA_ENABLE_IPC_MODE_LEGACY="0",
               FSNAP_COMM_TIMEOUT="120", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), **extra_env)
    return subprocess.Popen([sys.executable, "-m", "fitsnap3", "Ta.in", "--descriptors", str(tmp_path), "--overwrite", *extra_args],
                            cwd=tmp_path, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


def _check_outputs(tmp_path, ta_fits):
    coeffs = parse_snapcoeff(tmp_path / "Ta_pot.snapcoeff")
    standard = parse_snapcoeff(os.path.join(GOLDEN, "Ta_pot.snapcoeff"))
    assert len(coeffs) == len(standard) == 31
    assert np.max(np.abs(coeffs - standard)) < 1e-6 and np.max(np.abs(coeffs - standard) / np.abs(standard)) < 1e-6
    md = (tmp_path / "Ta_metrics.md").read_text()
    row = [ln for ln in md.splitlines() if "('*ALL', 'Unweighted', 'Training', 'Energy')" in ln][0].split("|")
    assert int(row[2]) == 363 and float(row[3]) == pytest.approx(ta_fits["metrics_all"][0][1], rel=6e-6)


def test_python_m_fitsnap3_in_a_one_rank_rccl_job_runs_the_collective_path(tmp_path, ta, ta_fits):
    # `--comm auto` with WORLD_SIZE = 1 would be the single-process flow; FSNAP_FORCE_MULTI=1 makes the entry point open
    # the native RCCL communicator and run fsnap_fit_dist + the pooled error analysis in a communicator of one rank --
    # everything a multi-GPU launch does except a second device
    from test_cli_dist_cpu import write_dump

    write_dump(tmp_path, ta)
    p = _run_cli(tmp_path, 1, 0, {"FSNAP_FORCE_MULTI": "1"}, ("--comm", "rccl"))
    log = p.communicate(timeout=600)[0]
    assert p.returncode == 0, log[-3000:]
    _check_outputs(tmp_path, ta_fits)


@pytest.mark.parametrize("transport", ["rccl", "p2p"])
def test_python_m_fitsnap3_with_two_ranks_shards_by_configuration(tmp_path, ta, ta_fits, transport):
    # RCCL: one GPU per rank.  Peer-to-peer transport: both ranks may share the one GPU of the box
    from test_cli_dist_cpu import write_dump

    if transport == "rccl" and __import__("fitsnap_amd._capi", fromlist=["x"]).device_count() < 2:
        pytest.skip("needs two GPUs: RCCL does not put two ranks on one device")
    write_dump(tmp_path, ta)
    procs = [_run_cli(tmp_path, 2, r, {}, ("--transport", transport)) for r in range(2)]
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-4000:]
    _check_outputs(tmp_path, ta_fits)
