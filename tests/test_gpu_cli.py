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
