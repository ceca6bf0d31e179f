"""CPU: host half of the assembly rows — row plans, factory discovery, config — against the
reference-generated goldens (tests/golden/assembly_reference.npz).  No GPU."""
import numpy as np
import pytest

from fitsnap_amd.calculators import calculator_factory
from fitsnap_amd.calculators.calculator import Calculator
from fitsnap_amd.calculators.lammps_base import LammpsBase, _extract_compute_np
from fitsnap_amd.calculators.row_plan import config_row_plan, type_fractions
from fitsnap_amd.config import Config
from fitsnap_amd.parallel_tools import ParallelTools

import fake_lammps
from assembly_cases import load_cases

CASES = load_cases()


def cfg_of(g):
    s = {k: dict(v) for k, v in g["settings"].items() if k != "REFERENCE"}
    return Config(ParallelTools(), s)


def test_factory_finds_grandchildren_by_name():
    # fitsnap3lib/calculators/calculator_factory.py:22-38
    for name, cls in (("LAMMPSSNAP", "LammpsSnap"), ("lammpspace", "LammpsPace")):
        inst = calculator_factory.search(name)
        assert type(inst).__name__ == cls and isinstance(inst, LammpsBase) and isinstance(inst, Calculator)
    with pytest.raises(IndexError, match="was not found in fitsnap calculators"):
        calculator_factory.search("nonesuch")


@pytest.mark.parametrize("name", sorted(CASES))
def test_config_width_and_blank2j_match_reference(name):
    g = CASES[name]
    cfg = cfg_of(g)
    bis = cfg.sections["BISPECTRUM"]
    assert bis.ncoeff == int(g["ncoeff"]) and np.array_equal(bis.blank2J, g["blank2J"])
    fake_lammps.install()
    calc = calculator_factory.calculator("LAMMPSSNAP", ParallelTools(), cfg)
    assert calc.get_width() == int(g["width"])          # lammps_snap.py:15-23


@pytest.mark.parametrize("name", sorted(CASES))
def test_row_plan_metadata_matches_reference(name):
    g = CASES[name]
    cfg = cfg_of(g)
    bis, calc = cfg.sections["BISPECTRUM"], cfg.sections["CALCULATOR"]
    rt, ai, at, n_rows, raw0 = [], [], [], 0, 0
    for i, c in enumerate(g["configs"]):
        plan, meta = config_row_plan(c["natoms"], c["types"], c["vol"], c["energy"], c["forces"], c["stress"], c["eweight"],
                                     c["fweight"], c["vweight"], calc.energy, calc.force, calc.stress, bis.bikflag, raw0,
                                     -1 if bis.bzeroflag else i)
        raw0 += c["raw"].shape[0]
        rt += meta["Row_Type"]; ai += meta["Atom_I"]; at += meta["Atom_Type"]
        n = len(plan["src_row"])
        n_rows += n
        assert all(len(plan[k]) == n for k in plan)
        assert plan["src_row"].max() < raw0
    assert n_rows == len(g["b"])
    assert rt == [str(x) for x in g["Row_Type"]]
    assert ai == [int(x) for x in g["Atom_I"]]
    assert at == [int(x) for x in g["Atom_Type"]]


def test_type_fractions():
    f = type_fractions(["W", "Be", "W", "W"], {"W": 1, "Be": 2}, 2)
    assert f.tolist() == [0.75, 0.25]


def test_extract_compute_np_is_a_view():
    # lammps_base.py:280-307: no copy of LAMMPS' memory
    fake_lammps.install()
    raw = np.arange(12.0).reshape(3, 4)
    fake_lammps.CURRENT.update(raw=raw)
    lmp = fake_lammps.FakeLammps()
    view = _extract_compute_np(lmp, "snap", 0, 2, (3, 4))
    assert np.array_equal(view, raw)
    raw[1, 2] = -5.0
    assert view[1, 2] == -5.0


def test_row_count_follows_create_a():
    # a_len = energy rows + 3 * atoms + 6 * configs (calculator.py:263-272)
    g = CASES["snap_1type_bzero0_efs"]
    cfg = cfg_of(g)
    fake_lammps.install()
    pt = ParallelTools()
    calc = calculator_factory.calculator("LAMMPSSNAP", pt, cfg)
    data = [{"Positions": np.zeros((c["natoms"], 3))} for c in g["configs"]]
    calc.allocate_per_config(data)
    calc.number_of_atoms = int(pt.shared_arrays["number_of_atoms"].array.sum())
    calc.number_of_files_per_node = len(data)
    assert calc.row_count() == len(g["b"])


def test_weighting_matches_reference_bitwise():
    # Scraper._weighting (scrape.py:323-353) — origin of the row weights; goldens produced by
    # the reference function itself (tests/golden/make_golden_assembly.py: weighting_goldens)
    import os
    from fitsnap_amd.scrapers import apply_weighting
    rows = np.load(os.path.join(os.path.dirname(__file__), "golden", "weighting_reference.npy"))
    assert len(rows) == 32
    for r in rows:
        boltz, smart, force, stress, test_bool, natoms = float(r[0]), int(r[1]), int(r[2]), int(r[3]), int(r[4]), int(r[5])
        # python floats, as the reference's group table holds them (np.float64 / 0 would give inf, not ZeroDivisionError)
        grp = {"eweight": float(r[6]), "fweight": float(r[7]), "vweight": float(r[8]), "training_size": int(r[9]),
               "testing_size": int(r[10])}
        data = {"Energy": float(r[11]), "test_bool": test_bool}
        apply_weighting(data, grp, natoms, boltz=boltz, smartweights=bool(smart), use_force=bool(force), use_stress=bool(stress))
        assert (data["eweight"], data["fweight"], data["vweight"]) == (r[12], r[13], r[14])
