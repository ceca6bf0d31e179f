"""CPU: pin the assembly oracle bit-exactly to what the reference's own LammpsSnap class
produced when driven by a fake lammps object."""
import numpy as np
import pytest

from oracle import assembly_oracle as ao

from assembly_cases import load_cases, written_w_rows

CASES = load_cases()


@pytest.mark.parametrize("name", sorted(CASES))
def test_assembly_oracle_matches_reference_bitwise(name):
    g = CASES[name]
    s = g["settings"]
    bis, calc = s["BISPECTRUM"], s["CALCULATOR"]
    A, b, w, rt, ai, at = [], [], [], [], [], []
    for c in g["configs"]:
        a_, b_, w_, rt_, ai_, at_ = ao.snap_config_rows(
            c["raw"], c["natoms"], c["types"], c["vol"], c["energy"], c["forces"], c["stress"], c["eweight"],
            c["fweight"], c["vweight"], int(bis["numTypes"]), int(g["ncoeff"]), bool(bis["bzeroflag"]), g["blank2J"],
            bool(calc["energy"]), bool(calc["force"]), bool(calc["stress"]), bool(bis.get("bikflag", 0)))
        A.append(a_); b.append(b_); w.append(w_); rt += rt_; ai += ai_; at += at_
    A, b, w = np.concatenate(A), np.concatenate(b), np.concatenate(w)
    assert A.shape == g["A"].shape == (len(g["b"]), int(g["width"]))
    assert np.array_equal(A, g["A"])
    assert np.array_equal(b, g["b"])
    ok = written_w_rows(g)
    assert np.array_equal(w[ok], g["w"][ok])
    assert rt == [str(x) for x in g["Row_Type"]]
    assert ai == [int(x) for x in g["Atom_I"]]
    assert at == [int(x) for x in g["Atom_Type"]]
