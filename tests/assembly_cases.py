"""Helpers shared by the assembly tests: unpack the golden cases of
tests/golden/assembly_reference.npz (produced by the reference itself, see
tests/golden/make_golden_assembly.py)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "assembly_reference.npz")


def load_cases():
    z = np.load(GOLDEN, allow_pickle=False)
    out = {}
    for name in z["cases"]:
        name = str(name)
        g = {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(name + "/")}
        g["settings"] = json.loads(str(g["settings_json"]))
        # per-configuration inputs
        natoms = g["natoms"]
        cfgs, ro, ao = [], 0, 0
        ncols = int(g["raw_cols"])
        for i, n in enumerate(natoms):
            nr = int(g["raw_rows"][i])
            cfgs.append(dict(
                raw=g["raw_concat"][ro:ro + nr * ncols].reshape(nr, ncols).copy(), natoms=int(n), vol=float(g["vols"][i]),
                types=g["types_concat"][ao:ao + n].astype(np.int32), atomtypes=[str(a) for a in g["atomtypes_concat"][ao:ao + n]],
                energy=float(g["energy"][i]), forces=g["forces_concat"][3 * ao:3 * (ao + n)].reshape(n, 3).copy(),
                stress=g["stress"][i].copy(), eweight=float(g["eweight"][i]), fweight=float(g["fweight"][i]),
                vweight=float(g["vweight"][i]), test_bool=int(g["test_bool"][i]), group=str(g["group"][i]), file=str(g["file"][i])))
            ro += nr * ncols
            ao += n
        g["configs"] = cfgs
        out[name] = g
    return out


def written_w_rows(g):
    """Rows whose weight the reference actually writes (bik energy rows 1.. are left as
    uninitialised memory by lammps_snap.py:477)."""
    ok = np.ones(len(g["w"]), dtype=bool)
    bik = bool(g["settings"]["BISPECTRUM"].get("bikflag", 0))
    if bik:
        ok &= ~((g["Row_Type"] == "Energy") & (g["Atom_I"] > 0))
    return ok
