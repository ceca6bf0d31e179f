"""Golden vectors for the post-LAMMPS assembly (`_collect_lammps`) by RUNNING THE REFERENCE'S
OWN calculator classes against a fake ``lammps`` object.

Run in the build container only:   python tests/golden/make_golden_assembly.py

The fake ``lammps`` class hands the reference a ctypes ``double**`` over a synthetic
``compute snap`` / ``compute pace`` global array (the layout LAMMPS produces:
(bik_rows + 3N + 6) x (ncoeff*ntypes + 1), last column = reference potential), atom ids /
types / positions and a cell volume.  The reference's LammpsSnap / LammpsPace then run
``allocate_per_config -> create_a -> process_configs -> collect_distributed_lists`` exactly
as in ``FitSnap.process_configs`` (fitsnap3lib/fitsnap.py:134-188); ``_prepare_lammps`` and
``_run_lammps`` are no-ops (no descriptor physics is involved in what is pinned here).
Stored per case: the inputs (raw arrays, config dicts as arrays, settings) and the
reference's outputs (A, b, w, Row_Type, Atom_I, Atom_Type, Groups, Configs, Testing).
Only data is written.
"""
from __future__ import annotations

import ctypes
import json
import os
import sys
import types
import warnings

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

CURRENT = {}  # what the fake LAMMPS instance serves right now


class _FakeNumpy:
    def extract_atom(self, name, nelem=None, dim=1, **kw):
        if name == "id":
            return CURRENT["ids"]
        if name == "type":
            return CURRENT["types"]
        if name == "x":
            return CURRENT["pos"]
        raise KeyError(name)


class FakeLammps:
    has_exceptions = True
    installed_packages = []

    def __init__(self, *a, **k):
        self.numpy = _FakeNumpy()
        self._keep = None

    def command(self, s):
        pass

    def close(self):
        pass

    def version(self):
        return 20250722

    def get_natoms(self):
        return len(CURRENT["ids"])

    def get_thermo(self, key):
        assert key == "vol"
        return CURRENT["vol"]

    def create_atoms(self, **kw):
        pass

    def extract_compute(self, name, style, rtype):
        arr = CURRENT["raw"]
        assert arr.flags["C_CONTIGUOUS"] and arr.dtype == np.float64
        rows = (ctypes.POINTER(ctypes.c_double) * arr.shape[0])()
        for r in range(arr.shape[0]):
            rows[r] = ctypes.cast(arr.ctypes.data + r * arr.strides[0], ctypes.POINTER(ctypes.c_double))
        self._keep = rows
        return ctypes.cast(rows, ctypes.POINTER(ctypes.POINTER(ctypes.c_double)))


def import_reference():
    stub = types.ModuleType("lammps")
    stub.lammps = FakeLammps
    sys.modules["lammps"] = stub
    sys.path.insert(0, REF)
    from fitsnap3lib.parallel_tools import ParallelTools
    from fitsnap3lib.io.input import Config
    from fitsnap3lib.calculators import calculator_factory
    return ParallelTools, Config, calculator_factory


def snap_settings(numtypes, twojmax, bzeroflag, energy, force, stress, bikflag=0, quadratic=0):
    tj = " ".join(str(t) for t in twojmax)
    s = {
        "BISPECTRUM": {"numTypes": numtypes, "twojmax": tj, "rcutfac": 4.67637, "rfac0": 0.99363, "rmin0": 0.0,
                       "wj": " ".join(["1.0"] * numtypes), "radelem": " ".join(["0.5"] * numtypes),
                       "type": " ".join(["Ta", "W", "Be"][:numtypes]), "wselfallflag": 0, "chemflag": 0,
                       "bzeroflag": bzeroflag, "quadraticflag": quadratic, "bikflag": bikflag},
        "CALCULATOR": {"calculator": "LAMMPSSNAP", "energy": energy, "force": force, "stress": stress,
                       "per_atom_energy": bikflag},
        "SOLVER": {"solver": "SVD"},
        "OUTFILE": {"metrics": "m.md", "potential": "pot"},
        "REFERENCE": {"units": "metal", "atom_style": "atomic", "pair_style": "zero 10.0", "pair_coeff": "* *"},
    }
    return s


def make_configs(rng, natoms_list, elems, testing_every=3):
    data = []
    for i, n in enumerate(natoms_list):
        data.append({
            "Group": f"grp{i % 2}", "File": f"cfg{i}.json", "NumAtoms": n,
            "AtomTypes": [elems[int(j)] for j in rng.integers(0, len(elems), n)],
            "Positions": rng.uniform(0, 5, (n, 3)),
            "Energy": float(rng.normal(-10 * n, 1.0)),
            "Forces": rng.normal(0, 1, (n, 3)),
            "Stress": (lambda s: (s + s.T) / 2)(rng.normal(0, 1e4, (3, 3))),
            "Lattice": np.diag([5.0, 5.0, 5.0]),
            "eweight": float(10 ** rng.uniform(-1, 2)), "fweight": float(10 ** rng.uniform(-2, 1)),
            "vweight": float(10 ** rng.uniform(-9, -6)), "test_bool": int(i % testing_every == testing_every - 1),
        })
    return data


def run_case(name, ParallelTools, Config, calculator_factory, settings, natoms_list, seed):
    rng = np.random.default_rng(seed)
    pt = ParallelTools()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cfg = Config(pt, settings, arguments_lst=["--overwrite"])
    calc = calculator_factory.calculator("LAMMPSSNAP", pt, cfg)
    calc._prepare_lammps = lambda: None
    calc._run_lammps = lambda: None
    bis = cfg.sections["BISPECTRUM"]
    numtypes, ncoeff = bis.numtypes, bis.ncoeff
    elems = list(bis.type_mapping.keys())
    data = make_configs(rng, natoms_list, elems)
    bik = bool(bis.bikflag)
    raws, vols, types_l = [], [], []
    for d in data:
        n = d["NumAtoms"]
        nrows = (n if bik else 1) + 3 * n + 6
        raw = rng.normal(0, 1, (nrows, ncoeff * numtypes + 1)) * (10.0 ** rng.uniform(-3, 3, (1, ncoeff * numtypes + 1)))
        raws.append(np.ascontiguousarray(raw))
        vols.append(float(rng.uniform(50, 500)))
        types_l.append(np.array([bis.type_mapping[a] for a in d["AtomTypes"]], dtype=np.int32))
    # the reference flow of FitSnap.process_configs (fitsnap.py:134-188)
    calc.allocate_per_config(data)
    pt.add_2_fitsnap("energy", cfg.sections["CALCULATOR"].energy)
    pt.add_2_fitsnap("force", cfg.sections["CALCULATOR"].force)
    pt.add_2_fitsnap("stress", cfg.sections["CALCULATOR"].stress)
    pt.add_2_fitsnap("per_atom_energy", cfg.sections["CALCULATOR"].per_atom_energy)
    pt.add_2_fitsnap("per_atom_scalar", getattr(cfg.sections["CALCULATOR"], "per_atom_scalar", False))
    pt.add_2_fitsnap("nonlinear", getattr(cfg.sections["CALCULATOR"], "nonlinear", False))
    calc.create_a()
    calc.shared_index = 0
    calc.distributed_index = 0
    for i, d in enumerate(data):
        CURRENT.update(raw=raws[i], vol=vols[i], types=types_l[i], ids=1 + np.arange(d["NumAtoms"]), pos=d["Positions"])
        calc.process_configs(d, i)
    calc.collect_distributed_lists()
    out = {
        "A": pt.shared_arrays["a"].array.copy(), "b": pt.shared_arrays["b"].array.copy(),
        "w": pt.shared_arrays["w"].array.copy(),
        "Row_Type": np.array(pt.fitsnap_dict["Row_Type"]), "Atom_I": np.array(pt.fitsnap_dict["Atom_I"]),
        "Atom_Type": np.array(pt.fitsnap_dict["Atom_Type"]), "Groups": np.array(pt.fitsnap_dict["Groups"]),
        "Configs": np.array(pt.fitsnap_dict["Configs"]), "Testing": np.array(pt.fitsnap_dict["Testing"]),
        "blank2J": np.asarray(bis.blank2J, dtype=np.float64), "ncoeff": ncoeff, "width": calc.get_width(),
        "settings_json": json.dumps(settings),
        "raw_concat": np.concatenate([r.ravel() for r in raws]),
        "raw_rows": np.array([r.shape[0] for r in raws]), "raw_cols": raws[0].shape[1],
        "vols": np.array(vols), "natoms": np.array(natoms_list),
        "types_concat": np.concatenate(types_l),
        "energy": np.array([d["Energy"] for d in data]),
        "forces_concat": np.concatenate([d["Forces"].ravel() for d in data]),
        "stress": np.array([d["Stress"] for d in data]),
        "eweight": np.array([d["eweight"] for d in data]), "fweight": np.array([d["fweight"] for d in data]),
        "vweight": np.array([d["vweight"] for d in data]), "test_bool": np.array([d["test_bool"] for d in data]),
        "group": np.array([d["Group"] for d in data]), "file": np.array([d["File"] for d in data]),
        "atomtypes_concat": np.array([a for d in data for a in d["AtomTypes"]]),
    }
    print(f"{name:28s} A {out['A'].shape} width {out['width']} ncoeff {ncoeff}")
    return out


def main():
    ParallelTools, Config, calculator_factory = import_reference()
    cases = {
        # name: (settings, natoms per config, seed)
        "snap_1type_bzero0_efs": (snap_settings(1, [6], 0, 1, 1, 1), [2, 3, 5, 1], 11),
        "snap_1type_bzero1_efs": (snap_settings(1, [6], 1, 1, 1, 1), [2, 4, 3], 12),
        "snap_2type_bzero0_efs": (snap_settings(2, [4, 2], 0, 1, 1, 1), [3, 2, 6], 13),      # blank2J zeros (mixed 2J)
        "snap_2type_bzero1_efs": (snap_settings(2, [4, 4], 1, 1, 1, 1), [3, 5], 14),
        "snap_1type_bzero0_e": (snap_settings(1, [4], 0, 1, 0, 0), [2, 3, 4], 15),
        "snap_1type_bzero0_f": (snap_settings(1, [4], 0, 0, 1, 0), [2, 3], 16),
        "snap_1type_bzero0_es": (snap_settings(1, [4], 0, 1, 0, 1), [2, 3], 17),
        "snap_2type_bzero0_fs": (snap_settings(2, [2, 2], 0, 0, 1, 1), [4, 2], 18),
        "snap_1type_bik_bzero1_ef": (snap_settings(1, [4], 1, 1, 1, 0, bikflag=1), [3, 2], 19),
        "snap_1type_quad_bzero0_efs": (snap_settings(1, [2], 0, 1, 1, 1, quadratic=1), [2, 3], 20),
    }
    out = {}
    for name, (settings, natoms, seed) in cases.items():
        res = run_case(name, ParallelTools, Config, calculator_factory, settings, natoms, seed)
        for k, v in res.items():
            out[f"{name}/{k}"] = v
    out["cases"] = np.array(list(cases.keys()))
    np.savez_compressed(os.path.join(HERE, "assembly_reference.npz"), **out)
    print("wrote", os.path.join(HERE, "assembly_reference.npz"))




def pace_goldens():
    """LammpsPace._collect_lammps (lammps_pace.py:369-509).  The reference's [ACE] section needs
    sympy (absent here) only to GENERATE the basis; the collect step reads just numtypes /
    ncoeff / bzeroflag / type_mapping / blank2J / bikflag, so those are injected."""
    import types as _t

    ParallelTools, Config, calculator_factory = import_reference()
    from fitsnap3lib.calculators.lammps_pace import LammpsPace
    out = {}
    cases = {"pace_1type_bzero0_efs": (1, 7, 0, 1, 1, 1, [2, 3, 4], 31),
             "pace_2type_bzero1_ef": (2, 5, 1, 1, 1, 0, [3, 2], 32),
             "pace_1type_bzero0_efs_nan": (1, 4, 0, 1, 1, 1, [2, 2], 33)}
    for name, (nt, nc, bzero, e, f, st, natoms, seed) in cases.items():
        rng = np.random.default_rng(seed)
        settings = snap_settings(nt, [2] * nt, bzero, e, f, st)
        pt = ParallelTools()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            cfg = Config(pt, settings, arguments_lst=["--overwrite"])
        elems = ["Ta", "W", "Be"][:nt]
        width = nt * (nc + (0 if bzero else 1))
        cfg.sections["ACE"] = _t.SimpleNamespace(numtypes=nt, ncoeff=nc, bzeroflag=bzero, bikflag=0,
                                                 type_mapping={el: i + 1 for i, el in enumerate(elems)},
                                                 blank2J=np.ones(width))
        calc = LammpsPace.__new__(LammpsPace)
        LammpsPace.__init__(calc, "LAMMPSPACE", pt, cfg)
        calc._prepare_lammps = lambda: None
        calc._run_lammps = lambda: None
        data = make_configs(rng, natoms, elems)
        raws, vols, types_l = [], [], []
        for d in data:
            n = d["NumAtoms"]
            raw = rng.normal(0, 1, (1 + 3 * n + 6, nc * nt + 1))
            if name.endswith("_nan") and not raws:
                raw[2, 1] = np.nan
                raw[4, 0] = np.inf
            raws.append(np.ascontiguousarray(raw))
            vols.append(float(rng.uniform(50, 500)))
            types_l.append(np.array([cfg.sections["ACE"].type_mapping[a] for a in d["AtomTypes"]], dtype=np.int32))
        calc.allocate_per_config(data)
        for k, v in (("energy", e), ("force", f), ("stress", st), ("per_atom_energy", 0), ("per_atom_scalar", 0), ("nonlinear", 0)):
            pt.add_2_fitsnap(k, v)
        calc.create_a()
        calc.shared_index = 0
        calc.distributed_index = 0
        for i, d in enumerate(data):
            CURRENT.update(raw=raws[i].copy(), vol=vols[i], types=types_l[i], ids=1 + np.arange(d["NumAtoms"]), pos=d["Positions"])
            calc.process_configs(d, i)
        calc.collect_distributed_lists()
        res = {"A": pt.shared_arrays["a"].array.copy(), "b": pt.shared_arrays["b"].array.copy(),
               "w": pt.shared_arrays["w"].array.copy(), "Row_Type": np.array(pt.fitsnap_dict["Row_Type"]),
               "Atom_I": np.array(pt.fitsnap_dict["Atom_I"]), "Testing": np.array(pt.fitsnap_dict["Testing"]),
               "ntypes": nt, "ncoeff": nc, "bzeroflag": bzero, "efs": np.array([e, f, st]),
               "raw_concat": np.concatenate([r.ravel() for r in raws]), "raw_rows": np.array([r.shape[0] for r in raws]),
               "raw_cols": raws[0].shape[1], "vols": np.array(vols), "natoms": np.array(natoms),
               "types_concat": np.concatenate(types_l), "energy": np.array([d["Energy"] for d in data]),
               "forces_concat": np.concatenate([d["Forces"].ravel() for d in data]),
               "stress": np.array([d["Stress"] for d in data]), "eweight": np.array([d["eweight"] for d in data]),
               "fweight": np.array([d["fweight"] for d in data]), "vweight": np.array([d["vweight"] for d in data]),
               "test_bool": np.array([d["test_bool"] for d in data]),
               "atomtypes_concat": np.array([a for d in data for a in d["AtomTypes"]])}
        print(f"{name:28s} A {res['A'].shape}")
        for k, v in res.items():
            out[f"{name}/{k}"] = v
    out["cases"] = np.array(list(cases.keys()))
    np.savez_compressed(os.path.join(HERE, "assembly_pace_reference.npz"), **out)


def weighting_goldens():
    """Scraper._weighting (scrape.py:323-353) run unbound on a stand-in `self`."""
    import types as _t

    sys.path.insert(0, REF)
    if "lammps" not in sys.modules:
        stub = types.ModuleType("lammps")
        stub.lammps = FakeLammps
        sys.modules["lammps"] = stub
    from fitsnap3lib.scrapers.scrape import Scraper
    rng = np.random.default_rng(77)
    rows = []
    for boltz in (0.0, 300.0):
        for smart in (0, 1):
            for force in (0, 1):
                for stress in (0, 1):
                    for test_bool in (0, 1):
                        natoms = int(rng.integers(1, 40))
                        grp = {"eweight": float(10 ** rng.uniform(-1, 2)), "fweight": float(10 ** rng.uniform(-2, 1)),
                               "vweight": float(10 ** rng.uniform(-9, -6)), "training_size": int(rng.integers(0, 30)),
                               "testing_size": int(rng.integers(1, 9))}
                        energy = float(rng.normal(-5.0 * natoms, 0.5))
                        me = _t.SimpleNamespace(
                            config=_t.SimpleNamespace(sections={"GROUPS": _t.SimpleNamespace(boltz=boltz, smartweights=smart),
                                                                "CALCULATOR": _t.SimpleNamespace(force=force, stress=stress)}),
                            group_table={"g": dict(grp)}, data={"Group": "g", "Energy": energy, "test_bool": test_bool},
                            kb=0.00008617333262145)
                        Scraper._weighting(me, natoms)
                        rows.append([boltz, smart, force, stress, test_bool, natoms, grp["eweight"], grp["fweight"],
                                     grp["vweight"], grp["training_size"], grp["testing_size"], energy,
                                     me.data["eweight"], me.data["fweight"], me.data["vweight"]])
    np.save(os.path.join(HERE, "weighting_reference.npy"), np.array(rows, dtype=np.float64))
    print("wrote weighting_reference.npy", len(rows), "cases")



if __name__ == "__main__":
    main()
    weighting_goldens()
    pace_goldens()
