"""Minimal counterpart of the reference's config system for the sections the linear-fit
hot path reads (fitsnap3lib/io/input.py:10-235, io/sections/**).

Same shape as the reference: ``config.sections["NAME"].attr`` and ``config.args.*``;
input is either a dict of dicts (library mode, input.py:141-151) or an INI file parsed
with configparser (input.py:110-140).  Only keys the hot path needs are typed and
defaulted here, with the reference's defaults:

  [SOLVER] solver=SVD, compute_testerrs, detailed_errors   (solver_sections/solver.py:15-30)
  [RIDGE]  alpha=1.0E-8, local_solver=0                     (solver_sections/ridge.py:13-14)
  [ARD]    alphabig, alphasmall, lambdabig, lambdasmall, threshold_lambda, directmethod,
           scap=1e-3, scai=1e-3, logcut=0.3                 (solver_sections/ard.py:13-21)
  [LASSO]  alpha=1.0E-8, max_iter=2000                      (solver_sections/lasso.py:13-14)
  [EXTRAS] apply_transpose, multinode_testing, only_test, dump_*   (extras.py:19-45)
  [CALCULATOR] calculator, energy, force, stress, per_atom_energy, linear
                                                            (calculator_sections/calculator.py:17-32)
  [BISPECTRUM] numTypes, twojmax, bzeroflag, quadraticflag, type, wj, radelem -> ncoeff,
           blank2J                                          (calculator_sections/bispectrum.py:80-125)
  [OUTFILE] metrics, potential                              (outfile.py)
A section that belongs to an un-selected solver raises UserWarning like the reference
(sections.py:93-97).
"""
from __future__ import annotations

import configparser
from types import SimpleNamespace

_BOOL_TRUE = {"1", "true", "yes", "on"}


def _get(d, key, default, kind):
    # INI keys are case-insensitive in configparser; library dicts may use any case
    val = None
    for k, v in d.items():
        if k.lower() == key.lower():
            val = v
            break
    if val is None:
        val = default
    if kind == "float":
        return float(val)
    if kind == "int":
        return int(float(val)) if not isinstance(val, bool) else int(val)
    if kind == "bool":
        if isinstance(val, str):
            return val.strip().lower() in _BOOL_TRUE
        return bool(val)
    return str(val)


def _check_keys(name, d, allowed):
    for k in d:
        if k.lower() not in {a.lower() for a in allowed}:
            raise RuntimeError(f">>> Found unmatched variable in {name} section of input: {k}")


def snap_ncoeff(twojmax: int) -> int:
    """Number of bispectrum components for a given 2J_max
    (fitsnap3lib/io/sections/calculator_sections/bispectrum.py:80-91: twojmax 6 -> 30, 8 -> 55)."""
    n = 0
    for j1 in range(twojmax + 1):
        for j2 in range(j1 + 1):
            for j in range(j1 - j2, min(twojmax, j1 + j2) + 1, 2):
                if j >= j1:
                    n += 1
    return n


def snap_blank2j(numtypes, twojmax, quadratic, bzeroflag):
    """0/1 column mask for per-type 2J_max (fitsnap3lib/io/sections/calculator_sections/
    bispectrum.py:69-125): a bispectrum component of type t is kept iff all of its (j1, j2, j)
    are <= twojmax[t]; quadratic terms iff all six indices are; with bzeroflag = 0 a 1 is
    prepended per type (the offset column)."""
    from itertools import combinations_with_replacement

    import numpy as np

    jmax = int(max(twojmax))
    blank, nper = [], 0
    for atype in range(numtypes):
        lin = []
        for j1 in range(jmax + 1):
            for j2 in range(j1 + 1):
                for j in range(abs(j1 - j2), min(jmax, j1 + j2) + 1, 2):
                    if j >= j1:
                        lin.append((j1, j2, j))
        row = [1.0 if all(ind <= int(twojmax[atype]) for ind in t) else 0.0 for t in lin]
        if quadratic:
            for a, b in combinations_with_replacement(lin, r=2):
                row.append(1.0 if all(ind <= int(twojmax[atype]) for ind in a + b) else 0.0)
        nper = len(row)
        blank.append(row)
    blank = np.array(blank, dtype=np.float64).reshape(numtypes, nper)
    if not bzeroflag:
        blank = np.concatenate((np.ones((numtypes, 1)), blank), axis=1)
    return blank.reshape(-1)


def snap_blist(numtypes, twojmax, quadratic):
    """Names of the bispectrum components [i, j1, j2, j] per type (bispectrum.py:69-119), used
    for the ``#  B[...]`` comments of the .snapcoeff file."""
    from itertools import combinations_with_replacement

    jmax = int(max(twojmax))
    out = []
    for _atype in range(numtypes):
        lin, i = [], 0
        for j1 in range(jmax + 1):
            for j2 in range(j1 + 1):
                for j in range(abs(j1 - j2), min(jmax, j1 + j2) + 1, 2):
                    if j >= j1:
                        i += 1
                        lin.append([i, j1, j2, j])
        per_type = list(lin)
        if quadratic:
            per_type += [[k, a, b] for k, (a, b) in enumerate(combinations_with_replacement(lin, r=2), start=len(lin))]
        out += per_type
    return out


class Config:
    def __init__(self, pt=None, input=None, arguments_lst=None):
        self.pt = pt
        self.input = input
        args = {"perform_fit": True, "overwrite": False, "verbose": False, "relative": False,
                "nofit": False, "infile": None}
        for a in arguments_lst or []:
            if a == "--overwrite":
                args["overwrite"] = True
            elif a == "--nofit":
                args["nofit"] = True
                args["perform_fit"] = False
            elif a in ("--verbose", "-v"):
                args["verbose"] = True
            elif a in ("--relative", "-r"):
                args["relative"] = True
        self.args = SimpleNamespace(**args)
        raw = self._read(input)
        self.sections = {}
        self._build(raw)

    @staticmethod
    def _read(input):
        if input is None:
            return {}
        if isinstance(input, dict):
            return {str(k).upper(): dict(v) for k, v in input.items()}
        cp = configparser.ConfigParser(inline_comment_prefixes=("#", ";"), interpolation=None)
        cp.optionxform = str
        with open(input) as f:
            cp.read_file(f)
        return {s.upper(): dict(cp.items(s)) for s in cp.sections()}

    def _build(self, raw):
        sol = raw.get("SOLVER", {})
        _check_keys("SOLVER", sol, ["solver", "normalweight", "normratio", "compute_testerrs", "detailed_errors",
                                   "nsam", "cov_nugget", "mcmc_num", "mcmc_gamma", "mcmc_sigma", "merr_mult",
                                   "merr_method", "merr_cfs"])
        solver = _get(sol, "solver", "SVD", "str")
        self.sections["SOLVER"] = SimpleNamespace(
            name="SOLVER", solver=solver,
            compute_testerrs=_get(sol, "compute_testerrs", "0", "bool"),
            detailed_errors=_get(sol, "detailed_errors", "0", "bool"),
            nsam=_get(sol, "nsam", "0", "int"), cov_nugget=_get(sol, "cov_nugget", "0.0", "float"),
            true_multinode=1 if solver == "ScaLAPACK" else 0)

        def not_used(section):
            # sections.py:93-97
            raise UserWarning(f"{section} section is in input, but not set as solver. Common mistake.")

        if "RIDGE" in raw or solver.upper() == "RIDGE":
            rd = raw.get("RIDGE", {})
            _check_keys("RIDGE", rd, ["alpha", "local_solver"])
            if solver.upper() != "RIDGE":
                not_used("RIDGE")
            self.sections["RIDGE"] = SimpleNamespace(name="RIDGE", alpha=_get(rd, "alpha", "1.0E-8", "float"),
                                                     local_solver=_get(rd, "local_solver", "0", "bool"))
        if "ARD" in raw or solver.upper() == "ARD":
            ad = raw.get("ARD", {})
            _check_keys("ARD", ad, ["alphabig", "alphasmall", "lambdabig", "lambdasmall", "threshold_lambda",
                                    "directmethod", "scap", "scai", "logcut"])
            if solver.upper() != "ARD":
                not_used("ARD")
            self.sections["ARD"] = SimpleNamespace(
                name="ARD",
                alphabig=_get(ad, "alphabig", "1.0E-12", "float"), alphasmall=_get(ad, "alphasmall", "1.0E-14", "float"),
                lambdabig=_get(ad, "lambdabig", "1.0E-6", "float"), lambdasmall=_get(ad, "lambdasmall", "1.0E-6", "float"),
                threshold_lambda=_get(ad, "threshold_lambda", "100000", "int"),
                directmethod=_get(ad, "directmethod", "0", "int"),
                scap=_get(ad, "scap", "1.e-3", "float"), scai=_get(ad, "scai", "1.e-3", "float"),
                logcut=_get(ad, "logcut", "0.3", "float"))

        if "LASSO" in raw or solver.upper() == "LASSO":
            ld = raw.get("LASSO", {})
            _check_keys("LASSO", ld, ["alpha", "max_iter"])
            if solver.upper() != "LASSO":
                not_used("LASSO")
            self.sections["LASSO"] = SimpleNamespace(name="LASSO", alpha=_get(ld, "alpha", "1.0E-8", "float"),
                                                     max_iter=_get(ld, "max_iter", "2000", "int"))

        ex = raw.get("EXTRAS", {})
        _check_keys("EXTRAS", ex, ["multinode_testing", "apply_transpose", "only_test", "dump_descriptors", "dump_truth",
                                   "dump_weights", "dump_dataframe", "dump_peratom", "dump_perconfig", "dump_configs"])
        out = raw.get("OUTFILE", {})
        self.sections["EXTRAS"] = SimpleNamespace(
            name="EXTRAS",
            multinode_testing=_get(ex, "multinode_testing", "0", "bool"),
            apply_transpose=_get(ex, "apply_transpose", "0", "bool"),
            only_test=_get(ex, "only_test", "0", "bool"),
            dump_a=_get(ex, "dump_descriptors", "0", "bool"), dump_b=_get(ex, "dump_truth", "0", "bool"),
            dump_w=_get(ex, "dump_weights", "0", "bool"), dump_dataframe=_get(ex, "dump_dataframe", "0", "bool"),
            descriptor_file=_get(out, "descriptors", "Descriptors.npy", "str"),
            truth_file=_get(out, "truth", "Truth-Ref.npy", "str"),
            weights_file=_get(out, "weights", "Weights.npy", "str"),
            dataframe_file=_get(out, "dataframe", "FitSNAP.df", "str"))
        self.sections["OUTFILE"] = SimpleNamespace(
            name="OUTFILE", metrics=_get(out, "metrics", "fitsnap_metrics.md", "str"),
            metrics_style=_get(out, "metrics_style", "MD", "str"),
            potential_name=_get(out, "potential", "fitsnap_potential", "str"))

        ca = raw.get("CALCULATOR", {})
        calc = _get(ca, "calculator", "LAMMPSSNAP", "str")
        self.sections["CALCULATOR"] = SimpleNamespace(
            name="CALCULATOR", calculator=calc,
            energy=_get(ca, "energy", "True", "bool"), force=_get(ca, "force", "True", "bool"),
            stress=_get(ca, "stress", "True", "bool"), per_atom_energy=_get(ca, "per_atom_energy", "False", "bool"),
            per_atom_scalar=_get(ca, "per_atom_scalar", "False", "bool"),
            nonlinear=_get(ca, "nonlinear", "False", "bool"))
        self.sections["CALCULATOR"].linear = not self.sections["CALCULATOR"].nonlinear

        if "BISPECTRUM" in raw:
            bi = raw["BISPECTRUM"]
            _check_keys("BISPECTRUM", bi, ["numTypes", "twojmax", "rcutfac", "rfac0", "rmin0", "wj", "radelem", "type",
                                           "wselfallflag", "chemflag", "bzeroflag", "quadraticflag", "bnormflag", "bikflag",
                                           "switchinnerflag", "switchflag", "sinner", "dinner", "dgradflag"])
            numtypes = _get(bi, "numTypes", "1", "int")
            twojmax = [int(x) for x in _get(bi, "twojmax", "6", "str").split()]
            if len(twojmax) == 1:
                twojmax = twojmax * numtypes
            quad = _get(bi, "quadraticflag", "0", "bool")
            ncoeff = snap_ncoeff(max(twojmax))
            if quad:
                ncoeff += ncoeff * (ncoeff + 1) // 2
            bzero = _get(bi, "bzeroflag", "0", "bool")
            types = _get(bi, "type", "H", "str").split()
            chem = _get(bi, "chemflag", "0", "bool")
            blank2j = snap_blank2j(numtypes, twojmax, quad, bzero)
            blist = snap_blist(numtypes, twojmax, quad)
            if chem:
                # explicit multi-element (EME) model, bispectrum.py:104-110, 127-134: every component exists once per
                # ordered element triple -> numTypes^3 times the descriptors per type; LAMMPS gets "N e0 e1 ..."
                if quad:
                    raise ValueError("Quadratic chemsnap not impelemented.")
                if min(twojmax) != max(twojmax):
                    raise RuntimeError("Still working on the capability to do mixed 2J values per-element and explicit "
                                       "multi-element descriptors \n Aborting...!")
                blist = blist * numtypes ** 3
                ncoeff = len(blist) // numtypes
                import numpy as _np
                blank2j = _np.ones(numtypes * (ncoeff + (0 if bzero else 1)))
                chem = " ".join([str(numtypes)] + [str(i) for i in range(len(types))])
            # inner cutoff switching (bispectrum.py:55-63): one sinner / dinner value per type, kept as the strings
            # that go into the .snapparam file
            inner = _get(bi, "switchinnerflag", "0", "bool")
            sinner = dinner = None
            if inner:
                sinner = _get(bi, "sinner", " ".join(["0.9"] * numtypes), "str")
                dinner = _get(bi, "dinner", " ".join(["0.1"] * numtypes), "str")
                if len(sinner.split()) != numtypes or len(dinner.split()) != numtypes:
                    raise ValueError("Number of sinner/dinner args must be number of types.")
            self.sections["BISPECTRUM"] = SimpleNamespace(
                name="BISPECTRUM", numtypes=numtypes, twojmax=twojmax, ncoeff=ncoeff,
                bzeroflag=bzero, quadraticflag=quad, types=types,
                type_mapping={t: i + 1 for i, t in enumerate(types)},      # bispectrum.py:29-37
                bikflag=_get(bi, "bikflag", "0", "bool"), chemflag=chem,
                wselfallflag=_get(bi, "wselfallflag", "0", "bool"),
                blank2J=blank2j,
                blist=blist,
                rcutfac=_get(bi, "rcutfac", "4.67637", "float"), rfac0=_get(bi, "rfac0", "0.99363", "float"),
                rmin0=_get(bi, "rmin0", "0.0", "float"), bnormflag=_get(bi, "bnormflag", "0", "bool"),
                switchinnerflag=inner, sinner=sinner, dinner=dinner,
                wj=[float(x) for x in _get(bi, "wj", "1.0", "str").split()],
                radelem=[float(x) for x in _get(bi, "radelem", "0.5", "str").split()])
        if "ACE" in raw:
            ac = raw["ACE"]
            numtypes = _get(ac, "numTypes", "1", "int")
            types = _get(ac, "type", "H", "str").split()
            ncoeff = _get(ac, "ncoeff", "0", "int")
            if ncoeff <= 0:
                raise NotImplementedError("[ACE] needs an explicit `ncoeff` (descriptors per type): ACE basis "
                                          "generation (fitsnap3lib/lib/sym_ACE) is outside this repository's scope")
            bzero = _get(ac, "bzeroflag", "0", "bool")
            width = numtypes * (ncoeff + (0 if bzero else 1))
            import numpy as _np
            self.sections["ACE"] = SimpleNamespace(
                name="ACE", numtypes=numtypes, ncoeff=ncoeff, bzeroflag=bzero, types=types,
                type_mapping={t: i + 1 for i, t in enumerate(types)}, bikflag=_get(ac, "bikflag", "0", "bool"),
                blank2J=_np.ones(width))
        if "REFERENCE" in raw:
            rf = raw["REFERENCE"]
            decl = ["pair_style " + _get(rf, "pair_style", "zero 10.0", "str")]      # reference.py:18-29
            decl += ["pair_coeff " + v for k, v in rf.items() if k.lower().startswith("pair_coeff")]
            self.sections["REFERENCE"] = SimpleNamespace(
                name="REFERENCE", units=_get(rf, "units", "metal", "str").lower(),
                atom_style=_get(rf, "atom_style", "atomic", "str").lower(), lmp_pairdecl=decl)
        import hashlib
        self.hash = hashlib.sha1(repr(sorted((k, sorted(v.items())) for k, v in raw.items())).encode()).hexdigest()[:30]
        mem = raw.get("MEMORY", {})
        self.sections["MEMORY"] = SimpleNamespace(name="MEMORY", override=_get(mem, "override", "0", "bool"))
