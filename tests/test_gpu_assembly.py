"""GPU (-m gpu): the post-LAMMPS assembly through the plugin API and the C ABI
(fsnap_rows_alloc / fsnap_assemble / fsnap_download_rows) against rows produced by the
reference's own LammpsSnap class (tests/golden/assembly_reference.npz).  Bit-exact."""
import numpy as np
import pytest

from fitsnap_amd.calculators import calculator_factory
from fitsnap_amd.config import Config
from fitsnap_amd.parallel_tools import ParallelTools
from fitsnap_amd.solvers import solver_factory
from oracle import fitsnap_oracle as orc

import fake_lammps
from assembly_cases import load_cases, written_w_rows

pytestmark = pytest.mark.gpu
CASES = load_cases()


def data_dicts(g):
    return [dict(Group=c["group"], File=c["file"], NumAtoms=c["natoms"], AtomTypes=c["atomtypes"],
                 Positions=np.zeros((c["natoms"], 3)), Energy=c["energy"], Forces=c["forces"], Stress=c["stress"],
                 eweight=c["eweight"], fweight=c["fweight"], vweight=c["vweight"], test_bool=c["test_bool"])
            for c in g["configs"]]


def serve(c):
    fake_lammps.CURRENT.update(raw=c["raw"], vol=c["vol"], types=c["types"], ids=1 + np.arange(c["natoms"]),
                               pos=np.zeros((c["natoms"], 3)))


def build(g, solver="SVD", extra=None):
    fake_lammps.install()
    s = {k: dict(v) for k, v in g["settings"].items() if k != "REFERENCE"}
    s["SOLVER"] = {"solver": solver}
    s.update(extra or {})
    pt = ParallelTools()
    cfg = Config(pt, s)
    calc = calculator_factory.calculator("LAMMPSSNAP", pt, cfg)
    calc._prepare_lammps = lambda: None
    calc._run_lammps = lambda: None
    return pt, cfg, calc


def run_flow(g, batch_bytes=None, **kw):
    pt, cfg, calc = build(g, **kw)
    if batch_bytes is not None:
        calc.BATCH_BYTES = batch_bytes
    data = data_dicts(g)
    # FitSnap.process_configs (fitsnap.py:134-188)
    calc.allocate_per_config(data)
    calc.create_a()
    calc.shared_index = 0
    calc.distributed_index = 0
    for i, d in enumerate(data):
        serve(g["configs"][i])
        calc.process_configs(d, i)
    calc.collect_distributed_lists()
    return pt, cfg, calc


@pytest.mark.parametrize("batch_bytes", [None, 1])          # one batch for everything / one kernel call per configuration
@pytest.mark.parametrize("name", sorted(CASES))
def test_assembled_rows_match_reference_bitwise(name, batch_bytes):
    g = CASES[name]
    pt, cfg, calc = run_flow(g, batch_bytes)
    A, b, w = pt.shared_arrays["a"].array, pt.shared_arrays["b"].array, pt.shared_arrays["w"].array
    assert A.shape == g["A"].shape
    assert np.array_equal(A, g["A"])
    assert np.array_equal(b, g["b"])
    ok = written_w_rows(g)
    assert np.array_equal(w[ok], g["w"][ok]) and not w[~ok].any()
    fd = pt.fitsnap_dict
    assert fd["Row_Type"] == [str(x) for x in g["Row_Type"]]
    assert fd["Atom_I"] == [int(x) for x in g["Atom_I"]]
    assert fd["Atom_Type"] == [int(x) for x in g["Atom_Type"]]
    assert fd["Groups"] == [str(x) for x in g["Groups"]]
    assert fd["Configs"] == [str(x) for x in g["Configs"]]
    assert fd["Testing"] == [bool(x) for x in g["Testing"]]
    pt.free()


def test_process_single_returns_fresh_rows():
    # lammps_base.py:101-125 / lammps_snap.py:224-389: (a, b, w) of one configuration
    g = CASES["snap_2type_bzero0_efs"]
    pt, cfg, calc = build(g)
    d = data_dicts(g)
    row = 0
    for i, c in enumerate(g["configs"]):
        serve(c)
        a, b, w = calc.process_single(d[i], i)
        n = len(b)
        assert np.array_equal(a, g["A"][row:row + n]) and np.array_equal(b, g["b"][row:row + n])
        assert np.array_equal(w, g["w"][row:row + n])
        row += n
    assert row == len(g["b"])
    pt.free()


def test_fit_runs_on_device_assembled_rows_without_upload(monkeypatch):
    # rows assembled in HBM are used by the solver as they are: no H2D of A
    g = CASES["snap_1type_bzero0_efs"]
    pt, cfg, calc = run_flow(g, solver="RIDGE", extra={"RIDGE": {"alpha": 1e-6}})
    s = solver_factory.solver("RIDGE", pt, cfg)
    calls = []
    monkeypatch.setattr(type(pt.hip()), "upload_rows", lambda self, A, b: calls.append(1))
    s.perform_fit()
    assert not calls
    A, b, w = g["A"], g["b"], np.where(written_w_rows(g), g["w"], 0.0)
    G, c, sc = s.last_statistics
    Gr, cr, scr = orc.normal_eq(A, b, w, g["Testing"])
    d = np.sqrt(np.maximum(np.diag(Gr), 1e-300))
    assert np.max(np.abs(G - Gr) / (d[:, None] * d[None, :])) < 1e-12
    assert sc[2] == scr[2]
    # writing into the host view invalidates the resident copy -> next fit uploads
    pt.shared_arrays["a"].array[0, 0] += 1.0
    pt.shared_arrays["a"].touch()
    monkeypatch.undo()
    s.perform_fit()
    G2, _, _ = s.last_statistics
    assert G2[0, 0] != G[0, 0]
    pt.free()


def test_extras_dump_files(tmp_path):
    # calculator.py:329-348: the on-disk A/b/w hand-off
    g = CASES["snap_1type_bzero1_efs"]
    out = {"descriptors": str(tmp_path / "Descriptors.npy"), "truth": str(tmp_path / "Truth-Ref.npy"),
           "weights": str(tmp_path / "Weights.npy"), "dataframe": str(tmp_path / "FitSNAP.df")}
    pt, cfg, calc = run_flow(g, extra={"EXTRAS": {"dump_descriptors": 1, "dump_truth": 1, "dump_weights": 1,
                                                   "dump_dataframe": 1}, "OUTFILE": out})
    calc.extras()
    assert np.array_equal(np.load(out["descriptors"]), g["A"])
    assert np.array_equal(np.load(out["truth"]), g["b"])
    assert np.array_equal(np.load(out["weights"]), g["w"])
    import pandas as pd
    df = pd.read_pickle(out["dataframe"])
    assert list(df["Row_Type"]) == [str(x) for x in g["Row_Type"]] and len(df) == len(g["b"])
    pt.free()


def test_nan_in_lammps_output_raises_like_reference():
    g = CASES["snap_1type_bzero0_e"]
    pt, cfg, calc = build(g)
    data = data_dicts(g)
    calc.allocate_per_config(data)
    calc.create_a()
    calc.shared_index = 0
    bad = dict(g["configs"][0])
    bad["raw"] = bad["raw"].copy()
    bad["raw"][0, 1] = np.nan
    serve(bad)
    with pytest.raises(ValueError, match="Nan in computed data"):        # lammps_snap.py:426-428
        calc.process_configs(data[0], 0)
    pt.free()


# ---------------------------------------------------------------------------------------
# PACE (lammps_pace.py:369-509): goldens from the reference's LammpsPace with an injected
# [ACE] section (tests/golden/make_golden_assembly.py: pace_goldens)
# ---------------------------------------------------------------------------------------
def _pace_cases():
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "assembly_pace_reference.npz"))
    return {str(n): {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(str(n) + "/")} for n in z["cases"]}


PACE = _pace_cases()


@pytest.mark.parametrize("name", sorted(PACE))
def test_pace_assembled_rows_match_reference_bitwise(name, capsys):
    g = PACE[name]
    fake_lammps.install()
    nt, nc, bzero = int(g["ntypes"]), int(g["ncoeff"]), int(g["bzeroflag"])
    e, f, st = (int(x) for x in g["efs"])
    elems = ["Ta", "W", "Be"][:nt]
    pt = ParallelTools()
    cfg = Config(pt, {"SOLVER": {"solver": "SVD"},
                      "CALCULATOR": {"calculator": "LAMMPSPACE", "energy": e, "force": f, "stress": st},
                      "ACE": {"numTypes": nt, "ncoeff": nc, "bzeroflag": bzero, "type": " ".join(elems)}})
    calc = calculator_factory.calculator("LAMMPSPACE", pt, cfg)
    assert type(calc).__name__ == "LammpsPace" and calc.get_width() == g["A"].shape[1]
    calc._prepare_lammps = lambda: None
    calc._run_lammps = lambda: None
    ncols = int(g["raw_cols"])
    data, raws, ro, ao = [], [], 0, 0
    for i, n in enumerate(g["natoms"]):
        n = int(n)
        nr = int(g["raw_rows"][i])
        raws.append(g["raw_concat"][ro:ro + nr * ncols].reshape(nr, ncols).copy())
        data.append(dict(Group="g", File=f"c{i}", NumAtoms=n, AtomTypes=[str(a) for a in g["atomtypes_concat"][ao:ao + n]],
                         Positions=np.zeros((n, 3)), Energy=float(g["energy"][i]),
                         Forces=g["forces_concat"][3 * ao:3 * (ao + n)].reshape(n, 3), Stress=g["stress"][i],
                         eweight=float(g["eweight"][i]), fweight=float(g["fweight"][i]), vweight=float(g["vweight"][i]),
                         test_bool=int(g["test_bool"][i])))
        ro += nr * ncols
        ao += n
    calc.allocate_per_config(data)
    calc.create_a()
    calc.shared_index = 0
    ao = 0
    for i, d in enumerate(data):
        n = d["NumAtoms"]
        fake_lammps.CURRENT.update(raw=raws[i], vol=float(g["vols"][i]), types=g["types_concat"][ao:ao + n].astype(np.int32),
                                   ids=1 + np.arange(n), pos=np.zeros((n, 3)))
        calc.process_configs(d, i)
        ao += n
    calc.collect_distributed_lists()
    assert np.array_equal(pt.shared_arrays["a"].array, g["A"])
    assert np.array_equal(pt.shared_arrays["b"].array, g["b"])
    assert np.array_equal(pt.shared_arrays["w"].array, g["w"])
    assert pt.fitsnap_dict["Row_Type"] == [str(x) for x in g["Row_Type"]]
    assert pt.fitsnap_dict["Atom_I"] == [int(x) for x in g["Atom_I"]]
    if name.endswith("_nan"):
        assert "applying np.nan_to_num()" in capsys.readouterr().out        # lammps_pace.py:399-403
    pt.free()


# ---- f3 in its literal form: assembly fused into the accumulation (fsnap_assemble_accumulate) ------------------------

def _two_step_accumulate(ctx2, plan_args, total_ptr):
    """fsnap_rows_alloc + fsnap_assemble + fsnap_normal_eq_accumulate: the rows go through HBM."""
    raw, src_row, kind, frac, d, truth, weight, fractions, blank2J, ntypes, ncoeff, offcol = plan_args
    ctx2.rows_alloc(len(src_row), ntypes * (ncoeff + offcol))
    ctx2.assemble(raw, 0, src_row, kind, frac, d, truth, weight, fractions, blank2J, ntypes, ncoeff, offcol)
    ctx2.normal_eq_accumulate(total_ptr)


def _synthetic_batch(rng, nconf, natoms, ntypes, ncoeff, offcol, bik=False):
    """Raw `compute snap` blocks + row plans of `nconf` configurations in the shapes LAMMPS hands over
    (bik rows | 3 N force rows | 6 virial rows; last column = reference potential)."""
    from fitsnap_amd.calculators.row_plan import config_row_plan
    raws, plans, fracs = [], [], []
    row0 = 0
    for ic in range(nconf):
        n = int(natoms[ic % len(natoms)])
        nraw = (n if bik else 1) + 3 * n + 6
        raw = rng.standard_normal((nraw, ntypes * ncoeff + 1)) * rng.uniform(0.1, 30.0)
        types = rng.integers(1, ntypes + 1, n).astype(np.int32)
        plan, _ = config_row_plan(n, types, rng.uniform(50.0, 500.0), rng.standard_normal() * n,
                                  rng.standard_normal((n, 3)), rng.standard_normal((3, 3)), rng.uniform(10.0, 200.0),
                                  rng.uniform(0.5, 2.0), rng.uniform(1e-8, 1e-6), True, True, True, bik, row0,
                                  ic if offcol else -1)
        raws.append(raw)
        plans.append(plan)
        fr = np.bincount(types - 1, minlength=ntypes) / n
        fracs.append(fr)
        row0 += nraw
    plan = {k: np.concatenate([p[k] for p in plans]) for k in plans[0]}
    fractions = np.array(fracs) if offcol else np.zeros((0, ntypes))
    blank2J = np.ones(ntypes * (ncoeff + offcol))
    blank2J[rng.random(len(blank2J)) < 0.1] = 0.0
    return (np.concatenate(raws, axis=0), plan["src_row"], plan["kind"], plan["frac"], plan["d"], plan["truth"],
            plan["weight"], fractions, blank2J, ntypes, ncoeff, offcol)


@pytest.mark.parametrize("shape", [
    dict(nconf=3, natoms=[7, 12, 5], ntypes=1, ncoeff=30, offcol=1),                 # K = 31, one superblock
    dict(nconf=40, natoms=[54, 31, 64], ntypes=2, ncoeff=55, offcol=0),              # K = 110: two superblocks, ragged
    dict(nconf=25, natoms=[40, 64], ntypes=1, ncoeff=128, offcol=0),                 # K = 128: two full superblocks
    dict(nconf=30, natoms=[100, 87], ntypes=2, ncoeff=90, offcol=1, bik=True),       # K = 182 > 128: tiled is the default
    dict(nconf=6, natoms=[16], ntypes=1, ncoeff=299, offcol=1),                      # K = 300, few rows per wave
])
def test_fused_assembly_accumulation_is_bitwise_the_two_step_path(shape):
    # fsnap_assemble_accumulate vs fsnap_assemble into resident rows + fsnap_normal_eq_accumulate on the tiled kernel:
    # identical bits in G, c and the scalars, batch after batch; and the sums are the oracle's
    import torch
    from fitsnap_amd import _capi
    rng = np.random.default_rng(4100 + shape["ncoeff"])
    K = shape["ntypes"] * (shape["ncoeff"] + shape["offcol"])
    dev = torch.device("cuda", 0)
    fused = torch.zeros(K * K + K + 3, dtype=torch.float64, device=dev)
    steps = torch.zeros_like(fused)
    c1, c2 = _capi.HipContext(0), _capi.HipContext(0)
    c2.set_option("tiled", 1)
    rows = []
    for batch in range(3):
        args = _synthetic_batch(rng, **shape)
        c1.assemble_accumulate(*args, fused.data_ptr())
        _two_step_accumulate(c2, args, steps.data_ptr())
        rows.append(c2.download_rows())
        torch.cuda.synchronize()
        assert np.array_equal(fused.cpu().numpy(), steps.cpu().numpy()), f"batch {batch}"
    A = np.concatenate([r[0] for r in rows])
    b = np.concatenate([r[1] for r in rows])
    w = np.concatenate([r[2] for r in rows])
    G, c, s = c1.download_packed(fused.data_ptr(), K)
    Go, co, so = orc.normal_eq(A, b, w)
    scale = np.sqrt(np.outer(np.diag(Go), np.diag(Go))) + 1e-300
    assert np.max(np.abs(G - Go) / scale) < 1e-12
    assert np.max(np.abs(c - co)) <= 1e-12 * np.sqrt(np.max(np.diag(Go)) * so[0])
    assert np.allclose(s, so, rtol=1e-12, atol=0.0)
    c1.close()
    c2.close()


@pytest.mark.parametrize("name", sorted(CASES))
def test_accumulate_single_matches_process_single_statistics_bitwise(name):
    # plugin level: calc.accumulate_single(data, i, ptr) == process_single + `c += aw.T @ aw; d += aw.T @ bw`
    # (transpose_trick/example.py:230-237) on the reference's own golden configurations
    import torch
    from fitsnap_amd import _capi
    g = CASES[name]
    pt, cfg, calc = build(g)
    data = data_dicts(g)
    K = calc.get_width()
    dev = torch.device("cuda", 0)
    fused = torch.zeros(K * K + K + 3, dtype=torch.float64, device=dev)
    steps = torch.zeros_like(fused)
    c2 = _capi.HipContext(0)
    c2.set_option("tiled", 1)
    nrows = 0
    Gn, cn = np.zeros((K, K)), np.zeros(K)
    for i, c in enumerate(g["configs"]):
        serve(c)
        nrows += calc.accumulate_single(data[i], i, fused.data_ptr())
        serve(c)
        a, b, w = calc.process_single(data[i], i)
        c2.upload_rows(a, b)
        c2.set_weights(w)
        c2.normal_eq_accumulate(steps.data_ptr())
        aw, bw = w[:, None] * a, w * b
        Gn += aw.T @ aw
        cn += aw.T @ bw
    torch.cuda.synchronize()
    assert nrows == len(g["b"])
    assert np.array_equal(fused.cpu().numpy(), steps.cpu().numpy())
    G, cc, s = pt.hip().download_packed(fused.data_ptr(), K)
    scale = np.sqrt(np.outer(np.diag(Gn), np.diag(Gn))) + 1e-300
    assert np.max(np.abs(G - Gn) / scale) < 1e-12
    assert np.max(np.abs(cc - cn)) <= 1e-12 * max(np.max(np.abs(cn)), 1e-300) + 1e-12 * np.sqrt(np.max(np.diag(Gn)) * s[0])
    assert s[2] == nrows                                  # every row of the loop is a training row
    c2.close()
    pt.free()


def test_assemble_accumulate_rejects_bad_plans():
    import torch
    from fitsnap_amd import _capi
    rng = np.random.default_rng(5)
    args = list(_synthetic_batch(rng, nconf=1, natoms=[4], ntypes=1, ncoeff=5, offcol=0))
    K = 5
    total = torch.zeros(K * K + K + 3, dtype=torch.float64, device=torch.device("cuda", 0))
    ctx = _capi.HipContext(0)
    bad = list(args)
    bad[1] = args[1].copy()
    bad[1][0] = len(args[0])                     # source row past the raw block
    with pytest.raises(ValueError, match="plan entry 0 out of range"):
        ctx.assemble_accumulate(*bad, total.data_ptr())
    with pytest.raises(ValueError, match="d_packed is NULL"):
        ctx.assemble_accumulate(*args, 0)         # no destination
    assert not total.cpu().numpy().any()
    ctx.close()
