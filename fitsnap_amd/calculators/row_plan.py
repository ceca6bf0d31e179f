"""Row plan of one configuration for the device-side assembly kernel (fsnap_assemble):
the bookkeeping half of ``_collect_lammps`` (fitsnap3lib/calculators/lammps_snap.py:391-556,
lammps_pace.py:369-509).  The arithmetic half (scaling, offset columns, blank2J mask,
reference-potential subtraction) is done by the kernel from this plan."""
from __future__ import annotations

import numpy as np

KIND_ENERGY, KIND_FORCE, KIND_STRESS, KIND_BIK_EXTRA = 0, 1, 2, 3
VOIGT = ([0, 1, 2, 1, 0, 0], [0, 1, 2, 2, 2, 1])      # xx yy zz yz xz xy (lammps_snap.py:539-540)


def config_row_plan(natoms, type_ids, vol, energy, forces, stress, eweight, fweight, vweight, use_energy, use_force,
                    use_stress, bikflag, raw_row0, frac_index, with_atom_type=True):
    """Returns (plan, meta) for one configuration.

    plan = dict(src_row, kind, frac, d, truth, weight) — numpy arrays, one entry per output row;
    meta = dict(Row_Type, Atom_I, Atom_Type) — Python lists like the reference's DistributedLists.
    ``raw_row0`` is the row of this configuration's raw block in the batch; ``frac_index`` the
    row of its atom-type fractions (or -1 when there is no offset column)."""
    bik_rows = natoms if bikflag else 1
    src, kind, frac, d, truth, weight = [], [], [], [], [], []
    rt, ai, at = [], [], []
    irow = 0
    if use_energy:
        src.append(raw_row0 + irow + np.arange(bik_rows))
        k = np.full(bik_rows, KIND_BIK_EXTRA, dtype=np.int32)
        k[0] = KIND_ENERGY
        kind.append(k)
        f = np.full(bik_rows, -1, dtype=np.int32)
        f[0] = frac_index
        frac.append(f)
        d.append(np.full(bik_rows, float(natoms)))
        t = np.zeros(bik_rows)
        t[0] = energy
        truth.append(t)
        w = np.zeros(bik_rows)
        w[0] = eweight
        weight.append(w)
        rt += ["Energy"] * bik_rows
        ai += [int(i) for i in range(bik_rows)]
        at += [int(i) for i in type_ids] if bikflag else [0]
    irow += bik_rows
    nf = 3 * natoms
    if use_force:
        src.append(raw_row0 + irow + np.arange(nf))
        kind.append(np.full(nf, KIND_FORCE, dtype=np.int32))
        frac.append(np.full(nf, -1, dtype=np.int32))
        d.append(np.ones(nf))
        truth.append(np.asarray(forces, dtype=np.float64).ravel())
        weight.append(np.full(nf, float(fweight)))
        rt += ["Force"] * nf
        ai += [int(np.floor(i / 3)) for i in range(nf)]
        at += [int(t_) for t_ in type_ids for _ in range(3)]
    irow += nf
    if use_stress:
        src.append(raw_row0 + irow + np.arange(6))
        kind.append(np.full(6, KIND_STRESS, dtype=np.int32))
        frac.append(np.full(6, -1, dtype=np.int32))
        d.append(np.full(6, float(vol)))
        truth.append(np.asarray(stress, dtype=np.float64)[VOIGT[0], VOIGT[1]].ravel())
        weight.append(np.full(6, float(vweight)))
        rt += ["Stress"] * 6
        ai += [0] * 6
        at += [0] * 6

    def cat(parts, dtype):
        return np.concatenate(parts).astype(dtype) if parts else np.zeros(0, dtype=dtype)

    plan = dict(src_row=cat(src, np.int64), kind=cat(kind, np.int32), frac=cat(frac, np.int32), d=cat(d, np.float64),
                truth=cat(truth, np.float64), weight=cat(weight, np.float64))
    meta = dict(Row_Type=rt, Atom_I=ai)
    if with_atom_type:
        meta["Atom_Type"] = at
    return plan, meta


def type_fractions(atom_types, type_mapping, ntypes):
    """Per-type atom fractions of a configuration (lammps_snap.py:459-462)."""
    f = np.zeros(ntypes)
    for a in atom_types:
        f[type_mapping[a] - 1] += 1
    return f / len(atom_types)
