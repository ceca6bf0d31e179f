"""CPU oracle for the post-LAMMPS A-matrix assembly.  TEST INFRASTRUCTURE ONLY (see the
header of fitsnap_oracle.py for the import rules).

Restates ``LammpsSnap._collect_lammps`` (fitsnap3lib/calculators/lammps_snap.py:391-556) in
numpy: per configuration the LAMMPS ``compute snap`` global array of shape
``(bik_rows + 3N + 6, ncoeff*ntypes + 1)`` is cut into energy / force / virial row blocks,
scaled, given the per-type offset columns and masked by ``blank2J``; the last column is the
reference potential that is subtracted from the truths.

Pinned bit-exactly (``tests/test_oracle_assembly.py``) to outputs of the reference's own
class driven by a fake ``lammps`` object (``tests/golden/make_golden_assembly.py`` ->
``tests/golden/assembly_reference.npz``), for bzeroflag in {0,1}, 1 and 2 atom types, mixed
2J (blank2J zeros), energy/force/stress toggles, bikflag and quadraticflag.
"""
from __future__ import annotations

import numpy as np

VOIGT = ([0, 1, 2, 1, 0, 0], [0, 1, 2, 2, 2, 1])   # xx yy zz yz xz xy  (lammps_snap.py:539-540)
STRESS_UNIT = 1.6021765e6                            # lammps_snap.py:526


def snap_config_rows(raw, natoms, atom_type_ids, vol, energy, forces, stress, eweight, fweight, vweight,
                     ntypes, ncoeff, bzeroflag, blank2J, use_energy=True, use_force=True, use_stress=True,
                     bikflag=False):
    """Rows of (A, b, w, Row_Type, Atom_I, Atom_Type) for ONE configuration.

    ``atom_type_ids``: 1-based LAMMPS type of every atom (type_mapping applied)."""
    raw = np.asarray(raw, dtype=np.float64)
    ncols_bis = ncoeff * ntypes
    icolref = ncols_bis
    bik_rows = natoms if bikflag else 1
    off = 0 if bzeroflag else 1
    K = ntypes * (ncoeff + off)
    A, b, w, rt, ai, at = [], [], [], [], [], []

    def widen(block, onehot):
        # insert the per-type offset column (lammps_snap.py:455-464, 495-499, 528-533)
        if not off:
            return block.reshape(block.shape[0], K)
        blk = block.reshape(block.shape[0], ntypes, ncoeff)
        return np.concatenate([onehot, blk], axis=2).reshape(block.shape[0], K)

    irow = 0
    if use_energy:
        bsum = raw[irow:irow + bik_rows, :ncols_bis] / natoms                       # :435
        if off:
            if bikflag:
                raise NotImplementedError("per atom energy is not implemented without bzeroflag")   # :457
            frac = np.zeros((1, ntypes, 1))
            for t in atom_type_ids:
                frac[0, t - 1, 0] += 1
            frac /= len(atom_type_ids)                                                # :459-462
            bsum = widen(bsum, frac)
        A.append(bsum * blank2J[np.newaxis, :])                                       # :467-468
        bb = np.zeros(bik_rows)
        bb[0] = (energy - raw[irow, icolref]) / natoms                                # :470-474
        b.append(bb)
        ww = np.zeros(bik_rows)          # the reference leaves bik rows 1.. unwritten (garbage); zero here
        ww[0] = eweight                                                               # :477
        w.append(ww)
        rt += ["Energy"] * bik_rows
        ai += list(range(bik_rows))
        at += [int(t) for t in atom_type_ids] if bikflag else [0]
    irow += bik_rows
    nf = 3 * natoms
    if use_force:
        blk = raw[irow:irow + nf, :ncols_bis]
        blk = widen(blk, np.zeros((nf, ntypes, 1)))
        A.append(np.matmul(blk, np.diag(blank2J)))                                    # :501-502
        b.append(np.asarray(forces, dtype=np.float64).ravel() - raw[irow:irow + nf, icolref])   # :504-507
        w.append(np.full(nf, fweight))
        rt += ["Force"] * nf
        ai += [int(np.floor(i / 3)) for i in range(nf)]
        at += [int(t) for t in atom_type_ids for _ in range(3)]
    irow += nf
    if use_stress:
        blk = STRESS_UNIT * raw[irow:irow + 6, :ncols_bis] / vol                      # :526
        blk = widen(blk, np.zeros((6, ntypes, 1)))
        A.append(np.matmul(blk, np.diag(blank2J)))
        b.append(np.asarray(stress)[VOIGT[0], VOIGT[1]].ravel() - raw[irow:irow + 6, icolref])
        w.append(np.full(6, vweight))
        rt += ["Stress"] * 6
        ai += [0] * 6
        at += [0] * 6
    if not A:
        return np.zeros((0, K)), np.zeros(0), np.zeros(0), [], [], []
    return np.concatenate(A), np.concatenate(b), np.concatenate(w), rt, ai, at
