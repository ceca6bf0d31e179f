#!/usr/bin/env python
"""Fused assembly + accumulation (fsnap_assemble_accumulate) against the two-step path (fsnap_assemble into resident
rows, fsnap_normal_eq_accumulate) on synthetic LAMMPS batches: wall time per batch (host staging included, both calls
are synchronous) -- run under `rocprofv3 --kernel-trace --stats` for the kernel times.
usage: fused_assembly_timing.py [nconf natoms ntypes ncoeff offcol]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from fitsnap_amd import _capi                                   # noqa: E402
from fitsnap_amd.calculators.row_plan import config_row_plan    # noqa: E402


def batch(rng, nconf, natoms, ntypes, ncoeff, offcol):
    raws, plans, fracs, row0 = [], [], [], 0
    for ic in range(nconf):
        nraw = 1 + 3 * natoms + 6
        types = rng.integers(1, ntypes + 1, natoms).astype(np.int32)
        plan, _ = config_row_plan(natoms, types, 300.0, -5.0 * natoms, rng.standard_normal((natoms, 3)),
                                  rng.standard_normal((3, 3)), 100.0, 1.0, 1e-7, True, True, True, False, row0,
                                  ic if offcol else -1)
        raws.append(rng.standard_normal((nraw, ntypes * ncoeff + 1)))
        plans.append(plan)
        fracs.append(np.bincount(types - 1, minlength=ntypes) / natoms)
        row0 += nraw
    plan = {k: np.concatenate([p[k] for p in plans]) for k in plans[0]}
    return (np.concatenate(raws, axis=0), plan["src_row"], plan["kind"], plan["frac"], plan["d"], plan["truth"],
            plan["weight"], np.array(fracs) if offcol else np.zeros((0, ntypes)), np.ones(ntypes * (ncoeff + offcol)),
            ntypes, ncoeff, offcol)


def main():
    nconf, natoms, ntypes, ncoeff, offcol = (int(x) for x in (sys.argv[1:6] if len(sys.argv) >= 6 else (160, 64, 2, 55, 0)))
    rng = np.random.default_rng(1)
    args = batch(rng, nconf, natoms, ntypes, ncoeff, offcol)
    n, K = len(args[1]), ntypes * (ncoeff + offcol)
    dev = torch.device("cuda", 0)
    tot_f = torch.zeros(K * K + K + 3, dtype=torch.float64, device=dev)
    tot_s = torch.zeros_like(tot_f)
    c1, c2 = _capi.HipContext(0), _capi.HipContext(0)
    c2.set_option("tiled", 1)
    c2.rows_alloc(n, K)
    reps = 20
    for it in range(3):
        c1.assemble_accumulate(*args, tot_f.data_ptr())
        c2.assemble(args[0], 0, *args[1:])
        c2.normal_eq_accumulate(tot_s.data_ptr())
    torch.cuda.synchronize()
    same = np.array_equal(tot_f.cpu().numpy(), tot_s.cpu().numpy())
    t0 = time.perf_counter()
    for it in range(reps):
        c1.assemble_accumulate(*args, tot_f.data_ptr())
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for it in range(reps):
        c2.assemble(args[0], 0, *args[1:])
        c2.normal_eq_accumulate(tot_s.data_ptr())
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{nconf} configurations x {natoms} atoms: {n} rows x {K}, raw {args[0].nbytes / 1e6:.1f} MB; "
          f"fused {1e3 * (t1 - t0) / reps:.3f} ms per batch, two-step {1e3 * (t2 - t1) / reps:.3f} ms per batch "
          f"(H2D staging of the raw arrays included in both); bit-identical sums: {same}")
    c1.close()
    c2.close()


if __name__ == "__main__":
    main()
