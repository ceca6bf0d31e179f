"""``python -m fitsnap3 <infile> [--descriptors DIR] [--overwrite] [--nofit] ...`` — entry point
with the reference's control flow (fitsnap3/__main__.py:44-57): FitSnap -> process_configs ->
all_barrier -> perform_fit -> write_output, errors routed through ``pt.exception``.

``--descriptors DIR`` (an addition of this build) skips the LAMMPS stage and ingests the
reference's dump files from DIR; without it a ``lammps`` Python module is required."""
from __future__ import annotations

import argparse
import sys


def main(argv=None):
    ap = argparse.ArgumentParser(prog="fitsnap3")
    ap.add_argument("infile")
    ap.add_argument("--descriptors", default=None, metavar="DIR",
                    help="ingest Descriptors.npy / Truth-Ref.npy / Weights.npy (/ FitSNAP.df) from DIR instead of running LAMMPS")
    ap.add_argument("--overwrite", action="store_true")
    ap.add_argument("--nofit", action="store_true")
    ap.add_argument("--verbose", "-v", action="store_true")
    ap.add_argument("--relative", "-r", action="store_true")
    ns = ap.parse_args(argv)
    from .fitsnap import FitSnap

    arglist = [f for f, on in (("--overwrite", ns.overwrite), ("--nofit", ns.nofit), ("--verbose", ns.verbose),
                               ("--relative", ns.relative)) if on]
    fs = FitSnap(ns.infile, comm=None, arglist=arglist)
    try:
        if ns.descriptors is not None:
            fs.load_descriptors(ns.descriptors)
        else:
            raise RuntimeError("the descriptor stage needs LAMMPS (outside this build): pass --descriptors DIR "
                               "with the reference's dumped Descriptors.npy / Truth-Ref.npy / Weights.npy")
        fs.pt.all_barrier()
        fs.perform_fit()
        fs.write_output()
    except Exception as e:
        fs.pt.exception(e)
    return 0


if __name__ == "__main__":
    sys.exit(main())
