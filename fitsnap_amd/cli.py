"""``python -m fitsnap3 <infile> [--descriptors DIR] [--overwrite] [--nofit] ...`` — entry point
with the reference's control flow (fitsnap3/__main__.py:44-57): FitSnap -> process_configs ->
all_barrier -> perform_fit -> write_output, errors routed through ``pt.exception``.

``--descriptors DIR`` (an addition of this build) skips the LAMMPS stage and ingests the
reference's dump files from DIR; without it a ``lammps`` Python module is required.

Like the reference's entry point, which takes ``MPI.COMM_WORLD`` when mpi4py is there (fitsnap3/__main__.py:34-41), this
one joins the job it was launched in by itself: under ``python -m torch.distributed.run --nproc-per-node N -m fitsnap3 ...``
(``WORLD_SIZE`` > 1) every process opens the native RCCL communicator on ITS GPU (``LOCAL_RANK``), keeps the rows of its
configurations only (``FitSnap.load_descriptors``), and rank 0 writes the potential.  ``--comm`` overrides the choice:
``none`` (N independent single-process fits -- never what a launcher wants), ``rccl``, or ``torch`` (a
``torch.distributed`` group, gloo on a machine without GPUs: the CPU tests)."""
from __future__ import annotations

import argparse
import os
import sys


def pick_comm(choice="auto"):
    """The ``comm`` argument of ``FitSnap`` / ``ParallelTools`` for this process (see module docstring)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if choice == "none" or (choice == "auto" and world <= 1):
        return None
    if choice in ("auto", "rccl"):
        return "rccl"
    import torch.distributed as dist

    if not dist.is_initialized():
        import torch

        dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo")      # env:// rendezvous of the launcher
    return "torch"


def main(argv=None):
    ap = argparse.ArgumentParser(prog="fitsnap3")
    ap.add_argument("infile")
    ap.add_argument("--descriptors", default=None, metavar="DIR",
                    help="ingest Descriptors.npy / Truth-Ref.npy / Weights.npy (/ FitSNAP.df) from DIR instead of running LAMMPS")
    ap.add_argument("--overwrite", action="store_true")
    ap.add_argument("--nofit", action="store_true")
    ap.add_argument("--verbose", "-v", action="store_true")
    ap.add_argument("--relative", "-r", action="store_true")
    ap.add_argument("--comm", default="auto", choices=["auto", "none", "rccl", "torch"],
                    help="communicator of a multi-process launch (auto: native RCCL when WORLD_SIZE > 1)")
    ns = ap.parse_args(argv)
    from .fitsnap import FitSnap

    arglist = [f for f, on in (("--overwrite", ns.overwrite), ("--nofit", ns.nofit), ("--verbose", ns.verbose),
                               ("--relative", ns.relative)) if on]
    fs = FitSnap(ns.infile, comm=pick_comm(ns.comm), arglist=arglist)
    try:
        if ns.descriptors is not None:
            fs.load_descriptors(ns.descriptors)
        else:
            raise RuntimeError("the descriptor stage needs LAMMPS (outside this build): pass --descriptors DIR "
                               "with the reference's dumped Descriptors.npy / Truth-Ref.npy / Weights.npy")
        fs.pt.all_barrier()
        fs.perform_fit()
        fs.write_output()
        fs.pt.all_barrier()                 # nobody leaves (and tears the communicator down) while rank 0 still writes
    except Exception as e:
        fs.pt.exception(e)
    return 0


if __name__ == "__main__":
    sys.exit(main())
