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
    ap.add_argument("--transport", default=None, choices=["rccl", "p2p"],
                    help="exchange step of the native communicator: RCCL (default) or the one-shot peer-to-peer all-reduce over "
                         "hipIpc windows (one node; ranks may share a device).  Same as FSNAP_DIST_TRANSPORT")
    ns = ap.parse_args(argv)
    if ns.transport:
        os.environ["FSNAP_DIST_TRANSPORT"] = ns.transport      # rank 0 creates the id; the transport travels with it
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
        # only rank 0 writes.  Nobody leaves (and tears the communicator down) while it still does -- and when it FAILS
        # there (an existing file without --overwrite), every rank hears of it in the same collective instead of sitting in a
        # barrier until FSNAP_COMM_TIMEOUT
        failure = None
        try:
            fs.write_output()
        except Exception as e:              # noqa: BLE001 -- re-raised below, on every rank
            failure = e
        if fs.pt.multi:
            said = fs.pt.bcast_object(None if failure is None or fs.pt.get_rank() != 0 else f"{type(failure).__name__}: {failure}", src=0)
            if failure is None and said is not None:
                failure = RuntimeError(f"rank 0 failed while writing the output: {said}")
        if failure is not None:
            raise failure
    except Exception as e:
        fs.pt.exception(e)
    return 0


if __name__ == "__main__":
    sys.exit(main())
