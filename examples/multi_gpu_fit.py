"""Data-parallel fit over several MI355X, one process per GPU, native RCCL behind the C ABI (no torch in the processes):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        examples/multi_gpu_fit.py [--rows-per-gpu 1000000] [--cols 128] [--solver RIDGE|SVD] [--transport rccl|p2p]

(the launcher is only used to start the processes and set RANK / LOCAL_RANK / WORLD_SIZE.)  The reference's form of this
algorithm is examples/library/transpose_trick/example.py:230-254: every MPI rank accumulates aw.T aw and aw.T bw over ITS
configurations, then comm.Allreduce(c), comm.Allreduce(d) and one solve.  Here every rank keeps the rows of its
configurations in the HBM of its GPU; ``solver.perform_fit`` is collective: fused statistics kernel, ONE in-place
all-reduce of K^2 + K + 3 doubles on the kernels' stream (``ncclAllReduce``, or with ``--transport p2p`` one launch of the
peer-to-peer transport over hipIpc windows, which also puts several ranks on ONE GPU), K x K solve; ``solver.fit`` ends up on
rank 0, like the reference.  ``error_analysis`` pools per-group error sums (a fixed-size table per rank), no row ever moves."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows-per-gpu", type=int, default=1_000_000)
    ap.add_argument("--cols", type=int, default=128)
    ap.add_argument("--solver", default="RIDGE", choices=["RIDGE", "SVD"])
    ap.add_argument("--transport", default=None, choices=["rccl", "p2p"])
    args = ap.parse_args()
    from fitsnap_amd.config import Config
    from fitsnap_amd.parallel_tools import ParallelTools
    from fitsnap_amd.solvers import solver_factory
    from fitsnap_amd.synthetic import synth_problem

    pt = ParallelTools(comm="rccl" if int(os.environ.get("WORLD_SIZE", "1")) > 1 else None, transport=args.transport)
    rank, world = pt.get_rank(), pt.get_size()
    m, K = args.rows_per_gpu, args.cols
    # this rank's "configurations": a disjoint block of the synthetic rows (a real run fills pt.shared_arrays from LAMMPS)
    A, b, w = synth_problem(m, K, row_offset=rank * 16 * 65536)
    groups = [f"g{(i // 1000) % 7}" for i in range(m)]
    fsd = {"Testing": (np.arange(m) % 10 == 9).tolist(), "Groups": groups, "Row_Type": ["Energy"] * m}
    settings = {"SOLVER": {"solver": args.solver}}
    if args.solver == "RIDGE":
        settings["RIDGE"] = {"alpha": 1e-8}
    solver = solver_factory.solver(args.solver, pt, Config(pt, settings))
    solver.keep_resident = True
    train = ~np.asarray(fsd["Testing"])
    solver.perform_fit(A, b, w[train], fs_dict=fsd)                 # first call: upload + fit
    pt.all_barrier()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        solver.perform_fit(A, b, w[train], fs_dict=fsd)             # rows resident: weights + fit
    pt.all_barrier()
    dt = (time.perf_counter() - t0) / reps
    solver.error_analysis(A, b, w, fsd)
    if rank == 0:
        print(f"{world} GPU(s), {world * m} rows x {K}: {dt * 1e3:.3f} ms per fit ({world * m / dt:.3e} rows/s), "
              f"|fit| = {np.linalg.norm(solver.fit):.6e}, torch imported: {'torch' in sys.modules}")
        print(solver.errors.head(6).to_string())
    pt.free()


if __name__ == "__main__":
    main()
