"""One rank of `python -m torch.distributed.run --nproc-per-node 2 tests/cli_dist_worker.py <outdir> <fitsnap3 argv...>`:
the drop-in entry point (`fitsnap_amd.cli.main`, what `python -m fitsnap3` runs) inside a launcher, on a box WITHOUT a
GPU.  The only thing replaced is the part the HIP kernels compute on a rank's OWN rows -- local statistics and local
predictions, supplied by the checker (oracle/), as in tests/test_dist_gloo.py; communicator pick-up, row sharding by
configuration, the collective fit, the pooled error analysis and rank 0's file output are the product code.  Every rank
writes what it held to <outdir>/shard<rank>.json."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    outdir, argv = sys.argv[1], sys.argv[2:]
    from fitsnap_amd import cli
    from fitsnap_amd import fitsnap as fitsnap_mod
    from fitsnap_amd.solvers.solver import Solver
    from oracle import fitsnap_oracle as orc

    Solver._local_statistics = lambda self, a, b, wf, mask, shared: orc.normal_eq(a, b, wf, testing=(mask == 0))

    def predict_rows(self, a=None, b=None):
        a = self.pt.shared_arrays["a"].array if a is None else np.asarray(a)
        return a @ np.asarray(self.fit, dtype=np.float64).reshape(-1)

    Solver.predict_rows = predict_rows
    seen = {}
    real_load = fitsnap_mod.FitSnap.load_descriptors

    def load(self, directory=".", shard=True):
        got = real_load(self, directory, shard)
        seen.update(rows=int(got[0]), total=int(len(self.row_owner)), rank=self.pt.get_rank(), size=self.pt.get_size(),
                    comm_kind=self.pt.comm_kind, owner_runs=int(np.count_nonzero(np.diff(self.row_owner)) + 1),
                    local_testing=len(self.pt.local_lists.get("Testing", [])),
                    global_testing=len(self.pt.fitsnap_dict["Testing"]))
        return got

    fitsnap_mod.FitSnap.load_descriptors = load
    rc = cli.main(argv)
    with open(os.path.join(outdir, f"shard{seen['rank']}.json"), "w") as f:
        json.dump(seen, f)
    import torch.distributed as dist

    if dist.is_initialized():
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
