#!/usr/bin/env python
"""Two (or N) ranks on ONE GPU through the peer-to-peer transport (csrc/fsnap_p2p.cpp): what an all-reduce of a fit's payload
costs, and the per-fit numbers of bench.py's multi-rank step.  Writes a text record (profiles/r06_two_ranks_one_gpu.txt).

    python scripts/two_ranks_one_gpu.py [--ranks 2] [--out FILE]

Both ranks share the device, so kernel times are NOT those of a two-GPU job (the ranks' SYRK kernels compete for the CUs);
the collective's latency and the code path are what this measures."""
import argparse
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--out", default=None)
    ns = ap.parse_args()
    lines = [f"{ns.ranks} ranks on one MI355X, peer-to-peer transport (hipIpc windows, one launch per all-reduce)", ""]
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for rank in range(ns.ranks):
            env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(ns.ranks), LOCAL_RANK="0", FSNAP_COMM_FILE=os.path.join(tmp, "id"),
                       FSNAP_COMM_TOKEN="two ranks one gpu", HSA_ENABLE_IPC_MODE_LEGACY="0", FSNAP_COMM_TIMEOUT="60")
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "p2p_worker.py"), tmp, "latency"], env=env))
        for p in procs:
            p.wait(timeout=600)
        lines.append("bare all-reduce of a fit's payload, back to back (wall time of 200 calls / 200, per rank):")
        for K in (128, 480, 1595):
            per = [float(np.load(os.path.join(tmp, f"rank{r}.npz"))[f"us_per_allreduce_{K}"]) for r in range(ns.ranks)]
            n = int(np.load(os.path.join(tmp, "rank0.npz"))[f"doubles_{K}"])
            lines.append(f"  K = {K:5d}: {n:8d} doubles ({n * 8 / 1024:8.1f} KiB)   " + "  ".join(f"rank {r}: {u:7.1f} us" for r, u in enumerate(per)))
    lines += ["", "bench.py --gpus N --transport p2p (strong scaling: the N = 1 rows split over the ranks; both ranks on device 0):"]
    for K, rows in ((128, 1000000), (480, 367900), (1595, 15213)):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ns.ranks), "--transport", "p2p", "--cols", str(K), "--rows", str(rows),
               "--steps", "20", "--warmup", "5", "--preheat", "50", "--scaling", "strong"]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        if out.returncode != 0 or not out.stdout.strip():
            lines.append(f"  {rows} x {K}: FAILED rc={out.returncode}: {out.stderr[-400:]}")
            continue
        rec = json.loads(out.stdout.strip().splitlines()[-1])
        pr = rec["per_rank"]
        lines.append(f"  {rows:8d} x {K:5d}: {rec['ms_per_step']:.4f} ms per fit, transport {rec.get('transport')}, ranks per device "
                     f"{rec.get('ranks_per_device')}, n_ranks_seen {rec['n_ranks_seen']}; per rank kernel_ms {pr['kernel_ms']} allreduce_ms {pr['allreduce_ms']}")
    text = "\n".join(lines) + "\n"
    sys.stdout.write(text)
    if ns.out:
        with open(ns.out, "w") as f:
            f.write(text)


if __name__ == "__main__":
    main()
