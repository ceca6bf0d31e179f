"""Worker of tests/test_gpu_p2p.py: one rank of a peer-to-peer communicator (csrc/fsnap_p2p.cpp) through the C ABI alone.
Launched with RANK / WORLD_SIZE / FSNAP_COMM_FILE / FSNAP_COMM_TOKEN in the environment; all ranks share device 0 (hipIpc
handles open between processes of one device -- the configuration RCCL refuses).  argv: <outdir> <scenario>."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rank_data(rank, n):
    return np.random.default_rng(1000 + rank).standard_normal(n) * 10.0 ** np.random.default_rng(7).integers(-8, 8, n)


def main(outdir, scenario):
    from fitsnap_amd import _capi, rendezvous

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ctx = _capi.HipContext(0)
    ctx.comm_init(world, rank, rendezvous.exchange(rank, world, lambda: _capi.comm_id("p2p")))
    rendezvous.done(rank)
    assert ctx.comm_transport() == "p2p" and ctx.comm_info() == (world, rank)
    out = {}
    if scenario == "collectives":
        # device all-reduce: odd and even lengths, one element, the K = 128 payload, and one that travels in pieces
        # (FSNAP_P2P_SLOT_MB = 1 in the environment: 131 072 doubles per piece)
        for n in (1, 2, 3, 16515, 131072, 300001):
            mine = rank_data(rank, n)
            d = ctx.dev_alloc(n * 8)
            ctx.dev_upload(d, mine)
            for rep in range(3):                # back to back: the double-buffered window is reused from the third call on
                ctx.allreduce_device(d, n)
            got = np.empty(n)
            ctx.sync()
            ctx.dev_download(d, got)
            out[f"dev_{n}"] = got
            ctx.dev_free(d)
        v = np.array([rank + 1.0, -rank, 0.5])
        for op, name in ((0, "sum"), (1, "max"), (2, "min")):
            w = v.copy()
            ctx.allreduce_host(w, op)
            out[f"host_{name}"] = w
        big = rank_data(rank, 700001)            # larger than a mailbox (FSNAP_P2P_MAILBOX_MB = 1)
        ctx.allreduce_host(big)
        out["host_big"] = big
        blob = bytes([rank]) * 5 + b"tail"
        out["gather"] = np.frombuffer(b"".join(ctx.allgather_bytes(blob, world)), dtype=np.uint8)
        big_blob = (bytes([65 + rank]) * 1500001)
        parts = ctx.allgather_bytes(big_blob, world)
        out["gather_big_ok"] = np.array([parts[q] == bytes([65 + q]) * 1500001 for q in range(world)])
        msg = b"from the last rank" if rank == world - 1 else None
        out["bcast"] = np.frombuffer(ctx.bcast_bytes(msg, 18, world - 1), dtype=np.uint8)
        for _ in range(50):
            ctx.barrier()
    elif scenario == "latency":
        # what one all-reduce of a fit's payload costs: HIP events are not exposed for bare collectives, so wall time of a
        # back-to-back train, per call (the kernels of the ranks overlap; one train = 200 calls)
        for K in (128, 480, 1595):
            n = K * K + K + 3 if K < 256 else K * (K + 1) // 2 + K + 3
            d = ctx.dev_alloc(n * 8)
            ctx.dev_upload(d, np.zeros(n))
            for _ in range(20):
                ctx.allreduce_device(d, n)
            ctx.sync()
            ctx.barrier()
            t0 = time.perf_counter()
            for _ in range(200):
                ctx.allreduce_device(d, n)
            ctx.sync()
            out[f"us_per_allreduce_{K}"] = np.array((time.perf_counter() - t0) / 200 * 1e6)
            out[f"doubles_{K}"] = np.array(n)
            ctx.dev_free(d)
            ctx.barrier()
    elif scenario.startswith("reupload"):
        # a job that fits one set of rows, then a LARGER one on the same contexts (bench.py: strong then weak scaling): every
        # buffer of the context is re-allocated between two series of collectives
        from fitsnap_amd.synthetic import synth_problem

        K = 128
        for m in (120000, 260000):
            A, b, w = synth_problem(m, K, row_offset=rank * 16 * 65536)
            ctx.upload_rows(A, b)
            ctx.set_weights(w)
            for _ in range(25):
                beta = ctx.fit_dist(_capi.SOLVE_RIDGE, 1e-8, K)[0]
            ctx.barrier()
            out[f"beta_{m}"] = beta
            t = np.zeros((world, 2))
            t[rank] = (m, rank)
            ctx.allreduce_host(t.reshape(-1))
            out[f"table_{m}"] = t
    elif scenario == "dead_peer":
        # rank 1 leaves before the collective: rank 0's all-reduce must come back with an error inside the bound
        # (FSNAP_COMM_TIMEOUT = 4 in the environment), not hang -- the wait INSIDE the kernel is bounded too
        ctx.barrier()
        if rank == 1:
            np.savez(os.path.join(outdir, f"rank{rank}.npz"), left=np.array(1))
            os._exit(0)
        d = ctx.dev_alloc(16515 * 8)
        ctx.dev_upload(d, np.ones(16515))
        t0 = time.perf_counter()
        ctx.allreduce_device(d, 16515)
        try:
            ctx.barrier()
            out["error"] = np.array("")
        except _capi.FsnapError as e:
            out["error"] = np.array(str(e))
        out["seconds"] = np.array(time.perf_counter() - t0)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
    if scenario != "dead_peer":
        ctx.barrier()
    ctx.close()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
