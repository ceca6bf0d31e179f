#!/usr/bin/env python
"""bench.py — BASELINE.json metric: training rows/sec through A^T A + solve on a synthetic
10^6 x 128 fp64 A-matrix per GPU (configs[1]: RIDGE normal equations), 1/2/4/8 MI355X.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

The launcher only provides RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*: this script imports neither torch nor
torch.distributed.  Everything on the GPU goes through the C ABI (include/fsnap_hip.h) via ctypes, including the
multi-GPU exchange (native RCCL: fsnap_comm_*, fsnap_fit_dist).

One "step" = one complete fit of the resident rows, ALL of it inside the timed region: packing of the per-row
weights (mask x w, mask x w x b and the three b-only scalars; option repack = 1 forces it on every fit although b, w
and the mask do not change between steps), fused mask x weight x fp64-MFMA normal equations, partial reduction,
(N > 1) one in-place ncclAllReduce of the packed K x K statistics on the same stream, K x K ridge solve through
fsnap_solve_device (K = 128: Jacobi-scaled Cholesky on the host from the page-locked mirror; K >= 384: blocked
Cholesky on the GPU) -> beta on the host of every rank.  A, b, w are resident in HBM before the timed region (the
PCIe-inclusive rate is reported separately, never as `value`).

Steady state: an MI355X that has been idle needs ~35 ms of sustained load before its clocks settle
(scripts/ramptest.py).  Before the W warm-up steps the benchmark therefore runs `--preheat` (default 300) additional
UNTIMED steps of the same workload -- a fixed count, so that every rank executes the same number of collectives.
The timed region is still exactly K steps between barrier + stream synchronisation on both sides.
Weak scaling: every rank owns 10^6 rows of its own (disjoint synthetic row blocks);
value = N * rows_per_gpu * steps / max-over-ranks wall time.

Rank 0 prints ONE JSON line (see the bench contract in the task statement) with two extra objects: `roofline`
(fp64-MFMA roofline of the SYRK kernel -- HBM roofline when K <= 80, where the kernel is bandwidth-bound -- measured
live with HIP events on the kernel's stream around every `--timing-every`-th launch of the timed region, the first
included: an event record between two dependent kernels idles the stream for ~5.6 us on this runtime, two per step =
10.4 us of a 0.33 ms step, so the instrument samples instead of bracketing every launch; `--timing-every 1` is the
old behaviour) and `cpu_baseline` (the oracle's restatement of the reference's numpy
path timed on this box's host cores; N = 1 only; a reported baseline, not the target).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROWS_PER_GPU = 1_000_000
K = 128
ALPHA = 1.0e-8                 # reference default, io/sections/solver_sections/ridge.py:13
PEAK_FP64_MFMA_TFLOPS = 78.6   # MI355X vendor fp64 matrix peak
PEAK_HBM_GBPS = 8000.0         # MI355X HBM3E (MI355X_MICROARCH.md)
RANK_ROW_STRIDE = 16 * 65536   # >= ROWS_PER_GPU, multiple of the generator's 64 Ki-row chunk


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--preheat", type=int, default=300,
                    help="untimed steps before the warm-up, to bring the GPU to its steady clock (0 = none)")
    ap.add_argument("--rows", type=int, default=ROWS_PER_GPU, help="rows per GPU (default: BASELINE config)")
    ap.add_argument("--cols", type=int, default=K)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-repack", action="store_true",
                    help="diagnostics: pack the per-row weights once instead of on every fit (NOT the headline)")
    ap.add_argument("--force-dist", action="store_true",
                    help="diagnostics: run the multi-GPU step (RCCL all-reduce, fsnap_fit_dist) in a communicator of ONE "
                         "rank, to measure its fixed overhead against the single-GPU step")
    ap.add_argument("--timing-every", type=int, default=8,
                    help="HIP events bracket every N-th kernel launch of the timed region (an event record between two "
                         "dependent kernels idles the stream ~5.6 us; 1 = every launch)")
    ap.add_argument("--option", action="append", default=[], help="kernel option key=value (split, nontemporal, nblocks)")
    return ap.parse_args()


def cpu_baseline(A, b, w, beta_gpu):
    """Reference algorithm (oracle restatement) on the host cores, bounded sample."""
    from oracle import fitsnap_oracle as orc

    m = len(b)
    try:
        from threadpoolctl import threadpool_info
        pools = [(p.get("internal_api"), p.get("num_threads")) for p in threadpool_info()]
        threads = max([p[1] for p in pools] + [1])
    except Exception:
        pools, threads = [], os.cpu_count() or 1
    # RIDGE path of the reference (ridge.py:37-59): weighting + normal equations + Cholesky, all rows
    t0 = time.perf_counter()
    beta = orc.ridge_fit(A, b, w, ALPHA)
    t_ridge = time.perf_counter() - t0
    # SVD path (svd.py:44-54, lstsq/gelsd) on all rows of the headline workload (~2-4 s on the GPU box's host)
    ms = min(m, 1_000_000)
    t0 = time.perf_counter()
    orc.svd_fit(A[:ms], b[:ms], w[:ms])
    t_svd = time.perf_counter() - t0
    rel = float(np.max(np.abs(beta_gpu - beta) / np.maximum(np.abs(beta), 1e-300)))
    return {
        "value": m / t_ridge, "unit": "rows/s", "cores": int(threads), "kind": "port",
        "sample": f"oracle ridge_fit (weight + X^T X + Cholesky, reference ridge.py:37-59) on all {m} rows: "
                  f"{t_ridge:.2f} s; oracle svd_fit (lstsq, svd.py:54) on {ms} rows: {t_svd:.2f} s",
        "svd_lstsq_rows_per_s": ms / t_svd, "host_cpu_count": os.cpu_count(), "blas": pools,
        "gpu_vs_oracle_max_rel_err": rel,
    }


def kernel_source_digest():
    """sha256 of the SYRK kernel sources: ties a recorded PMC traffic number to the code it was measured on."""
    h = hashlib.sha256()
    for name in ("fsnap_syrk.hip", "fsnap_device_common.h"):
        with open(os.path.join(ROOT, "fitsnap_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def recorded_traffic(m, Kc, info, kernel_name):
    """HBM bytes per launch of the dominant kernel from the PMC passes (scripts/pmc_traffic.py ->
    profiles/pmc_traffic.json) -- only if they were collected for THIS kernel source, shape and launch geometry;
    otherwise null (a regressed or re-tuned kernel must be re-measured, not inherit an old number)."""
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(tfile))
    except Exception:
        return None, "no profiles/pmc_traffic.json"
    want = {"rows": m, "K": Kc, "kernel": kernel_name, "workgroups": info["workgroups"], "threads": info["threads"],
            "chunks_per_wave": info["chunks_per_wave"], "source_sha256": kernel_source_digest()}
    for k, v in want.items():
        if rec.get(k) != v:
            return None, f"profiles/pmc_traffic.json was recorded for {k} = {rec.get(k)!r}, this run has {v!r}"
    return rec.get("hbm_bytes_per_launch"), rec.get("source")


def main():
    args = parse()
    # exactly ONE line on stdout: libraries underneath (RCCL prints a version banner through C stdio, which surfaces
    # at exit, after everything Python printed) get stderr as their fd 1; the JSON line goes to the real stdout
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    from fitsnap_amd import _capi, rendezvous
    from fitsnap_amd.synthetic import synth_problem   # input data; oracle/ is imported by the cpu_baseline leg only

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if _capi.device_count() < 1:
        raise SystemExit("bench.py needs a gfx950 GPU (no CPU fallback)")
    multi = world > 1 or args.force_dist

    m, Kc = args.rows, args.cols
    A, b, w = synth_problem(m, Kc, row_offset=rank * RANK_ROW_STRIDE)

    ctx = _capi.HipContext(local_rank % _capi.device_count())
    if multi:
        ctx.comm_init(world, rank, rendezvous.exchange(rank, world, _capi.comm_id))     # native RCCL, no torch
        rendezvous.done(rank)
    for kv in args.option:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    repack = not args.no_repack
    ctx.set_option("repack", 1 if repack else 0)
    ctx.upload_rows(A, b)
    ctx.set_weights(w)
    upload_ms = ctx.timing()["upload_ms"]
    info = ctx.launch_info()
    n = Kc * Kc + Kc + 3

    def step():
        if multi:
            return ctx.fit_dist(_capi.SOLVE_RIDGE, ALPHA, Kc)[0]     # kernel -> in-place ncclAllReduce -> solve, every rank
        return ctx.fit_resident(_capi.SOLVE_RIDGE, ALPHA)[0]         # the Solver classes' single-GPU path

    def fence():
        if multi:
            ctx.barrier()          # all ranks here + this rank's stream idle
        else:
            ctx.sync()

    ctx.set_option("timing_every", 0)
    for _ in range(max(0, args.preheat)):
        step()
    for _ in range(args.warmup):
        step()
    fence()
    # kernel timing of the timed region: HIP events on the kernel's stream around every N-th launch (always the first)
    every = max(1, args.timing_every)
    sampled0 = ctx.timing_count()[0]
    ctx.set_option("timing_every", every)             # the next launch (timed step 0) is a sampled one
    t0 = time.perf_counter()
    beta = None
    for _ in range(args.steps):
        beta = step()
    fence()
    elapsed = time.perf_counter() - t0
    # kernel times of the timed steps: HIP events recorded on the kernel's stream around every launch, read now
    nh = min(ctx.timing_count()[0] - sampled0, 256)
    syrk_hist, red_hist = ctx.timing_history(nh)
    syrk_avg_ms, red_avg_ms = float(np.mean(syrk_hist)), float(np.mean(red_hist))
    if multi:
        mx = np.array([elapsed, syrk_avg_ms])
        ctx.allreduce_host(mx, _capi.REDUCE_MAX)
        elapsed, syrk_avg_ms = float(mx[0]), float(mx[1])

    # stand-alone row-weighting kernel (north_star: achieved HBM GB/s), measured outside the timed region
    wk = None
    if rank == 0 and world == 1:
        try:
            d_aw = ctx.dev_alloc(m * Kc * 8)
            d_bw = ctx.dev_alloc(m * 8)
            wms = []
            for i in range(6):
                ctx.weight_rows_device(d_aw, Kc, d_bw)
                wms.append(ctx.timing()["weight_ms"])
            wms = float(np.mean(wms[1:]))
            wbytes = (16 * Kc + 24) * m                      # read A, b, w; write aw, bw
            wk = {"kernel": "fsnap_weight_rows_k", "bound": "hbm", "ms": wms, "achieved": wbytes / (wms * 1e-3) / 1e9,
                  "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": wbytes / (wms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                  "algorithmic_bytes_per_launch": wbytes}
            ctx.dev_free(d_aw)
            ctx.dev_free(d_bw)
        except Exception as e:  # pragma: no cover
            wk = {"error": str(e)}

    if rank == 0:
        total_rows = world * m
        flops_per_launch = (Kc * Kc + 3 * Kc) * m          # K^2 + 3K flop/row x rows per launch
        bytes_per_launch = (8 * Kc + 16) * m               # A row + b + w per row, A read once
        if info["split"] == 0:
            kernel_name = "fsnap_syrk_tiled"
        elif info["kernel_or_pairs"] == 4:
            kernel_name = f"fsnap_syrk_wave_p<{info['NB']}>"
        elif info["kernel_or_pairs"] == 3:
            kernel_name = f"fsnap_syrk_acc<{info['NB']}>"
        elif info["kernel_or_pairs"] == 2:
            kernel_name = f"fsnap_syrk_lds_static<{info['NB']},{info['threads'] // 64}>"
        else:
            kernel_name = f"fsnap_syrk_wave<{info['NB']},{info['split']}>"
        traffic, traffic_source = recorded_traffic(m, Kc, info, kernel_name)
        tf = flops_per_launch / (syrk_avg_ms * 1e-3) / 1e12
        gbs = bytes_per_launch / (syrk_avg_ms * 1e-3) / 1e9
        if Kc <= 80:
            # (K^2 + 3K) / (8K + 16) flop per byte is below the machine balance (~9.8): the bytes bound the kernel
            roofline = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBPS,
                        "achieved_TFLOPs_algorithmic": tf}
        else:
            roofline = {"bound": "mfma", "achieved": tf, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": tf / PEAK_FP64_MFMA_TFLOPS, "achieved_GBps_algorithmic": gbs}
        # what the matrix pipe executes: every 16 x 16 tile of the block triangle in full (the diagonal tiles' lower
        # halves are redundant), 2 * 16 * 16 flop per row and tile; kernels 1A / 1P / 1 / 1L only
        executed = None
        if info["split"] != 0:
            executed = info["NB"] * (info["NB"] + 1) // 2 * 512 * m
        roofline.update({"executed_mfma_flops_per_launch": executed,
                         "executed_over_algorithmic": (executed / flops_per_launch) if executed else None})
        roofline.update({"traffic": traffic, "traffic_source": traffic_source, "kernel": kernel_name,
                         "kernel_ms_avg": syrk_avg_ms, "reduce_kernel_ms_avg": red_avg_ms,
                         "kernel_timing": f"HIP events on the kernel's stream around every {every}. launch of the timed "
                                          f"region ({nh} of {args.steps} launches)",
                         "flops_per_launch": flops_per_launch, "algorithmic_bytes_per_launch": bytes_per_launch})
        out = {
            "metric": "training rows/sec through A^T A + solve, 10^6 x 128 fp64 per GPU",
            "value": total_rows * args.steps / elapsed,
            "unit": "rows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "preheat_steps": max(0, args.preheat),
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"synthetic {m} x {Kc} fp64 A per GPU (SURVEY 8d generator), RIDGE alpha=1e-8 "
                            "normal equations (BASELINE configs[1]), A/b/w resident in HBM",
                "rows_per_gpu": m, "K": Kc, "solver": "RIDGE",
                "parallelism": f"dp{world}: rows sharded by rank, one in-place ncclAllReduce of {n} doubles per fit "
                               "(native RCCL behind the C ABI, no torch), solve on every rank",
                "weights_packed_every_step": repack,
                "launch": info,
            },
            "roofline": roofline,
            "weighting_kernel": wk,
            "h2d_upload_ms": upload_ms,
            "h2d_inclusive_rows_per_s": m / ((upload_ms + elapsed / args.steps * 1e3) * 1e-3),
            "torch_imported": "torch" in sys.modules,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(A, b, w, beta)
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    if multi:
        ctx.barrier()
    ctx.close()


if __name__ == "__main__":
    main()
