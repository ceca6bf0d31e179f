#!/usr/bin/env python
"""bench.py — BASELINE.json metric: training rows/sec through A^T A + solve on a synthetic
10^6 x 128 fp64 A-matrix per GPU (configs[1]: RIDGE normal equations), 1/2/4/8 MI355X.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one complete fit of the resident rows: fused mask x weight x fp64-MFMA normal
equations on every GPU, (N > 1) RCCL all-reduce of the packed K x K statistics, K x K ridge
solve by rank 0 through fsnap_solve_device (K = 128: Jacobi-scaled Cholesky on the host from the
page-locked mirror the reduction kernel wrote; K >= 768: blocked Cholesky on the GPU) -> beta on
the host.  A, b, w are resident in HBM before the timed region (the PCIe-inclusive rate is
reported separately, never as `value`).

Steady state: an MI355X that has been idle needs ~35 ms of sustained load before its clocks
settle (scripts/ramptest.py: SYRK kernel 1.2 ms on the very first steps, 0.35 ms after 10 ms,
0.293 ms from ~35 ms on).  Before the W warm-up steps the benchmark therefore runs `--preheat`
(default 300) additional UNTIMED steps of the same workload -- a fixed count, so that every rank
of a multi-GPU run executes the same number of collectives.  The timed region is still exactly
K steps between barrier + synchronize.
Weak scaling: every rank owns 10^6 rows of its own (disjoint synthetic row blocks);
value = N * rows_per_gpu * steps / max-over-ranks wall time.

Rank 0 prints ONE JSON line (see the bench contract in the task statement) with two extra
objects: `roofline` (fp64-MFMA roofline of the SYRK kernel, measured live with HIP events on
the kernel's stream) and `cpu_baseline` (the oracle's restatement of the reference's numpy
path timed on this box's host cores; N = 1 only; a reported baseline, not the target).
"""
from __future__ import annotations

import argparse
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
    ap.add_argument("--host-solve", action="store_true", help="D2H via torch + fsnap_solve on the host (A/B)")
    ap.add_argument("--force-dist", action="store_true",
                    help="diagnostics: run the multi-GPU step (dedicated stream, RCCL all-reduce, solve from HBM) in a "
                         "process group of ONE rank, to measure its fixed overhead against the single-GPU step")
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


def main():
    args = parse()
    # exactly ONE line on stdout: libraries underneath (RCCL prints a version banner through C stdio, which surfaces
    # at exit, after everything Python printed) get stderr as their fd 1; the JSON line goes to the real stdout
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    from fitsnap_amd import _capi
    from fitsnap_amd.synthetic import synth_problem   # input data; oracle/ is imported by the cpu_baseline leg only

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a gfx950 GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    m, Kc = args.rows, args.cols
    A, b, w = synth_problem(m, Kc, row_offset=rank * RANK_ROW_STRIDE)

    ctx = _capi.HipContext(local_rank)
    for kv in args.option:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    ctx.upload_rows(A, b)
    ctx.set_weights(w)
    upload_ms = ctx.timing()["upload_ms"]
    info = ctx.launch_info()
    n = Kc * Kc + Kc + 3
    packed = torch.zeros(n, dtype=torch.float64, device=dev)
    host = torch.zeros(n, dtype=torch.float64).pin_memory()
    if multi:
        # kernels and the RCCL all-reduce share ONE non-default stream (the legacy default stream synchronises
        # implicitly with every other stream and costs several microseconds per launch)
        stream = torch.cuda.Stream(dev)
        torch.cuda.set_stream(stream)
        ctx.set_stream(stream.cuda_stream)
        torch.cuda.synchronize()                        # buffers above were created on the default stream
    else:
        stream = None                                   # single GPU: the context's own non-blocking stream

    brk = {"launch": 0.0, "sync": 0.0, "solve": 0.0}

    def step():
        t0 = time.perf_counter()
        if not multi and not args.host_solve:
            # the Solver classes' single-GPU path: statistics into a context-owned buffer + solve, one library call
            beta, _, _, _ = ctx.fit_resident(_capi.SOLVE_RIDGE, ALPHA)
            t1 = time.perf_counter()
            brk["launch"] += t1 - t0
            return beta
        else:
            ptr = packed.data_ptr()
            ctx.normal_eq_async(ptr)
            if multi:
                dist.all_reduce(packed)                   # RCCL over xGMI, same stream
                if rank == 0 and not args.host_solve:
                    ctx.mirror_packed(ptr, Kc)            # reduced statistics -> page-locked mirror (no D2H copy)
        t1 = time.perf_counter()
        beta = None
        if args.host_solve:
            if stream is None:
                ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
            host.copy_(packed, non_blocking=True)
            torch.cuda.current_stream(dev).synchronize()
            t2 = time.perf_counter()
            if rank == 0:
                h = host.numpy()
                beta, _, _ = _capi.solve(_capi.SOLVE_RIDGE, ALPHA, h[:Kc * Kc].reshape(Kc, Kc), h[Kc * Kc:Kc * Kc + Kc])
        else:
            # fsnap_solve_device: statistics in HBM -> beta (K <= 128: host factorisation of the page-locked mirror the
            # reduction kernel wrote, or of a D2H copy in the multi-GPU path; K >= 768: blocked Cholesky on the GPU)
            t2 = t1
            if rank == 0:
                beta, _, _ = ctx.solve_device(_capi.SOLVE_RIDGE, ALPHA, Kc, ptr)
            else:
                torch.cuda.current_stream(dev).synchronize()
        t3 = time.perf_counter()
        brk["launch"] += t1 - t0
        brk["sync"] += t2 - t1
        brk["solve"] += t3 - t2
        return beta

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(0, args.preheat)):
        step()
    for _ in range(args.warmup):
        step()
    fence()
    syrk_ms, red_ms = [], []
    for k in brk:
        brk[k] = 0.0
    t0 = time.perf_counter()
    beta = None
    for _ in range(args.steps):
        beta = step()
    fence()
    elapsed = time.perf_counter() - t0
    # kernel times of the timed steps: HIP events recorded on the kernel's stream around every launch, read now
    nh = min(args.steps, 256)
    syrk_hist, red_hist = ctx.timing_history(nh)
    syrk_ms, red_ms = list(syrk_hist), list(red_hist)
    if multi:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        kk = torch.tensor([float(np.mean(syrk_ms))], dtype=torch.float64, device=dev)
        dist.all_reduce(kk, op=dist.ReduceOp.MAX)
        syrk_avg_ms = float(kk.item())
    else:
        syrk_avg_ms = float(np.mean(syrk_ms))

    # Software-pipelined variant, measured OUTSIDE the timed region and reported next to `value` (never as `value`):
    # independent fits (a generation of re-weighting candidates, reference libmod_optimize.py:461-488) need not wait
    # for each other -- the host solves fit i while the kernel of fit i + 1 runs.  Two statistics buffers, the D2H
    # copy on a second stream.
    pipe = None
    if rank == 0 and not multi and not args.host_solve and Kc < 384 and args.steps >= 4:
        try:
            s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
            ctx.set_stream(s1.cuda_stream)
            bufs = [torch.zeros(n, dtype=torch.float64, device=dev) for _ in range(2)]
            hosts = [torch.zeros(n, dtype=torch.float64).pin_memory() for _ in range(2)]
            ev_k = [torch.cuda.Event() for _ in range(2)]
            ev_c = [torch.cuda.Event() for _ in range(2)]
            torch.cuda.synchronize()

            def launch(i):
                j = i & 1
                if i >= 2:
                    s1.wait_event(ev_c[j])                       # the copy of fit i - 2 has left this buffer
                ctx.normal_eq_async(bufs[j].data_ptr())
                ev_k[j].record(s1)
                s2.wait_event(ev_k[j])
                with torch.cuda.stream(s2):
                    hosts[j].copy_(bufs[j], non_blocking=True)
                ev_c[j].record(s2)

            pbrk = {"launch": 0.0, "wait": 0.0, "solve": 0.0}

            def run(nst):
                launch(0)
                bt = None
                for i in range(nst):
                    q0 = time.perf_counter()
                    if i + 1 < nst:
                        launch(i + 1)
                    q1 = time.perf_counter()
                    ev_c[i & 1].synchronize()
                    q2 = time.perf_counter()
                    h = hosts[i & 1].numpy()
                    bt, _, _ = _capi.solve(_capi.SOLVE_RIDGE, ALPHA, h[:Kc * Kc].reshape(Kc, Kc), h[Kc * Kc:Kc * Kc + Kc])
                    q3 = time.perf_counter()
                    pbrk["launch"] += q1 - q0
                    pbrk["wait"] += q2 - q1
                    pbrk["solve"] += q3 - q2
                torch.cuda.synchronize()
                return bt

            run(max(20, args.warmup))
            for kq in pbrk:
                pbrk[kq] = 0.0
            tp0 = time.perf_counter()
            bp = run(args.steps)
            tp = time.perf_counter() - tp0
            ph_s, ph_r = ctx.timing_history(min(args.steps, 256))
            pipe = {"rows_per_s": m * args.steps / tp, "ms_per_step": tp / args.steps * 1e3,
                    "kernel_ms_avg": float(np.mean(ph_s)), "reduce_kernel_ms_avg": float(np.mean(ph_r)),
                    "host_launch_ms_avg": pbrk["launch"] / args.steps * 1e3, "host_wait_ms_avg": pbrk["wait"] / args.steps * 1e3,
                    "host_solve_ms_avg": pbrk["solve"] / args.steps * 1e3,
                    "max_rel_diff_vs_sequential": float(np.max(np.abs(bp - beta)) / np.max(np.abs(beta))),
                    "what": "host solve of fit i overlapped with the kernel of fit i+1 (independent fits); not the headline"}
            ctx.use_own_stream()
        except Exception as e:  # pragma: no cover
            pipe = {"error": str(e)}

    # stand-alone row-weighting kernel (north_star: achieved HBM GB/s), measured outside the timed region
    wk = None
    if rank == 0 and world == 1:
        try:
            d_aw = torch.empty((m, Kc), dtype=torch.float64, device=dev)
            d_bw = torch.empty(m, dtype=torch.float64, device=dev)
            wms = []
            for i in range(6):
                ctx.weight_rows_device(d_aw.data_ptr(), Kc, d_bw.data_ptr())
                wms.append(ctx.timing()["weight_ms"])
            wms = float(np.mean(wms[1:]))
            wbytes = (16 * Kc + 24) * m                      # read A, b, w; write aw, bw
            wk = {"kernel": "fsnap_weight_rows_k", "bound": "hbm", "ms": wms, "achieved": wbytes / (wms * 1e-3) / 1e9,
                  "peak": 8000.0, "unit": "GB/s", "frac": wbytes / (wms * 1e-3) / 1e9 / 8000.0,
                  "algorithmic_bytes_per_launch": wbytes}
            del d_aw, d_bw
        except Exception as e:  # pragma: no cover
            wk = {"error": str(e)}

    if rank == 0:
        total_rows = world * m
        flops_per_launch = (Kc * Kc + 3 * Kc) * m          # K^2 + 3K flop/row x rows per launch
        achieved = flops_per_launch / (syrk_avg_ms * 1e-3) / 1e12
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile) and m == ROWS_PER_GPU and Kc == K:
            try:
                traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        if info["split"] == 0:
            kernel_name = "fsnap_syrk_tiled"
        elif info["kernel_or_pairs"] == 4:
            kernel_name = f"fsnap_syrk_wave_p<{info['NB']}>"
        elif info["kernel_or_pairs"] == 3:
            kernel_name = f"fsnap_syrk_acc<{info['NB']}>"
        elif info["kernel_or_pairs"] == 2:
            kernel_name = f"fsnap_syrk_lds_static<{info['NB']},{info['threads'] // 64}>" if dict(kv.split("=") for kv in args.option).get("kernel", "0") in ("0", "2", "4") \
                else f"fsnap_syrk_lds<{info['NB']},{info['threads'] // 64}>"
        else:
            kernel_name = f"fsnap_syrk_wave<{info['NB']},{info['split']}>"
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
                "parallelism": f"dp{world}: rows sharded by rank, one RCCL all-reduce of {n} doubles per fit",
                "launch": info,
            },
            "roofline": {
                "bound": "mfma", "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic,
                "kernel": kernel_name, "kernel_ms_avg": syrk_avg_ms, "reduce_kernel_ms_avg": float(np.mean(red_ms)),
                "flops_per_launch": flops_per_launch, "algorithmic_bytes_per_launch": (8 * Kc + 16) * m,
                "achieved_GBps_algorithmic": (8 * Kc + 16) * m / (syrk_avg_ms * 1e-3) / 1e9,
            },
            "step_host_launch_ms_avg": brk["launch"] / args.steps * 1e3,
            "step_wait_gpu_ms_avg": brk["sync"] / args.steps * 1e3,
            "step_host_solve_ms_avg": brk["solve"] / args.steps * 1e3,
            "weighting_kernel": wk,
            "pipelined": pipe,
            "h2d_upload_ms": upload_ms,
            "h2d_inclusive_rows_per_s": m / ((upload_ms + elapsed / args.steps * 1e3) * 1e-3),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(A, b, w, beta)
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    ctx.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
