#!/usr/bin/env python
"""bench.py — BASELINE.json metric: training rows/sec through A^T A + solve on a synthetic 10^6 x 128 fp64 A-matrix
(configs[1]: RIDGE normal equations) at 1 / 2 / 4 / 8 MI355X.

    python bench.py --gpus N --steps K --warmup W            # N > 1: this script starts the N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU.  Under a launcher (RANK / LOCAL_RANK / WORLD_SIZE in the environment) the process IS one rank;
without one, `--gpus N` > 1 makes this process the launcher: it starts N ranks of itself (one per visible device, a
private communicator-id file and job token in the environment), passes rank 0's JSON line through, and turns the
first failing rank into ONE line on stderr and a non-zero exit status for the whole job.  The script imports neither
torch nor torch.distributed: everything on the GPU goes through the C ABI (include/fsnap_hip.h) via ctypes, including
the multi-GPU exchange (native RCCL: fsnap_comm_*, fsnap_fit_dist).

One "step" = one complete fit of the rows resident on the GPUs, ALL of it inside the timed region: packing of the
per-row weights (mask x w, mask x w x b and the three b-only scalars; option repack = 1 forces it on every fit although
b, w and the mask do not change between steps), fused mask x weight x fp64-MFMA normal equations, partial reduction,
(N > 1) one in-place ncclAllReduce of the packed K x K statistics on the same stream, K x K ridge solve through
fsnap_solve_device (K = 128: Jacobi-scaled Cholesky on the host from the page-locked mirror; K >= 384: blocked
Cholesky on the GPU) -> beta on the host of every rank.  A, b, w are resident in HBM before the timed region (the
PCIe-inclusive rate is reported separately, never as `value`).

Two scalings of the same metric (`--scaling strong|weak|both`, default both):
* strong -- BASELINE.json's literal metric "10^6 x 128 ... at 1/2/4/8 GPU": the SAME 10^6 rows, row-sharded N ways
  (rank r owns rows [r 10^6 / N, (r+1) 10^6 / N) of the N = 1 problem, so every N fits the identical system);
  `value` = 10^6 * steps / max-over-ranks wall time, `"scaling": "strong"`;
* weak -- 10^6 rows PER GPU (disjoint synthetic row blocks): `weak_value` = N * 10^6 * steps / max-over-ranks time.
At N = 1 the two are the same run.  With `--scaling weak` the line's `value` is the weak number and `"scaling": "weak"`.

Steady state: an MI355X that has been idle needs ~35 ms of sustained load before its clocks settle
(scripts/ramptest.py).  Before the W warm-up steps every mode therefore runs `--preheat` (default 300) additional
UNTIMED steps of the same workload -- a fixed count, so that every rank executes the same number of collectives.
The timed region is exactly K steps between barrier + stream synchronisation on both sides.

Rank 0 prints ONE JSON line (see the bench contract in the task statement) with extra objects: `roofline`
(fp64-MFMA roofline of the SYRK kernel -- HBM roofline when K <= 80, where the kernel is bandwidth-bound -- measured
live with HIP events on the kernel's stream around every `--timing-every`-th launch of the timed region, the first
included: an event record between two dependent kernels idles the stream for ~5.6 us on this runtime, so the
instrument samples instead of bracketing every launch), `per_rank` (kernel_ms and allreduce_ms of every rank: HIP
events around the SYRK kernel and around the collective on each rank's own stream), `n_ranks_seen` (fsnap_comm_info),
`transport` (rccl | p2p: `--transport`) and, at N = 1, `cpu_baseline` (the oracle's restatement of the reference's numpy path timed on this
box's host cores; a reported baseline, not the target).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROWS = 1_000_000
K = 128
ALPHA = 1.0e-8                 # reference default, io/sections/solver_sections/ridge.py:13
PEAK_FP64_MFMA_TFLOPS = 78.6   # MI355X vendor fp64 matrix peak
PEAK_HBM_GBPS = 8000.0         # MI355X HBM3E (MI355X_MICROARCH.md)
RANK_ROW_STRIDE = 16 * 65536   # weak scaling: row-block stride between ranks, >= ROWS, multiple of the generator's chunk


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scaling", choices=("strong", "weak", "both"), default="both",
                    help="strong: --rows rows in total, split over the GPUs (BASELINE's metric, `value`); weak: --rows rows "
                         "per GPU (`weak_value`); both (default)")
    ap.add_argument("--preheat", type=int, default=300,
                    help="untimed steps before the warm-up, to bring the GPU to its steady clock (0 = none)")
    ap.add_argument("--rows", type=int, default=ROWS, help="rows of the problem (strong) / per GPU (weak)")
    ap.add_argument("--cols", type=int, default=K)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-repack", action="store_true",
                    help="diagnostics: pack the per-row weights once instead of on every fit (NOT the headline)")
    ap.add_argument("--force-dist", action="store_true",
                    help="diagnostics: run the multi-GPU step (RCCL all-reduce, fsnap_fit_dist) in a communicator of ONE "
                         "rank, to measure its fixed overhead against the single-GPU step")
    ap.add_argument("--pipelined", type=int, default=1,
                    help="1 / 0: also report the throughput with two fits in flight (N = 1 only; an extra object, never `value`)")
    ap.add_argument("--svd-solver", type=int, default=1,
                    help="N = 1: also time the reference's default solver (SVD: probe solve + up to 2 one-pass refinement steps, the "
                         "plugin class, and the row-space path on an ill-conditioned copy); reported as `svd_solver`, never `value`")
    ap.add_argument("--timing-every", type=int, default=4,
                    help="HIP events bracket every N-th kernel launch of the timed region (an event record between two "
                         "dependent kernels idles the stream ~5.6 us; 1 = every launch)")
    ap.add_argument("--transport", choices=["rccl", "p2p"], default=os.environ.get("FSNAP_DIST_TRANSPORT") or "rccl",
                    help="exchange step of a multi-rank run: native RCCL (default) or the one-shot peer-to-peer all-reduce over hipIpc "
                         "windows (one node; ranks may SHARE a device, which RCCL refuses -- how N > 1 runs on a one-GPU box)")
    ap.add_argument("--job-timeout", type=float, default=1800.0, help="launcher: seconds before a hung job is killed")
    ap.add_argument("--option", action="append", default=[], help="library option key=value (include/fsnap_hip.h: nblocks, nsplit, tiled, quad_min_rows, ...)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without a launcher's environment
# ---------------------------------------------------------------------------------------------------------------------
def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args):
    """Start args.gpus ranks of this script (one per device) and relay rank 0's JSON line.  Returns the exit status of
    the job: 0, or the status of the first rank that failed (the others are terminated), or 124 on --job-timeout."""
    n = args.gpus
    workdir = tempfile.mkdtemp(prefix="fsnap_bench_")
    base = dict(os.environ)
    base.update({
        "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
        "MASTER_PORT": base.get("MASTER_PORT") or str(_free_port()),
        "FSNAP_COMM_FILE": os.path.join(workdir, "comm_id"),           # a file this job owns ...
        "FSNAP_COMM_TOKEN": hashlib.sha256(os.urandom(32)).hexdigest(),  # ... and a token only its ranks know
        "FSNAP_BENCH_SPAWNED": "1",
        "HSA_ENABLE_IPC_MODE_LEGACY": base.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
    })
    procs = []
    for rank in range(n):
        env = dict(base, RANK=str(rank), LOCAL_RANK=str(rank))
        # rank 0 inherits stdout (the JSON line); everybody's stderr is the launcher's
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if rank == 0 else subprocess.DEVNULL))
    deadline = time.monotonic() + args.job_timeout
    status = 0
    failed = None
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                rc = procs[r].poll()
                if rc is None:
                    continue
                pending.discard(r)
                if rc != 0 and failed is None:
                    failed, status = r, rc
            if failed is not None and pending:
                # the survivors sit in (bounded) waits for the rank that is gone: give them a moment to report, then stop them
                grace = time.monotonic() + 10.0
                while pending and time.monotonic() < grace:
                    pending = {r for r in pending if procs[r].poll() is None}
                    time.sleep(0.05)
                for r in pending:
                    procs[r].terminate()
                for r in pending:
                    try:
                        procs[r].wait(timeout=10.0)
                    except subprocess.TimeoutExpired:
                        procs[r].kill()
                pending = set()
            if pending and time.monotonic() > deadline:
                failed, status = -1, 124
                for r in pending:
                    procs[r].kill()
                pending = set()
            time.sleep(0.02)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        try:
            for name in os.listdir(workdir):
                os.remove(os.path.join(workdir, name))
            os.rmdir(workdir)
        except OSError:
            pass
    if failed is not None:
        what = f"rank {failed} exited with status {status}" if failed >= 0 else f"no result after {args.job_timeout:.0f} s"
        sys.stderr.write(f"bench.py: {n}-GPU job failed: {what}; all ranks stopped\n")
        return status if status else 1
    return 0


# ---------------------------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------------------------
def cpu_quota_cpus():
    """CPUs of a cgroup quota (cpu.max "quota period" / cfs_quota_us, cfs_period_us), None without one."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = float(f.read())
        return q / p if q > 0 and p > 0 else None
    except Exception:
        return None


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
    # a cgroup CPU quota (the MI355X boxes: cpu.max = 16 CPUs of 256): BLAS threads beyond it only get throttled -- the baseline
    # runs with as many threads as the quota pays for, and `cores` says that number
    quota = cpu_quota_cpus()
    limiter = None
    if quota is not None and quota < threads:
        try:
            from threadpoolctl import threadpool_limits
            limiter = threadpool_limits(limits=max(1, int(quota)))
            threads = max(1, int(quota))
        except Exception:
            limiter = None
    # RIDGE path of the reference (ridge.py:37-59): weighting + normal equations + Cholesky, all rows
    t0 = time.perf_counter()
    beta = orc.ridge_fit(A, b, w, ALPHA)
    t_ridge = time.perf_counter() - t0
    # SVD path (svd.py:44-54, lstsq/gelsd) on all rows of the headline workload (~2-4 s on the GPU box's host)
    ms = min(m, 1_000_000)
    t0 = time.perf_counter()
    orc.svd_fit(A[:ms], b[:ms], w[:ms])
    t_svd = time.perf_counter() - t0
    if limiter is not None:
        limiter.restore_original_limits()
    rel = None if beta_gpu is None else float(np.max(np.abs(beta_gpu - beta) / np.maximum(np.abs(beta), 1e-300)))
    return {
        "value": m / t_ridge, "unit": "rows/s", "cores": int(threads), "kind": "port", "cpu_quota_cpus": quota,
        "sample": f"oracle ridge_fit (weight + X^T X + Cholesky, reference ridge.py:37-59) on all {m} rows: "
                  f"{t_ridge:.2f} s; oracle svd_fit (lstsq, svd.py:54) on {ms} rows: {t_svd:.2f} s",
        "svd_lstsq_rows_per_s": ms / t_svd, "host_cpu_count": os.cpu_count(), "blas": pools,
        "gpu_vs_oracle_max_rel_err": rel,
    }


def kernel_source_digest(kernel_name=""):
    """sha256 of the sources of the named SYRK kernel (fsnap_syrk_quad<...> lives in a file of its own): ties a recorded
    PMC traffic number to the code it was measured on."""
    h = hashlib.sha256()
    main = ("fsnap_syrk_quad.hip" if kernel_name.startswith("fsnap_syrk_quad") else
            "fsnap_syrk_short.hip" if kernel_name.startswith("fsnap_syrk_short") else "fsnap_syrk.hip")
    for name in (main, "fsnap_device_common.h"):
        with open(os.path.join(ROOT, "fitsnap_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def recorded_traffic(m, Kc, info, kernel_name):
    """HBM bytes per launch of the dominant kernel from the PMC passes (scripts/pmc_traffic.py ->
    profiles/pmc_traffic.json: one record, or a list of records for several shapes) -- only if a record was collected
    for THIS kernel source, shape and launch geometry; otherwise null (a regressed or re-tuned kernel must be
    re-measured, not inherit an old number)."""
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        recs = json.load(open(tfile))
    except Exception:
        return None, "no profiles/pmc_traffic.json"
    if isinstance(recs, dict):
        recs = [recs]
    want = {"rows": m, "K": Kc, "kernel": kernel_name, "workgroups": info["workgroups"], "threads": info["threads"],
            "chunks_per_wave": info["chunks_per_wave"], "source_sha256": kernel_source_digest(kernel_name)}
    shapes = ", ".join(f"{r.get('rows')} x {r.get('K')}" for r in recs)
    why = f"profiles/pmc_traffic.json was recorded for {shapes}, this run has {m} x {Kc}"
    for rec in recs:
        miss = [k for k, v in want.items() if rec.get(k) != v]
        if not miss:
            return rec.get("hbm_bytes_per_launch"), rec.get("source")
        if rec.get("rows") == m and rec.get("K") == Kc:
            k = miss[0]
            why = f"profiles/pmc_traffic.json was recorded for {k} = {rec.get(k)!r}, this run has {want[k]!r}"
    return None, why


def kernel_name_of(info):
    if info["split"] == 0:
        return "fsnap_syrk_tiled"
    if info["kernel_or_pairs"] == 7:
        return f"fsnap_syrk_short<{info['NB']}>"
    if info["kernel_or_pairs"] == 6:
        return f"fsnap_syrk_quadc<{info['NB']}>"
    if info["kernel_or_pairs"] == 5:
        return f"fsnap_syrk_quad<{info['NB']}>"
    if info["kernel_or_pairs"] == 4:
        return f"fsnap_syrk_wave_p<{info['NB']}>"
    return f"fsnap_syrk_acc<{info['NB']}>"


def _single_blas_thread():
    """Context manager: BLAS / OpenMP pools limited to one thread (no-op without threadpoolctl)."""
    try:
        from threadpoolctl import threadpool_limits

        return threadpool_limits(limits=1)
    except Exception:  # pragma: no cover
        import contextlib

        return contextlib.nullcontext()


def synth_rows(lo, hi, total, Kc):
    """Rows [lo, hi) of the `total`-row synthetic problem exactly as the one-GPU run generates them (the generator
    works in 64 Ki-row chunks; a partial last chunk draws its noise differently from a full one, so every chunk is
    generated at the length it has in the N = 1 problem and then cut)."""
    from fitsnap_amd.synthetic import SYNTH_CHUNK, synth_chunk, synth_params

    scales, beta_star = synth_params(Kc)
    A = np.empty((hi - lo, Kc))
    b = np.empty(hi - lo)
    w = np.empty(hi - lo)
    for ci in range(lo // SYNTH_CHUNK, (hi + SYNTH_CHUNK - 1) // SYNTH_CHUNK if hi > lo else 0):
        c0 = ci * SYNTH_CHUNK
        rows = min(SYNTH_CHUNK, total - c0)
        Ac, bc, wc = synth_chunk(ci, rows, Kc, beta_star, scales)
        a, z = max(lo, c0), min(hi, c0 + rows)
        A[a - lo:z - lo], b[a - lo:z - lo], w[a - lo:z - lo] = Ac[a - c0:z - c0], bc[a - c0:z - c0], wc[a - c0:z - c0]
    return A, b, w


def run_mode(ctx, args, mode, rank, world, multi, _capi):
    """One scaling mode on this rank: resident rows, pre-heat, warm-up, K timed steps.  Returns a dict (complete on
    every rank: the per-rank numbers are exchanged with one small all-reduce)."""
    from fitsnap_amd.synthetic import synth_problem

    Kc = args.cols
    # The boxes run under a CPU quota: a burst of BLAS threads (the generator's `A @ beta`, 64 OpenBLAS threads that keep
    # spinning after every call) can leave the process throttled -- descheduled for tens of milliseconds -- right when the
    # timed region starts (seen on small shapes, whose pre-heat is short: 1.5 ms per step instead of 0.065).  So the rows are
    # generated on one thread and the process idles for a quota period before it starts issuing fits.
    with _single_blas_thread():
        if mode == "weak":
            m_total = args.rows * world
            A, b, w = synth_problem(args.rows, Kc, row_offset=rank * RANK_ROW_STRIDE)
        else:
            m_total = args.rows
            lo, hi = args.rows * rank // world, args.rows * (rank + 1) // world
            A, b, w = synth_rows(lo, hi, args.rows, Kc)
    m = len(b)
    ctx.upload_rows(A, b)
    ctx.set_weights(w)
    time.sleep(0.25)
    tim = ctx.timing()
    upload_ms = tim["upload_ms"]
    upload_how = {"host_fill_probe_GBps": tim.get("upload_probe_GBps"), "page_locked_double_buffer": tim.get("upload_staged")}
    info = ctx.launch_info()

    def step():
        if multi:
            return ctx.fit_dist(_capi.SOLVE_RIDGE, ALPHA, Kc)[0]     # kernel -> in-place ncclAllReduce -> solve, every rank
        return ctx.fit_resident(_capi.SOLVE_RIDGE, ALPHA)[0]         # the Solver classes' single-GPU path

    def fence():
        if multi:
            ctx.barrier()          # all ranks here + this rank's stream idle
        else:
            ctx.sync()

    def timed():
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
        ctx.set_option("timing_every", 0)
        # kernel times of the timed steps: HIP events recorded on the kernel's stream, read now
        nh = min(ctx.timing_count()[0] - sampled0, 256)
        syrk_hist, red_hist = ctx.timing_history(nh)
        comm_hist = ctx.timing_history_comm(nh) if multi else np.full(nh, -1.0)
        comm_ms = float(np.mean(comm_hist[comm_hist >= 0])) if np.any(comm_hist >= 0) else 0.0
        return elapsed, beta, float(np.mean(syrk_hist)), float(np.mean(red_hist)), comm_ms, nh

    elapsed, beta, syrk_ms, red_ms, comm_ms, nh = timed()
    # per-rank numbers, max-over-ranks wall time
    per_rank = np.zeros((world, 4))
    per_rank[rank] = (elapsed, syrk_ms, comm_ms, float(m))
    if multi:
        ctx.allreduce_host(per_rank.reshape(-1), _capi.REDUCE_SUM)
    out = {
        "mode": mode, "rows_total": m_total, "rows_this_rank": m, "elapsed": float(per_rank[:, 0].max()), "beta": beta,
        "kernel_ms": [float(x) for x in per_rank[:, 1]], "allreduce_ms": [float(x) for x in per_rank[:, 2]],
        "rows_per_rank": [int(x) for x in per_rank[:, 3]], "elapsed_per_rank_s": [float(x) for x in per_rank[:, 0]],
        "reduce_ms": red_ms, "sampled": nh, "info": info, "upload_ms": upload_ms, "upload_how": upload_how, "A": A, "b": b, "w": w,
    }
    return out


def run_pipelined(args, head, dev, _capi):
    """Throughput with TWO fits in flight (outside the headline protocol, reported next to it): two contexts bound to the
    same resident rows, each with its own stream, weights and statistics; the host factorises the statistics of fit i
    while the GPU already runs fit i + 1 -- what a caller with independent candidates (the generation of a genetic
    re-weighting loop, examples/library/genetic_algorithm/libmod_optimize.py:461-488) can do through the C ABI as it is:
    fsnap_normal_eq_resident launches, fsnap_solve_device finishes.  Every fit is complete (pack, statistics, reduction,
    solve) inside the timed region."""
    A, b, w = head["A"], head["b"], head["w"]
    m, Kc = A.shape
    cs = [_capi.HipContext(dev), _capi.HipContext(dev)]
    try:
        d_a = cs[0].dev_alloc(A.nbytes + 256)
        d_b = cs[0].dev_alloc(b.nbytes)
        cs[0].dev_upload(d_a, A)
        cs[0].dev_upload(d_b, b)
        for c in cs:
            c.bind_rows(d_a, m, Kc, Kc, d_b)
            c.set_weights(w)
            c.set_option("timing_every", 0)
            c.set_option("repack", 1)

        def loop(n):
            ptr = [cs[0].normal_eq_resident(), None]
            beta = None
            for i in range(n):
                j = i & 1
                if i + 1 < n:
                    ptr[1 - j] = cs[1 - j].normal_eq_resident()
                beta = cs[j].solve_device(_capi.SOLVE_RIDGE, ALPHA, Kc, ptr[j])[0]
            return beta

        loop(max(20, args.preheat // 2))
        loop(args.warmup + 2)
        for c in cs:
            c.sync()
        t0 = time.perf_counter()
        beta = loop(args.steps)
        for c in cs:
            c.sync()
        elapsed = time.perf_counter() - t0
        same = bool(np.array_equal(beta, head["beta"]))
        cs[0].dev_free(d_a)
        cs[0].dev_free(d_b)
        return {"fits_in_flight": 2, "value": m * args.steps / elapsed, "ms_per_step": elapsed / args.steps * 1e3,
                "same_beta_as_headline": same,
                "protocol": "two contexts on the same resident rows; the host solve of fit i overlaps the kernels of fit i + 1"}
    except Exception as e:  # pragma: no cover - optional leg
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        for c in cs:
            c.close()


def run_svd_solver(ctx, args, head, dev, _capi):
    """The reference's DEFAULT solver (io/sections/solver_sections/solver.py:15 -> solvers/svd.py:54, lstsq on the weighted
    rows) on the same resident rows, outside the headline protocol (never `value`):
      * `steps`: what SVD.perform_fit runs per fit on a well-conditioned system, at the level of the headline step
        (weights resident, one library call per stage): probe solve of the statistics + up to 2 refinement steps (stop rule of Solver._refine), each ONE
        pass over the rows (fsnap_residual_rhs, kernels 4 + 7 fused) + a K x K solve;
      * `class_perform_fit`: the plugin class itself with keep_resident (adds what the class does per call: the staged
        upload of one weight per training row, label handling);
      * `row_space`: the same class on an ill-conditioned copy of the problem (one column nearly dependent on another,
        kappa ~ 1e9): CholeskyQR passes over the rows (fsnap_lstsq_rows)."""
    out = {}
    try:
        from fitsnap_amd.config import Config
        from fitsnap_amd.parallel_tools import ParallelTools
        from fitsnap_amd.solvers import solver_factory

        A, b, w = head["A"], head["b"], head["w"]
        m, Kc = A.shape
        RCOND, NREF = 1.0e-13, 2
        ctx.upload_rows(A, b)
        ctx.set_weights(w)
        ctx.set_option("timing_every", 0)

        from fitsnap_amd.solvers.solver import refinement_done, refinement_skip

        def svd_step():
            # the steps of SVD.perform_fit / Solver._refine, stop rule included: how many refinement passes a fit takes
            # follows the conditioning the Cholesky reports (refinement_skip / refinement_done), at most NREF
            beta, rank, rcond, ptr = ctx.fit_resident(_capi.SOLVE_LSTSQ_PROBE, RCOND)
            taken = 0
            if not refinement_skip(Kc, rcond):
                prev = float(np.max(np.abs(beta)))
                for _ in range(NREF):
                    s = ctx.residual_rhs(beta)[0]
                    delta = ctx.solve_device(_capi.SOLVE_LSTSQ, RCOND, Kc, ptr, rhs=s)[0]
                    beta = beta + delta
                    taken += 1
                    step = float(np.max(np.abs(delta)))
                    if refinement_done(Kc, step, prev, float(np.max(np.abs(beta))), rcond):
                        break
                    prev = step
            svd_step.taken, svd_step.rcond = taken, float(rcond)
            return beta, rank

        for _ in range(20):
            svd_step()
        ctx.sync()
        n = max(5, min(args.steps, 50))
        t0 = time.perf_counter()
        for _ in range(n):
            beta, rank = svd_step()
        ctx.sync()
        el = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(n):
            ctx.residual_rhs(beta)
        ctx.sync()
        el_res = time.perf_counter() - t0
        out["steps"] = {"ms_per_fit": el / n * 1e3, "rows_per_s": m * n / el, "refinement_steps": int(svd_step.taken), "refinement_steps_max": NREF,
                        "rcond_est": svd_step.rcond, "rank": int(rank),
                        "residual_rhs_ms_per_call": el_res / n * 1e3,
                        "residual_rhs_GBps": (8 * Kc + 17) * m / (el_res / n) / 1e9,
                        "protocol": "fsnap_fit_resident(LSTSQ_PROBE) + refinement_steps x (fsnap_residual_rhs: one pass over the "
                                    "rows + fsnap_solve_device_rhs), weights resident; the stop rule of Solver._refine "
                                    "(refinement_done: what is left is below lstsq's own kappa eps) decides the count"}
        pt = ParallelTools()
        cfg = Config(pt, {"SOLVER": {"solver": "SVD"}})
        sv = solver_factory.solver("SVD", pt, cfg)
        sv.keep_resident = True

        def class_fit(Ax, nrep):
            ts = []
            for _ in range(nrep + 8):      # (the first six uploads of a context are the timed probes of staged_h2d)
                t0 = time.perf_counter()
                sv.fit = None
                sv.perform_fit(Ax, b, w, trainall=True)
                ts.append(time.perf_counter() - t0)
            # median of the timed calls: on the driver's boxes ONE call of a series now and then takes 70+ ms inside a stream
            # wait (profiles/r05_lstsq_rows_phases.txt: the same five calls outside bench.py take 5.5 ms each)
            # -- the outliers stay visible: mean and maximum of the same calls are reported beside the median
            t = np.array(ts[8:]) * 1e3
            return float(np.median(t)), {"ms_mean": float(t.mean()), "ms_max": float(t.max()), "calls": int(len(t))}

        ms_cls, spread_cls = class_fit(A, 8)
        fit_cls = np.array(sv.fit)
        out["class_perform_fit"] = {"ms_per_fit": ms_cls, **spread_cls, "rows_per_s": m / (ms_cls * 1e-3),
                                    "refinement_steps": int(sv.last_refine_steps),
                                    "max_rel_diff_vs_steps": float(np.max(np.abs(fit_cls - beta) / np.maximum(np.abs(beta), 1e-300)))}
        if Kc >= 2:
            Ai = A.copy()
            Ai[:, Kc - 1] = Ai[:, 0] * (np.linalg.norm(A[:, Kc - 1]) / max(np.linalg.norm(A[:, 0]), 1e-300)) + 1.0e-9 * A[:, Kc - 1]
            ms_rs, spread_rs = class_fit(Ai, 5)
            rs = sv.last_row_space
            out["row_space"] = {"ms_per_fit": ms_rs, **spread_rs, "rows_per_s": m / (ms_rs * 1e-3), "used_row_space": rs is not None,
                                "info": {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in (rs or {}).items()}
                                if isinstance(rs, dict) else str(rs)}
        pt.free()
        # leave the headline rows in place for whatever follows
        ctx.upload_rows(A, b)
        ctx.set_weights(w)
    except Exception as e:  # pragma: no cover - optional leg
        out["error"] = f"{type(e).__name__}: {e}"
    return out


def run_rank(args):
    # exactly ONE line on stdout: libraries underneath (RCCL prints a version banner through C stdio, which surfaces
    # at exit, after everything Python printed) get stderr as their fd 1; the JSON line goes to the real stdout
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE = {world} ranks")

    from fitsnap_amd import rendezvous

    if os.environ.get("FSNAP_BENCH_DRYRUN"):
        # launch-path check without a GPU (tests/test_bench_launch_cpu.py): rendezvous only, with a stand-in id
        ident = rendezvous.exchange(rank, world, lambda: hashlib.sha256(os.urandom(16)).digest() * 4)
        time.sleep(0.2)                                        # every rank has read the id before rank 0 removes it
        rendezvous.done(rank)
        rec = {"dryrun": True, "rank": rank, "world": world, "local_rank": local_rank, "id_sha256": hashlib.sha256(ident).hexdigest()}
        sys.stderr.write("FSNAP_BENCH_DRYRUN " + json.dumps(rec) + "\n")
        if rank == 0:
            real_stdout.write(json.dumps(rec) + "\n")
            real_stdout.flush()
        return

    from fitsnap_amd import _capi

    ndev = _capi.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a gfx950 GPU (no CPU fallback)")
    if local_world > ndev and args.transport != "p2p":
        raise SystemExit(f"bench.py: {local_world} ranks on this node but only {ndev} GPU(s) visible (RCCL does not put two "
                         "ranks on one device)")
    multi = world > 1 or args.force_dist

    ctx = _capi.HipContext(local_rank % ndev)
    n_seen = 1
    if multi:
        ctx.comm_init(world, rank, rendezvous.exchange(rank, world, lambda: _capi.comm_id(args.transport)))     # native, no torch
        rendezvous.done(rank)
        n_seen = ctx.comm_info()[0]
    for kv in args.option:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    repack = not args.no_repack
    ctx.set_option("repack", 1 if repack else 0)

    Kc = args.cols
    modes = ["strong", "weak"] if args.scaling == "both" else [args.scaling]
    results = {}
    for mode in modes:
        if mode == "weak" and "strong" in results and world == 1:
            results["weak"] = results["strong"]            # one GPU: the same rows, the same run
            continue
        results[mode] = run_mode(ctx, args, mode, rank, world, multi, _capi)
    head = results[modes[0]]                               # what `value` reports

    pipelined = None
    if rank == 0 and world == 1 and not args.force_dist and args.pipelined:
        pipelined = run_pipelined(args, head, local_rank % ndev, _capi)

    svd_extra = None
    if rank == 0 and world == 1 and not args.force_dist and args.svd_solver:
        svd_extra = run_svd_solver(ctx, args, head, local_rank % ndev, _capi)

    # stand-alone row-weighting kernel (north_star: achieved HBM GB/s), measured outside the timed region (on rank 0's
    # rows; in a multi-GPU job the other ranks wait at the closing barrier meanwhile)
    wk = None
    if rank == 0:
        try:
            # the rows RESIDENT now are those of the mode that ran last (strong and weak hold different shards of a
            # multi-rank job: sizing the output by the first mode's rows overran it -- a GPU memory fault on rank 0 of every
            # `--scaling both` run with N > 1, found when round 6 first ran two ranks)
            m1 = results[modes[-1]]["rows_this_rank"]
            d_aw = ctx.dev_alloc(m1 * Kc * 8)
            d_bw = ctx.dev_alloc(m1 * 8)
            wms = []
            for i in range(6):
                ctx.weight_rows_device(d_aw, Kc, d_bw)
                wms.append(ctx.timing()["weight_ms"])
            wms = float(np.mean(wms[1:]))
            wbytes = (16 * Kc + 24) * m1                     # read A, b, w; write aw, bw
            wk = {"kernel": "fsnap_weight_rows_k", "bound": "hbm", "ms": wms, "achieved": wbytes / (wms * 1e-3) / 1e9,
                  "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": wbytes / (wms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                  "algorithmic_bytes_per_launch": wbytes}
            ctx.dev_free(d_aw)
            ctx.dev_free(d_bw)
        except Exception as e:  # pragma: no cover
            wk = {"error": str(e)}

    if rank == 0:
        info = head["info"]
        m0 = head["rows_this_rank"]                        # rows one launch of rank 0's kernel processes
        syrk_avg_ms = head["kernel_ms"][0]
        flops_per_launch = (Kc * Kc + 3 * Kc) * m0         # K^2 + 3K flop/row x rows per launch
        bytes_per_launch = (8 * Kc + 16) * m0              # A row + b + w per row, A read once
        kernel_name = kernel_name_of(info)
        traffic, traffic_source = recorded_traffic(m0, Kc, info, kernel_name)
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
            executed = info["NB"] * (info["NB"] + 1) // 2 * 512 * m0
        roofline.update({"executed_mfma_flops_per_launch": executed,
                         "executed_over_algorithmic": (executed / flops_per_launch) if executed else None})
        roofline.update({"traffic": traffic, "traffic_source": traffic_source, "kernel": kernel_name,
                         "kernel_ms_avg": syrk_avg_ms, "reduce_kernel_ms_avg": head["reduce_ms"],
                         "rows_per_launch": m0, "rank": 0,
                         "kernel_timing": f"HIP events on the kernel's stream around every {max(1, args.timing_every)}. launch "
                                          f"of the timed region ({head['sampled']} of {args.steps} launches)",
                         "flops_per_launch": flops_per_launch, "algorithmic_bytes_per_launch": bytes_per_launch})
        n = Kc * Kc + Kc + 3

        def rate(res):
            return res["rows_total"] * args.steps / res["elapsed"]

        out = {
            "metric": f"training rows/sec through A^T A + solve, {args.rows} x {Kc} fp64" +
                      (" in total over the GPUs (strong scaling)" if head["mode"] == "strong" else " per GPU (weak scaling)"),
            "value": rate(head),
            "unit": "rows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "preheat_steps": max(0, args.preheat),
            "ms_per_step": head["elapsed"] / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": head["mode"],
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (f"synthetic {args.rows} x {Kc} fp64 A (SURVEY 8d generator), RIDGE alpha=1e-8 normal equations "
                             f"(BASELINE configs[1]), A/b/w resident in HBM; strong: these rows split over {world} GPU(s); "
                             f"weak: {args.rows} rows on every GPU"),
                "rows": args.rows, "K": Kc, "solver": "RIDGE",
                "parallelism": f"dp{world}: rows sharded by rank, one in-place ncclAllReduce of {n} doubles per fit "
                               "(native RCCL behind the C ABI, no torch), solve on every rank",
                "weights_packed_every_step": repack,
                "launch": info,
            },
            "n_ranks_seen": n_seen,
            "transport": ctx.comm_transport() if multi else "none",
            "ranks_per_device": max(1, -(-local_world // ndev)),
            "per_rank": {"mode": head["mode"], "rows": head["rows_per_rank"], "kernel_ms": head["kernel_ms"],
                         "allreduce_ms": head["allreduce_ms"], "wall_s": head["elapsed_per_rank_s"]},
            "roofline": roofline,
            "weighting_kernel": wk,
            "h2d_upload_ms": head["upload_ms"],
            "h2d_upload_path": head["upload_how"],
            "h2d_inclusive_rows_per_s": m0 / ((head["upload_ms"] + head["elapsed"] / args.steps * 1e3) * 1e-3),
            "torch_imported": "torch" in sys.modules,
            "launched_by": "bench.py" if os.environ.get("FSNAP_BENCH_SPAWNED") else ("launcher" if "RANK" in os.environ else "direct"),
        }
        for mode in ("strong", "weak"):
            if mode in results:
                res = results[mode]
                out[f"{mode}_value"] = rate(res)
                out[f"{mode}_ms_per_step"] = res["elapsed"] / args.steps * 1e3
                if mode != head["mode"]:
                    out[f"{mode}_per_rank"] = {"rows": res["rows_per_rank"], "kernel_ms": res["kernel_ms"],
                                               "allreduce_ms": res["allreduce_ms"]}
        if pipelined is not None:
            out["pipelined"] = pipelined
        if svd_extra is not None:
            out["svd_solver"] = svd_extra
        if not args.no_cpu_baseline:
            if world == 1:
                out["cpu_baseline"] = cpu_baseline(head["A"], head["b"], head["w"], head["beta"])
            else:
                # rank 0 regenerates the whole problem of the strong-scaling run (the ranks hold slices of exactly these
                # rows) and times the oracle on it, like the N = 1 line; the other ranks wait at the closing barrier
                from fitsnap_amd.synthetic import synth_problem

                strong = results.get("strong")
                Af, bf, wf = synth_problem(args.rows, Kc)
                out["cpu_baseline"] = cpu_baseline(Af, bf, wf, strong["beta"] if strong is not None else None)
                del Af, bf, wf
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    if multi:
        ctx.barrier()
    ctx.close()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    rank = os.environ.get("RANK", "0")
    world = os.environ.get("WORLD_SIZE", "1")
    try:
        run_rank(args)
    except SystemExit:
        raise
    except BaseException as e:  # noqa: BLE001 - one line, then out: interpreter teardown may wait on a stuck stream
        sys.stderr.write(f"bench.py: rank {rank} of {world} failed: {type(e).__name__}: {e}\n")
        sys.stderr.flush()
        os._exit(1)


if __name__ == "__main__":
    main()
