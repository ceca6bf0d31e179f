"""GPU (-m gpu): the native RCCL transport behind the C ABI (fsnap_comm_*, fsnap_fit_dist) -- the reference's
comm.Allreduce(c), comm.Allreduce(d) + solve (examples/library/transpose_trick/example.py:245-254).

One GPU is enough for the API (a communicator of ONE rank runs every collective); the two-process test needs two
devices and is skipped otherwise (RCCL does not put two ranks on one device)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from fitsnap_amd import _capi
from fitsnap_amd.config import Config
from fitsnap_amd.parallel_tools import ParallelTools
from fitsnap_amd.solvers import solver_factory
from oracle import fitsnap_oracle as orc

from conftest import maxrel

pytestmark = pytest.mark.gpu


def _free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_rank_communicator_runs_every_collective(ta):
    A, b, w = ta
    ctx = _capi.HipContext(0)
    assert ctx.comm_info() == (1, 0)
    ctx.comm_init(1, 0, _capi.comm_id())
    assert ctx.comm_info() == (1, 0)
    x = np.arange(7, dtype=np.float64) - 3.0
    for op in (_capi.REDUCE_SUM, _capi.REDUCE_MAX, _capi.REDUCE_MIN):
        assert np.array_equal(ctx.allreduce_host(x.copy(), op), x)
    assert ctx.bcast_bytes(b"row labels", 10, 0) == b"row labels"
    assert ctx.allgather_bytes(b"\x01\x02\x03", 1) == [b"\x01\x02\x03"]
    ctx.barrier()
    # one fit: statistics -> in-place all-reduce in HBM -> solve; equals the single-GPU call bit for bit
    ctx.upload_rows(A, b)
    ctx.set_weights(w)
    plain, rank0, rc0, _ = ctx.fit_resident(_capi.SOLVE_RIDGE, 1e-8)
    dist, rank1, rc1, ptr = ctx.fit_dist(_capi.SOLVE_RIDGE, 1e-8, A.shape[1])
    assert np.array_equal(plain, dist) and rank0 == rank1 and rc0 == rc1
    G, c, s = ctx.download_packed(ptr, A.shape[1])
    Gr, cr, sr = orc.normal_eq(A, b, w)
    d = np.sqrt(np.diag(Gr))
    assert np.max(np.abs(G - Gr) / (d[:, None] * d[None, :])) < 1e-12 and s[2] == sr[2]
    with pytest.raises(ValueError):
        ctx.fit_dist(_capi.SOLVE_RIDGE, 1e-8, A.shape[1] + 1)
    with pytest.raises(_capi.FsnapError):
        ctx.comm_init(1, 0, _capi.comm_id())               # one communicator per context
    ctx.comm_destroy()
    assert ctx.comm_info() == (1, 0)
    with pytest.raises(_capi.FsnapError):
        ctx.allreduce_host(x.copy())                       # no communicator any more
    ctx.close()


@pytest.mark.parametrize("K,m", [(31, 4001), (142, 6007), (200, 12001), (264, 9001), (300, 5003), (480, 6000)])
def test_triangle_payload_of_the_all_reduce_gives_the_same_statistics(K, m):
    # option reduce_triangle: the multi-GPU fit all-reduces [upper triangle | c | scalars] (K (K + 1) / 2 + K + 3 doubles)
    # between a pack and an unpack kernel instead of the mirrored K^2 + K + 3 (default from 256 columns on).  In a
    # communicator of one rank: the reduced buffer, the fit, rank and conditioning carry the same bits in both forms, G is
    # exactly symmetric, and the device Cholesky (K = 480) reads the unpacked buffer.  200 / 264 columns: statistics from kernel 1Q
    # (>= 8 192 rows), mirror filled by the copy kernel behind the all-reduce, host solve
    A, b, w = orc.synth_problem(m, K)
    ctx = _capi.HipContext(0)
    ctx.comm_init(1, 0, _capi.comm_id())
    ctx.upload_rows(A, b)
    ctx.set_weights(w)
    got = []
    for tri in (0, 1, -1):
        ctx.set_option("reduce_triangle", tri)
        beta, rank, rc, ptr = ctx.fit_dist(_capi.SOLVE_RIDGE, 1e-8, K)
        G, c, s = ctx.download_packed(ptr, K)
        assert np.array_equal(G, G.T)
        got.append((beta, rank, rc, G, c, s))
    for g in got[1:]:
        assert all(np.array_equal(x, y) for x, y in zip(g, got[0]))
    Gr, cr, sr = orc.normal_eq(A, b, w)
    d = np.sqrt(np.diag(Gr))
    assert np.max(np.abs(got[1][3] - Gr) / (d[:, None] * d[None, :])) < 2e-12 and got[1][5][2] == sr[2]
    ref = orc.ridge_fit(A, b, w, 1e-8)
    assert maxrel(got[1][0], ref) < 1e-6
    ctx.close()


def test_fit_dist_needs_a_communicator_and_survives_a_rank_local_failure(ta):
    A, b, w = ta
    K = A.shape[1]
    ctx = _capi.HipContext(0)
    ctx.upload_rows(A, b)
    ctx.set_weights(w)
    with pytest.raises(_capi.FsnapError, match="no communicator"):
        ctx.fit_dist(_capi.SOLVE_RIDGE, 1e-8, K)                    # no silent single-GPU fallback
    ctx.comm_init(1, 0, _capi.comm_id())
    ctx.set_option("timing_every", 1)
    ref = ctx.fit_dist(_capi.SOLVE_RIDGE, 1e-8, K)[0]
    # a failure only this rank sees (wrong K): it still takes part in the collective (NaN statistics), reports ITS error,
    # and the communicator is in step for the next fit
    with pytest.raises(ValueError, match="columns"):
        ctx.fit_dist(_capi.SOLVE_RIDGE, 1e-8, K + 1)
    again = ctx.fit_dist(_capi.SOLVE_RIDGE, 1e-8, K)[0]
    assert np.array_equal(ref, again)
    # the collective of every sampled fit was timed on the stream
    again, rank, rcond, ptr = ctx.fit_dist(_capi.SOLVE_RIDGE, 1e-8, K)
    assert np.array_equal(ref, again) and rank == K and ptr
    t = ctx.timing_history_comm(3)
    assert t.shape == (3,) and np.all(t >= 0.0) and np.all(t < 50.0)
    # a rank without rows: zeros into the collective (here: a singular system -> the ridge term alone)
    ctx.drop_rows()
    beta0 = ctx.fit_dist(_capi.SOLVE_RIDGE, 1e-8, K)[0]
    assert np.array_equal(beta0, np.zeros(K))
    ctx.close()


def test_a_context_recreated_after_free_joins_the_communicator_again(ta, monkeypatch):
    # ParallelTools.free() destroys the context and with it the communicator; the next pt.hip() must not hand out a
    # context without one (fsnap_fit_dist would have fitted this rank's shard alone)
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("LOCAL_RANK", "0")
    A, b, w = ta
    pt = ParallelTools(comm="rccl")
    pt.force_multi = True
    first = pt.hip()
    assert pt._transport._joined
    pt.free()
    assert pt._hip is None and not pt._transport._joined
    second = pt.hip()
    assert second is not first and pt._transport._joined
    second.upload_rows(A, b)
    second.set_weights(w)
    beta = second.fit_dist(_capi.SOLVE_RIDGE, 1e-8, A.shape[1])[0]          # raises FSNAP_E_STATE without a communicator
    assert maxrel(beta, orc.ridge_fit(A, b, w, 1e-8)) < 1e-6
    pt.free()


def test_large_k_fit_through_the_communicator():
    # K = 480: the reduced statistics are factorised on the GPU (no mirror), straight after the all-reduce
    rng = np.random.default_rng(480)
    m, K = 5000, 480
    A = rng.standard_normal((m, K))
    b = rng.standard_normal(m)
    w = rng.uniform(0.5, 2.0, m)
    ctx = _capi.HipContext(0)
    ctx.comm_init(1, 0, _capi.comm_id())
    ctx.upload_rows(A, b)
    ctx.set_weights(w)
    beta = ctx.fit_dist(_capi.SOLVE_RIDGE, 1e-8, K)[0]
    assert maxrel(beta, orc.ridge_fit(A, b, w, 1e-8)) < 1e-6
    ctx.close()


def test_solver_classes_on_the_native_transport_in_a_one_rank_job(ta, ta_fits, monkeypatch):
    # ParallelTools(comm="rccl") with the collective code paths forced on: fit_dist, all-reduced refinement,
    # fixed-size error tables -- same results as the single-process run
    from pandas.testing import assert_frame_equal

    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("LOCAL_RANK", "0")
    A, b, w = ta
    pt = ParallelTools(comm="rccl")
    pt.force_multi = True
    assert pt.comm_kind == "rccl" and pt.multi and pt.hip().comm_info() == (1, 0)
    assert pt.bcast_object({"a": [1, 2.5, "x"]}) == {"a": [1, 2.5, "x"]}
    assert pt.allgather_object(("k", 3)) == [("k", 3)]
    assert pt.get_ncpn(363) == 363
    t = ta_fits["testing_mask"]
    m = len(b)
    fsd = {"Groups": [f"g{(i // 43) % 5}" for i in range(m)], "Testing": t.tolist(),
           "Row_Type": ["Energy"] * 363 + ["Force"] * 12672 + ["Stress"] * 2178}
    cfg = Config(pt, {"SOLVER": {"solver": "SVD"}})
    s = solver_factory.solver("SVD", pt, cfg)
    s.perform_fit(A, b, w[~t], fs_dict=fsd)
    assert maxrel(s.fit, ta_fits["svd_mask"]) < 1e-9        # statistics + two all-reduced refinement steps
    fit = s.fit.copy()
    s.error_analysis(A, b, w, fsd)
    multi_errors = s.errors.copy()
    cfg = Config(pt, {"SOLVER": {"solver": "RIDGE"}, "RIDGE": {"alpha": 1e-8}})
    r = solver_factory.solver("RIDGE", pt, cfg)
    r.perform_fit(A, b, w, trainall=True)
    assert maxrel(r.fit, ta_fits["ridge_sklearn_1e-8_all"]) < 1e-6
    pt.free()
    pt1 = ParallelTools()
    s1 = solver_factory.solver("SVD", pt1, Config(pt1, {"SOLVER": {"solver": "SVD"}}))
    s1.fit = fit
    s1.error_analysis(A, b, w, fsd)
    assert_frame_equal(multi_errors, s1.errors, check_exact=False, rtol=1e-11, atol=1e-13)
    pt1.free()


def test_bench_runs_the_multi_gpu_step_without_torch(tmp_path):
    # bench.py --force-dist: communicator of one rank, fsnap_fit_dist per step; the process must not import torch
    import json

    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", FSNAP_COMM_FILE=str(tmp_path / "id"))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--rows", "65536", "--steps", "5",
                          "--warmup", "2", "--preheat", "10", "--no-cpu-baseline", "--timing-every", "1"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["torch_imported"] is False and rec["n_gpus"] == 1 and rec["value"] > 0
    assert rec["config"]["weights_packed_every_step"] is True
    # one GPU: strong = weak = the same run; the collective of the sampled fits was timed; fsnap_comm_info saw one rank
    assert rec["scaling"] == "strong" and rec["strong_value"] == rec["weak_value"] == rec["value"]
    assert rec["n_ranks_seen"] == 1 and len(rec["per_rank"]["kernel_ms"]) == 1 and rec["per_rank"]["allreduce_ms"][0] > 0.0
    assert rec["transport"] == "rccl"


def test_bench_two_ranks_both_scalings_on_the_peer_to_peer_transport():
    # `bench.py --gpus 2` as the driver runs it (both scalings in one job), with the two ranks on the peer-to-peer transport so
    # that they can share the one GPU of the box.  First executed in round 6 -- and it crashed: rank 0's weighting-kernel leg
    # sized its output by the FIRST mode's shard while the rows of the LAST mode were resident (a GPU memory fault in every
    # N > 1 run with --scaling both, whatever the transport).
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    env["FSNAP_COMM_TIMEOUT"] = "120"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--transport", "p2p", "--rows", "262144",
                          "--steps", "5", "--warmup", "2", "--preheat", "10"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["n_ranks_seen"] == 2 and rec["transport"] == "p2p" and rec["ranks_per_device"] == 2
    assert rec["scaling"] == "strong" and rec["strong_value"] == rec["value"] > 0 and rec["weak_value"] > 0
    assert len(rec["per_rank"]["kernel_ms"]) == 2 and all(t > 0 for t in rec["per_rank"]["allreduce_ms"])
    assert rec["per_rank"]["rows"] == [131072, 131072]
    assert rec["weighting_kernel"]["ms"] > 0 and rec["cpu_baseline"]["value"] > 0 and rec["torch_imported"] is False


def test_bench_single_gpu_line_carries_the_contract_fields_and_the_pipelined_leg():
    # plain `python bench.py` on one GPU: ONE JSON line with the driver's fields, roofline + cpu_baseline objects, and the
    # two-fits-in-flight leg as an extra object (same coefficients, never `value`)
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "131072", "--steps", "6", "--warmup", "2",
                          "--preheat", "10"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in rec, key
    assert rec["n_gpus"] == 1 and rec["steps"] == 6 and rec["dtype"] == "f64" and rec["roofline"]["bound"] in ("mfma", "hbm")
    assert rec["cpu_baseline"]["kind"] in ("port", "reference") and rec["cpu_baseline"]["value"] > 0
    pl = rec["pipelined"]
    assert pl["fits_in_flight"] == 2 and pl["same_beta_as_headline"] is True and pl["value"] > 0


def test_bench_launcher_refuses_more_ranks_than_devices():
    # `bench.py --gpus N` starts its own ranks; with fewer devices than ranks every rank says so and the job fails with
    # one summary line instead of hanging in RCCL's bootstrap
    n = _capi.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    env["FSNAP_COMM_TIMEOUT"] = "60"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--rows", "65536", "--steps", "2",
                          "--warmup", "1", "--preheat", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and out.stdout.strip() == ""
    assert f"only {n - 1} GPU(s) visible" in out.stderr and f"bench.py: {n}-GPU job failed" in out.stderr


@pytest.mark.parametrize("transport", ["rccl", "p2p"])
def test_two_process_native_fit_matches_the_reference(tmp_path, ta, ta_fits, transport):
    # RCCL needs a device per rank; the peer-to-peer transport (csrc/fsnap_p2p.cpp) also runs both ranks on ONE GPU, which
    # is what a one-GPU box executes: rows by configuration i % 2, fits against the reference goldens, the triangle
    # payload at K >= 256, a rank with zero rows, pooled error tables, the row-space solve
    if transport == "rccl" and _capi.device_count() < 2:
        pytest.skip("needs two GPUs: RCCL does not put two ranks on one device")
    A, b, w = ta
    world = 2
    procs = []
    port = str(_free_port())          # one port for the job, free now (a fixed number collides with whatever else runs on the box)
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world),
                   FSNAP_COMM_FILE=str(tmp_path / "comm_id"), FSNAP_COMM_TOKEN="two-process test", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0", FSNAP_COMM_TIMEOUT="120", FSNAP_DIST_TRANSPORT=transport)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_native_worker.py"), str(tmp_path)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-4000:]
    r0, r1 = (dict(np.load(tmp_path / f"rank{r}.npz")) for r in range(world))
    assert not r0["torch_imported"] and not r1["torch_imported"]
    t = np.random.default_rng(12345).random(len(b)) < 0.1
    Gr, cr, sr = orc.normal_eq(A, b, w, t)
    d = np.sqrt(np.diag(Gr))
    for name in ("SVD", "RIDGE"):
        assert np.array_equal(r0[f"{name}_G"], r1[f"{name}_G"])                 # bit-identical sums on every rank
        assert np.max(np.abs(r0[f"{name}_G"] - Gr) / (d[:, None] * d[None, :])) < 1e-12 and r0[f"{name}_sc"][2] == sr[2]
        assert f"{name}_fit" in r0 and f"{name}_fit" not in r1                  # fit on rank 0 only
    assert maxrel(r0["SVD_fit"], orc.svd_fit(A, b, w, t)) < 1e-6
    assert maxrel(r0["RIDGE_fit"], orc.ridge_fit(A, b, w, 1e-8, testing=t)) < 1e-6
    # the ill-conditioned system of the worker, solved by the reference's lstsq on all rows
    r = np.random.default_rng(77)
    mm, K = 16000, 40
    U, _ = np.linalg.qr(r.standard_normal((mm, K)))
    V, _ = np.linalg.qr(r.standard_normal((K, K)))
    X = (U * np.logspace(0, -10, K)) @ V.T
    y = X @ r.standard_normal(K) + 1e-3 * r.standard_normal(mm)
    ref = orc.svd_fit(X, y, np.ones(mm))
    assert np.linalg.norm(r0["ill_fit"] - ref) <= 50 * 1e10 * np.finfo(float).eps * np.linalg.norm(ref)
    # a rank with ZERO rows (rank 1 owns nothing), right after fits in which it did own rows
    assert maxrel(r0["zero_fit"], orc.ridge_fit(A, b, w, 1e-8)) < 1e-6 and np.array_equal(r0["zero_G"], r1["zero_G"])
    assert int(r1["zero_rows_resident"]) == 0
    # K = 480: the all-reduced statistics are factorised by the device Cholesky on every rank
    r = np.random.default_rng(480)
    A4, b4, w4 = r.standard_normal((6000, 480)), r.standard_normal(6000), r.uniform(0.5, 2.0, 6000)
    ref4 = orc.ridge_fit(A4, b4, w4, 1e-8)
    for rr in (r0, r1):
        assert str(rr["transport"]) == transport
        assert maxrel(rr["k480_beta"], ref4) < 1e-6
    assert np.array_equal(r0["k480_beta"], r1["k480_beta"])                    # deterministic solve of identical sums
    # K = 1595 through the default solver: the ranks agree on the condition estimate (bit for bit) and hence on the steps
    r = np.random.default_rng(1595)
    A5, b5 = r.standard_normal((4000, 1595)), r.standard_normal(4000)
    assert maxrel(r0["k1595_fit"], orc.svd_fit(A5, b5, np.ones(4000))) < 1e-6
    assert r0["k1595_rcond"] == r1["k1595_rcond"] and r0["k1595_steps"] == r1["k1595_steps"]


def test_process_exits_cleanly_when_rccl_is_loaded_before_torch():
    # librccl is dlopen'ed RTLD_LOCAL: with RTLD_GLOBAL its symbols interposed on the libraries a LATER `import torch`
    # maps, and the process died at exit with "double free or corruption" (exit status 134) although everything had worked
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from fitsnap_amd import _capi\n"
        "c = _capi.HipContext(0); c.comm_init(1, 0, _capi.comm_id()); c.barrier(); c.close()\n"
        "import torch\n"
        "x = torch.ones(8, device='cuda'); assert float(x.sum()) == 8.0\n"
        "print('done')\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "done" in out.stdout, (out.returncode, out.stderr[-1500:])
