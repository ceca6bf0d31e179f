"""CPU: the launch path of `bench.py --gpus N` without a launcher (the scaling run's first contact): the script starts
N ranks of itself with RANK / LOCAL_RANK / WORLD_SIZE, a communicator-id file and a job token of its own; every rank
gets through the rendezvous with the SAME id; a rank that dies takes the job down with one line and a non-zero status.
FSNAP_BENCH_DRYRUN stops each rank after the rendezvous (no GPU here); on a GPU box the same path continues into
fsnap_comm_init (tests/test_gpu_native_comm.py)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE",
                                                            "FSNAP_COMM_FILE", "FSNAP_COMM_TOKEN")}
    env.update(extra)
    return env


def test_bench_spawns_its_own_ranks_and_they_agree_on_the_id():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1"],
                         env=_env(FSNAP_BENCH_DRYRUN="1", FSNAP_COMM_TIMEOUT="60"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    recs = [json.loads(l.split(" ", 1)[1]) for l in out.stderr.splitlines() if l.startswith("FSNAP_BENCH_DRYRUN ")]
    assert sorted(r["rank"] for r in recs) == [0, 1]
    assert {r["world"] for r in recs} == {2} and sorted(r["local_rank"] for r in recs) == [0, 1]
    assert len({r["id_sha256"] for r in recs}) == 1                       # both ranks hold rank 0's id
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0])["rank"] == 0          # exactly one line on stdout, from rank 0


def test_four_ranks_and_two_jobs_side_by_side_do_not_mix_ids():
    cmd = [sys.executable, BENCH, "--gpus", "4"]
    env = _env(FSNAP_BENCH_DRYRUN="1", FSNAP_COMM_TIMEOUT="60")
    a = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    b = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    ids = []
    for p in (a, b):
        so, se = p.communicate(timeout=300)
        assert p.returncode == 0, se[-2000:]
        recs = [json.loads(l.split(" ", 1)[1]) for l in se.splitlines() if l.startswith("FSNAP_BENCH_DRYRUN ")]
        assert sorted(r["rank"] for r in recs) == [0, 1, 2, 3] and len({r["id_sha256"] for r in recs}) == 1
        ids.append(recs[0]["id_sha256"])
    assert ids[0] != ids[1]


def test_a_missing_rank_fails_the_job_with_one_line_instead_of_hanging():
    # ranks started by a "launcher" that lost rank 1: rank 0 publishes, nobody can complete -- here the dry run has no
    # collective to wait in, so the failure is provoked in the rendezvous itself: rank 1 alone, no rank 0, short timeout
    env = _env(FSNAP_BENCH_DRYRUN="1", FSNAP_COMM_TIMEOUT="2", RANK="1", WORLD_SIZE="2", LOCAL_RANK="1",
               FSNAP_COMM_FILE=os.path.join(ROOT, "tests", "_no_such_dir_", "id"))
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0
    msg = [l for l in out.stderr.splitlines() if l.startswith("bench.py: rank 1 of 2 failed")]
    assert len(msg) == 1 and "TimeoutError" in msg[0] and "after 2 s" in msg[0]
    assert out.stdout.strip() == ""


def test_world_size_mismatch_is_refused():
    env = _env(RANK="0", WORLD_SIZE="4", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE = 4" in out.stderr


def test_strong_scaling_rows_are_the_rows_of_the_one_gpu_problem():
    sys.path.insert(0, ROOT)
    import numpy as np

    import bench
    from fitsnap_amd.synthetic import synth_problem

    total, Kc = 3 * 65536 + 1234, 5
    A, b, w = synth_problem(total, Kc)
    for world in (1, 3, 8):
        parts = [bench.synth_rows(total * r // world, total * (r + 1) // world, total, Kc) for r in range(world)]
        assert np.array_equal(np.concatenate([p[0] for p in parts]), A)
        assert np.array_equal(np.concatenate([p[1] for p in parts]), b)
        assert np.array_equal(np.concatenate([p[2] for p in parts]), w)
