"""GPU (-m gpu): every BASELINE config exercised at its own size (SURVEY.md 8a sizes).

C2 (10^6 x 128) lives in test_gpu_parity.py::test_full_size_linearity_and_parity.  Here:
  C3  WBe shape 1 772 880 x 110   (kernel 1A <7>: partial last column block)    examples/WBe_PRB2019/.../screen.out:63-65
  quadratic-SNAP shape 15 213 x 1 595 (tiled kernel + device Cholesky)          examples/Ta_Quadratic_JCP2018/.../screen.out:31-33
  InP shape 367 900 x 480  (tiled kernel, many row splits)                      BASELINE.md
  the 32-bit buffer-offset clamp of the launch plans (rows per wave limited to < 4 GiB of A)
Real A for these needs LAMMPS: the matrices are the synthetic generator's at the reference's shapes; the
checks are the size-independent properties (additivity over a row split, mask == zero weight, exact power-of-two
scaling) plus statistics and fits against the oracle on the same rows."""
import numpy as np
import pytest

from fitsnap_amd import _capi
from fitsnap_amd.config import Config
from fitsnap_amd.parallel_tools import ParallelTools
from fitsnap_amd.solvers import solver_factory
from oracle import fitsnap_oracle as orc

from conftest import maxrel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = _capi.HipContext(0)
    yield c
    c.close()


def run_stats(ctx, A, b, w, testing=None):
    ctx.upload_rows(A, b)
    ctx.set_weights(w, None if testing is None else (~np.asarray(testing, dtype=bool)).astype(np.uint8))
    return ctx.normal_eq()


def scaled_diff(G, Gr):
    d = np.sqrt(np.diag(Gr))
    return np.max(np.abs(G - Gr) / (d[:, None] * d[None, :]))


def make_solver(name, extra=None):
    pt = ParallelTools()
    d = {"SOLVER": {"solver": name}}
    d.update(extra or {})
    return pt, solver_factory.solver(name, pt, Config(pt, d))


def test_wbe_shape_at_full_size(ctx):
    m, K = 1_772_880, 110
    A, b, w = orc.synth_problem(m, K)
    G, c, s = run_stats(ctx, A, b, w)
    info = ctx.launch_info()
    assert info["kernel_or_pairs"] == 3 and info["NB"] == 7                    # kernel 1A, 7 column blocks (110 = 6*16 + 14)
    # (1) additivity over an uneven three-way row split
    cuts = [0, 411_003, 1_299_998, m]
    parts = [run_stats(ctx, A[a:z], b[a:z], w[a:z]) for a, z in zip(cuts[:-1], cuts[1:])]
    assert scaled_diff(sum(p[0] for p in parts), G) < 1e-12
    assert sum(p[2][2] for p in parts) == s[2] == m
    # (2) masking a row == zero weight on that row
    t = orc.synth_testing_mask(m)
    Gm, cm, sm = run_stats(ctx, A, b, w, t)
    Gz, cz, sz = run_stats(ctx, A, b, np.where(t, 0.0, w))
    assert scaled_diff(Gm, Gz) < 1e-13 and sm[2] == (~t).sum()
    # (3) exact power-of-two weight scaling
    G4, c4, _ = run_stats(ctx, A, b, 2.0 * w)
    assert np.array_equal(G4, 4.0 * G) and np.array_equal(c4, 4.0 * c)
    # (4) statistics against the oracle's BLAS
    Gr, cr, sr = orc.normal_eq(A, b, w)
    assert scaled_diff(G, Gr) < 1e-12 and s[2] == sr[2]
    assert np.max(np.abs(c - cr) / (np.sqrt(np.diag(Gr)) * np.sqrt(sr[0]))) < 1e-12
    # (5) fits through the plugin API: RIDGE (configs[2] solver family) and SVD vs the oracle
    pt, sol = make_solver("RIDGE", {"RIDGE": {"alpha": 1e-8}})
    sol.perform_fit(A, b, w, trainall=True)
    assert maxrel(sol.fit, orc.ridge_fit(A, b, w, 1e-8)) < 1e-6
    pt.free()
    pt, sol = make_solver("SVD")
    sol.perform_fit(A, b, w[~t], fs_dict={"Testing": t.tolist()})
    assert maxrel(sol.fit, orc.svd_fit(A, b, w, t)) < 1e-6
    pt.free()


def test_quadratic_snap_shape_at_full_size():
    # 15 213 x 1 595 end to end through solver_factory: tiled SYRK, reduction, blocked Cholesky on the GPU, refinement
    m, K = 15_213, 1_595
    A, b, w = orc.synth_problem(m, K)
    pt, sol = make_solver("SVD")
    sol.perform_fit(A, b, w, trainall=True)
    ref = orc.svd_fit(A, b, w)
    assert maxrel(sol.fit, ref) < 1e-6
    G, c, s = sol.last_statistics
    Gr, cr, sr = orc.normal_eq(A, b, w)
    assert scaled_diff(G, Gr) < 1e-12 and s[2] == m
    pt.free()
    pt, sol = make_solver("RIDGE", {"RIDGE": {"alpha": 1e-8}})
    sol.perform_fit(A, b, w, trainall=True)
    assert maxrel(sol.fit, orc.ridge_fit(A, b, w, 1e-8)) < 1e-6
    pt.free()


def test_quadratic_snap_shape_ill_conditioned_takes_the_row_space_path():
    # the same shape with kappa(A_w) = 1e9 (descriptor products of a quadratic model are nearly dependent): the SVD solver
    # must leave the normal equations for the row-space solve (K > 256: factors kept apart on the host, 16-row tiles in the
    # orthogonalisation kernel) and still match lstsq(1e-13) on the rows -- and must get there quickly: the K x K solve is
    # asked for with FSNAP_SOLVE_LSTSQ_PROBE, so the statistics the Cholesky cannot resolve come straight back instead of
    # going through the library's Jacobi eigen-truncation first (58 s at this K before that)
    import time
    m, K = 4_001, 1_595
    rng = np.random.default_rng(1595)
    U, _ = np.linalg.qr(rng.standard_normal((m, K)))
    V, _ = np.linalg.qr(rng.standard_normal((K, K)))
    A = (U * np.logspace(0, -9, K)) @ V.T
    b = A @ rng.standard_normal(K) + 1e-4 * rng.standard_normal(m)
    w = rng.uniform(0.5, 2.0, m)
    pt, sol = make_solver("SVD")
    t0 = time.perf_counter()
    sol.perform_fit(A, b, w, trainall=True)
    assert time.perf_counter() - t0 < 10.0
    info = sol.last_row_space
    assert info is not None and info["converged"] == 1.0 and sol.last_rank == K
    ref = orc.svd_fit(A, b, w)
    aw, bw = w[:, None] * A, w * b
    kw = 2.0 * 1e9 / 0.5                                              # kappa(A) x the spread of the row weights: an upper bound
    assert np.linalg.norm(sol.fit - ref) <= max(1e-6, 50 * kw * np.finfo(float).eps) * np.linalg.norm(ref)
    res, res_ref = np.linalg.norm(aw @ sol.fit - bw), np.linalg.norm(aw @ ref - bw)
    assert abs(res - res_ref) <= 1e-6 * res_ref
    pt.free()


def test_inp_shape_at_full_size(ctx):
    m, K = 367_900, 480
    A, b, w = orc.synth_problem(m, K)
    t = orc.synth_testing_mask(m)
    G, c, s = run_stats(ctx, A, b, w, t)
    Gr, cr, sr = orc.normal_eq(A, b, w, t)
    assert scaled_diff(G, Gr) < 1e-12 and s[2] == sr[2]
    h = 150_001
    G1 = run_stats(ctx, A[:h], b[:h], w[:h], t[:h])[0]
    G2 = run_stats(ctx, A[h:], b[h:], w[h:], t[h:])[0]
    assert scaled_diff(G1 + G2, G) < 1e-12
    pt, sol = make_solver("RIDGE", {"RIDGE": {"alpha": 1e-4}})
    sol.perform_fit(A, b, w[~t], fs_dict={"Testing": t.tolist()})
    assert maxrel(sol.fit, orc.ridge_fit(A, b, w, 1e-4, testing=t)) < 1e-6
    pt.free()


@pytest.mark.parametrize("K", [31, 128, 200])
def test_row_range_of_a_wave_is_clamped_below_4_gib(ctx, K):
    # lda = 60 000 doubles (480 KB per row) and one workgroup: a wave's rows would span 40000/4 * 480 KB = 4.8 GB of A,
    # beyond what a 32-bit buffer offset reaches; the launch plan must split the range (fsnap_capi.cpp: max_cpw clamp)
    import torch

    m, lda = 40_000, 60_000
    rng = np.random.default_rng(K)
    A = rng.standard_normal((m, K))
    b = rng.standard_normal(m)
    w = rng.uniform(0.5, 2.0, m)
    dev = torch.device("cuda", 0)
    dA = torch.zeros((m, lda), dtype=torch.float64, device=dev)          # 19.2 GB
    dA[:, :K] = torch.from_numpy(A).to(dev)
    dA[:, K:K + 8] = float("nan")                                        # anything right of column K must not be read into G
    db = torch.from_numpy(b).to(dev)
    torch.cuda.synchronize()
    c2 = _capi.HipContext(0)
    try:
        c2.set_option("nblocks", 1)
        c2.set_option("nsplit", 1)
        c2.bind_rows(dA.data_ptr(), m, K, lda, db.data_ptr())
        c2.set_weights(w)
        Gr, cr, sr = orc.normal_eq(A, b, w)
        if K == 128:
            # 40 000 rows of 128 columns are kernel 1S's by default: ONE chunk of 40 000 rows here, whose 32-bit offsets
            # start afresh with every 128-row phase (a descriptor per phase) -- nothing to clamp
            info = c2.launch_info()
            assert info["kernel_or_pairs"] == 7 and info["workgroups"] == 1 and info["chunks_per_wave"] == m
            G, c, s = c2.normal_eq()
            assert scaled_diff(G, Gr) < 1e-12 and s[2] == m
            c2.set_option("short", 0)            # ... and kernel 1A on the same rows: the clamp this test is about
        info = c2.launch_info()
        rows_per_wave = info["chunks_per_wave"] * 4
        assert rows_per_wave * lda * 8 < 2 ** 32 and info["workgroups"] >= 2
        G, c, s = c2.normal_eq()
        assert scaled_diff(G, Gr) < 1e-12 and s[2] == m
    finally:
        c2.close()
        del dA, db
        torch.cuda.empty_cache()


def test_repack_option_keeps_results_and_packs_every_fit(ctx, ta):
    A, b, w = ta
    ctx.upload_rows(A, b)
    ctx.set_weights(w)
    ref = ctx.normal_eq()
    ctx.set_option("repack", 1)
    try:
        for _ in range(3):
            G, c, s = ctx.normal_eq()
            assert np.array_equal(G, ref[0]) and np.array_equal(c, ref[1]) and np.array_equal(s, ref[2])
    finally:
        ctx.set_option("repack", 0)


def test_training_weights_are_spread_on_the_device(ctx, ta, ta_fits):
    # fsnap_set_weights_train: one weight per TRAINING row (the reference's explicit-array quirk, svd.py:46), mask and
    # its prefix sum resident across re-weightings -- same statistics as the host-side scatter, bit for bit
    A, b, w = ta
    t = ta_fits["testing_mask"]
    mask = (~t).astype(np.uint8)
    rank = (np.cumsum(mask, dtype=np.int64) - mask).astype(np.int32)
    ctx.upload_rows(A, b)
    ctx.set_weights(np.where(t, 0.0, w), mask)
    ref = ctx.normal_eq()
    ctx.set_weights_train(np.ascontiguousarray(w[~t]), mask, rank)
    got = ctx.normal_eq()
    assert all(np.array_equal(x, y) for x, y in zip(got, ref))
    for scale in (0.5, 3.0):                                   # resident mask: only the weights travel
        ctx.set_weights_train(np.ascontiguousarray(scale * w[~t]))
        G, c, s = ctx.normal_eq()
        if scale == 0.5:                                       # power of two: exact
            assert np.array_equal(G, 0.25 * ref[0])
        else:
            assert np.max(np.abs(G - 9.0 * ref[0]) / np.sqrt(np.outer(np.diag(ref[0]), np.diag(ref[0])))) < 1e-13 * 9
        assert s[2] == ref[2][2]
    ctx.set_weights(w)                                         # full weights, no mask: the resident mask survives
    assert ctx.normal_eq()[2][2] == len(b)
    ctx.set_weights_train(np.ascontiguousarray(w[~t]))
    assert all(np.array_equal(x, y) for x, y in zip(ctx.normal_eq(), ref))
    ctx.set_weights(w, mask)                                   # a mask uploaded without its prefix sum invalidates it
    with pytest.raises(_capi.FsnapError):
        ctx.set_weights_train(np.ascontiguousarray(w[~t]))
    with pytest.raises(ValueError):
        ctx.set_weights_train(np.ones(5), mask, rank[:-1])


def test_reweighting_loop_through_the_solver_matches_independent_fits(ta, ta_fits):
    # keep_resident: candidate weights travel compactly, the mask stays on the device; every candidate equals a fresh fit
    A, b, w = ta
    t = ta_fits["testing_mask"]
    fsd = {"Testing": t.tolist()}
    pt, s = make_solver("RIDGE", {"RIDGE": {"alpha": 1e-8}})
    s.keep_resident = True
    rng = np.random.default_rng(1)
    for it in range(4):
        wi = w * rng.uniform(0.5, 2.0, len(w))
        s.perform_fit(A, b, wi[~t], fs_dict=fsd)
        assert maxrel(s.fit, orc.ridge_fit(A, b, wi, 1e-8, testing=t)) < 1e-6
        if it == 1:
            s.pt.hip().set_weights(w, (t).astype(np.uint8))    # somebody else used the context meanwhile
    pt.free()
