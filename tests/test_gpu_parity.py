"""GPU (-m gpu): parity of the HIP path, called through the C ABI, against the oracle and
the reference-generated golden vectors.

Tolerances: the statistics G, c are fp64 sums of the same products the oracle forms, in a
different order -> compared relative to the natural scale sqrt(G_ii G_jj) at 1e-12;
fitted coefficients must match the reference's SVD / RIDGE within 1e-6 relative
(BASELINE.json north_star) and 1e-6 absolute (the reference's own test bar,
tests/example_checker.py:62)."""
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


def stats_close(G, c, s, Gr, cr, sr, tol=1e-12):
    d = np.sqrt(np.maximum(np.diag(Gr), 1e-300))
    assert np.max(np.abs(G - Gr) / (d[:, None] * d[None, :])) < tol
    bscale = np.sqrt(max(sr[0], 1e-300))
    assert np.max(np.abs(c - cr) / (d * bscale)) < tol
    assert abs(s[0] - sr[0]) <= tol * max(abs(sr[0]), 1e-300) * 10
    assert abs(s[1] - sr[1]) <= 1e-9 * max(np.sqrt(sr[0] * sr[2]), 1e-300)
    assert s[2] == sr[2]
    assert np.array_equal(G, G.T)


def run_stats(ctx, A, b, w, testing=None):
    ctx.upload_rows(A, b)
    ctx.set_weights(w, None if testing is None else (~np.asarray(testing, dtype=bool)).astype(np.uint8))
    return ctx.normal_eq()


# ---------------------------------------------------------------------------------------
# statistics kernel
# ---------------------------------------------------------------------------------------
def test_ta_golden_statistics(ctx, ta, ta_fits):
    A, b, w = ta
    G, c, s = run_stats(ctx, A, b, w)
    stats_close(G, c, s, *orc.normal_eq(A, b, w))
    t = ta_fits["testing_mask"]
    G, c, s = run_stats(ctx, A, b, w, t)
    stats_close(G, c, s, *orc.normal_eq(A, b, w, t))


@pytest.mark.parametrize("K", [1, 2, 15, 16, 17, 30, 31, 32, 33, 48, 55, 64, 79, 80, 96, 97, 110, 112, 113, 127, 128,
                               129, 137, 142, 143, 144])
def test_statistics_all_column_block_shapes(ctx, K):
    # every NB (1..9), odd/even NB tails, K odd (unaligned 16-byte loads), both SPLIT paths; 129 ... 144: kernel 1A with
    # nine column blocks (45 tiles: 32 in the accumulation registers, 13 in VGPRs; the ACE width 142 of Ta_PACE_RIDGE)
    rng = np.random.default_rng(K)
    m = 4099 + 7 * K                      # ragged: not a multiple of 4 or of the chunk pipeline
    A = rng.standard_normal((m, K)) * (10.0 ** rng.uniform(-3, 3, size=K))
    b = rng.standard_normal(m)
    w = rng.choice([100.0, 1.0, 1e-8], size=m, p=[0.03, 0.83, 0.14])
    t = rng.random(m) < 0.2
    G, c, s = run_stats(ctx, A, b, w, t)
    ref = orc.normal_eq(A, b, w, t)
    stats_close(G, c, s, *ref)
    if K > 80:
        # systems this short run on kernel 1S by default (tests/test_gpu_short.py); kernel 1A on the same rows
        assert ctx.launch_info()["kernel_or_pairs"] == 7
        ctx.set_option("short", 0)
        try:
            G, c, s = run_stats(ctx, A, b, w, t)
            assert ctx.launch_info()["kernel_or_pairs"] == 3
        finally:
            ctx.set_option("short", -1)
        stats_close(G, c, s, *ref)


# ---------------------------------------------------------------------------------------
# kernel 1Q: 145 ... 256 columns, the tile triangle dealt to the four waves of a workgroup (fsnap_syrk_quad.hip)
# ---------------------------------------------------------------------------------------
@pytest.fixture()
def qctx(ctx):
    ctx.set_option("quad_min_rows", 0)          # short test systems: kernel 1Q takes them too
    yield ctx
    ctx.set_option("quad_min_rows", -1)


@pytest.mark.parametrize("K", [145, 150, 159, 160, 161, 168, 175, 176, 177, 191, 192, 193, 200, 207, 208, 209, 223, 224, 225,
                               239, 240, 241, 255, 256, 257, 264, 272, 273, 275, 287, 288])
def test_quad_kernel_statistics_all_column_block_shapes(qctx, K):
    # NB = 10 ... 18: odd / even block counts, K a multiple of 16 (no column select) and not, the last block nearly empty;
    # 17 and 18 blocks run the two-set load pipeline with up to 11 accumulator tiles in VGPRs
    rng = np.random.default_rng(K)
    m = 6151 + 11 * K                     # ragged: not a multiple of 4, uneven chunk ranges per workgroup
    A = rng.standard_normal((m, K)) * (10.0 ** rng.uniform(-3, 3, size=K))
    b = rng.standard_normal(m)
    w = rng.choice([100.0, 1.0, 1e-8], size=m, p=[0.03, 0.83, 0.14])
    t = rng.random(m) < 0.2
    G, c, s = run_stats(qctx, A, b, w, t)
    info = qctx.launch_info()
    assert info["kernel_or_pairs"] == 5 and info["NB"] == (K + 15) // 16
    stats_close(G, c, s, *orc.normal_eq(A, b, w, t))


@pytest.mark.parametrize("K", [289, 300, 304, 305, 320, 321, 336, 337, 352, 353, 368, 369, 384, 385, 400, 401, 416, 417, 432, 433,
                               448, 449, 464, 465, 479, 480, 481, 496, 497, 511, 512])
def test_cluster_quad_kernel_statistics_all_column_block_shapes(qctx, K):
    # kernel 1QC, NB = 19 ... 32: kernel 1Q's plan on a cluster of 2 (up to 23 column blocks) or 4 workgroups of one XCD --
    # odd / even block counts, K a multiple of 16 and not, the last block nearly empty; ragged row ranges per cluster;
    # rows of the testing set and tiny / huge weights as in the reference's weighting (svd.py:35-46)
    rng = np.random.default_rng(K)
    m = 9151 + 13 * K
    A = rng.standard_normal((m, K)) * (10.0 ** rng.uniform(-3, 3, size=K))
    b = rng.standard_normal(m)
    w = rng.choice([100.0, 1.0, 1e-8], size=m, p=[0.03, 0.83, 0.14])
    t = rng.random(m) < 0.2
    G, c, s = run_stats(qctx, A, b, w, t)
    info = qctx.launch_info()
    assert info["kernel_or_pairs"] == 6 and info["NB"] == (K + 15) // 16
    stats_close(G, c, s, *orc.normal_eq(A, b, w, t))
    # the tiled kernel (what shorter systems of these widths run on) on the same rows: same statistics to rounding
    qctx.set_option("tiled", 1)
    try:
        G2, c2, s2 = run_stats(qctx, A, b, w, t)
        assert qctx.launch_info()["kernel_or_pairs"] != 6
    finally:
        qctx.set_option("tiled", 0)
    stats_close(G2, c2, s2, G, c, s)


@pytest.mark.parametrize("K,m", [(480, 367900), (320, 200003)])
def test_cluster_quad_kernel_is_bit_identical_run_to_run_and_fits_like_the_oracle(qctx, K, m):
    ctx = qctx
    # InP's width (examples/InP_JPCA2020: 367 900 x 480) at full size: fixed-order sums -> the same bits on every launch (the
    # flow control between the members of a cluster only paces them); fit through the C ABI vs a dense solve of the oracle's
    # statistics
    A, b, w = orc.synth_problem(m, K)
    ctx.upload_rows(A, b)
    ctx.set_weights(w)
    first = ctx.normal_eq()
    assert ctx.launch_info()["kernel_or_pairs"] == 6
    for _ in range(3):
        again = ctx.normal_eq()
        assert all(np.array_equal(x, y) for x, y in zip(first, again))
    beta = ctx.fit_resident(_capi.SOLVE_RIDGE, 1e-8)[0]
    G, c, s = orc.normal_eq(A, b, w)
    stats_close(*first, G, c, s)
    ref = np.linalg.solve(G + 1e-8 * np.eye(K), c)
    assert np.max(np.abs(beta - ref)) <= 1e-9 * np.max(np.abs(ref))


@pytest.mark.parametrize("m", [1, 2, 3, 4, 5, 7, 63, 64, 65, 257, 1023, 1025])
@pytest.mark.parametrize("K", [168, 256, 275])
def test_quad_kernel_tiny_and_ragged_row_counts(qctx, m, K):
    rng = np.random.default_rng(1000 + m + K)
    A = rng.standard_normal((m, K))
    b = rng.standard_normal(m)
    w = rng.uniform(0.5, 2.0, m)
    G, c, s = run_stats(qctx, A, b, w)
    if m >= 4:
        assert qctx.launch_info()["kernel_or_pairs"] == 5
    stats_close(G, c, s, *orc.normal_eq(A, b, w), tol=1e-11)


def test_quad_kernel_never_fetches_masked_rows_and_pads(qctx):
    # NaN / Inf in test rows and zero-weight rows, a leading dimension wider than the row with NaN in the padding columns
    rng = np.random.default_rng(77)
    m, K, lda = 9001, 200, 211
    Abig = np.full((m, lda), np.nan)
    A = rng.standard_normal((m, K))
    b = rng.standard_normal(m)
    w = rng.uniform(0.5, 2.0, m)
    t = rng.random(m) < 0.25
    w[rng.random(m) < 0.1] = 0.0
    A[t] = np.nan
    A[w == 0.0] = np.inf
    Abig[:, :K] = A
    qctx.upload_rows(Abig[:, :K], b)            # (a strided view: the leading dimension is lda)
    qctx.set_weights(w, (~t).astype(np.uint8))
    G, c, s = qctx.normal_eq()
    assert qctx.launch_info()["kernel_or_pairs"] == 5
    Aclean = np.where((t | (w == 0.0))[:, None], 0.0, A)
    stats_close(G, c, s, *orc.normal_eq(Aclean, b, w, t))


@pytest.mark.parametrize("K", [168, 200, 256, 275])
def test_quad_kernel_agrees_with_the_tiled_kernel_and_the_oracle(ctx, K):
    A, b, w = orc.synth_problem(60013, K)
    t = np.random.default_rng(K).random(len(b)) < 0.15
    ref = orc.normal_eq(A, b, w, t)
    got = run_stats(ctx, A, b, w, t)
    assert ctx.launch_info()["kernel_or_pairs"] == 5        # 60 013 rows: kernel 1Q by default
    stats_close(*got, *ref)
    again = ctx.normal_eq()
    assert all(np.array_equal(x, y) for x, y in zip(got, again))           # run-to-run bit-identical
    ctx.set_option("tiled", 1)
    try:
        tiled = ctx.normal_eq()
        assert ctx.launch_info()["split"] == 0                              # the tiled kernel
    finally:
        ctx.set_option("tiled", 0)
    stats_close(*tiled, *ref)
    # streaming accumulation (fsnap_normal_eq_accumulate) through kernel 1Q: two batches add up to the whole
    import torch
    total = torch.zeros(K * K + K + 3, dtype=torch.float64, device=torch.device("cuda", 0))
    h = len(b) // 2 + 3
    for lo, hi in ((0, h), (h, len(b))):
        ctx.upload_rows(A[lo:hi], b[lo:hi])
        ctx.set_weights(w[lo:hi], (~t[lo:hi]).astype(np.uint8))
        ctx.normal_eq_accumulate(total.data_ptr())
        assert ctx.launch_info()["kernel_or_pairs"] == 5
    stats_close(*ctx.download_packed(total.data_ptr(), K), *ref, tol=2e-12)


@pytest.mark.parametrize("K", [150, 176, 256, 288])
def test_quad_kernel_on_pairs_packed_in_hbm_gives_the_bits_of_the_fused_form(qctx, K):
    # option fused_pack = 0: fsnap_pack_weights_k writes the per-row pairs to HBM and kernel 1Q reads them from there (what a
    # workgroup does whose rows' pairs do not fit the LDS, and what the row-space passes do with pairs of their own): the same
    # pairs, the same arithmetic in the same order -> identical G and c
    rng = np.random.default_rng(300 + K)
    m = 30011
    A = rng.standard_normal((m, K))
    b = rng.standard_normal(m)
    w = rng.uniform(0.1, 3.0, m)
    t = rng.random(m) < 0.3
    fused = run_stats(qctx, A, b, w, t)
    assert qctx.launch_info()["kernel_or_pairs"] == 5 and qctx.launch_info()["fused_pack"] == 1
    qctx.set_option("fused_pack", 0)
    try:
        plain = qctx.normal_eq()
        assert qctx.launch_info()["kernel_or_pairs"] == 5 and qctx.launch_info()["fused_pack"] == 0
    finally:
        qctx.set_option("fused_pack", 1)
    assert np.array_equal(fused[0], plain[0]) and np.array_equal(fused[1], plain[1])
    stats_close(*plain, *orc.normal_eq(A, b, w, t))


@pytest.mark.parametrize("K,m", [(168, 80000), (275, 40000), (288, 40000)])
def test_quad_kernel_fit_matches_the_oracle_solve(ctx, K, m):
    # the whole fit: statistics from kernel 1Q; 168 columns: host mirror written by the reduction, host solve;
    # 275 / 288 columns: no mirror, the GPU factorises (DEVICE_CHOL_MIN_K = 232)
    A, b, w = orc.synth_problem(m, K)
    ctx.upload_rows(A, b)
    ctx.set_weights(w)
    beta = ctx.fit_resident(_capi.SOLVE_RIDGE, 1e-8)[0]
    assert ctx.launch_info()["kernel_or_pairs"] == 5
    G, c, s = orc.normal_eq(A, b, w)
    ref = np.linalg.solve(G + 1e-8 * np.eye(K), c)
    assert np.max(np.abs(beta - ref)) <= 1e-9 * np.max(np.abs(ref))
    again = ctx.fit_resident(_capi.SOLVE_RIDGE, 1e-8)[0]
    assert np.array_equal(beta, again)


@pytest.mark.parametrize("m", [1, 2, 3, 4, 5, 7, 8, 9, 63, 64, 65, 1023, 1025])
def test_statistics_tiny_and_ragged_row_counts(ctx, m):
    rng = np.random.default_rng(100 + m)
    A = rng.standard_normal((m, 31))
    b = rng.standard_normal(m)
    w = rng.uniform(0.5, 2.0, m)
    G, c, s = run_stats(ctx, A, b, w)
    stats_close(G, c, s, *orc.normal_eq(A, b, w), tol=1e-11)


def test_masked_rows_may_hold_garbage(ctx):
    # test rows are excluded by fancy indexing in the reference (svd.py:44-46): NaN/Inf in
    # them must not reach G
    rng = np.random.default_rng(5)
    A = rng.standard_normal((2000, 40))
    b = rng.standard_normal(2000)
    w = np.ones(2000)
    t = np.zeros(2000, dtype=bool)
    t[::7] = True
    A2, b2 = A.copy(), b.copy()
    A2[t] = np.nan
    b2[t] = np.inf
    G, c, s = run_stats(ctx, A2, b2, w, t)
    stats_close(G, c, s, *orc.normal_eq(A, b, w, t))
    assert np.isfinite(G).all()


def test_all_rows_masked_and_zero_weights(ctx):
    rng = np.random.default_rng(6)
    A = rng.standard_normal((100, 20))
    b = rng.standard_normal(100)
    G, c, s = run_stats(ctx, A, b, np.ones(100), np.ones(100, dtype=bool))
    assert not G.any() and not c.any() and s.tolist() == [0.0, 0.0, 0.0]
    G, c, s = run_stats(ctx, A, b, np.zeros(100))
    assert not G.any() and s[2] == 100


def test_strided_rows_lda_greater_than_k(ctx):
    rng = np.random.default_rng(7)
    big = rng.standard_normal((3000, 80))
    A = big[:, :50]                                 # row stride 80 doubles
    b = rng.standard_normal(3000)
    w = rng.uniform(0.1, 3, 3000)
    G, c, s = run_stats(ctx, A, b, w)
    stats_close(G, c, s, *orc.normal_eq(A, b, w))


def test_reweighting_resident_rows(ctx, ta):
    # GA use case (libmod_optimize.py:461-488): same A, b, new w per call, A stays in HBM
    A, b, w = ta
    ctx.upload_rows(A, b)
    for seed in range(3):
        w2 = w * np.random.default_rng(seed).uniform(0.5, 2.0, len(w))
        ctx.set_weights(w2)
        G, c, s = ctx.normal_eq()
        stats_close(G, c, s, *orc.normal_eq(A, b, w2))


def test_run_to_run_bit_identical(ctx, ta):
    A, b, w = ta
    G1, c1, s1 = run_stats(ctx, A, b, w)
    G2, c2, s2 = run_stats(ctx, A, b, w)
    assert np.array_equal(G1, G2) and np.array_equal(c1, c2) and np.array_equal(s1, s2)


@pytest.mark.parametrize("m", [1, 3, 4, 13, 47, 48, 49, 191, 193, 1000, 12289])
@pytest.mark.parametrize("K", [97, 128, 142])
def test_one_wave_triangle_kernel_tiny_and_ragged_row_counts(ctx, m, K):
    # kernel 1A (default for 80 < K <= 128): row-waves without rows, partial 3-chunk pipeline groups, masked tails
    rng = np.random.default_rng(7000 + 13 * m + K)
    A = rng.standard_normal((m, K))
    b = rng.standard_normal(m)
    w = rng.uniform(0.5, 2.0, m)
    t = rng.random(m) < 0.3
    ctx.set_option("short", 0)          # (kernel 1S takes systems this short by default: tests/test_gpu_short.py)
    try:
        G, c, s = run_stats(ctx, A, b, w, t)
        assert ctx.launch_info()["kernel_or_pairs"] == 3
    finally:
        ctx.set_option("short", -1)
    stats_close(G, c, s, *orc.normal_eq(A, b, w, t), tol=1e-11)


@pytest.mark.parametrize("K,m", [(129, 5003), (142, 13035), (192, 4001), (200, 3000), (257, 2049), (300, 4100), (448, 3000),
                                 (480, 6000), (1000, 3000), (1595, 2500)])
def test_general_k_tiled_kernel(ctx, K, m):
    # K > 128: ACE (142), EME (480) and quadratic SNAP (1595) widths on the tiled kernel 1T (even / odd superblock counts,
    # full / partly filled / half-empty last superblock); option tiled = 1 keeps it covered at the widths kernels 1A / 1Q /
    # 1QC take by default
    rng = np.random.default_rng(2000 + K)
    A = rng.standard_normal((m, K)) * (10.0 ** rng.uniform(-2, 2, size=K))
    b = rng.standard_normal(m)
    w = rng.choice([100.0, 1.0, 1e-8], size=m, p=[0.03, 0.83, 0.14])
    t = rng.random(m) < 0.1
    ctx.set_option("tiled", 1)
    try:
        G, c, s = run_stats(ctx, A, b, w, t)
        assert ctx.launch_info()["split"] == 0
    finally:
        ctx.set_option("tiled", 0)
    stats_close(G, c, s, *orc.normal_eq(A, b, w, t), tol=2e-12)


@pytest.mark.parametrize("K", [142, 300, 448])
def test_tiled_kernel_masked_rows_may_hold_garbage(ctx, K):
    # the tiled kernel reads packed (w_eff, w_eff b) pairs and masks rows through out-of-range load offsets: NaN / Inf
    # in A, b and w of test rows must not reach the statistics; K = 142 / 300 / 448: last superblock half empty
    # (its empty blocks are skipped) / partly filled (selects) / full
    rng = np.random.default_rng(70 + K)
    m = 9001
    A = rng.standard_normal((m, K))
    b = rng.standard_normal(m)
    w = rng.uniform(0.5, 2.0, m)
    t = rng.random(m) < 0.25
    A2, b2, w2 = A.copy(), b.copy(), w.copy()
    A2[t] = np.nan
    b2[t] = np.inf
    w2[t] = -np.inf
    ctx.set_option("tiled", 1)
    try:
        G, c, s = run_stats(ctx, A2, b2, w2, t)
        assert ctx.launch_info()["split"] == 0
    finally:
        ctx.set_option("tiled", 0)
    stats_close(G, c, s, *orc.normal_eq(A, b, w, t), tol=2e-12)
    assert np.isfinite(G).all() and np.isfinite(c).all() and np.isfinite(s).all()


@pytest.mark.parametrize("K", [31, 64, 100, 128])
def test_tiled_kernel_forced_on_small_k(ctx, K):
    A, b, w = orc.synth_problem(9001, K)
    ctx.set_option("tiled", 1)
    try:
        for nsplit in (0, 1, 7):
            ctx.set_option("nsplit", nsplit)
            G, c, s = run_stats(ctx, A, b, w)
            stats_close(G, c, s, *orc.normal_eq(A, b, w))
    finally:
        ctx.set_option("tiled", 0)
        ctx.set_option("nsplit", 0)


def test_ridge_fit_k142_ace_shape():
    # BASELINE configs[3] shape (Ta_PACE_RIDGE: 13035 x 142, alpha 1e-4, local solver)
    A, b, w = orc.synth_problem(13035, 142)
    pt, s = make_solver("RIDGE", {"RIDGE": {"alpha": 1e-4, "local_solver": 1}})
    s.perform_fit(A, b, w, trainall=True)
    ref = orc.ridge_fit(A, b, w, 1e-4, local_solver=True)
    assert np.max(np.abs(s.fit - ref)) / np.max(np.abs(ref)) < 1e-6
    assert maxrel(s.fit, ref) < 1e-6                                          # element-wise, like every other RIDGE / SVD test
    pt.free()


def test_one_context_alternates_between_device_and_host_factorisation():
    # a context that has factorised on the GPU (page-locked result block of the device Cholesky) and then fits a narrow
    # problem (page-locked mirror of the statistics, allocated on first use) must still own the first block afterwards:
    # the mirror's (re)allocation used to free it and leave the pointer behind
    c = _capi.HipContext(0)
    for K, m in ((480, 6000), (31, 5000), (480, 6000), (128, 9000), (512, 4000), (64, 3000), (480, 6000)):
        rng = np.random.default_rng(7000 + K)
        A = rng.standard_normal((m, K))
        b = rng.standard_normal(m)
        w = rng.uniform(0.5, 2.0, m)
        c.upload_rows(A, b)
        c.set_weights(w)
        beta, rank, _, _ = c.fit_resident(_capi.SOLVE_RIDGE, 1e-8)
        ref = orc.ridge_fit(A, b, w, 1e-8, local_solver=True)
        assert rank == K and np.max(np.abs(beta - ref)) / np.max(np.abs(ref)) < 1e-9, (K, m)
    c.close()


@pytest.mark.parametrize("K,m", [(257, 3001), (320, 4000), (480, 6000), (1000, 5000), (1595, 7000)])
def test_large_k_device_cholesky_matches_host_solve(ctx, K, m):
    # fsnap_solve_device factorises large systems on the GPU (blocked kernels 8a-8e); same answer as the host solver and
    # as a dense numpy solve of the same statistics
    rng = np.random.default_rng(4000 + K)
    A = rng.standard_normal((m, K)) * (10.0 ** rng.uniform(-2, 2, size=K))
    b = rng.standard_normal(m)
    w = rng.uniform(0.5, 2.0, m)
    ctx.upload_rows(A, b)
    ctx.set_weights(w)
    ptr = ctx.normal_eq_resident()
    G, c, _ = ctx.download_packed(ptr, K)
    for kind, param in ((_capi.SOLVE_RIDGE, 1e-6), (_capi.SOLVE_LSTSQ, 1e-13), (_capi.SOLVE_CHOL, 0.0)):
        ctx.set_option("device_solve", 1)          # force the GPU factorisation also below the automatic threshold
        try:
            beta_dev, rank, rcond = ctx.solve_device(kind, param, K, ptr)
        finally:
            ctx.set_option("device_solve", 0)
        assert rank == K and rcond > 1e-3
        ctx.set_option("device_solve", 2)
        try:
            beta_host, rank_h, _ = ctx.solve_device(kind, param, K, ptr)
        finally:
            ctx.set_option("device_solve", 0)
        alpha = param if kind == _capi.SOLVE_RIDGE else 0.0
        ref = np.linalg.solve(G + alpha * np.eye(K), c)
        scale = np.max(np.abs(ref))
        assert np.max(np.abs(beta_dev - beta_host)) / scale < 1e-9
        assert np.max(np.abs(beta_dev - ref)) / scale < 1e-8


@pytest.mark.parametrize("K,m", [(232, 2500), (257, 3001), (480, 6000), (1000, 5000), (1595, 7000)])
def test_device_cholesky_factor_is_reused_for_further_right_hand_sides(ctx, K, m):
    # the refinement steps of a fit solve G delta = s with the G that was just factorised on the GPU: fsnap_solve_device_rhs
    # then runs a forward (kernel 8f) and a backward sweep with the factor left on the device (option "chol_reuse", default 1)
    # instead of factorising again -- same answers as the full path and as a dense solve, for RIDGE and LSTSQ; new statistics, a
    # different shift or another buffer forget the factor
    rng = np.random.default_rng(9000 + K)
    A = rng.standard_normal((m, K)) * (10.0 ** rng.uniform(-1, 1, size=K))
    b = rng.standard_normal(m)
    w = rng.uniform(0.5, 2.0, m)
    ctx.upload_rows(A, b)
    ctx.set_weights(w)
    ctx.set_option("device_solve", 1)
    try:
        ptr = ctx.normal_eq_resident()
        G, c, _ = ctx.download_packed(ptr, K)
        for kind, param in ((_capi.SOLVE_RIDGE, 1e-6), (_capi.SOLVE_LSTSQ, 1e-13)):
            alpha = param if kind == _capi.SOLVE_RIDGE else 0.0
            M = G + alpha * np.eye(K)
            beta0, rank, rcond = ctx.solve_device(kind, param, K, ptr)
            assert rank == K
            for trial in range(3):
                rhs = rng.standard_normal(K) * np.abs(c).max()
                ctx.set_option("chol_reuse", 1)
                ctx.solve_device(kind, param, K, ptr)                       # (option changes forget the factor: factorise)
                x_reuse, rk1, rc1 = ctx.solve_device(kind, param, K, ptr, rhs=rhs)
                ctx.set_option("chol_reuse", 0)
                x_full, rk0, rc0 = ctx.solve_device(kind, param, K, ptr, rhs=rhs)
                ref = np.linalg.solve(M, rhs)
                scale = np.abs(ref).max()
                assert rk1 == rk0 == K and rc1 == rc0
                assert np.abs(x_reuse - x_full).max() <= 1e-10 * scale
                assert np.abs(x_reuse - ref).max() <= 1e-8 * scale
            # the first right-hand side again through the sweeps: the fit's own coefficients
            ctx.set_option("chol_reuse", 1)
            ctx.solve_device(kind, param, K, ptr)
            again = ctx.solve_device(kind, param, K, ptr, rhs=c)[0]
            assert np.abs(again - beta0).max() <= 1e-10 * np.abs(beta0).max()
        # new statistics behind the same pointer: the factor of the old ones must not answer
        ctx.set_option("chol_reuse", 1)
        ctx.solve_device(_capi.SOLVE_RIDGE, 1e-6, K, ptr)
        w2 = w * rng.uniform(0.2, 5.0, m)
        ctx.set_weights(w2)
        ptr2 = ctx.normal_eq_resident()
        G2, c2, _ = ctx.download_packed(ptr2, K)
        rhs = rng.standard_normal(K)
        x = ctx.solve_device(_capi.SOLVE_RIDGE, 1e-6, K, ptr2, rhs=rhs)[0]
        ref = np.linalg.solve(G2 + 1e-6 * np.eye(K), rhs)
        assert np.abs(x - ref).max() <= 1e-8 * np.abs(ref).max()
        # ... nor for another shift
        ctx.solve_device(_capi.SOLVE_RIDGE, 1e-6, K, ptr2)
        x = ctx.solve_device(_capi.SOLVE_RIDGE, 1e-2, K, ptr2, rhs=rhs)[0]
        ref = np.linalg.solve(G2 + 1e-2 * np.eye(K), rhs)
        assert np.abs(x - ref).max() <= 1e-8 * np.abs(ref).max()
    finally:
        ctx.set_option("device_solve", 0)
        ctx.set_option("chol_reuse", 1)


@pytest.mark.parametrize("K", [40, 128, 200])
def test_mirror_packed_serves_statistics_modified_in_hbm(ctx, K):
    # fsnap_mirror_packed: the multi-GPU path all-reduces the packed statistics in place; the page-locked host mirror
    # must then reflect the REDUCED buffer.  Here the "all-reduce" is a doubling on the same stream.
    import torch

    rng = np.random.default_rng(300 + K)
    m = 5000
    A = rng.standard_normal((m, K))
    b = rng.standard_normal(m)
    ctx.upload_rows(A, b)
    ctx.set_weights(np.ones(m))
    dev = torch.device("cuda", 0)
    packed = torch.zeros(K * K + K + 3, dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream(dev)
    ctx.set_stream(st.cuda_stream)
    try:
        ctx.normal_eq_async(packed.data_ptr())
        packed.mul_(2.0)                                  # stands in for the sum over two identical ranks
        ctx.mirror_packed(packed.data_ptr(), K)
        beta, rank, _ = ctx.solve_device(_capi.SOLVE_RIDGE, 0.5, K, packed.data_ptr())
        G, c, _ = orc.normal_eq(A, b, np.ones(m))
        ref = np.linalg.solve(2 * G + 0.5 * np.eye(K), 2 * c)
        assert rank == K and np.max(np.abs(beta - ref)) / np.max(np.abs(ref)) < 1e-9
        # a new kernel launch invalidates the mirror: the solve reads the fresh (undoubled) statistics
        ctx.normal_eq_async(packed.data_ptr())
        beta1, _, _ = ctx.solve_device(_capi.SOLVE_RIDGE, 0.5, K, packed.data_ptr())
        ref1 = np.linalg.solve(G + 0.5 * np.eye(K), c)
        assert np.max(np.abs(beta1 - ref1)) / np.max(np.abs(ref1)) < 1e-9
    finally:
        ctx.use_own_stream()


@pytest.mark.parametrize("K,m", [(96, 3000), (800, 4000)])
def test_solve_device_with_replacement_rhs(ctx, K, m):
    # fsnap_solve_device_rhs: G delta = s (refinement step) with G resident in HBM -- host path (K = 96) and
    # blocked GPU factorisation (K = 800)
    rng = np.random.default_rng(900 + K)
    A = rng.standard_normal((m, K))
    b = rng.standard_normal(m)
    ctx.upload_rows(A, b)
    ctx.set_weights(np.ones(m))
    ptr = ctx.normal_eq_resident()
    G, c, _ = ctx.download_packed(ptr, K)
    rhs = rng.standard_normal(K)
    x, rank, _ = ctx.solve_device(_capi.SOLVE_RIDGE, 1e-6, K, ptr, rhs=rhs)
    ref = np.linalg.solve(G + 1e-6 * np.eye(K), rhs)
    assert rank == K and np.max(np.abs(x - ref)) / np.max(np.abs(ref)) < 1e-9
    x0, _, _ = ctx.solve_device(_capi.SOLVE_RIDGE, 1e-6, K, ptr)
    assert np.max(np.abs(x0 - np.linalg.solve(G + 1e-6 * np.eye(K), c))) / np.max(np.abs(x0)) < 1e-9
    with pytest.raises(ValueError):
        ctx.solve_device(_capi.SOLVE_RIDGE, 1e-6, K, ptr, rhs=np.zeros(K + 1))


def test_svd_fit_large_k_with_refinement_matches_lstsq():
    # K = 800: statistics, blocked GPU Cholesky and the two refinement solves all stay in HBM
    rng = np.random.default_rng(800)
    m, K = 6000, 800
    A = rng.standard_normal((m, K)) * (10.0 ** rng.uniform(-2, 2, size=K))
    b = rng.standard_normal(m)
    w = rng.uniform(0.5, 2.0, m)
    pt, s = make_solver("SVD")
    s.perform_fit(A, b, w, trainall=True)
    ref = orc.svd_fit(A, b, w)
    assert np.max(np.abs(s.fit - ref) / np.abs(ref)) < 1e-6
    pt.free()


def test_large_k_device_cholesky_falls_back_when_ill_conditioned(ctx):
    # two identical columns: the scaled matrix is singular -> tiny / failed pivot on the GPU -> general host path
    rng = np.random.default_rng(77)
    K, m = 300, 2000
    A = rng.standard_normal((m, K))
    A[:, 299] = A[:, 7]
    b = rng.standard_normal(m)
    ctx.upload_rows(A, b)
    ctx.set_weights(np.ones(m))
    ptr = ctx.normal_eq_resident()
    ctx.set_option("device_solve", 1)
    try:
        beta, rank, _ = ctx.solve_device(_capi.SOLVE_LSTSQ, 1e-13, K, ptr)
        with pytest.raises(np.linalg.LinAlgError):
            ctx.solve_device(_capi.SOLVE_CHOL, 0.0, K, ptr)
    finally:
        ctx.set_option("device_solve", 0)
    assert rank == K - 1
    ref = np.linalg.lstsq(A, b, rcond=1e-13)[0]
    assert np.max(np.abs(A @ beta - A @ ref)) < 1e-8 * np.max(np.abs(b))


def test_call_order_errors(ctx):
    c2 = _capi.HipContext(0)
    try:
        with pytest.raises(_capi.FsnapError, match="no rows"):
            c2.normal_eq()
    finally:
        c2.close()


# ---------------------------------------------------------------------------------------
# stand-alone weighting kernel and GEMV
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("K", [31, 128, 55])
def test_weight_rows_bit_exact(ctx, K):
    rng = np.random.default_rng(K)
    m = 5001
    A = rng.standard_normal((m, K))
    b = rng.standard_normal(m)
    w = rng.uniform(0.1, 10, m)
    t = rng.random(m) < 0.3
    ctx.upload_rows(A, b)
    ctx.set_weights(w, (~t).astype(np.uint8))
    aw, bw = ctx.weight_rows()
    awr, bwr = orc.weight_rows_full(A, b, w, t)
    assert np.array_equal(aw, awr) and np.array_equal(bw, bwr)     # one IEEE multiply per element


def test_predict_and_sse(ctx, ta, ta_fits):
    A, b, w = ta
    beta = ta_fits["svd_all"]
    ctx.upload_rows(A, b)
    ctx.set_weights(w)
    preds, sse = ctx.predict(beta, want_preds=True, want_sse=True)
    ref = orc.predict(A, beta)
    scale = np.maximum(np.abs(A) @ np.abs(beta), 1e-300)
    assert np.max(np.abs(preds - ref) / scale) < 1e-14
    aw, bw = orc.weight_rows(A, b, w)
    assert sse == pytest.approx(np.sum((bw - aw @ beta) ** 2), rel=1e-10)


# ---------------------------------------------------------------------------------------
# solver classes through the plugin API vs the reference's own fits
# ---------------------------------------------------------------------------------------
def make_solver(name, extra=None):
    pt = ParallelTools()
    d = {"SOLVER": {"solver": name}}
    d.update(extra or {})
    cfg = Config(pt, d)
    return pt, solver_factory.solver(name, pt, cfg)


def check_fit(fit, ref, elementwise=1e-6):
    assert fit.shape == ref.shape and fit.dtype == np.float64
    assert maxrel(fit, ref) < elementwise                                 # north_star: 1e-6 relative
    assert np.max(np.abs(fit - ref)) / np.max(np.abs(ref)) < 1e-6         # norm-wise
    assert np.max(np.abs(fit - ref)) < 1e-6                               # example_checker.py:62


def test_svd_solver_matches_reference(ta, ta_fits):
    A, b, w = ta
    t = ta_fits["testing_mask"]
    pt, s = make_solver("SVD")
    s.perform_fit(A, b, w, trainall=True)
    check_fit(s.fit, ta_fits["svd_all"])
    check_fit(s.fit, ta_fits["snapcoeff"])
    s.perform_fit(A, b, w[~t], fs_dict={"Testing": t.tolist()})
    check_fit(s.fit, ta_fits["svd_mask"])
    pt.free()


def test_svd_solver_shared_array_path(ta, ta_fits):
    A, b, w = ta
    t = ta_fits["testing_mask"]
    pt, s = make_solver("SVD")
    m, K = A.shape
    pt.create_shared_array("a", m, K)
    pt.create_shared_array("b", m)
    pt.create_shared_array("w", m)
    pt.shared_arrays["a"].array[:] = A
    pt.shared_arrays["b"].array[:] = b
    pt.shared_arrays["w"].array[:] = w
    pt.fitsnap_dict["Testing"] = t.tolist()
    s.perform_fit()
    check_fit(s.fit, ta_fits["svd_mask_shared"])
    pt.free()


def test_svd_transpose_trick_flag(ta, ta_fits):
    A, b, w = ta
    pt, s = make_solver("SVD", {"EXTRAS": {"apply_transpose": 1}})
    s.perform_fit(A, b, w, trainall=True)
    check_fit(s.fit, ta_fits["svd_transpose_all"])
    pt.free()


@pytest.mark.parametrize("tag,alpha", [("1e-8", 1e-8), ("1e-4", 1e-4)])
@pytest.mark.parametrize("local", [0, 1])
def test_ridge_solver_matches_reference(ta, ta_fits, tag, alpha, local):
    A, b, w = ta
    t = ta_fits["testing_mask"]
    pt, s = make_solver("RIDGE", {"RIDGE": {"alpha": alpha, "local_solver": local}})
    kind = "local" if local else "sklearn"
    # The reference's Local_Ridge inverts the unscaled, ill-conditioned normal matrix explicitly
    # (regressor.py:15); its own answer differs ELEMENTWISE from the reference's sklearn ridge on
    # the same system by up to 1.0e-6 (ta golden, alpha 1e-4, masked).  Against that solver the
    # elementwise bar is 1e-6 plus the reference's own local-vs-sklearn disagreement.
    for m in ("all", "mask"):
        ref = ta_fits[f"ridge_{kind}_{tag}_{m}"]
        tol = 1e-6 + (maxrel(ta_fits[f"ridge_local_{tag}_{m}"], ta_fits[f"ridge_sklearn_{tag}_{m}"]) if local else 0.0)
        if m == "all":
            s.perform_fit(A, b, w, trainall=True)
        else:
            s.perform_fit(A, b, w[~t], fs_dict={"Testing": t.tolist()})
        check_fit(s.fit, ref, elementwise=tol)
        check_fit(s.fit, ta_fits[f"ridge_sklearn_{tag}_{m}"])               # the accurate reference solver: plain 1e-6
    pt.free()


def test_second_golden_set(ta):
    import os
    from conftest import GOLDEN
    A, b, w = ta
    d = np.load(os.path.join(GOLDEN, "ta_xyz_delta.npz"))
    pt, s = make_solver("SVD")
    s.perform_fit(A + d["dA"], b + d["db"], w + d["dw"], trainall=True)
    check_fit(s.fit, d["svd_all"])
    pt.free()


@pytest.mark.parametrize("key,mask,extra", [
    ("ard_class_all", False, {}),
    ("ard_class_mask", True, {}),
    ("ard_class_direct", False, {"ARD": {"directmethod": 1}}),
    ("ard_class_scaled", True, {"ARD": {"scap": 1.0e-2, "scai": 1.0e-4, "logcut": 1.0}}),
])
def test_ard_solver_matches_the_reference_class(ta, ta_fits, key, mask, extra):
    # goldens: the reference's ARD class itself (ard.py:15-49; make_golden.py forwards ARDRegression's renamed
    # iteration keyword).  The solver iterates on the GPU statistics (K x K inverses of the column-equilibrated matrix),
    # sklearn on the rows (pinvh of the unscaled one): equal support; north_star's 1e-6 element-wise on the kept
    # coefficients against the SAME iteration in extended precision (oracle.ard_fit_extended) -- the class's own float64
    # vectors are 2e-5 ... 3e-4 from that yardstick, and the distance to them is theirs, not ours
    A, b, w = ta
    pt, s = make_solver("ARD", extra)
    m, K = A.shape
    for name, arr in (("a", A), ("b", b), ("w", w)):
        pt.create_shared_array(name, m, K if name == "a" else 1)
        pt.shared_arrays[name].array[:] = arr
    t = ta_fits["testing_mask"] if mask else None
    pt.fitsnap_dict["Testing"] = t.tolist() if mask else [False] * m
    s.perform_fit()
    ref = ta_fits[key]
    assert np.array_equal(s.fit != 0, ref != 0)
    nz = ref != 0
    sec = s.config.sections["ARD"]
    ext, ext_iter = orc.ard_fit_extended(A, b, w, testing=t, directmethod=bool(sec.directmethod), scap=sec.scap, scai=sec.scai,
                                         logcut=sec.logcut)

    def elementwise(x, y):
        return np.max(np.abs(x[nz] - y[nz]) / np.abs(y[nz]))

    assert np.array_equal(ext != 0, nz) and s.n_iter_ == ext_iter
    assert elementwise(s.fit, ext) < 1e-6
    assert elementwise(s.fit, ref) < 1.01 * elementwise(ref, ext) + 1e-6
    assert np.max(np.abs(s.fit - ref)) < 2e-5 * np.max(np.abs(ref))
    pt.free()


def test_ard_apply_transpose_on_the_golden_rows(ta, ta_fits):
    # ard.py:22-24 on Ta: the regression runs on (G, c) with kappa(G) = 7e10 -- the iteration never converges, in the
    # reference class either (1000 iterations), and its coefficients move by percents under 1e-15 perturbations of G.
    # What is reproducible is the regression's own residual |G x - c| / |c|: ours must be as small as the class's
    A, b, w = ta
    pt, s = make_solver("ARD", {"EXTRAS": {"apply_transpose": 1}})
    m, K = A.shape
    for name, arr in (("a", A), ("b", b), ("w", w)):
        pt.create_shared_array(name, m, K if name == "a" else 1)
        pt.shared_arrays[name].array[:] = arr
    pt.fitsnap_dict["Testing"] = [False] * m
    s.perform_fit()
    G, c, _ = orc.normal_eq(A, b, w)
    ref = ta_fits["ard_class_transpose"]
    res = np.linalg.norm(G @ s.fit - c) / np.linalg.norm(c)
    res_ref = np.linalg.norm(G @ ref - c) / np.linalg.norm(c)
    assert res <= 5.0 * res_ref and res_ref < 1e-5
    pt.free()


@pytest.mark.parametrize("key,mask,extra", [
    ("lasso_class_all", False, {}),
    ("lasso_class_mask", True, {}),
    ("lasso_class_alpha1e-2_mask", True, {"LASSO": {"alpha": 1.0e-2}}),
    ("lasso_class_alpha1_all", False, {"LASSO": {"alpha": 1.0}}),
    ("lasso_class_alpha1_iter50_all", False, {"LASSO": {"alpha": 1.0, "max_iter": 50}}),
    ("lasso_class_transpose", False, {"LASSO": {"alpha": 1.0e-2}, "EXTRAS": {"apply_transpose": 1}}),
])
def test_lasso_solver_matches_the_reference_class(ta, ta_fits, key, mask, extra):
    # goldens: the reference's LASSO class (lasso.py:15-29).  Statistics from the GPU, scikit-learn's coordinate descent
    # on them inside the library: same support, north_star's 1e-6 on the values (1e-5 norm-wise for the transposed
    # problem, whose Gram matrix squares kappa(G) = 7e10 once more)
    A, b, w = ta
    pt, s = make_solver("LASSO", extra)
    m, K = A.shape
    for name, arr in (("a", A), ("b", b), ("w", w)):
        pt.create_shared_array(name, m, K if name == "a" else 1)
        pt.shared_arrays[name].array[:] = arr
    pt.fitsnap_dict["Testing"] = ta_fits["testing_mask"].tolist() if mask else [False] * m
    s.perform_fit()
    ref = ta_fits[key]
    assert np.array_equal(s.fit != 0, ref != 0)
    tol = 1e-5 if "transpose" in key else 1e-6
    assert np.max(np.abs(s.fit - ref)) <= tol * np.max(np.abs(ref))
    if "transpose" not in key:
        nz = ref != 0
        assert np.max(np.abs(s.fit[nz] - ref[nz]) / np.abs(ref[nz])) < 1e-6
    assert 1 <= s.n_iter_ <= s.config.sections["LASSO"].max_iter
    pt.free()


def test_error_analysis_all_rows(ta, ta_fits):
    # '*ALL' rows of the committed Ta_metrics.md through Solver.error_analysis (GPU GEMV)
    A, b, w = ta
    pt, s = make_solver("SVD")
    s.perform_fit(A, b, w, trainall=True)
    m = len(b)
    row_type = ["Energy"] * 363 + ["Force"] * 12672 + ["Stress"] * 2178
    fs = {"Groups": ["g"] * m, "Testing": [False] * m, "Row_Type": row_type}
    s.error_analysis(A, b, w, fs)
    for wi, wt in enumerate(("Unweighted", "weighted")):
        for ri, rt in enumerate(("Energy", "Force", "Stress")):
            n, mae, rmse, rsq = ta_fits["metrics_all"][wi * 3 + ri]
            row = s.errors.loc[("*ALL", wt, "Training", rt)]
            assert row["ncount"] == n
            assert row["mae"] == pytest.approx(mae, rel=6e-6)
            assert row["rmse"] == pytest.approx(rmse, rel=6e-6)
            assert row["rsq"] == pytest.approx(rsq, abs=6e-6)
    pt.free()


# ---------------------------------------------------------------------------------------
# BASELINE.json full size (10^6 x 128): size-independent properties + fit vs the oracle
# ---------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def big():
    return orc.synth_problem(1_000_000, 128)


def test_full_size_linearity_and_parity(ctx, big):
    A, b, w = big
    m = len(b)
    G, c, s = run_stats(ctx, A, b, w)
    # (1) additivity over a row split: stats(top) + stats(bottom) == stats(all)
    h = 499_999
    G1, c1, s1 = run_stats(ctx, A[:h], b[:h], w[:h])
    G2, c2, s2 = run_stats(ctx, A[h:], b[h:], w[h:])
    d = np.sqrt(np.diag(G))
    assert np.max(np.abs(G1 + G2 - G) / (d[:, None] * d[None, :])) < 1e-12
    assert s1[2] + s2[2] == s[2] == m
    # (2) masking a row == zero weight on that row == deleting it
    t = orc.synth_testing_mask(m)
    Gm, cm, sm = run_stats(ctx, A, b, w, t)
    Gz, cz, sz = run_stats(ctx, A, b, np.where(t, 0.0, w))
    assert np.max(np.abs(Gm - Gz) / (d[:, None] * d[None, :])) < 1e-13 and sm[2] == (~t).sum()
    # (3) weight scaling: stats(2w) == 4 stats(w) exactly (power-of-two scaling is exact in fp64)
    G4, c4, s4 = run_stats(ctx, A, b, 2.0 * w)
    assert np.array_equal(G4, 4.0 * G) and np.array_equal(c4, 4.0 * c)
    # (4) against the oracle's BLAS
    stats_close(G, c, s, *orc.normal_eq(A, b, w))
    # (5) RIDGE fit through the plugin API vs the oracle's restatement of the reference
    pt, sol = make_solver("RIDGE", {"RIDGE": {"alpha": 1e-8}})
    sol.perform_fit(A, b, w, trainall=True)
    ref = orc.ridge_fit(A, b, w, 1e-8)
    assert maxrel(sol.fit, ref) < 1e-6
    pt.free()


@pytest.mark.parametrize("masked", [False, True], ids=["all_train", "ten_percent_testing"])
def test_full_size_svd_and_ridge_variants_match_the_reference_solvers(big, masked):
    # SURVEY 8(d)'s variants of BASELINE configs[1] at 10^6 x 128: the reference's DEFAULT solver (SVD = lstsq on the
    # weighted training rows, svd.py:44-54) and RIDGE with the reference's default alpha = 1e-4
    # (io/sections/solver_sections/ridge.py:13), with all rows training and with the 10 % testing mask; tolerance 1e-6
    # relative per coefficient (north_star).  The weights go in as the reference passes them: one per TRAINING row.
    A, b, w = big
    t = orc.synth_testing_mask(len(b)) if masked else None
    fsd = {"Testing": t.tolist()} if masked else None
    w_train = w[~t] if masked else w
    pt, sol = make_solver("SVD")
    sol.perform_fit(A, b, w_train, fs_dict=fsd, trainall=not masked)
    assert sol.last_row_space is None and sol.last_rank == 128       # well conditioned: statistics + refinement, not the row-space solve
    assert sol.last_refine_steps <= sol.refine_steps
    assert maxrel(sol.fit, orc.svd_fit(A, b, w, t)) < 1e-6
    pt.free()
    for alpha in (1e-4, 1e-8):
        pt, sol = make_solver("RIDGE", {"RIDGE": {"alpha": alpha}})
        sol.perform_fit(A, b, w_train, fs_dict=fsd, trainall=not masked)
        assert maxrel(sol.fit, orc.ridge_fit(A, b, w, alpha, testing=t)) < 1e-6
        pt.free()


# ---------------------------------------------------------------------------------------
# K x K solve on the device (fsnap_solve_device) vs the host solver
# ---------------------------------------------------------------------------------------
def test_device_solve_falls_back_for_hard_systems(ctx, ta, ta_fits):
    import torch
    A, b, w = ta
    ctx.set_option("device_solve", 1)
    # ill-conditioned after scaling (min pivot 4e-5): host path with refinement, same answer as before
    ctx.upload_rows(A, b)
    ctx.set_weights(w)
    ptr = ctx.normal_eq_resident()
    beta, rank, _ = ctx.solve_device(_capi.SOLVE_LSTSQ, 1e-13, 31, ptr)
    assert rank == 31 and maxrel(beta, ta_fits["svd_all"]) < 1e-6
    # zero column -> beta_j = 0 (lstsq minimum norm); indefinite -> LinAlgError; NaN -> ValueError
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(9)
    X = rng.standard_normal((200, 6))
    X[:, 2] = 0.0
    y = rng.standard_normal(200)
    pk = torch.tensor(np.concatenate([(X.T @ X).ravel(), X.T @ y, np.zeros(3)]), device=dev)
    beta, rank, _ = ctx.solve_device(_capi.SOLVE_LSTSQ, 1e-13, 6, pk.data_ptr())
    assert rank == 5 and beta[2] == 0.0
    pk = torch.tensor(np.concatenate([np.array([[1.0, 2.0], [2.0, 1.0]]).ravel(), np.ones(2), np.zeros(3)]), device=dev)
    with pytest.raises(np.linalg.LinAlgError):
        ctx.solve_device(_capi.SOLVE_CHOL, 0.0, 2, pk.data_ptr())
    pk = torch.tensor(np.concatenate([np.array([[np.nan, 0.0], [0.0, 1.0]]).ravel(), np.ones(2), np.zeros(3)]), device=dev)
    with pytest.raises(ValueError):
        ctx.solve_device(_capi.SOLVE_RIDGE, 1e-8, 2, pk.data_ptr())
    ctx.set_option("device_solve", 0)


def test_collective_path_on_one_gpu(ta, ta_fits):
    # torch.distributed/nccl (= RCCL) process group of ONE rank: exercises the multi-GPU code path
    # (async kernel into a torch tensor on torch's stream -> all_reduce -> D2H -> rank-0 solve)
    import torch.distributed as dist
    A, b, w = ta
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29617", rank=0, world_size=1)
    try:
        pt = ParallelTools(comm="torch")
        pt._size = 2          # force the collective branch; the group itself has one member
        cfg = Config(pt, {"SOLVER": {"solver": "RIDGE"}, "RIDGE": {"alpha": 1e-8}})
        s = solver_factory.solver("RIDGE", pt, cfg)
        s.perform_fit(A, b, w, trainall=True)
        check_fit(s.fit, ta_fits["ridge_sklearn_1e-8_all"])
        G, c, sc = s.last_statistics
        stats_close(G, c, sc, *orc.normal_eq(A, b, w))
        # error analysis through the collective branch: every rank reduces its rows to per-group sums on its GPU,
        # the small tables are gathered and pooled on rank 0 -- same table as the single-process evaluation
        m = len(b)
        fsd = {"Groups": [f"g{(i // 43) % 5}" for i in range(m)], "Testing": [bool((i // 43) % 7 == 0) for i in range(m)],
               "Row_Type": ["Energy"] * 363 + ["Force"] * 12672 + ["Stress"] * 2178}
        fit = s.fit.copy()
        s.error_analysis(A, b, w, fsd)
        multi_errors = s.errors.copy()
        pt.free()
        pt1, s1 = make_solver("RIDGE", {"RIDGE": {"alpha": 1e-8}})
        s1.fit = fit
        s1.device_error_stats = False                      # the reference's pandas evaluation of the GPU predictions
        s1.error_analysis(A, b, w, fsd)
        from pandas.testing import assert_frame_equal
        assert_frame_equal(multi_errors, s1.errors, check_exact=False, rtol=1e-9, atol=1e-12)
        pt1.free()
    finally:
        dist.destroy_process_group()


def test_anl_solver_matches_reference(ta, ta_fits, tmp_path, monkeypatch):
    # fitsnap3lib/solvers/anl.py: posterior mean and covariance from the same GPU statistics.
    # The reference inverts the UNSCALED normal matrix with pinv (cond ~7e10): its own answer moves by
    # ~4e-8 under a 1e-16 perturbation of G, so 1e-6 is the meaningful bar here too.
    A, b, w = ta
    monkeypatch.chdir(tmp_path)          # the class writes covariance.npy / mean.npy like the reference
    pt, s = make_solver("ANL", {"SOLVER": {"solver": "ANL", "nsam": 0, "cov_nugget": 1.0e-10}})
    s.perform_fit(A, b, w, trainall=True)
    check_fit(s.fit, ta_fits["anl_fit"])
    ref = ta_fits["anl_cov"]
    dscale = np.sqrt(np.abs(np.diag(ref)))
    assert np.max(np.abs(s.cov - ref) / (dscale[:, None] * dscale[None, :])) < 1e-5
    assert (tmp_path / "covariance.npy").exists() and (tmp_path / "mean.npy").exists()
    pt.free()


def test_error_analysis_per_group_rows_match_reference(ta, ta_fits):
    # Solver.error_analysis (solver.py:137-435) with group labels: every (group, weighting, train/test,
    # row type) row vs the reference's own DataFrame (goldens: tests/golden/make_golden.py)
    A, b, w = ta
    t = ta_fits["testing_mask"]
    m = len(b)
    row_type = ["Energy"] * 363 + ["Force"] * 12672 + ["Stress"] * 2178
    fsd = {"Groups": [str(g) for g in ta_fits["ea_groups"]], "Testing": t.tolist(), "Row_Type": row_type}
    pt, s = make_solver("SVD")
    s.perform_fit(A, b, w[~t], fs_dict=fsd)
    s.error_analysis(A, b, w, fsd)
    err = s.errors
    idx = ["|".join(str(x) for x in ix) for ix in err.index]
    assert idx == [str(x) for x in ta_fits["ea_index"]]
    ours = err[["ncount", "mae", "rmse", "rsq"]].to_numpy(dtype=np.float64)
    ref = ta_fits["ea_values"]
    assert np.array_equal(ours[:, 0], ref[:, 0])
    # mae / rmse to 1e-6 relative (coefficients agree to ~1e-7); rsq absolute
    assert np.max(np.abs(ours[:, 1:3] - ref[:, 1:3]) / np.abs(ref[:, 1:3])) < 1e-6
    assert np.nanmax(np.abs(ours[:, 3] - ref[:, 3])) < 1e-6
    pt.free()


def test_randomised_shapes_masks_and_extreme_values(ctx):
    # 40 random problems: ragged m, every K class, heavy masking, weights spanning 40 decades,
    # denormal / huge descriptor magnitudes, negative zero, lda > K
    rng = np.random.default_rng(20250926)
    for it in range(40):
        K = int(rng.choice([1, 3, 16, 30, 31, 55, 80, 81, 96, 110, 127, 128, 129, 200]))
        m = int(rng.integers(1, 6000))
        scale = 10.0 ** rng.uniform(-150, 100, size=K) if it % 4 == 0 else 10.0 ** rng.uniform(-3, 3, size=K)
        big = rng.standard_normal((m, K + int(rng.integers(0, 9)))) if it % 3 == 0 else None
        A = (big[:, :K] if big is not None else rng.standard_normal((m, K))) * scale
        if it % 5 == 0:
            A[rng.random((m, K)) < 0.3] = -0.0
        b = rng.standard_normal(m) * 10.0 ** rng.uniform(-5, 5)
        w = 10.0 ** rng.uniform(-20, 20, size=m) if it % 2 else rng.choice([0.0, 1.0, 467.0, 1e-9], size=m)
        t = rng.random(m) < rng.choice([0.0, 0.1, 0.9, 1.0])
        G, c, s = run_stats(ctx, A, b, w, t)
        with np.errstate(over="ignore", invalid="ignore"):
            Gr, cr, sr = orc.normal_eq(A, b, w, t)
        fin = np.isfinite(Gr).all() and np.isfinite(cr).all() and np.isfinite(sr).all()
        if not fin:          # overflow to inf in both implementations
            assert not (np.isfinite(G).all() and np.isfinite(c).all() and np.isfinite(s).all())
            continue
        d = np.sqrt(np.maximum(np.diag(Gr), 1e-300))
        ok = d > 1e-150      # products below the denormal range flush differently in different summation orders
        if ok.any():
            assert np.max(np.abs(G - Gr)[np.ix_(ok, ok)] / (d[ok][:, None] * d[ok][None, :])) < 1e-11, (it, m, K)
        assert s[2] == sr[2], (it, m, K)


# ---------------------------------------------------------------------------------------
# iterative refinement with the row-space residual (fsnap_residual_rhs)
# ---------------------------------------------------------------------------------------
def test_residual_rhs_matches_oracle(ctx, ta, ta_fits):
    A, b, w = ta
    t = ta_fits["testing_mask"]
    beta = ta_fits["ridge_sklearn_1e-4_mask"]          # any vector that is not the LS solution
    ctx.upload_rows(A, b)
    ctx.set_weights(w, (~t).astype(np.uint8))
    s, sse = ctx.residual_rhs(beta, want_sse=True)
    aw, bw = orc.weight_rows(A, b, w, t)
    r = bw - aw @ beta
    ref = aw.T @ r
    scale = np.abs(aw).T @ np.abs(r)
    # r = bw - aw @ beta cancels four digits on these rows: the rounding of the predictions (K eps |aw| |beta| per row, in
    # numpy's sum as much as in the kernel's) shows up in s undamped -- the bar is the error of s given exact
    # predictions (1e-13 of |aw|^T |r|) plus a twentieth of that bound
    K = A.shape[1]
    rounding = 0.05 * K * np.finfo(float).eps * (np.abs(aw).T @ (np.abs(aw) @ np.abs(beta)))
    assert np.max(np.abs(s - ref) / (1e-13 * scale + rounding)) < 1.0
    assert sse == pytest.approx(r @ r, rel=1e-12)


@pytest.mark.parametrize("K", [40, 128, 200])
def test_residual_rhs_general_k(ctx, K):
    A, b, w = orc.synth_problem(5003, K)
    beta = np.random.default_rng(K).standard_normal(K)
    ctx.upload_rows(A, b)
    ctx.set_weights(w)
    s, _ = ctx.residual_rhs(beta)
    aw, bw = orc.weight_rows(A, b, w)
    r = bw - aw @ beta
    assert np.max(np.abs(s - aw.T @ r) / (np.abs(aw).T @ np.abs(r))) < 1e-13


@pytest.mark.parametrize("K,m", [(7, 1003), (31, 15213), (64, 9001), (110, 20011), (128, 50001), (142, 13035), (200, 7001),
                                 (256, 4099), (257, 3001), (275, 9001), (288, 5003)])
def test_one_pass_residual_rhs(ctx, K, m):
    # fsnap_residual_rhs for K <= 288: kernels 4 + 7 fused, every row read once (option fused_residual: 1 = the default
    # form, 0 = the two-kernel form that wider systems take).  Both against the
    # oracle's s = aw^T (bw - aw beta) and SSE on the training rows; NaN / Inf in A, b, w of test rows reach nothing in the
    # fused forms (the reference drops those rows by fancy indexing, svd.py:44-46).
    rng = np.random.default_rng(6000 + K)
    A, b, w = orc.synth_problem(m, K)
    t = rng.random(m) < 0.2
    beta = rng.standard_normal(K) * 0.1
    aw, bw = orc.weight_rows(A, b, w, t)
    r = bw - aw @ beta
    ref, scale = aw.T @ r, np.abs(aw).T @ np.abs(r)
    A2, b2, w2 = A.copy(), b.copy(), w.copy()
    A2[t] = np.nan
    b2[t] = np.inf
    w2[t] = -np.inf
    got = {}
    try:
        for mode in (1, 0):
            ctx.set_option("fused_residual", mode)
            dirty = mode != 0                      # the two-kernel form multiplies masked rows by u = 0: finite rows only
            ctx.upload_rows(A2 if dirty else A, b2 if dirty else b)
            ctx.set_weights(w2 if dirty else w, (~t).astype(np.uint8))
            s, sse = ctx.residual_rhs(beta, want_sse=True)
            assert np.isfinite(s).all()
            assert np.max(np.abs(s - ref) / scale) < 1e-13
            assert abs(sse - r @ r) <= 1e-12 * (r @ r)
            got[mode] = s
    finally:
        ctx.set_option("fused_residual", 1)
    assert np.max(np.abs(got[1] - got[0]) / scale) < 1e-14


def test_refinement_recovers_lstsq_accuracy(ta, ta_fits):
    # golden Ta set: plain normal equations 7e-8 from the reference SVD; refined: < 1e-10
    A, b, w = ta
    pt, s = make_solver("SVD")
    s.refine_steps = 0
    s.perform_fit(A, b, w, trainall=True)
    plain = maxrel(s.fit, ta_fits["svd_all"])
    s.refine_steps = 2
    s.perform_fit(A, b, w, trainall=True)
    refined = maxrel(s.fit, ta_fits["svd_all"])
    assert plain < 1e-6 and refined < 1e-10 and refined < plain
    pt.free()


@pytest.mark.parametrize("kappa", [1e5, 1e6, 1e7])
def test_refinement_on_ill_conditioned_problem(kappa):
    # kappa(A) up to 1e7: kappa^2 eps ~ 1e-2 for the plain normal equations; lstsq (the reference,
    # svd.py:54) is accurate to ~kappa eps.  Two refinement steps must land within 1e-6 of it.
    rng = np.random.default_rng(int(np.log10(kappa)))
    m, K = 20000, 40
    U, _ = np.linalg.qr(rng.standard_normal((m, K)))
    V, _ = np.linalg.qr(rng.standard_normal((K, K)))
    X = (U * np.logspace(0, -np.log10(kappa), K)) @ V.T
    y = X @ rng.standard_normal(K) + 1e-6 * rng.standard_normal(m)
    w = np.ones(m)
    ref = orc.svd_fit(X, y, w)
    pt, s = make_solver("SVD")
    s.perform_fit(X, y, w, trainall=True)
    assert np.max(np.abs(s.fit - ref)) / np.max(np.abs(ref)) < 1e-6
    pt.free()


def test_error_stats_kernel_matches_numpy(ctx):
    # fsnap_error_stats: the ten per-category sums vs a direct numpy evaluation (two-pass centred sums)
    rng = np.random.default_rng(321)
    m, K, ncat = 50021, 40, 37
    A = rng.standard_normal((m, K))
    b = rng.standard_normal(m) * 50 + 1000.0          # large mean: the centred sums must not cancel
    w = rng.choice([0.0, 1.0, 1e-3, 250.0], size=m)
    beta = rng.standard_normal(K)
    cat = rng.integers(-1, ncat, size=m).astype(np.int32)   # -1 = row not in any category
    ctx.upload_rows(A, b)
    ctx.set_weights(w)
    st = ctx.error_stats(beta, cat, ncat)
    r = b - A @ beta
    for c in range(ncat):
        sel = cat == c
        t, ww, rr = b[sel], w[sel], r[sel]
        n, nw = sel.sum(), np.count_nonzero(ww)
        ref = [n, nw, t.sum(), (ww * t).sum(), np.abs(rr).sum(), (rr ** 2).sum(), ((t - t.sum() / n) ** 2).sum(),
               np.abs(ww * rr).sum(), ((ww * rr) ** 2).sum(), ((ww * t - (ww * t).sum() / max(nw, 1)) ** 2).sum()]
        assert st[c, 0] == n and st[c, 1] == nw
        assert np.allclose(st[c, 2:], ref[2:], rtol=1e-11, atol=0)


def test_error_analysis_device_and_pandas_paths_agree(ta, ta_fits):
    A, b, w = ta
    t = ta_fits["testing_mask"]
    row_type = ["Energy"] * 363 + ["Force"] * 12672 + ["Stress"] * 2178
    fsd = {"Groups": [str(g) for g in ta_fits["ea_groups"]], "Testing": t.tolist(), "Row_Type": row_type}
    tables = []
    for device in (True, False):
        pt, s = make_solver("SVD")
        s.device_error_stats = device
        s.perform_fit(A, b, w[~t], fs_dict=fsd)
        s.error_analysis(A, b, w, fsd)
        tables.append(s.errors)
        pt.free()
    assert list(tables[0].index) == list(tables[1].index)
    x = tables[0][["ncount", "mae", "rmse", "rsq"]].to_numpy(dtype=np.float64)
    y = tables[1][["ncount", "mae", "rmse", "rsq"]].to_numpy(dtype=np.float64)
    ok = np.isclose(x, y, rtol=1e-9, atol=1e-300) | (np.isnan(x) & np.isnan(y)) | (np.isinf(x) & np.isinf(y))
    assert ok.all()


@pytest.mark.parametrize("K", [100, 128, 142])
def test_one_wave_triangle_kernel_masked_rows_may_hold_garbage(ctx, K):
    # kernel 1A applies the row mask through out-of-range load offsets: NaN / Inf in A, b AND w of test rows must not
    # reach the statistics (the reference drops those rows by fancy indexing, svd.py:44-46)
    rng = np.random.default_rng(50 + K)
    m = 20003
    A = rng.standard_normal((m, K))
    b = rng.standard_normal(m)
    w = rng.uniform(0.5, 2.0, m)
    t = rng.random(m) < 0.25
    A2, b2, w2 = A.copy(), b.copy(), w.copy()
    A2[t] = np.nan
    b2[t] = np.inf
    w2[t] = -np.inf
    ctx.set_option("short", 0)          # (kernel 1S takes systems this short by default: tests/test_gpu_short.py has its own)
    try:
        G, c, s = run_stats(ctx, A2, b2, w2, t)
        assert ctx.launch_info()["kernel_or_pairs"] == 3
    finally:
        ctx.set_option("short", -1)
    stats_close(G, c, s, *orc.normal_eq(A, b, w, t))
    assert np.isfinite(G).all() and np.isfinite(c).all() and np.isfinite(s).all()


def test_one_wave_triangle_kernel_strided_rows(ctx):
    # rows bound in place (fsnap_bind_rows): odd leading dimension, first column at a 24-byte offset -> rows are only
    # 8-byte aligned and the 16-byte buffer loads straddle them
    import torch
    rng = np.random.default_rng(8)
    big = rng.standard_normal((9001, 131))
    A = np.ascontiguousarray(big[:, 3:103])         # K = 100 view of the 131-wide rows
    b = rng.standard_normal(9001)
    w = rng.uniform(0.1, 3, 9001)
    dev = torch.device("cuda", 0)
    dbig = torch.from_numpy(big).to(dev)
    db = torch.from_numpy(b).to(dev)
    ctx.bind_rows(dbig.data_ptr() + 3 * 8, 9001, 100, 131, db.data_ptr())
    ctx.set_weights(w)
    ref = orc.normal_eq(A, b, w)
    for short, kernel in ((0, 3), (-1, 7)):         # kernel 1A, then kernel 1S (the default of a system this short)
        ctx.set_option("short", short)
        try:
            G, c, s = ctx.normal_eq()
            assert ctx.launch_info()["kernel_or_pairs"] == kernel
        finally:
            ctx.set_option("short", -1)
        stats_close(G, c, s, *ref)


@pytest.mark.parametrize("K,m", [(96, 30011), (110, 1772), (128, 250003), (142, 13035), (144, 70001),
                                 (31, 15213), (31, 400003), (16, 9001), (55, 120001), (80, 60007), (1, 5000)])
def test_fused_packing_gives_the_bits_of_the_packing_kernel(K, m):
    # kernels 1A (80 < K <= 144) and 1P (K <= 80) form (w_eff, w_eff b) of their rows in LDS themselves (option fused_pack, default) instead of reading the pairs
    # fsnap_pack_weights_k wrote to HBM: the same numbers reach the same instructions in the same order -> G and c carry
    # the same bits; the three b-only scalars are summed per row-wave instead of per packing workgroup (same to rounding,
    # the training-row count exactly).  NaN / Inf in b and w of masked rows stay out in both forms.
    rng = np.random.default_rng(4000 + K)
    A, b, w = orc.synth_problem(m, K)
    t = rng.random(m) < 0.2
    b2, w2 = b.copy(), w.copy()
    b2[t] = np.nan
    w2[t] = np.inf
    ref = orc.normal_eq(A, b, w, t)
    c = _capi.HipContext(0)
    got = []
    for fused in (1, 0):
        c.set_option("fused_pack", fused)
        got.append(run_stats(c, A, b2, w2, t))
        info = c.launch_info()
        short = 80 < K <= 144 and m <= (112 if K > 128 else 128) * 384          # kernel 1S's default range on 256 CUs
        assert info["kernel_or_pairs"] == (7 if short else 3 if K > 80 else 4) and info["fused_pack"] == fused
        stats_close(*got[-1], *ref)
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1])
    assert got[0][2][2] == got[1][2][2]
    assert np.allclose(got[0][2], got[1][2], rtol=1e-13, atol=0)
    # a second weight set on the resident rows: nothing stale survives in the LDS / HBM pairs
    w3 = w * rng.uniform(0.5, 2.0, m)
    for fused in (1, 0):
        c.set_option("fused_pack", fused)
        c.set_weights(w3, (~t).astype(np.uint8))
        stats_close(*c.normal_eq(), *orc.normal_eq(A, b, w3, t))
    c.close()


@pytest.mark.parametrize("mode,m,K,pad", [(2, 70001, 128, 0), (2, 50003, 110, 7), (1, 300007, 128, 0), (1, 270001, 120, 5)])
def test_row_upload_paths_deliver_the_same_rows(mode, m, K, pad):
    # fsnap_upload_rows through the page-locked double buffer (option staged_upload = 2; 1 = double buffer that may hand the
    # rest to the pageable copy after looking at its first two slots): the rows on the device are the caller's rows,
    # bit for bit, also when a leading dimension wider than the row is packed on the way
    rng = np.random.default_rng(10 * mode + K)
    big = rng.standard_normal((m, K + pad))
    A = big[:, :K]                                   # a view with lda = K + pad
    b = rng.standard_normal(m)
    c = _capi.HipContext(0)
    c.set_option("staged_upload", mode)
    c.upload_rows(A, b)
    A2, b2, _ = c.download_rows(want_w=False)
    assert np.array_equal(A2, A) and np.array_equal(b2, b)
    tim = c.timing()
    assert tim["upload_ms"] > 0.0
    if mode == 2:
        assert tim["upload_staged"]
    c.set_weights(np.ones(m))
    G, cc, sc = c.normal_eq()
    c.set_option("staged_upload", 0)
    c.upload_rows(A, b)
    c.set_weights(np.ones(m))
    G0, c0, s0 = c.normal_eq()
    assert np.array_equal(G, G0) and np.array_equal(cc, c0) and np.array_equal(sc, s0)
    c.close()


@pytest.mark.parametrize("K", [31, 128, 142])
def test_weights_bound_in_caller_memory_are_read_afresh_by_every_fit(K):
    # fsnap_bind_rows / fsnap_bind_weights: rows, weights and mask live in memory the caller owns and may change between fits
    # without telling the library.  The kernels that pack the per-row pairs themselves read b, w and the mask in every launch;
    # the packing-kernel form (fused_pack = 0) re-packs on every launch for caller-owned memory.  Both see an in-place edit.
    import torch

    rng = np.random.default_rng(700 + K)
    m = 40007
    A, b, w = orc.synth_problem(m, K)
    t = rng.random(m) < 0.15
    dev = torch.device("cuda", 0)
    dA = torch.zeros(m * K + 4, dtype=torch.float64, device=dev)     # 16 readable bytes (and more) behind the last row
    dA[: m * K].copy_(torch.from_numpy(np.ascontiguousarray(A).reshape(-1)))
    db = torch.from_numpy(b).to(dev)
    dw = torch.from_numpy(w).to(dev)
    dm = torch.from_numpy((~t).astype(np.uint8)).to(dev)
    torch.cuda.synchronize()
    for fused in (1, 0):
        c = _capi.HipContext(0)
        c.set_option("fused_pack", fused)
        c.bind_rows(dA.data_ptr(), m, K, K, db.data_ptr())
        c.bind_weights(dw.data_ptr(), dm.data_ptr())
        stats_close(*c.normal_eq(), *orc.normal_eq(A, b, w, t))
        w2 = w * rng.uniform(0.5, 2.0, m)
        t2 = rng.random(m) < 0.3
        dw.copy_(torch.from_numpy(w2))                               # in place, same addresses
        dm.copy_(torch.from_numpy((~t2).astype(np.uint8)))
        torch.cuda.synchronize()
        stats_close(*c.normal_eq(), *orc.normal_eq(A, b, w2, t2))
        dw.copy_(torch.from_numpy(w))
        dm.copy_(torch.from_numpy((~t).astype(np.uint8)))
        torch.cuda.synchronize()
        c.close()


def test_fused_packing_falls_back_when_a_workgroups_rows_do_not_fit_the_lds(ctx):
    # few workgroups (option nblocks) -> thousands of chunks per row-wave: their pairs do not fit the 160 KiB of LDS, the
    # launch reads pairs packed by fsnap_pack_weights_k instead -- same statistics
    A, b, w = orc.synth_problem(120001, 128)
    t = np.random.default_rng(5).random(len(b)) < 0.2
    ref = orc.normal_eq(A, b, w, t)
    ctx.set_option("nblocks", 8)
    try:
        got = run_stats(ctx, A, b, w, t)
        info = ctx.launch_info()
        assert info["kernel_or_pairs"] == 3 and info["fused_pack"] == 0 and info["chunks_per_wave"] > 628
    finally:
        ctx.set_option("nblocks", 0)
    stats_close(*got, *ref)
    got2 = run_stats(ctx, A, b, w, t)
    assert ctx.launch_info()["fused_pack"] == 1
    stats_close(*got2, *ref)


@pytest.mark.parametrize("K", [31, 128, 200])
def test_streaming_accumulation_matches_one_shot(ctx, K):
    # fsnap_normal_eq_accumulate: per-batch `c += cm; d += dm` (transpose_trick/example.py:230-237) on the device;
    # covers the wave-triangle (31), one-wave-triangle (128) and tiled (200) kernels' reductions
    import torch
    rng = np.random.default_rng(600 + K)
    m = 9001
    A = rng.standard_normal((m, K))
    b = rng.standard_normal(m)
    w = rng.uniform(0.5, 2.0, m)
    t = rng.random(m) < 0.2
    dev = torch.device("cuda", 0)
    total = torch.zeros(K * K + K + 3, dtype=torch.float64, device=dev)
    for lo in range(0, m, 2500):
        hi = min(lo + 2500, m)
        ctx.upload_rows(A[lo:hi], b[lo:hi])
        ctx.set_weights(w[lo:hi], (~t[lo:hi]).astype(np.uint8))
        ctx.normal_eq_accumulate(total.data_ptr())
    G, c, s = ctx.download_packed(total.data_ptr(), K)
    stats_close(G, c, s, *orc.normal_eq(A, b, w, t), tol=2e-12)
    beta, rank, _ = ctx.solve_device(_capi.SOLVE_RIDGE, 1e-6, K, total.data_ptr())
    ref = np.linalg.solve(G + 1e-6 * np.eye(K), c)
    assert rank == K and np.max(np.abs(beta - ref)) / np.max(np.abs(ref)) < 1e-9


def test_reweighting_loop_error_analysis_reuses_labels_and_builds_df_lazily(ta, ta_fits):
    # GA-style loop (libmod_optimize.py:461-488): same rows and label lists, new weights per candidate.  The row
    # categories are converted / uploaded once, the DataFrame is only built when somebody reads it.
    A, b, w = ta
    t = ta_fits["testing_mask"]
    row_type = ["Energy"] * 363 + ["Force"] * 12672 + ["Stress"] * 2178
    fsd = {"Groups": [str(g) for g in ta_fits["ea_groups"]], "Testing": t.tolist(), "Row_Type": row_type}
    pt, s = make_solver("SVD")
    s.keep_resident = True
    tables = []
    for scale in (1.0, 3.0, 1.0):
        s.fit = None
        s.perform_fit(A, b, (w * scale)[~t], fs_dict=fsd)
        s.error_analysis(A, b, w * scale, fsd)
        assert s._df is None                                    # not built by error_analysis itself
        tables.append(s.errors[["ncount", "mae", "rmse", "rsq"]].to_numpy(dtype=np.float64))
    # unweighted rows are scale invariant, weighted MAE / RMSE scale with the weights; first and third candidate agree
    assert np.allclose(tables[0], tables[2], rtol=1e-10, atol=1e-300, equal_nan=True)
    assert not np.allclose(tables[0], tables[1], rtol=1e-3, atol=1e-300, equal_nan=True)
    df = s.df
    assert len(df.index) == len(b) and {"truths", "preds", "weights", "Groups", "Testing", "Row_Type"} <= set(df.columns)
    assert np.allclose(df["preds"].to_numpy(), A @ s.fit[:A.shape[1]] if len(s.fit) == A.shape[1] else df["preds"].to_numpy())
    pt.free()


def test_prepare_data_is_bitwise_the_reference_product(ta, ta_fits):
    # Solver.prepare_data (reference solvers/solver.py:50-76): aw, bw of the training rows through fsnap_weight_rows
    A, b, w = ta
    t = ta_fits["testing_mask"]
    pt, s = make_solver("SVD")
    aw, bw = s.prepare_data(A, b, w[~t], {"Testing": t.tolist()})          # one weight per training row, as the reference
    aw_ref, bw_ref = orc.weight_rows(A, b, w, t)
    assert aw.shape == aw_ref.shape and np.array_equal(aw, aw_ref) and np.array_equal(bw, bw_ref)
    aw, bw = s.prepare_data(A, b, w, None)                                   # no dictionary: every row trains
    aw_ref, bw_ref = orc.weight_rows(A, b, w)
    assert np.array_equal(aw, aw_ref) and np.array_equal(bw, bw_ref)
    # shared arrays + the job's own Testing list
    pt.create_shared_array("a", A.shape[0], A.shape[1])
    pt.create_shared_array("b", len(b))
    pt.create_shared_array("w", len(b))
    pt.shared_arrays["a"].array[:] = A
    pt.shared_arrays["b"].array[:] = b
    pt.shared_arrays["w"].array[:] = w
    aw, bw = s.prepare_data(None, None, None, {"Testing": t.tolist()})
    aw_ref, bw_ref = orc.weight_rows(A, b, w, t)
    assert np.array_equal(aw, aw_ref) and np.array_equal(bw, bw_ref)
    pt.free()


def test_one_label_flipped_in_place_changes_the_fit_and_the_error_table(ta, ta_fits):
    # keep_resident caches (training mask on the device, category ids) must follow an in-place edit of ONE entry of the
    # label lists at any index: fit and error table equal a fresh solver's (the reference re-reads the labels every call)
    from pandas.testing import assert_frame_equal

    A, b, w = ta
    t = ta_fits["testing_mask"].copy()
    row_type = ["Energy"] * 363 + ["Force"] * 12672 + ["Stress"] * 2178
    fsd = {"Groups": [str(g) for g in ta_fits["ea_groups"]], "Testing": t.tolist(), "Row_Type": row_type}
    pt, s = make_solver("RIDGE")
    s.keep_resident = True
    s.perform_fit(A, b, w[~t], fs_dict=fsd)
    s.error_analysis(A, b, w, fsd)
    before = s.fit.copy()
    m = len(b)
    sampled = {int(i * (m / 257)) for i in range(257)} | {m - 1}
    # the training row with the largest weighted norm among those the old probe never looked at (a row whose w^2 |a|^2
    # drowns in the rounding of G would change nothing)
    cand = np.array([i for i in range(m) if i not in sampled and not t[i]])
    k = int(cand[np.argmax((w[cand] ** 2) * np.einsum("ij,ij->i", A[cand], A[cand]))])
    fsd["Testing"][k] = True                                                    # in place: the list objects stay the same
    t[k] = True
    s.perform_fit(A, b, w[~t], fs_dict=fsd)
    s.error_analysis(A, b, w, fsd)
    pt2, s2 = make_solver("RIDGE")
    s2.perform_fit(A, b, w[~t], fs_dict={key: list(v) for key, v in fsd.items()})
    s2.error_analysis(A, b, w, {key: list(v) for key, v in fsd.items()})
    assert np.array_equal(s.fit, s2.fit) and not np.array_equal(s.fit, before)
    assert_frame_equal(s.errors, s2.errors, check_exact=False, rtol=1e-12, atol=0.0)
    pt.free()
    pt2.free()


def test_tiled_partials_survive_interleaved_geometries(ctx):
    # the tiled kernel's c partials are cleared only when (splits, superblocks) change: alternate two tiled shapes and a
    # one-wave-triangle shape on ONE context (they share the partial buffers) and check every result
    rng = np.random.default_rng(77)
    shapes = [(6000, 200), (900, 200), (5000, 96), (6000, 200), (30000, 200), (900, 200)]
    for m, K in shapes:
        A = rng.standard_normal((m, K))
        b = rng.standard_normal(m)
        w = rng.uniform(0.5, 2.0, m)
        t = rng.random(m) < 0.3
        G, c, s = run_stats(ctx, A, b, w, t)
        stats_close(G, c, s, *orc.normal_eq(A, b, w, t))
        G2, c2, s2 = ctx.normal_eq()                       # same geometry again: no clearing in between
        assert np.array_equal(G, G2) and np.array_equal(c, c2) and np.array_equal(s, s2)


def test_timing_events_are_sampled_by_option():
    c = _capi.HipContext(0)
    rng = np.random.default_rng(5)
    c.upload_rows(rng.standard_normal((4096, 64)), rng.standard_normal(4096))
    c.set_weights(np.ones(4096))
    sampled0, launches0 = c.timing_count()
    for _ in range(3):
        c.normal_eq()                                      # default: every launch is bracketed by events
    assert c.timing_count() == (sampled0 + 3, launches0 + 3)
    c.set_option("timing_every", 4)                        # the next launch is a sampled one, then every 4th
    for _ in range(9):
        c.normal_eq()
    assert c.timing_count() == (sampled0 + 6, launches0 + 12)
    syrk, red = c.timing_history(3)
    assert np.all(syrk > 0) and np.all(red > 0)
    c.set_option("timing_every", 0)
    G, cc, s = c.normal_eq()
    assert c.timing_count() == (sampled0 + 6, launches0 + 13) and np.isfinite(G).all()
    with pytest.raises(ValueError):
        c.set_option("timing_every", -1)
    c.close()
