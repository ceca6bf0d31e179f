"""GPU (-m gpu): kernel 1S (`fsnap_syrk_short`, fitsnap_amd/csrc/fsnap_syrk_short.hip) -- the statistics kernel of short
systems of 81 ... 144 columns (BASELINE configs[3]'s own shape: examples/Ta_PACE_RIDGE, 13 035 x 142) -- through the C ABI
against the oracle.  Same bars as tests/test_gpu_parity.py: statistics relative to sqrt(G_ii G_jj) at 1e-12, fits within
1e-6 of the reference's solvers (BASELINE.json north_star)."""
import numpy as np
import pytest

from fitsnap_amd import _capi
from oracle import fitsnap_oracle as orc

from conftest import maxrel
from test_gpu_parity import run_stats, stats_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = _capi.HipContext(0)
    yield c
    c.close()


def problem(m, K, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((m, K)) * (10.0 ** rng.uniform(-3, 3, size=K))
    b = rng.standard_normal(m)
    w = rng.choice([100.0, 1.0, 1e-8], size=m, p=[0.03, 0.83, 0.14])
    t = rng.random(m) < 0.2
    return A, b, w, t


@pytest.mark.parametrize("K", [81, 88, 95, 96, 97, 104, 111, 112, 113, 120, 127, 128, 129, 136, 141, 142, 143, 144])
def test_short_kernel_statistics_all_column_block_shapes(ctx, K):
    # NB = 6 ... 9: even block counts, odd ones (the last 16-column block staged beside a zero block), K odd (rows only 8-byte
    # aligned), K a multiple of 16 and not; ragged row count; rows of the testing set, tiny / huge weights (svd.py:35-46)
    m = 4099 + 7 * K
    A, b, w, t = problem(m, K, K)
    G, c, s = run_stats(ctx, A, b, w, t)
    info = ctx.launch_info()
    assert info["kernel_or_pairs"] == 7 and info["NB"] == (K + 15) // 16 and info["threads"] == 512
    stats_close(G, c, s, *orc.normal_eq(A, b, w, t))


@pytest.mark.parametrize("m", [1, 2, 3, 4, 5, 15, 16, 17, 31, 33, 111, 112, 113, 127, 128, 129, 1000, 14336, 14337, 28672, 28673, 43008])
@pytest.mark.parametrize("K", [96, 110, 142])
def test_short_kernel_tiny_and_ragged_row_counts(ctx, m, K):
    # one chunk of a few rows, chunks that end inside an 8-row pair of steps, a full phase, the first systems of two and three phases, the
    # longest default system at 142 columns
    A, b, w, t = problem(m, K, 7000 + 13 * m + K)
    G, c, s = run_stats(ctx, A, b, w, t)
    assert ctx.launch_info()["kernel_or_pairs"] == 7
    stats_close(G, c, s, *orc.normal_eq(A, b, w, t), tol=1e-11)


@pytest.mark.parametrize("K,m,nblocks", [(142, 40003, 0), (142, 5001, 3), (128, 43333, 0), (91, 20011, 7), (110, 100003, 0)])
def test_short_kernel_forced_on_long_systems_runs_in_phases(ctx, K, m, nblocks):
    # option short = 1: chunks longer than a phase (128 rows, 112 at NB = 9) -- stage, multiply, stage again; the accumulators,
    # the c sums and the b-only scalars carry over
    A, b, w, t = problem(m, K, 31 * K + m)
    ref = orc.normal_eq(A, b, w, t)
    ctx.set_option("short", 1)
    ctx.set_option("nblocks", nblocks)
    try:
        G, c, s = run_stats(ctx, A, b, w, t)
        info = ctx.launch_info()
        assert info["kernel_or_pairs"] == 7 and info["chunks_per_wave"] > 128
    finally:
        ctx.set_option("short", -1)
        ctx.set_option("nblocks", 0)
    stats_close(G, c, s, *ref)
    # and kernel 1A (the default of the long ones) on the same rows
    ctx.set_option("short", 0)
    try:
        G, c, s = run_stats(ctx, A, b, w, t)
        assert ctx.launch_info()["kernel_or_pairs"] == 3
    finally:
        ctx.set_option("short", -1)
    stats_close(G, c, s, *ref)


def test_short_kernel_ace_shape_bit_identical_run_to_run_and_fits_like_the_oracle(ctx):
    # BASELINE configs[3]'s own shape (examples/Ta_PACE_RIDGE/Ta.in: 13 035 rows x 142 ACE descriptors), RIDGE as there
    m, K = 13035, 142
    A, b, w = orc.synth_problem(m, K)
    t = np.random.default_rng(3).random(m) < 0.1
    first = run_stats(ctx, A, b, w, t)
    assert ctx.launch_info()["kernel_or_pairs"] == 7
    for _ in range(3):
        again = ctx.normal_eq()
        assert all(np.array_equal(x, y) for x, y in zip(first, again))
    stats_close(*first, *orc.normal_eq(A, b, w, t))
    for alpha in (1e-8, 1e-4):
        beta = ctx.fit_resident(_capi.SOLVE_RIDGE, alpha)[0]
        assert maxrel(beta, orc.ridge_fit(A, b, w, alpha, testing=t)) < 1e-6
    beta = ctx.fit_resident(_capi.SOLVE_LSTSQ, 1e-13)[0]
    assert maxrel(beta, orc.svd_fit(A, b, w, t)) < 1e-6


def test_short_kernel_masked_rows_never_reach_the_statistics(ctx):
    # NaN / Inf in the rows, b and w of the testing set stay out (the reference drops those rows by fancy indexing, svd.py:44-46)
    m, K = 9001, 142
    A, b, w, t = problem(m, K, 99)
    A2, b2, w2 = A.copy(), b.copy(), w.copy()
    A2[t] = np.nan
    b2[t] = np.inf
    w2[t] = -np.inf
    G, c, s = run_stats(ctx, A2, b2, w2, t)
    assert ctx.launch_info()["kernel_or_pairs"] == 7
    stats_close(G, c, s, *orc.normal_eq(A, b, w, t))
    assert np.isfinite(G).all() and np.isfinite(c).all() and np.isfinite(s).all()


def test_short_kernel_on_packed_pairs_gives_the_bits_of_its_own_packing(ctx):
    # option fused_pack = 0: the pairs come from fsnap_pack_weights_k's array in HBM (the form the row-space passes use) -- the
    # same numbers reach the same instructions in the same order
    m, K = 13035, 142
    A, b, w, t = problem(m, K, 5)
    got = []
    for fused in (1, 0):
        ctx.set_option("fused_pack", fused)
        try:
            got.append(run_stats(ctx, A, b, w, t))
            info = ctx.launch_info()
            assert info["kernel_or_pairs"] == 7 and info["fused_pack"] == fused
        finally:
            ctx.set_option("fused_pack", 1)
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1])
    assert got[0][2][2] == got[1][2][2] and np.allclose(got[0][2], got[1][2], rtol=1e-13, atol=0)


def test_short_kernel_strided_rows_bound_in_place(ctx):
    # fsnap_bind_rows with an odd leading dimension and the first column at a 24-byte offset: rows only 8-byte aligned, the
    # 16-byte loads straddle them; K = 142 of 151
    import torch
    rng = np.random.default_rng(8)
    m, K, lda = 7003, 142, 151
    big = rng.standard_normal((m, lda))
    A = np.ascontiguousarray(big[:, 3:3 + K])
    b = rng.standard_normal(m)
    w = rng.uniform(0.1, 3, m)
    dev = torch.device("cuda", 0)
    dbig = torch.from_numpy(big).to(dev)
    db = torch.from_numpy(b).to(dev)
    c = _capi.HipContext(0)
    try:
        c.bind_rows(dbig.data_ptr() + 3 * 8, m, K, lda, db.data_ptr())
        c.set_weights(w)
        G, cc, s = c.normal_eq()
        assert c.launch_info()["kernel_or_pairs"] == 7
        stats_close(G, cc, s, *orc.normal_eq(A, b, w))
    finally:
        torch.cuda.synchronize()
        c.close()


def test_short_kernel_row_space_solve_of_an_ill_conditioned_short_system(ctx):
    # the row-space passes bring per-row pairs of their own (statistics of Q): kernel 1S on pairs from HBM inside fsnap_lstsq_rows
    m, K = 9000, 142
    rng = np.random.default_rng(17)
    U, _ = np.linalg.qr(rng.standard_normal((m, K)))
    V, _ = np.linalg.qr(rng.standard_normal((K, K)))
    sv = np.logspace(0, -9, K)
    A = (U * sv) @ V.T
    beta_star = rng.standard_normal(K)
    b = A @ beta_star + 1e-6 * rng.standard_normal(m)
    w = rng.uniform(0.5, 2.0, m)
    ctx.upload_rows(A, b)
    ctx.set_weights(w)
    beta, rank, info = ctx.lstsq_rows(1e-13)
    assert rank == K
    ref = orc.svd_fit(A, b, w)
    assert maxrel(beta, ref) < max(1e-6, 50 * 1e9 * 2.2e-16)


def test_row_space_solve_at_the_ace_width_on_the_two_wave_pass_kernel(ctx):
    # 129 ... 144 columns from 32 768 rows on: the row-space passes run on kernel 13C with nine column blocks (three 64-column
    # panels, the last one a single block) instead of kernel 13B -- first pass (weighted rows of A) and in-place passes; the
    # statistics of Q on kernel 1S (pairs from HBM, three phases per workgroup)
    m, K = 40000, 142
    rng = np.random.default_rng(23)
    U, _ = np.linalg.qr(rng.standard_normal((m, K)))
    V, _ = np.linalg.qr(rng.standard_normal((K, K)))
    sv = np.logspace(0, -8, K)
    A = (U * sv) @ V.T
    b = A @ rng.standard_normal(K) + 1e-6 * rng.standard_normal(m)
    w = rng.uniform(0.5, 2.0, m)
    w[rng.random(m) < 0.05] = 0.0                     # zero-weight rows become zero rows of Q
    ctx.upload_rows(A, b)
    ctx.set_weights(w)
    beta, rank, info = ctx.lstsq_rows(1e-13)
    assert rank == K and info["passes"] >= 2
    kappa = np.linalg.cond(w[:, None] * A)            # (the weights stretch the 1e9 of the unweighted rows)
    assert maxrel(beta, orc.svd_fit(A, b, w)) < max(1e-6, 50 * kappa * 2.2e-16)
