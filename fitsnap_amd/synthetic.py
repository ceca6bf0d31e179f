"""Synthetic workload of SURVEY.md 8(d) / BASELINE.md 2: the 10^6 x 128 fp64 problem of the headline benchmark (and its
smaller / wider relatives), generated in 64 Ki-row chunks so that every rank of a multi-GPU run can produce its own row
range.  Input data only -- shared by ``bench.py``, ``__graft_entry__.smoke()`` and the tests; the checker lives in
``oracle/``."""
import numpy as np

SYNTH_SEED = 20250926
SYNTH_CHUNK = 65536


def synth_chunk(chunk_index, rows, K, beta_star, scales):
    """One <=64Ki-row chunk of the synthetic A/b/w: A = N(0,1)*s_j, b = A beta* + 1e-3 N(0,1),
    w in {100, 1, 1e-8} with p = {0.03, 0.83, 0.14}."""
    rng = np.random.default_rng([SYNTH_SEED, chunk_index])
    A = rng.standard_normal((rows, K)) * scales
    b = A @ beta_star + 1.0e-3 * rng.standard_normal(rows)
    u = rng.random(rows)
    w = np.where(u < 0.03, 100.0, np.where(u < 0.86, 1.0, 1.0e-8))
    return A, b, w


def synth_params(K):
    scales = 10.0 ** (-4.0 * np.arange(K) / max(K - 1, 1))
    beta_star = np.random.default_rng(1).standard_normal(K) / scales
    return scales, beta_star


def synth_problem(m, K, row_offset=0):
    """Rows [row_offset, row_offset+m) of the synthetic problem (chunk-aligned offsets)."""
    assert row_offset % SYNTH_CHUNK == 0
    scales, beta_star = synth_params(K)
    A = np.empty((m, K))
    b = np.empty(m)
    w = np.empty(m)
    done = 0
    ci = row_offset // SYNTH_CHUNK
    while done < m:
        rows = min(SYNTH_CHUNK, m - done)
        A[done:done + rows], b[done:done + rows], w[done:done + rows] = synth_chunk(ci, rows, K, beta_star, scales)
        done += rows
        ci += 1
    return A, b, w


def synth_testing_mask(m, frac=0.1):
    return np.random.default_rng(2).random(m) < frac
