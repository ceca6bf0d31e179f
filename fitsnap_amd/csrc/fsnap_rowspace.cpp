// fsnap_rowspace.cpp — least squares on the ROWS for systems the K x K statistics cannot resolve.
//
// Reference semantics: fit = scipy.linalg.lstsq(aw, bw, 1.0e-13)   (fitsnap3lib/solvers/svd.py:44-54; LAPACK dgelsd:
// QR of aw, Q^T bw, then the SVD of the K x K factor with singular values below 1e-13 sigma_max dropped, minimum-norm
// solution).  Its error grows like kappa(A_w) eps; the normal equations the fast path solves grow like kappa^2 eps and
// are numerically singular from kappa ~ 1e7-1e8 on.  This file reproduces dgelsd's structure with the QR done on the GPU:
//
//   pass k = 1, 2, ...:   G_k = Q_{k-1}^T Q_{k-1}            the SYRK kernels (Q_0 = A_w; pass 1 reuses the fit's G)
//                         R_k = chol(D^-1 G_k D^-1 + s I) D    host, Jacobi-scaled, shift s ~ K eps ||.|| (never fails,
//                                                             also for rank-deficient A: shifted CholeskyQR3)
//                         Q_k = Q_{k-1} R_k^-1               kernel 13 (fsnap_trsm.hip), by substitution
//                         R_hat = R_k R_hat                   so that A_w = Q_k R_hat throughout
//   until max |Q_k^T Q_k - I| <= 1e-10 (2-3 passes; each amplifies what the previous one could not resolve by ~1e7)
//   z = Q^T (w b)                                            comes out of the last SYRK pass as its "c" vector
//   beta = pinv_{rcond}(R_hat) z                             host: back substitution when no singular value can be
//                                                            below the cut, else one-sided Jacobi SVD of R_hat with
//                                                            dgelsd's truncation (minimum-norm solution)
//   one refinement step with the residual of the ORIGINAL rows: r = w (b - A beta), beta += pinv(R_hat) Q^T r
//
// A_w = Q R_hat holds to ~K eps ||A_w|| because every pass divides by R_k with a substitution (backward error
// eps |R_k|); with Q orthonormal the singular values of R_hat are those of A_w, so the 1e-13 cut acts on the same
// numbers as in dgelsd.  Multi-GPU: every rank orthogonalises ITS rows with the SAME R_k (the G_k are all-reduced, the
// host algebra is deterministic), z and the refinement right-hand side are all-reduced.
//
// The two host steps are also exported on their own (fsnap_rowspace_factor / fsnap_rowspace_solve): a caller that
// streams its rows through passes of its own can use them, and the CPU tests drive them with numpy standing in for the
// kernels.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

#include "fsnap_ctx.h"
#include "fsnap_kernels.h"

#include "fsnap_rowspace_host.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>

using fsnap_rs::FactorChain;
using fsnap_rs::FactorSolver;
using fsnap_rs::factor_pass;
using fsnap_rs::finite_all;
using fsnap_rs::gram_deviation;
using vec = std::vector<double>;

namespace fsnap {

struct RowSpace {
    DevBuf Q, qpack, Rdev, packed, rvec, dz, dzpart, beta, scan;
    // page-locked K x K blocks, one per pass, that receive the factors of the device factorisation and stay in place for
    // the K x K end (FactorChain::push_view)
    std::vector<double*> pinfac;
    std::vector<size_t> pinfac_doubles;
    double* pinfac_get(size_t idx, size_t n) {
        if (pinfac.size() <= idx) {
            pinfac.resize(idx + 1, nullptr);
            pinfac_doubles.resize(idx + 1, 0);
        }
        if (pinfac_doubles[idx] < n) {
            if (pinfac[idx]) (void)hipHostFree(pinfac[idx]);
            pinfac[idx] = nullptr;
            pinfac_doubles[idx] = 0;
            if (hipHostMalloc((void**)&pinfac[idx], n * sizeof(double), hipHostMallocDefault) != hipSuccess) return nullptr;
            pinfac_doubles[idx] = n;
        }
        return pinfac[idx];
    }
    // page-locked staging for what crosses PCIe every pass: the K x K statistics down, the factor up (from pageable
    // memory each of these copies went through the runtime's own staging, ~40 us apiece)
    // the factor of a device-factorised pass goes home on a stream of its own, behind an event, while the pass kernel and the
    // statistics of Q run (20 MB at K = 1595: 0.4 ms that used to sit between the two on the fit's stream)
    hipStream_t copy_stream = nullptr;
    hipEvent_t fac_ready = nullptr, fac_copied = nullptr;
    bool copy_pending = false;
    bool copy_ensure() {
        if (copy_stream) return true;
        if (hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking) != hipSuccess) {
            copy_stream = nullptr;
            return false;
        }
        if (hipEventCreateWithFlags(&fac_ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&fac_copied, hipEventDisableTiming) != hipSuccess) {
            if (fac_ready) (void)hipEventDestroy(fac_ready);
            if (fac_copied) (void)hipEventDestroy(fac_copied);
            fac_ready = fac_copied = nullptr;
            (void)hipStreamDestroy(copy_stream);
            copy_stream = nullptr;
            return false;
        }
        return true;
    }
    double* pin = nullptr;
    size_t pin_doubles = 0;
    bool pin_ensure(size_t n) {
        if (n <= pin_doubles) return true;
        if (pin) (void)hipHostFree(pin);
        pin = nullptr;
        pin_doubles = 0;
        if (hipHostMalloc((void**)&pin, n * sizeof(double), hipHostMallocDefault) != hipSuccess) return false;
        pin_doubles = n;
        return true;
    }
};

void rowspace_release(fsnap_ctx* ctx) {
    if (!ctx->rowspace) return;
    RowSpace* rs = ctx->rowspace;
    DevBuf* bufs[] = {&rs->Q, &rs->qpack, &rs->Rdev, &rs->packed, &rs->rvec, &rs->dz, &rs->dzpart, &rs->beta, &rs->scan};
    for (DevBuf* b : bufs) b->release();
    if (rs->copy_stream) {
        (void)hipStreamSynchronize(rs->copy_stream);
        (void)hipEventDestroy(rs->fac_ready);
        (void)hipEventDestroy(rs->fac_copied);
        (void)hipStreamDestroy(rs->copy_stream);
    }
    if (rs->pin) (void)hipHostFree(rs->pin);
    for (double* p : rs->pinfac)
        if (p) (void)hipHostFree(p);
    delete rs;
    ctx->rowspace = nullptr;
}

}  // namespace fsnap

extern "C" {

int fsnap_rowspace_factor(int64_t K, const double* G, int first, double tol, double* Rhat, double* Rp, double* info) {
    if (!G || !Rhat || !Rp || K <= 0 || K > (1 << 20)) return FSNAP_E_ARG;
    double dev = 0.0, shift = 0.0;
    int conv = 0;
    const int rc = factor_pass((int)K, G, first, tol, Rhat, Rp, &dev, &conv, &shift);
    if (info) {
        info[0] = dev;
        info[1] = conv;
        info[2] = shift;
    }
    return rc;
}

int fsnap_rowspace_solve(int64_t K, const double* Rhat, const double* z, double rcond, double* beta, int* rank, double* info) {
    if (!Rhat || !z || !beta || K <= 0 || K > (1 << 20)) return FSNAP_E_ARG;
    if (!finite_all(Rhat, (size_t)K * K) || !finite_all(z, (size_t)K)) return FSNAP_NUM_NONFINITE;
    FactorSolver fs;
    fs.prepare((int)K, Rhat, rcond);
    fs.apply(z, beta);
    if (rank) *rank = fs.rank;
    if (info) {
        info[0] = fs.triangular ? 0.0 : (fs.deflated ? 3.0 : 1.0);
        info[1] = fs.smax;
        info[2] = fs.smin;
        info[3] = fs.sweeps;
    }
    return finite_all(beta, (size_t)K) ? FSNAP_OK : FSNAP_NUM_NONFINITE;
}

int fsnap_rowspace_chain(int64_t K64, int64_t nfac, const double* R, const unsigned char* active, const double* z, double rcond,
                         double* beta, int* rank, double* info) {
    if (!R || !z || !beta || K64 <= 0 || K64 > (1 << 20) || nfac < 1 || nfac > 16) return FSNAP_E_ARG;
    const int K = (int)K64;
    if (!finite_all(R, (size_t)nfac * K * K) || !finite_all(z, (size_t)K)) return FSNAP_NUM_NONFINITE;
    FactorChain chain;
    chain.K = K;
    chain.active.assign((size_t)K, 1);
    if (active)
        for (int j = 0; j < K; ++j) chain.active[j] = active[j] ? 1 : 0;
    for (int64_t k = 0; k < nfac; ++k) chain.push(R + (size_t)k * K * K);
    double nrm = 0.0, inv = 0.0, bound = 0.0;
    const bool use_chain = chain.certified(rcond, &nrm, &inv, &bound);
    int rk = 0;
    if (use_chain) {
        chain.solve(z, beta);
        for (int j = 0; j < K; ++j) rk += chain.active[j] ? 1 : 0;
    } else {
        fsnap_rs::Scratch Rhat((size_t)K * K);
        chain.product(Rhat.data());
        FactorSolver fs;
        fs.prepare(K, Rhat.data(), rcond);
        fs.apply(z, beta);
        rk = fs.rank;
    }
    if (rank) *rank = rk;
    if (info) {
        info[0] = use_chain ? 1.0 : 0.0;
        info[1] = nrm;
        info[2] = inv;
        info[3] = bound;
    }
    return finite_all(beta, (size_t)K) ? FSNAP_OK : FSNAP_NUM_NONFINITE;
}

int fsnap_set_dense_pinv(fsnap_ctx* ctx, fsnap_dense_pinv_fn fn, void* user) {
    if (!ctx) return FSNAP_E_ARG;
    ctx->dense_pinv = fn;
    ctx->dense_pinv_user = user;
    return FSNAP_OK;
}

int fsnap_lstsq_rows(fsnap_ctx* ctx, double rcond, int64_t K64, double* beta, int* rank_out, double* info) {
    if (!ctx) return FSNAP_E_ARG;
    if (!beta || K64 <= 0) return ctx->fail(FSNAP_E_ARG, "fsnap_lstsq_rows: bad argument");
    const int K = (int)K64, K16 = (K + 15) & ~15;
    const int64_t npk = FSNAP_PACKED_LEN(K64);
    const int nranks = ctx->comm ? 2 : 1;     // "> 1" = collective (a communicator of one rank takes the same path)
    bool have_rows = ctx->dA && ctx->m > 0;
    const bool reuse_asked = ctx->opt_rowspace_reuse != 0;      // one-shot (option "rowspace_reuse_stats")
    ctx->opt_rowspace_reuse = 0;
    // Failures that only THIS rank sees must not keep it out of the first collective (its peers would wait until
    // FSNAP_COMM_TIMEOUT): the rank then contributes NaN statistics, every rank stops at the first factorisation with
    // FSNAP_NUM_NONFINITE, and this rank reports its own error.
    int local_rc = FSNAP_OK;
    std::string local_err;
    auto local_fail = [&](int code) {
        if (local_rc == FSNAP_OK) {
            local_rc = code;
            local_err = ctx->err;
        }
    };
    if (have_rows && ctx->K != K64)
        local_fail(ctx->fail(FSNAP_E_ARG, "fsnap_lstsq_rows: K = %d but the resident rows have %lld columns", K, (long long)ctx->K));
    if (!have_rows && nranks == 1) return ctx->fail(FSNAP_E_STATE, "no rows: call fsnap_upload_rows/fsnap_bind_rows first");
    if (have_rows && !ctx->dw) local_fail(ctx->fail(FSNAP_E_STATE, "no weights: call fsnap_set_weights/fsnap_bind_weights first"));
    if (local_rc != FSNAP_OK && nranks == 1) return local_rc;
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    if (!ctx->rowspace) {
        ctx->rowspace = new (std::nothrow) fsnap::RowSpace();
        if (!ctx->rowspace) return ctx->fail(FSNAP_E_NOMEM, "out of host memory");
    }
    fsnap::RowSpace* rs = ctx->rowspace;
    const size_t m = have_rows ? (size_t)ctx->m : 0;
    // the small K x K workspaces: without them this rank cannot even take part in the collective
    const size_t rdoubles = fsnap::trsm_factor_doubles(K16);       // padded factor + the inverses of its 16 x 16 diagonal blocks
    if (!rs->packed.ensure((size_t)npk * 8) || (nranks > 1 && !fsnap::allreduce_packed_reserve(ctx, K)) || !rs->Rdev.ensure(rdoubles * 8) || !rs->beta.ensure((size_t)K * 8) ||
        !rs->dz.ensure((size_t)K * 8) || !rs->pin_ensure((size_t)npk + rdoubles + 2 * (size_t)K))
        return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(row-space workspace) failed");
    if (have_rows && local_rc == FSNAP_OK) {
        const int nbt = fsnap::gemvT_num_blocks(ctx->m);
        if (!rs->Q.ensure(m * (size_t)K * 8 + 256) || !rs->qpack.ensure(m * 16 + 64) || !rs->rvec.ensure(m * 8) ||
            !rs->dzpart.ensure((size_t)nbt * K * 8)) {
            local_fail(ctx->fail(FSNAP_E_NOMEM, "hipMalloc of %zu bytes for the orthogonalised rows failed", m * (size_t)K * 8));
            if (nranks == 1) return local_rc;
        }
    }
    double* dp = (double*)rs->packed.p;
    double* dQ = (double*)rs->Q.p;
    hipStream_t st = ctx->stream;
    int rc;
    // K > 256: the factors of the passes stay apart (FactorChain) and R_hat is only multiplied out when a truncation turns
    // out to be needed; small K keeps the accumulated factor (the product is cheap there and the exact Frobenius bound of
    // FactorSolver is sharper than the chain's estimate)
    const bool chained = K > 256;
    // K >= 384 with the factors kept apart: the Gram matrix is factorised where it is, in HBM (kernels 8b-8d in their
    // factor-only form), the factor lands in the layout the pass reads, and one D2H copy per pass brings it to the host for
    // the K x K end -- the host factorisation (one core) was 2 x 22 ms of a 125 ms call at K = 1595.  The host then needs
    // only the diagonal, Q^T (w b) and two steering numbers per pass (fsnap_gram_scan_k), not the K x K matrix.
    const bool device_factor = chained && K >= 384 && ctx->opt_device_solve != 2;
    if (device_factor && !rs->scan.ensure((size_t)2 * K * 8)) return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(row-space workspace) failed");
    vec Rhat(chained ? 0 : (size_t)K * K), Rp(device_factor ? 0 : (size_t)K * K), z((size_t)K), scanh(device_factor ? (size_t)2 * K : 0);
    double scan_dev = 0.0, scan_fro = 0.0;
    FactorChain chain;
    double* const host = rs->pin;                        // [npk] statistics of the current Q
    double* const Rpad = rs->pin + npk;                  // [K16 x K16] padded factor of the pass | [K16 / 16][16][16] inverse blocks
    double* const hvec = Rpad + rdoubles;                // [2 K] beta up, Q^T r down

    // statistics of the current Q (pass 0: of A_w), summed over the ranks, on the host
    auto gather_stats = [&](bool of_rows) -> int {
        if (local_rc == FSNAP_OK && have_rows) {
            int r2 = of_rows ? fsnap_normal_eq_async(ctx, dp) : fsnap::normal_eq_launch_on(ctx, dQ, K, (const double*)rs->qpack.p, dp);
            if (r2) {
                if (nranks == 1) return r2;
                local_fail(r2);
            }
        } else if (local_rc == FSNAP_OK) {
            FSNAP_HIP(hipMemsetAsync(dp, 0, (size_t)npk * 8, st), "hipMemsetAsync(packed)");
        }
        if (local_rc != FSNAP_OK) (void)hipMemsetAsync(dp, 0xFF, (size_t)npk * 8, st);     // NaN in every double
        if (nranks > 1) {
            int r3 = fsnap::allreduce_packed(ctx, dp, K);
            if (r3) return r3;
        }
        if (device_factor) {
            // diagonal (to its places in the host copy), Q^T (w b) + scalars, row maxima / sums of the scan
            FSNAP_HIP(fsnap::launch_gram_scan(dp, K, (double*)rs->scan.p, st), "launch fsnap_gram_scan_k");
            FSNAP_HIP(hipMemcpy2DAsync(host, (size_t)(K + 1) * 8, dp, (size_t)(K + 1) * 8, 8, (size_t)K, hipMemcpyDeviceToHost, st),
                      "hipMemcpy(diagonal)");
            FSNAP_HIP(hipMemcpyAsync(host + (size_t)K * K, dp + (size_t)K * K, (size_t)(K + 3) * 8, hipMemcpyDeviceToHost, st),
                      "hipMemcpy(statistics)");
            FSNAP_HIP(hipMemcpyAsync(scanh.data(), rs->scan.p, (size_t)2 * K * 8, hipMemcpyDeviceToHost, st), "hipMemcpy(scan)");
        } else {
            FSNAP_HIP(hipMemcpyAsync(host, dp, (size_t)npk * 8, hipMemcpyDeviceToHost, st), "hipMemcpy(statistics)");
        }
        int r4 = fsnap::wait_stream(ctx, nullptr, "statistics of a row-space pass");
        if (r4) return r4;
        if (local_rc != FSNAP_OK) return ctx->fail(local_rc, "%s", local_err.c_str());
        if (device_factor) {
            double dv = 0.0, f2 = 0.0;
            bool finite = true;
            for (int a = 0; a < K; ++a) {
                finite = finite && (scanh[a] == scanh[a]);
                dv = std::fmax(dv, scanh[a]);
                f2 += scanh[(size_t)K + a];
            }
            scan_dev = dv;
            scan_fro = std::sqrt(f2);
            if (!finite || !std::isfinite(f2))
                return ctx->fail(FSNAP_NUM_NONFINITE, nranks > 1 ? "row-space solve: non-finite statistics after the all-reduce (NaN/Inf in a training row of "
                                                                   "some rank, or a rank failed before the collective: see that rank's error)"
                                                                 : "row-space solve: non-finite Gram matrix");
        }
        return FSNAP_OK;
    };
    // FSNAP_ROWSPACE_TIMING=1: wall-clock marks of the host phases on stderr
    const bool timing = getenv("FSNAP_ROWSPACE_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[fsnap_lstsq_rows] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    // (w_eff, w_eff b) per row in HBM: the first pass and the qpack pairs read them.  The statistics launch does not leave
    // them behind any more (kernel 1A packs its rows' pairs in LDS): pack them unless they are current -- BEFORE the first
    // collective, so that a rank that fails here stops everybody at the first factorisation (NaN statistics)
    if (have_rows && local_rc == FSNAP_OK && (rc = fsnap::wpack_current(ctx))) {
        if (nranks == 1) return rc;
        local_fail(rc);
    }
    // The fit from the statistics that sent the caller here computed G = A_w^T A_w of these very rows a moment ago, and for
    // systems the host factorises it still sits in the page-locked mirror: the first pass starts from it (0.35 ms of a
    // 10^6 x 128 call).  Only on the caller's word (one-shot option), single rank, and while the mirror reflects ctx->packed.
    const bool reuse = reuse_asked && nranks == 1 && !device_factor && ctx->mirror && ctx->mirror_of &&
                       ctx->mirror_of == (const double*)ctx->packed.p && ctx->mirror_K == K64;
    if (reuse) {
        if ((rc = fsnap::wait_stream(ctx, ctx->mirror_ev, "statistics mirror"))) return rc;
        memcpy(host, ctx->mirror, (size_t)npk * 8);
        if (ctx->mirror_upper)
            for (int i = 1; i < K; ++i)
                for (int j = 0; j < i; ++j) host[(size_t)i * K + j] = host[(size_t)j * K + i];
        mark("statistics of the rows (kept)");
    } else {
        if ((rc = gather_stats(true))) return rc;
        mark("statistics of the rows");
    }
    if (have_rows) {
        if (local_rc == FSNAP_OK)
            FSNAP_HIP(fsnap::launch_qpack((const double*)ctx->wpack.p, ctx->m, (double*)rs->qpack.p, st), "launch fsnap_qpack_k");
        FSNAP_HIP(hipMemsetAsync((char*)rs->Q.p + m * (size_t)K * 8, 0, 256, st), "hipMemsetAsync");   // tail pad of the SYRK loads
    }
    const double tol = 1.0e-10;
    const int maxpass = 6;
    int passes = 0, converged = 0;
    double dev = 0.0, shift = 0.0;
    for (int pass = 1; pass <= maxpass; ++pass) {
        int conv = 0;
        if (pass == 1 && chained) chain.start(K, host);
        if (device_factor) {
            dev = scan_dev;
            if (pass > 1 && dev <= tol) {
                converged = 1;
                break;
            }
            const double fro = scan_fro;
            const int np64 = (K + 63) / 64 * 64, npanel = np64 / 64;
            if (!ctx->dchol.ensure(fsnap::chol_large_work_doubles(K) * 8) || !ctx->dsolve.ensure(((size_t)np64 + npanel + 2) * 8))
                return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(device Cholesky) failed");
            double* d_dsc = (double*)ctx->dsolve.p;
            double* d_minpiv = d_dsc + np64;
            int* d_status = (int*)(d_minpiv + npanel);
            ctx->chol_status_word = nullptr;                  // the buffer of fsnap_solve_device's chain is reused here
            ctx->chol_factor_of = nullptr;                    // ... and so is the work matrix: no factor of a fit in it any more
            // shift: a few times the rounding level of the Gram matrix, x 100 and again when a pivot fails (as factor_pass)
            shift = 4.0 * (K + 100.0) * std::numeric_limits<double>::epsilon() * fro;
            int status = 0;
            if (rs->copy_pending) {          // the previous pass's factor is still being read out of Rdev by the copy stream
                FSNAP_HIP(hipStreamWaitEvent(st, rs->fac_copied, 0), "hipStreamWaitEvent");
            }
            for (int attempt = 0; attempt < 10; ++attempt) {
                FSNAP_HIP(fsnap::launch_chol_factor(dp, K, shift, (double*)ctx->dchol.p, d_dsc, d_status, d_minpiv, K16,
                                                    (double*)rs->Rdev.p, st), "launch device Cholesky (factor)");
                FSNAP_HIP(hipMemcpyAsync(&status, d_status, sizeof(int), hipMemcpyDeviceToHost, st), "hipMemcpy(status)");
                if ((rc = fsnap::wait_stream(ctx, nullptr, "pass factor"))) return rc;
                if (status == 0) break;
                if (status & 1) return ctx->fail(FSNAP_NUM_NONFINITE, "row-space pass %d: non-finite Gram matrix", pass);
                shift *= 100.0;
            }
            if (status != 0) return ctx->fail(FSNAP_NUM_NOT_SPD, "row-space pass %d: the Gram matrix could not be factorised", pass);
            mark("factor (device)");
        } else {
        rc = factor_pass(K, host, pass == 1, tol, chained ? nullptr : Rhat.data(), Rp.data(), &dev, &conv, &shift);
        if (rc == FSNAP_NUM_NONFINITE && nranks > 1)
            return ctx->fail(rc, "row-space pass %d: non-finite statistics after the all-reduce (NaN/Inf in a training row of some "
                                 "rank, or a rank failed before the collective: see that rank's error)", pass);
        if (rc) return ctx->fail(rc, "row-space pass %d: the Gram matrix could not be factorised (status %d)", pass, rc);
        if (conv) {
            converged = 1;
            break;
        }
        mark("factor_pass");
        if (chained) chain.push(Rp.data());
        // padded copy of the factor for the kernel
        std::fill(Rpad, Rpad + (size_t)K16 * K16, 0.0);
        for (int i = 0; i < K16; ++i) Rpad[(size_t)i * K16 + i] = 1.0;
        for (int i = 0; i < K; ++i) memcpy(Rpad + (size_t)i * K16 + i, Rp.data() + (size_t)i * K + i, (size_t)(K - i) * 8);
        fsnap::trsm_invert_diagonal_blocks(Rpad, K16);
        FSNAP_HIP(hipMemcpyAsync(rs->Rdev.p, Rpad, rdoubles * 8, hipMemcpyHostToDevice, st), "hipMemcpy(R)");
        }
        if (device_factor) {
            // the factor for the K x K end: K x K compact into a page-locked block of its own, where it stays (no second copy).
            // On the copy stream, behind an event (the factorisation is complete: its status has been read): the download
            // runs beside the pass kernel and the statistics of Q; the next factorisation and the K x K end wait for it.
            double* hf = rs->pinfac_get((size_t)pass - 1, (size_t)K * K);
            if (!hf) return ctx->fail(FSNAP_E_NOMEM, "hipHostMalloc(factor staging) failed");
            hipStream_t cs = rs->copy_ensure() ? rs->copy_stream : st;
            if (cs != st) {
                FSNAP_HIP(hipEventRecord(rs->fac_ready, st), "hipEventRecord");
                FSNAP_HIP(hipStreamWaitEvent(cs, rs->fac_ready, 0), "hipStreamWaitEvent");
            }
            FSNAP_HIP(hipMemcpy2DAsync(hf, (size_t)K * 8, rs->Rdev.p, (size_t)K16 * 8, (size_t)K * 8, (size_t)K, hipMemcpyDeviceToHost, cs),
                      "hipMemcpy(factor)");
            if (cs != st) {
                FSNAP_HIP(hipEventRecord(rs->fac_copied, cs), "hipEventRecord");
                rs->copy_pending = true;
            }
            chain.push_view(hf);
        }
        if (have_rows) {
            if (pass == 1)
                FSNAP_HIP(fsnap::launch_trsm_rows(ctx->dA, ctx->lda, (const double*)ctx->wpack.p, dQ, K, ctx->m, K,
                                                  (const double*)rs->Rdev.p, K16, st), "launch fsnap_trsm_rows_k");
            else
                FSNAP_HIP(fsnap::launch_trsm_rows(dQ, K, nullptr, dQ, K, ctx->m, K, (const double*)rs->Rdev.p, K16, st),
                          "launch fsnap_trsm_rows_k");
        }
        // (host-factorised widths: Rpad is reused by the next pass; device-factorised: the statistics of Q queue behind the pass)
        if (!device_factor && (rc = fsnap::wait_stream(ctx, nullptr, "row-space pass"))) return rc;
        passes = pass;
        mark(device_factor ? "TRSM pass launched" : "factor upload + TRSM pass");
        if ((rc = gather_stats(false))) return rc;
        mark(device_factor ? "TRSM pass + statistics of Q" : "statistics of Q");
        memcpy(z.data(), host + (size_t)K * K, (size_t)K * 8);     // z = Q^T (w b)
    }
    if (rs->copy_pending) {                  // the factors are read on the host from here on
        rs->copy_pending = false;
        if ((rc = fsnap::wait_stream(ctx, rs->fac_copied, "factor download"))) return rc;
    }
    if (!converged) {
        // pass budget used up: judge the last Q as it is (the refinement step below absorbs what is left)
        if (device_factor) {
            dev = scan_dev;
        } else {
            if (!finite_all(host, (size_t)K * K)) return ctx->fail(FSNAP_NUM_NONFINITE, "row-space solve: non-finite Gram matrix");
            dev = gram_deviation(K, host);
        }
        converged = dev <= tol;
    }
    if (passes == 0) return ctx->fail(FSNAP_E_STATE, "row-space solve made no pass");
    if (!finite_all(z.data(), z.size())) return ctx->fail(FSNAP_NUM_NONFINITE, "row-space solve: non-finite Q^T b");

    FactorSolver fs;
    fsnap_rs::Scratch Rprod;
    bool use_chain = false;
    double chain_norm = 0.0, chain_inv = 0.0;
    if (chained) {
        use_chain = chain.certified(rcond, &chain_norm, &chain_inv, nullptr);         // no singular value can be cut
        if (!use_chain) {
            Rprod.reset((size_t)K * K);                         // (product() writes every entry)
            chain.product(Rprod.data());
        }
    }
    if (!use_chain) {
        if (ctx->dense_pinv) {
            fs.external = reinterpret_cast<FactorSolver::pinv_fn>(ctx->dense_pinv);
            fs.external_user = ctx->dense_pinv_user;
            fs.token = ++ctx->dense_pinv_token;
        }
        fs.prepare(K, Rprod.size() ? Rprod.data() : Rhat.data(), rcond);
    }
    mark("condition bound / prepare");
    auto apply = [&](const double* rhs, double* out) {
        if (use_chain) chain.solve(rhs, out);
        else fs.apply(rhs, out);
    };
    apply(z.data(), beta);
    mark("solve");
    // one refinement step with the residual of the original rows
    double rel_step = 0.0;
    {
        vec dzh((size_t)K, 0.0), dbeta((size_t)K);
        if (have_rows) {
            const unsigned char* mask = ctx->dmask;
            memcpy(hvec, beta, (size_t)K * 8);
            FSNAP_HIP(hipMemcpyAsync(rs->beta.p, hvec, (size_t)K * 8, hipMemcpyHostToDevice, st), "hipMemcpy(beta)");
            FSNAP_HIP(fsnap::launch_gemv_rows(ctx->dA, ctx->lda, (const double*)rs->beta.p, ctx->m, K, nullptr, ctx->db, ctx->dw,
                                              mask, nullptr, (double*)rs->rvec.p, st, true), "launch fsnap_gemv_rows_k");
            FSNAP_HIP(fsnap::launch_gemvT_rows(dQ, K, (const double*)rs->rvec.p, ctx->m, K, (double*)rs->dzpart.p,
                                               (double*)rs->dz.p, st), "launch fsnap_gemvT_rows_k");
            FSNAP_HIP(hipMemcpyAsync(hvec + K, rs->dz.p, (size_t)K * 8, hipMemcpyDeviceToHost, st), "hipMemcpy(dz)");
            if ((rc = fsnap::wait_stream(ctx, nullptr, "row-space refinement"))) return rc;
            memcpy(dzh.data(), hvec + K, (size_t)K * 8);
            mark("refinement: residual pass");
        }
        if (nranks > 1 && (rc = fsnap_allreduce_host(ctx, dzh.data(), K, 0))) return rc;
        if (finite_all(dzh.data(), dzh.size())) {
            apply(dzh.data(), dbeta.data());
            double nb = 0.0, nd = 0.0;
            for (int j = 0; j < K; ++j) {
                nb = std::fmax(nb, std::fabs(beta[j]));
                nd = std::fmax(nd, std::fabs(dbeta[j]));
            }
            rel_step = nb > 0.0 ? nd / nb : 0.0;
            // a step that is not small says Q was not orthonormal yet; it is then the better answer all the same
            for (int j = 0; j < K; ++j) beta[j] += dbeta[j];
        }
    }
    mark("refinement: solve");
    int nact = 0;
    if (use_chain)
        for (int j = 0; j < K; ++j) nact += chain.active[j] ? 1 : 0;
    if (rank_out) *rank_out = use_chain ? nact : fs.rank;
    if (info) {
        info[0] = passes;
        info[1] = dev;
        info[2] = converged;
        info[3] = (use_chain || fs.triangular) ? 0.0 : (fs.deflated ? 3.0 : (fs.use_external ? 2.0 : 1.0));      // 2: the host language's SVD,
                                                                                // 3: dropped triplets projected away (deflate)
        info[4] = use_chain ? chain_norm : fs.smax;                       // chain: bounds, not singular values
        info[5] = use_chain ? (chain_inv > 0.0 ? 1.0 / chain_inv : 0.0) : fs.smin;
        info[6] = rel_step;
        info[7] = shift;
    }
    if (!finite_all(beta, (size_t)K)) return ctx->fail(FSNAP_NUM_NONFINITE, "row-space solve: non-finite coefficients");
    return FSNAP_OK;
}

}  // extern "C"
