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
#include <vector>

#include "fsnap_ctx.h"
#include "fsnap_kernels.h"

extern "C" int fsnap_host_chol_upper(double* a, int n, double* min_piv);   // fsnap_solve.cpp

namespace {

using vec = std::vector<double>;
const double EPS = std::numeric_limits<double>::epsilon();

bool finite_all(const double* p, size_t n) {
    double t = 0.0;
    for (size_t i = 0; i < n; ++i) t += p[i] * 0.0;
    return t == 0.0;
}

// max |G_ij - delta_ij| over the columns with a non-zero diagonal entry: how far Q is from orthonormal columns
double gram_deviation(int K, const double* G) {
    double dev = 0.0;
    for (int a = 0; a < K; ++a) {
        if (!(G[(size_t)a * K + a] > 0.0)) continue;
        for (int b = 0; b < K; ++b) {
            if (!(G[(size_t)b * K + b] > 0.0)) continue;
            dev = std::fmax(dev, std::fabs(G[(size_t)a * K + b] - (a == b ? 1.0 : 0.0)));
        }
    }
    return dev;
}

// ---- pass factor --------------------------------------------------------------------------------------------------
// G: K x K Gram matrix of the current Q.  Columns with G_jj == 0 are inactive (zero columns of A_w: coefficient 0, as
// lstsq's minimum-norm solution gives them); their row / column of Rp is the unit vector and their diagonal entry of
// R_hat is set to 0.  Returns the deviation max |G_ij - delta_ij| over the active columns in *dev; when dev <= tol
// nothing is factorised (*converged = 1).  Otherwise Rp (K x K, upper) receives the factor to divide out and
// R_hat <- Rp R_hat.  first = 1: R_hat is initialised to the identity (and the deviation is not a stopping criterion).
int factor_pass(int K, const double* G, int first, double tol, double* Rhat, double* Rp, double* dev_out, int* converged,
                double* shift_out) {
    if (!finite_all(G, (size_t)K * K)) return FSNAP_NUM_NONFINITE;
    std::vector<int> act;
    act.reserve(K);
    for (int j = 0; j < K; ++j)
        if (G[(size_t)j * K + j] > 0.0) act.push_back(j);
    const int n = (int)act.size();
    if (first) {
        std::fill(Rhat, Rhat + (size_t)K * K, 0.0);
        for (int j : act) Rhat[(size_t)j * K + j] = 1.0;
    }
    const double dev = gram_deviation(K, G);
    if (dev_out) *dev_out = dev;
    if (converged) *converged = 0;
    if (shift_out) *shift_out = 0.0;
    if (!first && dev <= tol) {
        if (converged) *converged = 1;
        return FSNAP_OK;
    }
    // identity everywhere, then the active block
    std::fill(Rp, Rp + (size_t)K * K, 0.0);
    for (int j = 0; j < K; ++j) Rp[(size_t)j * K + j] = 1.0;
    if (n == 0) return FSNAP_OK;
    // Jacobi scaling d, scaled active block padded to a multiple of 32 with an identity (full-speed factorisation)
    vec d(n);
    for (int a = 0; a < n; ++a) d[a] = std::sqrt(G[(size_t)act[a] * K + act[a]]);
    const int np = (n >= 48 && (n & 31)) ? ((n + 31) & ~31) : n;
    vec S((size_t)np * np), U((size_t)np * np);
    double fro = 0.0;
    for (int a = 0; a < n; ++a)
        for (int b = a; b < n; ++b) {
            // symmetrise defensively; the GPU reduction mirrors the triangle exactly
            const double g = 0.5 * (G[(size_t)act[a] * K + act[b]] + G[(size_t)act[b] * K + act[a]]) / (d[a] * d[b]);
            S[(size_t)a * np + b] = (a == b) ? 1.0 : g;
            fro += (a == b ? 1.0 : 2.0) * g * g;
        }
    for (int a = n; a < np; ++a) S[(size_t)a * np + a] = 1.0;
    fro = std::sqrt(fro);
    // shift: a few times the rounding level of the Gram matrix (Fukaya et al. 2020 use 11 (mK + K(K+1)) u ||A||^2; the
    // fixed-order MFMA sums here are far below that worst case).  A failed factorisation retries with 100 x the shift.
    double shift = 4.0 * (n + 100.0) * EPS * fro;
    int fail = 0;
    for (int attempt = 0; attempt < 10; ++attempt) {
        U = S;
        for (int a = 0; a < n; ++a) U[(size_t)a * np + a] += shift;
        double mp = 0.0;
        fail = fsnap_host_chol_upper(U.data(), np, &mp);
        if (fail < 0 && finite_all(U.data(), U.size())) break;
        fail = 1;
        shift *= 100.0;
    }
    if (fail >= 0) return FSNAP_NUM_NOT_SPD;
    if (shift_out) *shift_out = shift;
    // Rp = U diag(d) on the active block
    for (int a = 0; a < n; ++a)
        for (int b = a; b < n; ++b) Rp[(size_t)act[a] * K + act[b]] = U[(size_t)a * np + b] * d[b];
    // R_hat <- Rp R_hat (both upper triangular; inactive rows of R_hat are zero rows and stay so)
    vec out((size_t)n * K, 0.0);
    for (int a = 0; a < n; ++a) {
        double* o = out.data() + (size_t)a * K;
        for (int b = a; b < n; ++b) {
            const double f = Rp[(size_t)act[a] * K + act[b]];
            if (f == 0.0) continue;
            const double* r = Rhat + (size_t)act[b] * K;
            for (int c = act[b]; c < K; ++c) o[c] += f * r[c];
        }
    }
    for (int a = 0; a < n; ++a) memcpy(Rhat + (size_t)act[a] * K, out.data() + (size_t)a * K, (size_t)K * sizeof(double));
    return FSNAP_OK;
}

// ---- the K x K end of dgelsd ---------------------------------------------------------------------------------------
struct FactorSolver {
    int K = 0, n = 0, rank = 0;
    std::vector<int> act;
    bool triangular = false;   // no singular value can be below the cut: back substitution
    vec T;                     // n x n active block of R_hat (row-major, upper)
    vec W, J, s2;              // SVD form: rows of W = sigma_i v_i^T (n x n), J = U^T (n x n), s2 = sigma_i^2
    std::vector<char> keep;
    double smax = 0.0, smin = 0.0;
    int sweeps = 0;

    void prepare(int K_, const double* Rhat, double rcond) {
        K = K_;
        act.clear();
        for (int j = 0; j < K; ++j)
            if (Rhat[(size_t)j * K + j] != 0.0) act.push_back(j);
        n = (int)act.size();
        rank = 0;
        T.assign((size_t)n * n, 0.0);
        for (int a = 0; a < n; ++a)
            for (int b = a; b < n; ++b) T[(size_t)a * n + b] = Rhat[(size_t)act[a] * K + act[b]];
        if (n == 0) return;
        // Frobenius bounds: sigma_max <= ||T||_F, sigma_min >= 1 / ||T^-1||_F.  If even these cannot put a singular
        // value below rcond * sigma_max, dgelsd would not truncate either and its solution is T^-1 z.
        double fro = 0.0;
        for (double v : T) fro += v * v;
        fro = std::sqrt(fro);
        double inv2 = 0.0;
        bool ok = true;
        {
            vec X((size_t)n * n, 0.0);     // X = T^-1 by back substitution, column by column of the identity
            for (int c = n - 1; c >= 0 && ok; --c) {
                // solve T x = e_c: x_c = 1 / T_cc, x_i = -(sum_{k>i} T_ik x_k) / T_ii for i < c
                X[(size_t)c * n + c] = 1.0 / T[(size_t)c * n + c];
                for (int i = c - 1; i >= 0; --i) {
                    double s = 0.0;
                    const double* ti = T.data() + (size_t)i * n;
                    for (int k = i + 1; k <= c; ++k) s += ti[k] * X[(size_t)k * n + c];
                    X[(size_t)i * n + c] = -s / ti[i];
                }
            }
            for (double v : X) inv2 += v * v;
            ok = std::isfinite(inv2);
        }
        const double rc = rcond > 0.0 ? rcond : 0.0;
        triangular = ok && (fro * std::sqrt(inv2) * rc < 0.5);
        if (triangular) {
            rank = n;
            smax = fro;
            smin = 1.0 / std::sqrt(inv2);
            return;
        }
        jacobi_svd(rc);
    }

    // one-sided Jacobi on the ROWS of W = T (left rotations): J T = diag(sigma) V^T with J orthogonal.  Rows instead of
    // columns because T is upper triangular (the preconditioned orientation of Drmac & Veselic) and rows are contiguous.
    void jacobi_svd(double rcond) {
        W = T;
        J.assign((size_t)n * n, 0.0);
        for (int i = 0; i < n; ++i) J[(size_t)i * n + i] = 1.0;
        const double tol = std::sqrt((double)n) * EPS;
        auto dot = [&](const double* x, const double* y) {
            double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            int k = 0;
            for (; k + 4 <= n; k += 4) {
                a0 += x[k] * y[k];
                a1 += x[k + 1] * y[k + 1];
                a2 += x[k + 2] * y[k + 2];
                a3 += x[k + 3] * y[k + 3];
            }
            for (; k < n; ++k) a0 += x[k] * y[k];
            return (a0 + a1) + (a2 + a3);
        };
        vec nrm(n);
        for (sweeps = 0; sweeps < 60; ++sweeps) {
            int rotated = 0;
            for (int i = 0; i < n; ++i) nrm[i] = dot(W.data() + (size_t)i * n, W.data() + (size_t)i * n);
            for (int p = 0; p < n - 1; ++p)
                for (int q = p + 1; q < n; ++q) {
                    double* wp = W.data() + (size_t)p * n;
                    double* wq = W.data() + (size_t)q * n;
                    const double al = nrm[p], be = nrm[q];
                    if (al == 0.0 || be == 0.0) continue;
                    const double ga = dot(wp, wq);
                    if (std::fabs(ga) <= tol * std::sqrt(al) * std::sqrt(be)) continue;
                    ++rotated;
                    const double zeta = (be - al) / (2.0 * ga);
                    const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                    const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
                    for (int k = 0; k < n; ++k) {
                        const double a = wp[k], b = wq[k];
                        wp[k] = c * a - s * b;
                        wq[k] = s * a + c * b;
                    }
                    double* jp = J.data() + (size_t)p * n;
                    double* jq = J.data() + (size_t)q * n;
                    for (int k = 0; k < n; ++k) {
                        const double a = jp[k], b = jq[k];
                        jp[k] = c * a - s * b;
                        jq[k] = s * a + c * b;
                    }
                    nrm[p] = dot(wp, wp);       // recomputed, not updated: graded rows lose digits in the update formula
                    nrm[q] = dot(wq, wq);
                }
            if (!rotated) break;
        }
        s2.resize(n);
        smax = 0.0;
        for (int i = 0; i < n; ++i) {
            s2[i] = dot(W.data() + (size_t)i * n, W.data() + (size_t)i * n);
            smax = std::fmax(smax, s2[i]);
        }
        keep.assign(n, 0);
        rank = 0;
        smin = std::sqrt(smax);
        const double cut2 = rcond * rcond * smax;      // sigma_i > rcond sigma_max  <=>  sigma_i^2 > rcond^2 sigma_max^2
        for (int i = 0; i < n; ++i)
            if (s2[i] > cut2 && s2[i] > 0.0) {
                keep[i] = 1;
                ++rank;
                smin = std::fmin(smin, std::sqrt(s2[i]));
            }
        smax = std::sqrt(smax);
    }

    // beta (K entries, zeros in inactive columns) = pinv(R_hat) z
    void apply(const double* z, double* beta) const {
        for (int j = 0; j < K; ++j) beta[j] = 0.0;
        if (n == 0) return;
        vec y(n);
        for (int a = 0; a < n; ++a) y[a] = z[act[a]];
        if (triangular) {
            for (int i = n - 1; i >= 0; --i) {
                const double* ti = T.data() + (size_t)i * n;
                double s = y[i];
                for (int k = i + 1; k < n; ++k) s -= ti[k] * y[k];
                y[i] = s / ti[i];
            }
            for (int a = 0; a < n; ++a) beta[act[a]] = y[a];
            return;
        }
        vec x(n, 0.0);
        for (int i = 0; i < n; ++i) {
            if (!keep[i]) continue;
            const double* ji = J.data() + (size_t)i * n;
            double t = 0.0;
            for (int k = 0; k < n; ++k) t += ji[k] * y[k];
            const double f = t / s2[i];
            const double* wi = W.data() + (size_t)i * n;
            for (int k = 0; k < n; ++k) x[k] += f * wi[k];
        }
        for (int a = 0; a < n; ++a) beta[act[a]] = x[a];
    }
};

}  // namespace

namespace fsnap {

struct RowSpace {
    DevBuf Q, qpack, Rdev, packed, rvec, dz, dzpart, beta;
};

void rowspace_release(fsnap_ctx* ctx) {
    if (!ctx->rowspace) return;
    RowSpace* rs = ctx->rowspace;
    DevBuf* bufs[] = {&rs->Q, &rs->qpack, &rs->Rdev, &rs->packed, &rs->rvec, &rs->dz, &rs->dzpart, &rs->beta};
    for (DevBuf* b : bufs) b->release();
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
        info[0] = fs.triangular ? 0.0 : 1.0;
        info[1] = fs.smax;
        info[2] = fs.smin;
        info[3] = fs.sweeps;
    }
    return finite_all(beta, (size_t)K) ? FSNAP_OK : FSNAP_NUM_NONFINITE;
}

int fsnap_lstsq_rows(fsnap_ctx* ctx, double rcond, int64_t K64, double* beta, int* rank_out, double* info) {
    if (!ctx) return FSNAP_E_ARG;
    if (!beta || K64 <= 0) return ctx->fail(FSNAP_E_ARG, "fsnap_lstsq_rows: bad argument");
    const int K = (int)K64, K16 = (K + 15) & ~15;
    const int64_t npk = FSNAP_PACKED_LEN(K64);
    const int nranks = ctx->comm ? 2 : 1;     // "> 1" = collective (a communicator of one rank takes the same path)
    const bool have_rows = ctx->dA && ctx->m > 0;
    if (have_rows && ctx->K != K64)
        return ctx->fail(FSNAP_E_ARG, "fsnap_lstsq_rows: K = %d but the resident rows have %lld columns", K, (long long)ctx->K);
    if (!have_rows && nranks == 1) return ctx->fail(FSNAP_E_STATE, "no rows: call fsnap_upload_rows/fsnap_bind_rows first");
    if (have_rows && !ctx->dw) return ctx->fail(FSNAP_E_STATE, "no weights: call fsnap_set_weights/fsnap_bind_weights first");
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    if (!ctx->rowspace) {
        ctx->rowspace = new (std::nothrow) fsnap::RowSpace();
        if (!ctx->rowspace) return ctx->fail(FSNAP_E_NOMEM, "out of host memory");
    }
    fsnap::RowSpace* rs = ctx->rowspace;
    const size_t m = have_rows ? (size_t)ctx->m : 0;
    if (!rs->packed.ensure((size_t)npk * 8) || !rs->Rdev.ensure((size_t)K16 * K16 * 8) || !rs->beta.ensure((size_t)K * 8) ||
        !rs->dz.ensure((size_t)K * 8))
        return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(row-space workspace) failed");
    if (have_rows) {
        const int nbt = fsnap::gemvT_num_blocks(ctx->m);
        if (!rs->Q.ensure(m * (size_t)K * 8 + 256) || !rs->qpack.ensure(m * 16 + 64) || !rs->rvec.ensure(m * 8) ||
            !rs->dzpart.ensure((size_t)nbt * K * 8))
            return ctx->fail(FSNAP_E_NOMEM, "hipMalloc of %zu bytes for the orthogonalised rows failed", m * (size_t)K * 8);
    }
    double* dp = (double*)rs->packed.p;
    double* dQ = (double*)rs->Q.p;
    hipStream_t st = ctx->stream;
    int rc;
    vec host((size_t)npk), Rhat((size_t)K * K), Rp((size_t)K * K), Rpad((size_t)K16 * K16), z((size_t)K);

    // statistics of the current Q (pass 0: of A_w), summed over the ranks, on the host
    auto gather_stats = [&](bool of_rows) -> int {
        if (have_rows) {
            int r2 = of_rows ? fsnap_normal_eq_async(ctx, dp) : fsnap::normal_eq_launch_on(ctx, dQ, K, (const double*)rs->qpack.p, dp);
            if (r2) return r2;
        } else {
            FSNAP_HIP(hipMemsetAsync(dp, 0, (size_t)npk * 8, st), "hipMemsetAsync(packed)");
        }
        if (nranks > 1) {
            int r3 = fsnap_allreduce_device(ctx, dp, npk);
            if (r3) return r3;
        }
        FSNAP_HIP(hipMemcpyAsync(host.data(), dp, (size_t)npk * 8, hipMemcpyDeviceToHost, st), "hipMemcpy(statistics)");
        FSNAP_HIP(hipStreamSynchronize(st), "hipStreamSynchronize");
        return FSNAP_OK;
    };
    if ((rc = gather_stats(true))) return rc;
    if (have_rows) {
        // (w_eff, w_eff b) of the fit are current now (the launch above packed them if needed)
        FSNAP_HIP(fsnap::launch_qpack((const double*)ctx->wpack.p, ctx->m, (double*)rs->qpack.p, st), "launch fsnap_qpack_k");
        FSNAP_HIP(hipMemsetAsync((char*)rs->Q.p + m * (size_t)K * 8, 0, 256, st), "hipMemsetAsync");   // tail pad of the SYRK loads
    }
    const double tol = 1.0e-10;
    const int maxpass = 6;
    int passes = 0, converged = 0;
    double dev = 0.0, shift = 0.0;
    for (int pass = 1; pass <= maxpass; ++pass) {
        int conv = 0;
        rc = factor_pass(K, host.data(), pass == 1, tol, Rhat.data(), Rp.data(), &dev, &conv, &shift);
        if (rc) return ctx->fail(rc, "row-space pass %d: the Gram matrix could not be factorised (status %d)", pass, rc);
        if (conv) {
            converged = 1;
            break;
        }
        // padded copy of the factor for the kernel
        std::fill(Rpad.begin(), Rpad.end(), 0.0);
        for (int i = 0; i < K16; ++i) Rpad[(size_t)i * K16 + i] = 1.0;
        for (int i = 0; i < K; ++i) memcpy(Rpad.data() + (size_t)i * K16 + i, Rp.data() + (size_t)i * K + i, (size_t)(K - i) * 8);
        FSNAP_HIP(hipMemcpyAsync(rs->Rdev.p, Rpad.data(), Rpad.size() * 8, hipMemcpyHostToDevice, st), "hipMemcpy(R)");
        if (have_rows) {
            if (pass == 1)
                FSNAP_HIP(fsnap::launch_trsm_rows(ctx->dA, ctx->lda, (const double*)ctx->wpack.p, dQ, K, ctx->m, K,
                                                  (const double*)rs->Rdev.p, K16, st), "launch fsnap_trsm_rows_k");
            else
                FSNAP_HIP(fsnap::launch_trsm_rows(dQ, K, nullptr, dQ, K, ctx->m, K, (const double*)rs->Rdev.p, K16, st),
                          "launch fsnap_trsm_rows_k");
        }
        FSNAP_HIP(hipStreamSynchronize(st), "hipStreamSynchronize");      // Rpad is reused by the next pass
        passes = pass;
        if ((rc = gather_stats(false))) return rc;
        memcpy(z.data(), host.data() + (size_t)K * K, (size_t)K * 8);     // z = Q^T (w b)
    }
    if (!converged) {
        // pass budget used up: judge the last Q as it is (the refinement step below absorbs what is left)
        if (!finite_all(host.data(), (size_t)K * K)) return ctx->fail(FSNAP_NUM_NONFINITE, "row-space solve: non-finite Gram matrix");
        dev = gram_deviation(K, host.data());
        converged = dev <= tol;
    }
    if (passes == 0) return ctx->fail(FSNAP_E_STATE, "row-space solve made no pass");
    if (!finite_all(z.data(), z.size())) return ctx->fail(FSNAP_NUM_NONFINITE, "row-space solve: non-finite Q^T b");

    FactorSolver fs;
    fs.prepare(K, Rhat.data(), rcond);
    fs.apply(z.data(), beta);
    // one refinement step with the residual of the original rows
    double rel_step = 0.0;
    {
        vec dzh((size_t)K, 0.0), dbeta((size_t)K);
        if (have_rows) {
            const unsigned char* mask = ctx->dmask;
            FSNAP_HIP(hipMemcpyAsync(rs->beta.p, beta, (size_t)K * 8, hipMemcpyHostToDevice, st), "hipMemcpy(beta)");
            FSNAP_HIP(fsnap::launch_gemv_rows(ctx->dA, ctx->lda, (const double*)rs->beta.p, ctx->m, K, nullptr, ctx->db, ctx->dw,
                                              mask, nullptr, (double*)rs->rvec.p, st, true), "launch fsnap_gemv_rows_k");
            FSNAP_HIP(fsnap::launch_gemvT_rows(dQ, K, (const double*)rs->rvec.p, ctx->m, K, (double*)rs->dzpart.p,
                                               (double*)rs->dz.p, st), "launch fsnap_gemvT_rows_k");
            FSNAP_HIP(hipMemcpyAsync(dzh.data(), rs->dz.p, (size_t)K * 8, hipMemcpyDeviceToHost, st), "hipMemcpy(dz)");
            FSNAP_HIP(hipStreamSynchronize(st), "hipStreamSynchronize");
        }
        if (nranks > 1 && (rc = fsnap_allreduce_host(ctx, dzh.data(), K, 0))) return rc;
        if (finite_all(dzh.data(), dzh.size())) {
            fs.apply(dzh.data(), dbeta.data());
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
    if (rank_out) *rank_out = fs.rank;
    if (info) {
        info[0] = passes;
        info[1] = dev;
        info[2] = converged;
        info[3] = fs.triangular ? 0.0 : 1.0;
        info[4] = fs.smax;
        info[5] = fs.smin;
        info[6] = rel_step;
        info[7] = shift;
    }
    if (!finite_all(beta, (size_t)K)) return ctx->fail(FSNAP_NUM_NONFINITE, "row-space solve: non-finite coefficients");
    return FSNAP_OK;
}

}  // extern "C"
