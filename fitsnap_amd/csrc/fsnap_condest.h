// Condition estimate behind a Cholesky factorisation (host code; shared by the host solve and the device solve).
//
// The reference's default solver is an SVD (fitsnap3lib/solvers/svd.py:54, scipy lstsq = dgelsd): it KNOWS the
// conditioning of the rows.  This path solves the Jacobi-scaled normal equations S = D (G + alpha I) D = U^T U by
// Cholesky; what it knew about the conditioning until round 5 was the smallest pivot, which bounds lambda_min(S) from
// ABOVE only -- without pivoting by a factor that grows exponentially with K (A = Z (I - triu(1, 1)), K = 26: smallest
// pivot 0.04, lambda_min 1e-16).  Every decision that depends on the conditioning (skip / stop the refinement, go to the
// row-space solve) now rests on lambda_min(S) estimated FROM THE FACTOR, the way LAPACK's dpocon looks behind dpotrf:
// a few applications of S^-1 = U^-1 U^-T (two triangular sweeps, 2 K^2 flops each).
//
// The estimator is the Lanczos process on S^-1 with full re-orthogonalisation: after j steps the largest Ritz value
// theta_j <= 1 / lambda_min, monotonically increasing in j, and converging like the Chebyshev polynomial of degree j in
// the gap -- one step already returns (v0 . e_min)^2 / lambda_min ~ 1 / (K lambda_min) for an isolated small eigenvalue,
// the second nearly all of it.  The estimate 1 / theta_j is therefore an estimate FROM ABOVE; callers divide by an explicit
// margin (solvers/solver.py: RCOND_MARGIN).  The start vector is a fixed pseudo-random sequence: same bits on every
// rank of a multi-GPU job (the factor comes from bit-identical all-reduced statistics).
#pragma once

#include <cmath>
#include <cstdint>
#include <vector>

namespace fsnap {

struct CondEstimate {
    double lambda_min = 0.0;   // estimate (from above) of the smallest eigenvalue of the scaled matrix; 0 = numerically singular
    int steps = 0;             // applications of S^-1 taken
};

// largest eigenvalue of the symmetric tridiagonal (a[0..n), b[0..n-1)) by bisection on the Sturm count
inline double tridiag_lambda_max(const double* a, const double* b, int n) {
    double lo = a[0], hi = a[0];
    for (int i = 0; i < n; ++i) {
        const double r = (i > 0 ? std::fabs(b[i - 1]) : 0.0) + (i + 1 < n ? std::fabs(b[i]) : 0.0);
        if (a[i] - r < lo) lo = a[i] - r;
        if (a[i] + r > hi) hi = a[i] + r;
    }
    if (n == 1) return a[0];
    for (int it = 0; it < 200 && hi - lo > 4.0e-16 * std::fmax(std::fabs(hi), std::fabs(lo)); ++it) {
        const double x = 0.5 * (lo + hi);
        // number of eigenvalues above x = number of positive terms of the LDL^T recurrence of (T - x I)
        int above = 0;
        double d = 1.0;
        for (int i = 0; i < n; ++i) {
            const double off = i > 0 ? b[i - 1] * b[i - 1] : 0.0;
            d = (a[i] - x) - (i > 0 ? off / d : 0.0);
            if (d == 0.0) d = -1.0e-300;
            if (d > 0.0) ++above;
        }
        if (above > 0) lo = x; else hi = x;
    }
    return 0.5 * (lo + hi);
}

// apply_inv(x): x <- S^-1 x in place (n doubles); returns false when the operator failed (non-finite result, device
// error): the estimate is then 0 (numerically singular) and the caller takes its general path.
// Stops after >= min_steps steps once a step raises the Ritz value by less than 25 %, at max_steps, or when the Krylov
// space closes (an invariant subspace: the Ritz value is exact).  `growth`: the stopping ratio (1.25 = the 25 % above);
// `theta_trace` (max_steps doubles, optional): the largest Ritz value after every step taken, zeros behind the last one.
template <class ApplyInv>
CondEstimate lanczos_lambda_min(int n, ApplyInv&& apply_inv, int min_steps = 2, int max_steps = 8, double growth = 1.25,
                                double* theta_trace = nullptr) {
    CondEstimate out;
    if (n <= 0) return out;
    if (max_steps > 16) max_steps = 16;
    if (theta_trace)
        for (int j = 0; j < max_steps; ++j) theta_trace[j] = 0.0;
    std::vector<double> V((size_t)(max_steps + 1) * n), w(n);
    double al[16], be[16];
    // fixed start vector: a 64-bit LCG mapped to (-1, 1), never near zero in every component at once
    uint64_t s = 0x9E3779B97F4A7C15ull;
    double nrm = 0.0;
    for (int i = 0; i < n; ++i) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const double u = (double)(int64_t)(s >> 11) * (1.0 / 9007199254740992.0);          // [0, 1)
        const double v = (u < 0.5 ? -1.0 : 1.0) * (0.25 + 0.75 * std::fabs(2.0 * u - 1.0));
        V[i] = v;
        nrm += v * v;
    }
    nrm = 1.0 / std::sqrt(nrm);
    for (int i = 0; i < n; ++i) V[i] *= nrm;
    double theta_prev = 0.0, theta = 0.0;
    for (int j = 0; j < max_steps; ++j) {
        double* vj = V.data() + (size_t)j * n;
        for (int i = 0; i < n; ++i) w[i] = vj[i];
        if (!apply_inv(w.data())) return CondEstimate{0.0, j + 1};
        double a = 0.0, chk = 0.0;
        for (int i = 0; i < n; ++i) {
            a += vj[i] * w[i];
            chk += w[i] * 0.0;
        }
        if (chk != 0.0 || !(a > 0.0)) return CondEstimate{0.0, j + 1};      // S^-1 is SPD: anything else is a broken factor
        al[j] = a;
        out.steps = j + 1;
        theta = tridiag_lambda_max(al, be, j + 1);
        if (!(theta > 0.0) || !std::isfinite(theta)) return CondEstimate{0.0, j + 1};
        if (theta_trace) theta_trace[j] = theta;
        if (j + 1 >= min_steps && theta <= growth * theta_prev) break;
        if (j + 1 == max_steps) break;
        theta_prev = theta;
        // next Lanczos vector: full re-orthogonalisation (twice), the Krylov space is at most 16 vectors
        for (int pass = 0; pass < 2; ++pass)
            for (int p = 0; p <= j; ++p) {
                const double* vp = V.data() + (size_t)p * n;
                double d = 0.0;
                for (int i = 0; i < n; ++i) d += vp[i] * w[i];
                for (int i = 0; i < n; ++i) w[i] -= d * vp[i];
            }
        double b = 0.0;
        for (int i = 0; i < n; ++i) b += w[i] * w[i];
        b = std::sqrt(b);
        if (!(b > 1.0e-14 * theta)) break;          // invariant subspace: theta is an eigenvalue of S^-1
        be[j] = b;
        double* vn = V.data() + (size_t)(j + 1) * n;
        const double ib = 1.0 / b;
        for (int i = 0; i < n; ++i) vn[i] = w[i] * ib;
    }
    out.lambda_min = 1.0 / theta;
    return out;
}

}  // namespace fsnap
