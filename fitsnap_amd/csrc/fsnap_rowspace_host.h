// fsnap_rowspace_host.h — the K x K host algebra of the row-space least-squares path (fsnap_rowspace_host.cpp), shared
// with its GPU orchestration (fsnap_rowspace.cpp).  Internal; the public entry points are in include/fsnap_hip.h.
#pragma once
#include <vector>

namespace fsnap_rs {

bool finite_all(const double* p, size_t n);
// max |G_ij - delta_ij| over the columns with a non-zero diagonal entry
double gram_deviation(int K, const double* G);
// one pass: see fsnap_rowspace_host.cpp
int factor_pass(int K, const double* G, int first, double tol, double* Rhat, double* Rp, double* dev_out, int* converged,
                double* shift_out);

// the K x K end of dgelsd on the accumulated factor R_hat
struct FactorSolver {
    int K = 0, n = 0, rank = 0;
    std::vector<int> act;
    bool triangular = false;   // no singular value can be below the cut: back substitution
    std::vector<double> T;     // n x n active block of R_hat (row-major, upper)
    std::vector<double> W, J, s2;   // SVD form: rows of W = sigma_i v_i^T (n x n), J = U^T (n x n), s2 = sigma_i^2
    std::vector<char> keep;
    double smax = 0.0, smin = 0.0;
    int sweeps = 0;
    void prepare(int K_, const double* Rhat, double rcond);
    void jacobi_svd(double rcond);
    void apply(const double* z, double* beta) const;   // beta (K entries, zeros in inactive columns) = pinv(R_hat) z
};

}  // namespace fsnap_rs
