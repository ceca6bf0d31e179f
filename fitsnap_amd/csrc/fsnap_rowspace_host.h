// fsnap_rowspace_host.h — the K x K host algebra of the row-space least-squares path (fsnap_rowspace_host.cpp), shared
// with its GPU orchestration (fsnap_rowspace.cpp).  Internal; the public entry points are in include/fsnap_hip.h.
#pragma once
#include <algorithm>
#include <cstddef>
#include <vector>

namespace fsnap_rs {

// n doubles whose CONTENTS ARE UNDEFINED, from a small per-thread pool of blocks that stay mapped between calls.  The K x K
// arrays of the large-K paths were fresh zero-filled vectors per call: 3 - 4 ms each of page faults + memset at K = 1595 (four of
// them in a fit with dependent columns: 14 of 67 ms); the users below overwrite or zero what they need on their threads.
class Scratch {
    double* p_ = nullptr;
    size_t n_ = 0, cap_ = 0;

public:
    Scratch() = default;
    explicit Scratch(size_t n) { reset(n); }
    ~Scratch() { release(); }
    Scratch(const Scratch&) = delete;
    Scratch& operator=(const Scratch&) = delete;
    Scratch(Scratch&& o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_) { o.p_ = nullptr; o.n_ = o.cap_ = 0; }
    Scratch& operator=(Scratch&& o) noexcept {
        if (this != &o) {
            release();
            p_ = o.p_; n_ = o.n_; cap_ = o.cap_;
            o.p_ = nullptr; o.n_ = o.cap_ = 0;
        }
        return *this;
    }
    void reset(size_t n);        // (throws std::bad_alloc like a vector)
    void release();
    double* data() { return p_; }
    const double* data() const { return p_; }
    size_t size() const { return n_; }
    double* begin() { return p_; }
    double* end() { return p_ + n_; }
    const double* begin() const { return p_; }
    const double* end() const { return p_ + n_; }
};


bool finite_all(const double* p, size_t n);
// max |G_ij - delta_ij| over the columns with a non-zero diagonal entry
double gram_deviation(int K, const double* G);
// one sweep over a Gram matrix: FSNAP_NUM_NONFINITE if it holds NaN / Inf; *dev = max |G_ij - delta_ij| and *fro = Frobenius norm
// of the Jacobi-scaled matrix (unit diagonal), both over the columns with a positive diagonal entry -- what a pass needs from
// the host when the factorisation itself runs on the GPU
int gram_scan(int K, const double* G, double* dev, double* fro);
// one pass: see fsnap_rowspace_host.cpp
// Rhat may be NULL: the factor is then only returned in Rp (FactorChain keeps the factors apart)
int factor_pass(int K, const double* G, int first, double tol, double* Rhat, double* Rp, double* dev_out, int* converged,
                double* shift_out);

// The factors of the passes kept apart: R_hat = R_p ... R_2 R_1 is never formed unless a truncation is needed.  Forming it
// costs K^3 / 3 flops per pass on one host core -- 110 ms per pass at K = 1595, where the two GPU passes themselves take
// 14 ms -- while solving through the chain is p back substitutions (K^2 / 2 flops each).
struct FactorChain {
    int K = 0;
    std::vector<const double*> R;          // the factors: K x K row-major (leading dimension K), upper triangular, unit rows /
                                           // columns for inactive columns
    std::vector<std::vector<double>> own;  // storage of the factors this chain copied (push); push_view keeps a pointer only
    std::vector<char> active;              // columns with a non-zero diagonal entry of the first Gram matrix
    void start(int K_, const double* G);
    void push(const double* Rp) {
        own.emplace_back(Rp, Rp + (size_t)K * K);
        R.push_back(own.back().data());
    }
    // a factor that stays where it is (page-locked staging of the device factorisation: a copy of 20 MB is 3 ms at
    // K = 1595); the memory must outlive the chain's use
    void push_view(const double* Rp) { R.push_back(Rp); }
    // upper estimate of ||R_hat||_2 ||R_hat^-1||_2, per factor: sqrt(||R||_1 ||R||_inf) (exact) x an estimate of ||R^-1||_2 --
    // the smaller of 3 x sqrt(est_1 est_inf) (Hager / Higham's 1-norm estimator) and 2 x inverse iteration on R^T R; a
    // Neumann bound 1 / (1 - ||R - I||) for the near-identity factors of the later passes
    double condition_bound(double* norm_out, double* inv_norm_out) const;
    // may the chain be solved through without a truncation for this rcond?  Only when the ESTIMATE (condition_bound:
    // lower-bound estimators x safety factors -- not a bound) leaves two orders of margin: estimate x rcond < 1e-2.
    // Otherwise the caller multiplies the factors out and FactorSolver decides with provable Frobenius bounds / the SVD.
    bool certified(double rcond, double* norm_out, double* inv_norm_out, double* bound_out) const;
    void solve(const double* z, double* beta) const;      // beta = R_1^-1 ... R_p^-1 z, zeros in inactive columns
    void product(double* Rhat) const;                     // R_hat, with zero diagonal entries for inactive columns
};

// the K x K end of dgelsd on the accumulated factor R_hat
struct FactorSolver {
    int K = 0, n = 0, rank = 0;
    std::vector<int> act;
    bool triangular = false;   // no singular value can be below the cut: back substitution
    bool deflated = false;     // a few (ncut <= 24) singular values are below the cut and every other one provably above it: back
                               // substitution between two projections (deflate), no SVD
    int ncut = 0;
    std::vector<double> Uc, Vc;   // the dropped triplets' left / right singular vectors (ncut x n each, orthonormal rows)
    Scratch T;                 // n x n active block of R_hat (row-major, upper; zero below the diagonal)
    std::vector<double> W, J, s2;   // SVD form: rows of W = sigma_i v_i^T (n x n), J = U^T (n x n), s2 = sigma_i^2
    std::vector<char> keep;
    double smax = 0.0, smin = 0.0;
    int sweeps = 0;
    // optional dense kernel of the host language for large truncated solves (include/fsnap_hip.h: fsnap_dense_pinv_fn)
    typedef int (*pinv_fn)(void* user, long long token, long long n, const double* T, double rcond, const double* y, double* x,
                           int* rank);
    pinv_fn external = nullptr;
    void* external_user = nullptr;
    long long token = 0;
    double rcond_used = 0.0;
    mutable bool use_external = false;
    void prepare(int K_, const double* Rhat, double rcond);
    void jacobi_svd(double rcond);
    bool deflate(double rcond, Scratch& X, double norm_bound);   // X = T^-1 on entry (overwritten)
    template <int DB, int MAXCUT>
    int deflate_width(double rcond, Scratch& X, double norm_bound, double lower);   // 1 done, 0 no, -1 more than MAXCUT
    void apply(const double* z, double* beta) const;   // beta (K entries, zeros in inactive columns) = pinv(R_hat) z
};

}  // namespace fsnap_rs
