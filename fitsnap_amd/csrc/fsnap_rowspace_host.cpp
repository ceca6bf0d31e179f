// fsnap_rowspace_host.cpp — host (K x K) half of the row-space least-squares path: the factor of a CholeskyQR pass and
// the K x K end of dgelsd (back substitution or one-sided Jacobi SVD of the accumulated triangular factor).  Plain C++
// compiled with AVX2 + FMA like fsnap_solve.cpp (the Jacobi sweeps are dot products and plane rotations of contiguous
// rows); the GPU orchestration is in fsnap_rowspace.cpp, the algorithm is described there.
// Reference semantics: scipy.linalg.lstsq(aw, bw, 1.0e-13), fitsnap3lib/solvers/svd.py:54.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <functional>
#include <limits>
#include <new>
#include <thread>
#include <vector>

#include "../../include/fsnap_hip.h"
#include "fsnap_condest.h"
#include "fsnap_rowspace_host.h"

extern "C" int fsnap_host_chol_upper(double* a, int n, double* min_piv);   // fsnap_solve.cpp

namespace fsnap_rs {
// FSNAP_ROWSPACE_TIMING=1 (the switch of fsnap_lstsq_rows' phase marks): wall time of the host phases below on stderr
struct HostProf {
    const char* what;
    std::chrono::steady_clock::time_point t0;
    explicit HostProf(const char* w) : what(w), t0(std::chrono::steady_clock::now()) {}
    ~HostProf() {
        if (getenv("FSNAP_ROWSPACE_TIMING"))
            fprintf(stderr, "[fsnap_rowspace_host]   %-30s %8.3f ms\n", what,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
};

// ... and a sum over the repetitions of a phase inside a loop (printed when it goes out of scope)
struct HostProfSum {
    const char* what;
    double ms = 0.0;
    int calls = 0;
    explicit HostProfSum(const char* w) : what(w) {}
    template <class F>
    void time(F&& f) {
        const auto t0 = std::chrono::steady_clock::now();
        f();
        ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        ++calls;
    }
    ~HostProfSum() {
        if (calls && getenv("FSNAP_ROWSPACE_TIMING")) fprintf(stderr, "[fsnap_rowspace_host]     %-28s %8.3f ms in %d calls\n", what, ms, calls);
    }
};

using vec = std::vector<double>;

// ---- Scratch: per-thread pool of large blocks ----------------------------------------------------------------------------------
namespace {
struct ScratchPool {
    static constexpr int SLOTS = 4;
    static constexpr size_t MAX_KEEP = (size_t)8 << 20;          // doubles: blocks above 64 MB are not kept
    double* p[SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    size_t cap[SLOTS] = {0, 0, 0, 0};
    bool alive = true;                                           // false once the thread's pool has been torn down: a Scratch that
                                                                 // outlives it (an object destroyed late in thread / process exit)
                                                                 // frees its block instead of parking it
    ~ScratchPool() {
        alive = false;
        for (int i = 0; i < SLOTS; ++i) {
            free(p[i]);
            p[i] = nullptr;
            cap[i] = 0;
        }
    }
};
thread_local ScratchPool g_scratch;
}  // namespace

void Scratch::reset(size_t n) {
    if (n <= cap_ && p_) {
        n_ = n;
        return;
    }
    release();
    if (n == 0) return;
    ScratchPool& pool = g_scratch;
    int best = -1;
    for (int i = 0; pool.alive && i < ScratchPool::SLOTS; ++i)
        if (pool.p[i] && pool.cap[i] >= n && (best < 0 || pool.cap[i] < pool.cap[best])) best = i;
    if (best >= 0) {
        p_ = pool.p[best];
        cap_ = pool.cap[best];
        pool.p[best] = nullptr;
        pool.cap[best] = 0;
    } else {
        const size_t bytes = ((n * sizeof(double) + 63) / 64) * 64;
        p_ = static_cast<double*>(aligned_alloc(64, bytes));
        if (!p_) throw std::bad_alloc();
        cap_ = bytes / sizeof(double);
    }
    n_ = n;
}

void Scratch::release() {
    if (!p_) return;
    ScratchPool& pool = g_scratch;
    int slot = -1;
    if (pool.alive && cap_ <= ScratchPool::MAX_KEEP) {
        for (int i = 0; i < ScratchPool::SLOTS && slot < 0; ++i)
            if (!pool.p[i]) slot = i;
        if (slot < 0) {                                 // full: replace the smallest block if this one is larger
            int smallest = 0;
            for (int i = 1; i < ScratchPool::SLOTS; ++i)
                if (pool.cap[i] < pool.cap[smallest]) smallest = i;
            if (pool.cap[smallest] < cap_) {
                free(pool.p[smallest]);
                pool.p[smallest] = nullptr;
                slot = smallest;
            }
        }
    }
    if (slot >= 0) {
        pool.p[slot] = p_;
        pool.cap[slot] = cap_;
    } else {
        free(p_);
    }
    p_ = nullptr;
    n_ = cap_ = 0;
}
static const double EPS = std::numeric_limits<double>::epsilon();

// ---- host threads for the O(n^3) / many-solve phases of large systems ---------------------------------------------------------
// At K = 1595 the K x K end of an ill-conditioned fit was 115 ms (condition estimate: ~100 triangular solves of 10 MB each) to
// 436 ms (product of the factors + explicit inverse + subspace iteration) of host time on ONE core behind 9 ms of GPU passes
// (profiles/r05_rowspace_large_k.txt).  The loops below split over FSNAP_HOST_THREADS threads (default: up to 16) from n = 384 on;
// threads are created per phase (a phase is milliseconds, a thread start ~50 us) and joined before it returns.  Every split is
// over independent outputs (rows, columns, whole estimators): the results do not depend on the thread count.
static int host_threads(int n) {
    if (n < 384) return 1;
    static const int cap = [] {
        const char* e = getenv("FSNAP_HOST_THREADS");
        int v = e && *e ? atoi(e) : 16;
        const unsigned hc = std::thread::hardware_concurrency();
        if (hc > 0 && v > (int)hc) v = (int)hc;
        if (!(e && *e)) {
            // a cgroup CPU quota (cpu.max "quota period", or cfs_quota_us / cfs_period_us): more runnable threads than the
            // quota pays for are throttled for the rest of the period -- never more threads than whole CPUs of quota
            double cpus = 0.0;
            if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
                char q[32] = {0};
                double period = 0.0;
                if (fscanf(f, "%31s %lf", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0.0) cpus = atof(q) / period;
                fclose(f);
            } else {
                double quota = -1.0, period = 0.0;
                if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
                    if (fscanf(g, "%lf", &quota) != 1) quota = -1.0;
                    fclose(g);
                }
                if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                    if (fscanf(g, "%lf", &period) != 1) period = 0.0;
                    fclose(g);
                }
                if (quota > 0.0 && period > 0.0) cpus = quota / period;
            }
            // ... and not all of it: the caller's other threads (the HIP runtime's, a spinning stream wait) run on the same quota, and
            // a cgroup that overdraws it is stopped for the rest of the 100 ms period (seen as 80 ms outliers of a 17 ms phase
            // with 16 threads on a 16-CPU quota): three quarters
            if (cpus > 0.0 && (double)v > 0.75 * cpus) v = 0.75 * cpus < 1.0 ? 1 : (int)(0.75 * cpus);
        }
        return v < 1 ? 1 : (v > 64 ? 64 : v);
    }();
    return cap;
}

// ... and no more threads than the work feeds: a thread start costs ~30 us, so every thread should get a few hundred microseconds
// of it (3e5 multiply-adds of these memory-bound loops).  At n = 480 a power-iteration step is 30 us of work: 16 threads per
// step made it 1 ms.
static int threads_for(int n, double work) {
    const int cap = host_threads(n);
    static const double grain = [] {
        const char* e = getenv("FSNAP_HOST_GRAIN");              // multiply-adds per thread (tuning aid)
        const double v = e && *e ? atof(e) : 3.0e5;
        return v >= 1.0e3 ? v : 3.0e5;
    }();
    const double want = work / grain;
    return want < 2.0 ? 1 : (want > (double)cap ? cap : (int)want);
}

// f(t) for t = 0 .. nt - 1, the caller being thread 0
template <class F>
static void run_threads(int nt, F&& f) {
    if (nt <= 1) {
        f(0);
        return;
    }
    std::vector<std::thread> th;
    th.reserve((size_t)nt - 1);
    int started = 1;
    for (int t = 1; t < nt; ++t) {
        try {
            th.emplace_back([&f, t] { f(t); });
            ++started;
        } catch (...) {
            break;                                  // no more threads to be had: the caller takes the rest
        }
    }
    f(0);
    for (int t = started; t < nt; ++t) f(t);
    for (auto& x : th) x.join();
}

// independent tasks, at most `nt` at a time (task i runs on thread i % nt)
static void run_tasks(int nt, std::vector<std::function<void()>>& tasks) {
    const int n = (int)tasks.size();
    if (nt > n) nt = n;
    run_threads(nt, [&](int t) {
        for (int i = t; i < n; i += nt) tasks[i]();
    });
}

// sum of squares with eight independent partial sums in a fixed order (one dependent chain costs 4 cycles per entry)
static double sum_squares(const double* p, size_t n) {
    double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t i = 0;
    for (; i + 8 <= n; i += 8)
        for (int j = 0; j < 8; ++j) t[j] += p[i + j] * p[i + j];
    double r = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    for (; i < n; ++i) r += p[i] * p[i];
    return r;
}

// (eight partial sums in a fixed order: one dependent chain costs 4 cycles per entry)
static inline double dot_n(const double* x, const double* y, int n) {
    double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int k = 0;
    for (; k + 8 <= n; k += 8)
        for (int j = 0; j < 8; ++j) t[j] += x[k + j] * y[k + j];
    double r = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    for (; k < n; ++k) r += x[k] * y[k];
    return r;
}

bool finite_all(const double* p, size_t n) {
    // x * 0 is 0 for finite x and NaN for NaN / Inf; eight independent sums (one dependent chain costs 4 cycles per entry:
    // 10 ms for the 1595 x 1595 statistics)
    double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t i = 0;
    for (; i + 8 <= n; i += 8)
        for (int k = 0; k < 8; ++k) t[k] += p[i + k] * 0.0;
    for (; i < n; ++i) t[0] += p[i] * 0.0;
    return ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7])) == 0.0;
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

int gram_scan(int K, const double* G, double* dev_out, double* fro_out) {
    if (!finite_all(G, (size_t)K * K)) return FSNAP_NUM_NONFINITE;
    vec dinv((size_t)K, 0.0);
    for (int j = 0; j < K; ++j)
        if (G[(size_t)j * K + j] > 0.0) dinv[j] = 1.0 / std::sqrt(G[(size_t)j * K + j]);
    double dev = 0.0, fro2 = 0.0;
    for (int a = 0; a < K; ++a) {
        if (!(dinv[a] > 0.0)) continue;
        const double* ga = G + (size_t)a * K;
        double dmax = 0.0, f = 0.0;
        for (int b = 0; b < K; ++b) {
            const double act = dinv[b] > 0.0 ? 1.0 : 0.0;
            const double g = ga[b];
            dmax = std::fmax(dmax, act * std::fabs(g - (a == b ? 1.0 : 0.0)));
            const double sc = (a == b) ? act : g * dinv[a] * dinv[b];
            f += sc * sc;
        }
        dev = std::fmax(dev, dmax);
        fro2 += f;
    }
    if (dev_out) *dev_out = dev;
    if (fro_out) *fro_out = std::sqrt(fro2);
    return FSNAP_OK;
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
    if (first && Rhat) {
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
    // scratch of two np x np matrices, kept between calls (a fresh 20 MB vector is 5 ms of page faults at K = 1595)
    static thread_local vec S, U;
    S.assign((size_t)np * np, 0.0);
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
    if (!Rhat) return FSNAP_OK;
    if (first) {
        // R_hat was the identity on the active block: the product is Rp's active block itself
        for (int a = 0; a < n; ++a)
            for (int b = a; b < n; ++b) Rhat[(size_t)act[a] * K + act[b]] = Rp[(size_t)act[a] * K + act[b]];
        return FSNAP_OK;
    }
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
// ---- factor chain -------------------------------------------------------------------------------------------------
namespace {

// x <- T^-1 x for an upper triangular T (row-oriented: contiguous dot products)
void solve_upper(int n, const double* T, double* x) {
    for (int i = n - 1; i >= 0; --i) {
        const double* ti = T + (size_t)i * n;
        x[i] = (x[i] - dot_n(ti + i + 1, x + i + 1, n - i - 1)) / ti[i];      // (dot_n: eight partial sums, not one chain of n - i)
    }
}

// x <- T^-T x (forward substitution on T^T, written as axpys of the contiguous rows of T)
void solve_upper_transposed(int n, const double* T, double* x) {
    for (int i = 0; i < n; ++i) {
        const double* ti = T + (size_t)i * n;
        const double xi = x[i] / ti[i];
        x[i] = xi;
        if (xi != 0.0)
            for (int k = i + 1; k < n; ++k) x[k] -= ti[k] * xi;
    }
}

// ---- cooperative triangular solves for large n ---------------------------------------------------------------------------------
// One substitution with a 1595 x 1595 factor streams 10 MB through one core: 0.7 ms, and the condition estimate of a factor chain
// is ~50 of them in sequence.  A TriTeam runs the substitutions of ONE estimator on a few threads: blocks of 64 unknowns are
// solved by the owner thread, then every thread applies them to its share of the remaining right-hand side (rows dealt round-robin
// for T^-1 x, contiguous ranges of the later entries for T^-T x), two spin barriers per block.  Every entry receives the same
// operations in the same order whatever the number of threads (also with one): the result does not depend on it.  The workers
// spin (then yield) between the solves of an estimator -- the serial work in between is microseconds -- and are joined with it.
class SpinBarrier {
    int n;
    std::atomic<int> arrived{0}, gen{0};

public:
    explicit SpinBarrier(int n_) : n(n_) {}
    void resize(int n_) { n = n_; }                     // before the first wait only
    void wait() {
        if (n <= 1) return;
        const int g = gen.load(std::memory_order_acquire);
        if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
            arrived.store(0, std::memory_order_relaxed);
            gen.store(g + 1, std::memory_order_release);
            return;
        }
        for (int spins = 0; gen.load(std::memory_order_acquire) == g; ++spins) {
            if (spins < 4000) __builtin_ia32_pause();
            else std::this_thread::yield();             // a descheduled team mate (CPU quota): do not burn its time slice
        }
    }
};

// f(t, team) for t = 0 .. team - 1, for bodies that meet at `bar`: the threads are created FIRST and the barrier is sized to the
// number that actually started (thread creation fails under a pids cgroup limit / EAGAIN: run_threads' fallback -- the caller
// runs the missing bodies afterwards -- would leave f(0) waiting at the barrier for threads that do not exist)
template <class F>
static void run_team(int nt, SpinBarrier& bar, F&& f) {
    if (nt <= 1) {
        bar.resize(1);
        f(0, 1);
        return;
    }
    std::vector<std::thread> th;
    th.reserve((size_t)nt - 1);
    std::atomic<int> team{0};
    for (int t = 1; t < nt; ++t) {
        try {
            th.emplace_back([&f, &team, t] {
                int n;
                for (int spins = 0; (n = team.load(std::memory_order_acquire)) == 0; ++spins) {
                    if (spins < 4000) __builtin_ia32_pause();
                    else std::this_thread::yield();
                }
                f(t, n);
            });
        } catch (...) {
            break;
        }
    }
    const int n = (int)th.size() + 1;
    bar.resize(n);
    team.store(n, std::memory_order_release);
    f(0, n);
    for (auto& x : th) x.join();
}

class TriTeam {
    static constexpr int NB = 64;
    const int n;
    int nt;
    SpinBarrier bar;
    std::vector<std::thread> workers;
    std::atomic<int> job_gen{0};
    const double* T = nullptr;
    double* x = nullptr;
    int kind = 0;                                       // 0: x <- T^-1 x, 1: x <- T^-T x, -1: leave

    void upper(int t) {
        for (int b1 = n; b1 > 0; b1 -= NB) {
            const int b0 = std::max(0, b1 - NB);
            if (t == 0)
                for (int i = b1 - 1; i >= b0; --i) {
                    const double* ti = T + (size_t)i * n;
                    x[i] = (x[i] - dot_n(ti + i + 1, x + i + 1, b1 - i - 1)) / ti[i];
                }
            bar.wait();
            for (int i = t; i < b0; i += nt) {
                x[i] -= dot_n(T + (size_t)i * n + b0, x + b0, b1 - b0);
            }
            bar.wait();
        }
    }
    void transposed(int t) {
        for (int b0 = 0; b0 < n; b0 += NB) {
            const int b1 = std::min(n, b0 + NB);
            if (t == 0)
                for (int i = b0; i < b1; ++i) {
                    const double* ti = T + (size_t)i * n;
                    const double xi = x[i] / ti[i];
                    x[i] = xi;
                    for (int k = i + 1; k < b1; ++k) x[k] -= ti[k] * xi;
                }
            bar.wait();
            const int rest = n - b1, chunk = (rest + nt - 1) / nt;
            const int c0 = std::min(n, b1 + t * chunk), c1 = std::min(n, c0 + chunk);
            if (c1 > c0)
                for (int i = b0; i < b1; ++i) {
                    const double* __restrict__ ti = T + (size_t)i * n;
                    const double xi = x[i];
                    for (int k = c0; k < c1; ++k) x[k] -= ti[k] * xi;
                }
            bar.wait();
        }
    }
    void run(int t) {
        if (kind == 0) upper(t);
        else transposed(t);
    }
    void worker(int t) {
        int seen = 0;
        for (;;) {
            for (int spins = 0; job_gen.load(std::memory_order_acquire) == seen; ++spins) {
                if (spins < 4000) __builtin_ia32_pause();
                else std::this_thread::yield();
            }
            seen = job_gen.load(std::memory_order_acquire);
            if (kind < 0) return;
            run(t);
        }
    }
    void post(int k, const double* T_, double* x_) {
        kind = k;
        T = T_;
        x = x_;
        job_gen.fetch_add(1, std::memory_order_release);
    }
    // (a solve at n = 480 is 30 us of work; beyond FSNAP_TRI_TEAM threads the two barriers per 64 unknowns cost more than the rows
    // they share out, now that the dot products run on eight partial sums)
    static int team_size(int n, int want) {
        static const int cap = [] {
            const char* e = getenv("FSNAP_TRI_TEAM");
            const int v = e && *e ? atoi(e) : 4;
            return v < 1 ? 1 : (v > 16 ? 16 : v);
        }();
        return n < 1024 ? 1 : std::max(1, std::min(want, cap));
    }

public:
    // `want` threads (the caller included); small systems keep the single-thread substitutions
    TriTeam(int n_, int want) : n(n_), nt(team_size(n_, want)), bar(team_size(n_, want)) {
        int started = 1;
        for (int t = 1; t < nt; ++t) {
            try {
                workers.emplace_back(&TriTeam::worker, this, t);
                ++started;
            } catch (...) {
                break;                                  // no more threads to be had: a smaller team (the workers are still parked)
            }
        }
        if (started != nt) {
            nt = started;
            bar.resize(started);
        }
    }
    ~TriTeam() {
        if (!workers.empty()) {
            post(-1, nullptr, nullptr);
            for (auto& w : workers) w.join();
        }
    }
    TriTeam(const TriTeam&) = delete;
    TriTeam& operator=(const TriTeam&) = delete;
    void solve_upper(const double* T_, double* x_) {
        if (n < 384) {
            fsnap_rs::solve_upper(n, T_, x_);
            return;
        }
        post(0, T_, x_);
        run(0);
    }
    void solve_upper_transposed(const double* T_, double* x_) {
        if (n < 384) {
            fsnap_rs::solve_upper_transposed(n, T_, x_);
            return;
        }
        post(1, T_, x_);
        run(0);
    }
};

// Hager's / Higham's estimate of ||B||_1 for B = T^-1 (transposed = false) or B = T^-T (true): LAPACK dlacon's iteration
double inverse_norm1_estimate(int n, const double* T, bool transposed, int threads = 1) {
    if (n == 0) return 0.0;
    TriTeam team(n, threads);
    auto apply = [&](double* v, bool tr) { (tr != transposed) ? team.solve_upper_transposed(T, v) : team.solve_upper(T, v); };
    vec x((size_t)n, 1.0 / n), y((size_t)n), zt((size_t)n);
    double est = 0.0;
    int jlast = -1;
    for (int it = 0; it < 5; ++it) {
        y = x;
        apply(y.data(), false);                       // y = B x
        double ny = 0.0;
        for (double v : y) ny += std::fabs(v);
        if (!(ny > est) && it > 0) break;
        est = std::fmax(est, ny);
        for (int i = 0; i < n; ++i) zt[i] = y[i] >= 0.0 ? 1.0 : -1.0;
        apply(zt.data(), true);                       // z = B^T sign(y)
        int j = 0;
        double zmax = 0.0, ztx = 0.0;
        for (int i = 0; i < n; ++i) {
            if (std::fabs(zt[i]) > zmax) {
                zmax = std::fabs(zt[i]);
                j = i;
            }
            ztx += zt[i] * x[i];
        }
        if ((it > 0 && zmax <= ztx) || j == jlast) break;
        jlast = j;
        std::fill(x.begin(), x.end(), 0.0);
        x[j] = 1.0;
    }
    // the alternating-sign probe that catches the cases the iteration misses
    for (int i = 0; i < n; ++i) x[i] = ((i & 1) ? -1.0 : 1.0) * (1.0 + (n > 1 ? (double)i / (n - 1) : 0.0));
    apply(x.data(), false);
    double alt = 0.0;
    for (double v : x) alt += std::fabs(v);
    return std::fmax(est, 2.0 * alt / (3.0 * n));
}

// ||T^-1||_2 = 1 / sigma_min(T) from the Lanczos process on (T^T T)^-1 (two triangular solves per step, fsnap_condest.h: fixed
// start vector, full re-orthogonalisation): the largest Ritz value increases towards 1 / sigma_min^2, so the result is a
// LOWER estimate -- after j steps at least what j steps of inverse iteration from the same start return, and within a few
// per cent once a step adds less than 10 %.  4 ... 10 steps; until round 6 this was 14 steps of plain inverse iteration,
// 28 sequential solves of K^2 / 2 that were the long pole of the chain's estimators (16 of 32 ms at K = 1595).
// `snap`: what the first `snap_steps` steps alone return (the quick look of certified()).
double inverse_norm2_estimate(int n, const double* T, int snap_steps = 0, double* snap = nullptr, int threads = 1) {
    if (snap) *snap = 0.0;
    if (n == 0) return 0.0;
    TriTeam team(n, threads);
    constexpr int MAX_STEPS = 10;
    double trace[MAX_STEPS];
    const fsnap::CondEstimate ce = fsnap::lanczos_lambda_min(
        n,
        [&](double* v) {
            team.solve_upper_transposed(T, v);     // w = T^-T v
            team.solve_upper(T, v);                // v = T^-1 w = (T^T T)^-1 v_old
            double chk = 0.0;
            for (int i = 0; i < n; ++i) chk += v[i] * 0.0;
            return chk == 0.0;
        },
        4, MAX_STEPS, 1.10, trace);
    const double inf = std::numeric_limits<double>::infinity();
    if (!(ce.lambda_min > 0.0)) {               // a singular or non-finite factor: ||T^-1|| is unbounded
        if (snap) *snap = inf;
        return inf;
    }
    if (snap && snap_steps > 0) {
        const int j = std::min(snap_steps, ce.steps) - 1;
        *snap = trace[j] > 0.0 ? std::sqrt(trace[j]) : 0.0;
    }
    return std::sqrt(1.0 / ce.lambda_min);
}

// ||T||_2 from the Lanczos process on T^T T (upper triangular T, row-major; uniform start vector, full re-orthogonalisation):
// a LOWER estimate, after j steps at least what j steps of power iteration return; 3 ... `steps` steps, stopping once a step
// adds less than 2 % (until round 6: 12 steps of power iteration, the long pole of the chain's estimators once the inverse
// iteration had gone).  One pass over T per step (w_i = t_i . v and v' += w_i t_i use the same row); the rows are cut into a
// FIXED number of chunks with a partial v' each, summed in chunk order -- the value does not depend on how many threads ran
// the chunks.
double norm2_estimate(int n, const double* T, int steps = 8, int threads = 0) {
    if (n == 0) return 0.0;
    if (steps > 16) steps = 16;
    // (one set of threads for all steps; `threads` > 0: the caller's share when other estimators run beside this one)
    const int nchunk = n >= 384 ? 16 : 1;
    const int nt = std::min(std::min(threads_for(n, 0.5 * (double)n * n * steps), threads > 0 ? threads : 64), nchunk);
    vec V((size_t)(steps + 1) * n), part((size_t)nchunk * n), w((size_t)n);
    std::fill(V.begin(), V.begin() + n, 1.0 / std::sqrt((double)n));
    double al[16], be[16];
    double theta = 0.0, theta_prev = 0.0, bad = 0.0;
    bool stop = false, broken = false;
    // ONE set of threads for all the steps (a barrier after the partial products, one after the serial part by thread 0)
    SpinBarrier bar(nt);
    run_team(nt, bar, [&](int t, int team) {
        for (int it = 0; it < steps; ++it) {
            const double* __restrict__ v = V.data() + (size_t)it * n;
            for (int ch = t; ch < nchunk; ch += team) {
                double* __restrict__ pv = part.data() + (size_t)ch * n;
                std::fill(pv, pv + n, 0.0);
                for (int i = ch; i < n; i += nchunk) {            // rows dealt round-robin: row i costs n - i
                    const double* __restrict__ ti = T + (size_t)i * n;
                    const double acc = dot_n(ti + i, v + i, n - i);
                    for (int k = i; k < n; ++k) pv[k] += ti[k] * acc;
                }
            }
            bar.wait();
            if (t == 0) {
                double a = 0.0;
                for (int k = 0; k < n; ++k) {
                    double t2 = 0.0;
                    for (int ch = 0; ch < nchunk; ++ch) t2 += part[(size_t)ch * n + k];
                    w[k] = t2;
                    a += t2 * v[k];
                }
                if (!(a > 0.0) || !std::isfinite(a)) {          // T^T T is positive semi-definite: zero = a zero factor
                    bad = a;
                    broken = stop = true;
                } else {
                    al[it] = a;
                    theta = fsnap::tridiag_lambda_max(al, be, it + 1);
                    if ((it + 1 >= 3 && theta <= 1.02 * theta_prev) || it + 1 == steps) stop = true;
                    theta_prev = theta;
                    if (!stop) {
                        for (int pass = 0; pass < 2; ++pass)
                            for (int p = 0; p <= it; ++p) {
                                const double* vp = V.data() + (size_t)p * n;
                                const double d = dot_n(vp, w.data(), n);
                                for (int k = 0; k < n; ++k) w[k] -= d * vp[k];
                            }
                        double b2 = 0.0;
                        for (int k = 0; k < n; ++k) b2 += w[k] * w[k];
                        const double bn = std::sqrt(b2);
                        if (!(bn > 1.0e-14 * theta)) {
                            stop = true;                          // invariant subspace: theta is an eigenvalue
                        } else {
                            be[it] = bn;
                            double* vn = V.data() + (size_t)(it + 1) * n;
                            for (int k = 0; k < n; ++k) vn[k] = w[k] / bn;
                        }
                    }
                }
            }
            bar.wait();
            if (stop) return;
        }
    });
    return broken ? bad : std::sqrt(theta);
}


// ---- 64 x 64 tiles of the O(n^3) phases (product of the factors, explicit inverse) ---------------------------------------
// Round 5 formed both by rows with axpys of whole rows: every multiply-add streamed 8 bytes of the other matrix through L2
// (20 / 16 ms at K = 1595 on twelve threads, ~7 GF/s a thread).  By tiles the operands of 2 x 64^3 flops are 3 x 32 KB, and a
// 4 x 8 register block does 8 multiply-adds per 6 loads from L1.  Every output element is accumulated in ONE fixed order
// (tiles of the inner dimension ascending, inside a tile k ascending): the bits do not depend on the thread count.
constexpr int TB = 64;
typedef double v4 __attribute__((vector_size(32)));
typedef double v4u __attribute__((vector_size(32), aligned(8)));

// C[mb x nb] += s * A[mb x kb] B[kb x nb]  (row-major pieces of larger arrays; s = +1 / -1)
template <bool SUB>
static void tile_gemm(double* __restrict__ C, size_t ldc, const double* __restrict__ A, size_t lda, const double* __restrict__ B, size_t ldb,
                      int mb, int nb, int kb) {
    int i = 0;
    for (; i + 4 <= mb; i += 4) {
        const double *a0 = A + (size_t)i * lda, *a1 = a0 + lda, *a2 = a1 + lda, *a3 = a2 + lda;
        int j = 0;
        for (; j + 8 <= nb; j += 8) {
            v4 c00 = {0, 0, 0, 0}, c01 = c00, c10 = c00, c11 = c00, c20 = c00, c21 = c00, c30 = c00, c31 = c00;
            const double* b = B + j;
            for (int k = 0; k < kb; ++k, b += ldb) {
                const v4 b0 = *(const v4u*)b, b1 = *(const v4u*)(b + 4);
                const v4 x0 = {a0[k], a0[k], a0[k], a0[k]}, x1 = {a1[k], a1[k], a1[k], a1[k]};
                const v4 x2 = {a2[k], a2[k], a2[k], a2[k]}, x3 = {a3[k], a3[k], a3[k], a3[k]};
                c00 += x0 * b0; c01 += x0 * b1;
                c10 += x1 * b0; c11 += x1 * b1;
                c20 += x2 * b0; c21 += x2 * b1;
                c30 += x3 * b0; c31 += x3 * b1;
            }
            double* c = C + (size_t)i * ldc + j;
            if (SUB) {
                *(v4u*)c -= c00; *(v4u*)(c + 4) -= c01; c += ldc;
                *(v4u*)c -= c10; *(v4u*)(c + 4) -= c11; c += ldc;
                *(v4u*)c -= c20; *(v4u*)(c + 4) -= c21; c += ldc;
                *(v4u*)c -= c30; *(v4u*)(c + 4) -= c31;
            } else {
                *(v4u*)c += c00; *(v4u*)(c + 4) += c01; c += ldc;
                *(v4u*)c += c10; *(v4u*)(c + 4) += c11; c += ldc;
                *(v4u*)c += c20; *(v4u*)(c + 4) += c21; c += ldc;
                *(v4u*)c += c30; *(v4u*)(c + 4) += c31;
            }
        }
        for (; j < nb; ++j) {                            // ragged columns
            double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
            for (int k = 0; k < kb; ++k) {
                const double bv = B[(size_t)k * ldb + j];
                t0 += a0[k] * bv; t1 += a1[k] * bv; t2 += a2[k] * bv; t3 += a3[k] * bv;
            }
            double* c = C + (size_t)i * ldc + j;
            if (SUB) { c[0] -= t0; c[ldc] -= t1; c[2 * ldc] -= t2; c[3 * ldc] -= t3; }
            else { c[0] += t0; c[ldc] += t1; c[2 * ldc] += t2; c[3 * ldc] += t3; }
        }
    }
    for (; i < mb; ++i) {                                // ragged rows
        const double* a = A + (size_t)i * lda;
        double* c = C + (size_t)i * ldc;
        for (int j = 0; j < nb; ++j) {
            double t = 0.0;
            for (int k = 0; k < kb; ++k) t += a[k] * B[(size_t)k * ldb + j];
            if (SUB) c[j] -= t; else c[j] += t;
        }
    }
}

// tasks of unequal cost, heaviest first, dealt to the threads in a snake (0 .. nt-1, nt-1 .. 0, ...): a static split that
// stays within a task of even
template <class F>
static void run_weighted(int nt, std::vector<std::pair<double, int>>& cost_and_id, F&& f) {
    std::stable_sort(cost_and_id.begin(), cost_and_id.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first > b.first; });
    const int n = (int)cost_and_id.size();
    if (nt > n) nt = n < 1 ? 1 : n;
    run_threads(nt, [&](int t) {
        for (int r = 0; r * nt < n; ++r) {
            const int i = r * nt + ((r & 1) ? nt - 1 - t : t);
            if (i < n) f(cost_and_id[i].second);
        }
    });
}

// C = A B for upper triangular n x n matrices (full row-major storage, zeros below the diagonal; C is overwritten, strictly
// lower part zero): tile (I, J) = sum_{I <= Kb <= J} A_{I Kb} B_{Kb J}
static void upper_product(int n, const double* A, const double* B, double* C) {
    const int nb = (n + TB - 1) / TB;
    const int nt = threads_for(n, (double)n * n * n / 3.0);
    std::vector<std::pair<double, int>> tiles;
    for (int I = 0; I < nb; ++I)
        for (int J = 0; J < nb; ++J) tiles.emplace_back(J >= I ? (double)(J - I + 1) : 0.01, I * nb + J);
    run_weighted(nt, tiles, [&](int id) {
        const int I = id / nb, J = id % nb;
        const int i0 = I * TB, j0 = J * TB, mi = std::min(TB, n - i0), nj = std::min(TB, n - j0);
        for (int i = 0; i < mi; ++i) std::fill(C + (size_t)(i0 + i) * n + j0, C + (size_t)(i0 + i) * n + j0 + nj, 0.0);
        if (J < I) return;
        double Bt[TB * TB];                                   // the B tile, contiguous: its 64 rows lie 8 n bytes apart (a page each)
        for (int Kb = I; Kb <= J; ++Kb) {
            const int k0 = Kb * TB, kk = std::min(TB, n - k0);
            for (int k = 0; k < kk; ++k) memcpy(Bt + (size_t)k * TB, B + (size_t)(k0 + k) * n + j0, (size_t)nj * sizeof(double));
            tile_gemm<false>(C + (size_t)i0 * n + j0, (size_t)n, A + (size_t)i0 * n + k0, (size_t)n, Bt, (size_t)TB, mi, nj, kk);
        }
    });
}

// X = T^-1 for an upper triangular n x n matrix (full row-major storage; X need not be initialised: every entry is written, the
// strictly lower part with zeros).  Diagonal tiles by back substitution (rows from the bottom, axpys of <= 64 entries); column strips of 32: for I = J-1 .. 0
// T_{II} X_{I, strip} = -(sum_{I < Kb <= J} T_{I Kb} X_{Kb, strip}), the sum by tiles, the solve by back substitution.
static void upper_inverse(int n, const double* T, double* X) {
    const int nb = (n + TB - 1) / TB;
    const int nt = threads_for(n, (double)n * n * n / 3.0);
    run_threads(std::min(nt, nb), [&](int t) {
        const int team = std::min(nt, nb);
        for (int I = t; I < nb; I += team) {
            const int c0 = I * TB, c1 = std::min(n, c0 + TB);
            for (int i = c0; i < c1; ++i) std::fill(X + (size_t)i * n, X + (size_t)i * n + c1, 0.0);      // left of and inside the diagonal tile
            for (int i = c1 - 1; i >= c0; --i) {
                double* __restrict__ xi = X + (size_t)i * n;
                const double* ti = T + (size_t)i * n;
                xi[i] = 1.0;
                for (int k = i + 1; k < c1; ++k) {
                    const double f = ti[k];
                    if (f == 0.0) continue;
                    const double* __restrict__ xk = X + (size_t)k * n;
                    for (int c = k; c < c1; ++c) xi[c] -= f * xk[c];
                }
                const double inv = 1.0 / ti[i];
                for (int c = i; c < c1; ++c) xi[c] *= inv;
            }
        }
    });
    constexpr int SW = 32;                               // strip width: two strips per tile column, twice the tasks to balance
    std::vector<std::pair<double, int>> strips;
    for (int c0 = TB; c0 < n; c0 += SW) strips.emplace_back((double)(c0 / TB) * (c0 / TB), c0);
    run_weighted(nt, strips, [&](int c0) {
        const int J = c0 / TB, w = std::min(SW, n - c0);
        const int rows = std::min(n, (J + 1) * TB);
        double S[TB * SW];
        // the strip of X as a contiguous rows x 32 panel (its rows lie a page apart in X): the diagonal tile's piece first, every
        // solved tile row is added as it is written
        vec P((size_t)rows * SW, 0.0);
        for (int k = J * TB; k < rows; ++k) memcpy(P.data() + (size_t)k * SW, X + (size_t)k * n + c0, (size_t)w * sizeof(double));
        for (int I = J - 1; I >= 0; --I) {
            const int i0 = I * TB;
            std::fill(S, S + TB * SW, 0.0);
            for (int Kb = I + 1; Kb <= J; ++Kb) {
                const int k0 = Kb * TB, kk = std::min(TB, n - k0);
                tile_gemm<false>(S, SW, T + (size_t)i0 * n + k0, (size_t)n, P.data() + (size_t)k0 * SW, SW, TB, w, kk);
            }
            // X_{I, strip} = -T_II^-1 S by back substitution inside the tile (NOT -X_II S: the product with the explicit inverse of
            // a diagonal tile that holds a rounding-level pivot leaves a residual T X - I of the size of that inverse, and the
            // deflation's certificate reads the rounding-level content of X)
            for (int i = TB - 1; i >= 0; --i) {
                const double* ti = T + (size_t)(i0 + i) * n + i0;
                double* __restrict__ si = S + (size_t)i * SW;
                for (int k = i + 1; k < TB; ++k) {
                    const double f = ti[k];
                    if (f == 0.0) continue;
                    const double* __restrict__ sk = S + (size_t)k * SW;
                    for (int c = 0; c < SW; ++c) si[c] += f * sk[c];
                }
                const double inv = -1.0 / ti[i];
                double* __restrict__ xi = X + (size_t)(i0 + i) * n + c0;
                for (int c = 0; c < SW; ++c) si[c] *= inv;                 // row i of S now holds x_i: the rows above add T_ik x_k
                for (int c = 0; c < w; ++c) xi[c] = si[c];
                memcpy(P.data() + (size_t)(i0 + i) * SW, si, (size_t)SW * sizeof(double));
            }
        }
    });
}

}  // namespace

void FactorChain::start(int K_, const double* G) {
    K = K_;
    R.clear();
    own.clear();
    own.reserve(16);                                  // (push keeps pointers into it: no reallocation)
    active.assign((size_t)K, 0);
    for (int j = 0; j < K; ++j) active[j] = G[(size_t)j * K + j] > 0.0;
}

// Everything condition_bound() and certified() need to know about the factors, gathered once: per factor ONE sweep for the exact
// norm pairs of R and of E = R - I, and -- unless the factor is near the identity (the later passes: Neumann bound, no iteration) --
// Hager's estimator for R^-1 and R^-T, the Lanczos process on (R^T R)^-1 (4 ... 10 steps, with the value after 3 steps kept: the
// quick look of certified() uses that one) and on R^T R.  The sweeps and the four estimators of every factor are independent tasks
// on the host threads (round 5: 14 steps of inverse iteration + 12 of power iteration, ~50 triangular solves of 10 MB each per
// factor at K = 1595 -- 115 ms on one core, 16 ms on the team; round 6: ~25 and 9 ms).
struct ChainLook {
    double n1 = 0.0, ninf = 0.0, enorm = 0.0, rmax = 0.0, dmax = 0.0;      // ||R||_1, ||R||_inf, sqrt(||E||_1 ||E||_inf), max |r_ij|, max 1 / |r_ii|
    double h1 = 0.0, hinf = 0.0, inv2_3 = 0.0, inv2_full = 0.0, norm2 = 0.0;      // (inv2_full, inv2_3: Lanczos on (R^T R)^-1, converged / after 3 steps; norm2: Lanczos on R^T R -- lower estimates)
    bool near_identity = false;
};

static std::vector<ChainLook> chain_looks(int K, const std::vector<const double*>& R) {
    HostProf hp_("chain: sweeps + estimators");
    std::vector<ChainLook> L(R.size());
    const int nt = threads_for(K, 30.0 * (double)K * K);          // (the estimators of one factor: ~50 substitutions of K^2 / 2)
    std::vector<std::function<void()>> tasks;
    // the sweeps: rows dealt round-robin to a FIXED number of parts per factor (row i costs K - i), partial column sums added in
    // part order -- the values do not depend on the thread count.  Off the diagonal |e_ij| = |r_ij|: the sums of E = R - I follow
    // from those of R and the diagonal.
    constexpr int NPART = 8;
    struct SweepPart {
        vec colsum;
        double ninf = 0.0, einf = 0.0, rmax = 0.0, dmax = 0.0;
    };
    std::vector<SweepPart> parts(R.size() * NPART);
    for (size_t f = 0; f < R.size(); ++f)
        for (int pt = 0; pt < NPART; ++pt)
            tasks.emplace_back([&, f, pt] {
                const double* Rk = R[f];
                SweepPart& sp = parts[f * NPART + pt];
                sp.colsum.assign((size_t)K, 0.0);
                double* __restrict__ cs = sp.colsum.data();
                double ninf = 0.0, einf = 0.0, rmax = 0.0, dmax = 0.0;
                for (int i = pt; i < K; i += NPART) {
                    const double* __restrict__ ri = Rk + (size_t)i * K;
                    double rs = 0.0, rm = 0.0;
                    for (int c = i + 1; c < K; ++c) {
                        const double a = std::fabs(ri[c]);
                        rs += a;
                        cs[c] += a;
                        rm = a > rm ? a : rm;
                    }
                    const double d = std::fabs(ri[i]);
                    cs[i] += d;
                    rmax = std::fmax(rmax, std::fmax(rm, d));
                    ninf = std::fmax(ninf, rs + d);
                    einf = std::fmax(einf, rs + std::fabs(ri[i] - 1.0));
                    dmax = std::fmax(dmax, d > 0.0 ? 1.0 / d : std::numeric_limits<double>::infinity());
                }
                sp.ninf = ninf;
                sp.einf = einf;
                sp.rmax = rmax;
                sp.dmax = dmax;
            });
    run_tasks(nt, tasks);
    tasks.clear();
    for (size_t f = 0; f < R.size(); ++f) {
        ChainLook& l = L[f];
        const double* Rk = R[f];
        double n1 = 0.0, e1n = 0.0, ninf = 0.0, einf = 0.0, rmax = 0.0, dmax = 0.0;
        for (int c = 0; c < K; ++c) {
            double t = 0.0;
            for (int pt = 0; pt < NPART; ++pt) t += parts[f * NPART + pt].colsum[c];
            const double rcc = Rk[(size_t)c * K + c];
            n1 = std::fmax(n1, t);
            e1n = std::fmax(e1n, std::fmax(t - std::fabs(rcc), 0.0) + std::fabs(rcc - 1.0));
        }
        for (int pt = 0; pt < NPART; ++pt) {
            const SweepPart& sp = parts[f * NPART + pt];
            ninf = std::fmax(ninf, sp.ninf);
            einf = std::fmax(einf, sp.einf);
            rmax = std::fmax(rmax, sp.rmax);
            dmax = std::fmax(dmax, sp.dmax);
        }
        // (the entries below the diagonal of a factor are zero: the full-matrix maximum of the earlier code is this one)
        l.n1 = n1;
        l.ninf = ninf;
        l.enorm = std::sqrt(e1n * einf);           // >= ||R - I||_2
        l.rmax = rmax;
        l.dmax = dmax;
        l.near_identity = l.enorm < 0.5;
    }
    int nest = 0;
    for (size_t f = 0; f < R.size(); ++f) nest += L[f].near_identity ? 0 : 3;
    // the estimators run side by side, each with a team for its substitutions; the Lanczos process on the inverse is the long one
    // (8 ... 20 solves against <= 11): it gets the largest share.  The one for ||R||_2 (only the sharper bound behind the quick look
    // uses it) runs beside them instead of after them.
    const int nfac = nest / 3;
    // shares of a factor's threads: a third each for the two Lanczos processes (<= 20 solves / <= 8 passes of twice a solve's
    // work), a sixth for each of Hager's (<= 11 solves)
    const int long_task = nfac > 0 ? std::max(1, nt / (3 * nfac)) : 1, power_task = long_task,
              per_task = nfac > 0 ? std::max(1, nt / (6 * nfac)) : 1;
    for (size_t f = 0; f < R.size(); ++f) {
        if (L[f].near_identity) continue;
        tasks.emplace_back([&, f] {
            HostProf hp("  estimator: Lanczos on (R^T R)^-1");
            L[f].inv2_full = inverse_norm2_estimate(K, R[f], 3, &L[f].inv2_3, long_task);
        });
        tasks.emplace_back([&, f] {
            HostProf hp("  estimator: Lanczos on R^T R");
            L[f].norm2 = norm2_estimate(K, R[f], 8, power_task);
        });
        tasks.emplace_back([&, f] {
            HostProf hp("  estimator: Hager ||R^-1||_1");
            L[f].h1 = inverse_norm1_estimate(K, R[f], false, per_task);
        });
        tasks.emplace_back([&, f] {
            HostProf hp("  estimator: Hager ||R^-1||_inf");
            L[f].hinf = inverse_norm1_estimate(K, R[f], true, per_task);
        });
    }
    run_tasks(nt, tasks);
    return L;
}

// the full estimate from the looks: nrm x inv
static double chain_bound_from(int K, const std::vector<const double*>& R, const std::vector<ChainLook>& L, double* norm_out,
                               double* inv_norm_out) {
    double nrm = 1.0, inv = 1.0;
    for (size_t f = 0; f < R.size(); ++f) {
        const ChainLook& l = L[f];
        // sqrt(||R||_1 ||R||_inf) is a true bound but overshoots a graded factor by one or two orders; the Lanczos value
        // approaches ||R||_2 from below: x 1.25, and never below the largest entry
        if (l.near_identity) nrm *= std::fmin(std::sqrt(l.n1 * l.ninf), 1.0 + l.enorm);          // ||I + E||_2 <= 1 + ||E||_2
        else nrm *= std::fmin(std::sqrt(l.n1 * l.ninf), std::fmax(1.25 * l.norm2, l.rmax));
        if (l.near_identity) {
            inv *= 1.0 / (1.0 - l.enorm);                      // Neumann series: the factors of the later passes
        } else {
            // the sharper of two upper estimates of ||R^-1||_2: the 1- / inf-norm pair (Hager / Higham's estimator x 3) and
            // the Lanczos value of (R^T R)^-1 (x 2: it approaches 1 / sigma_min from below)
            const double by_norm1 = 3.0 * std::sqrt(l.h1 * l.hinf);
            const double by_iteration = 2.0 * l.inv2_full;
            // never below what either estimator has actually SEEN (each is a lower bound of its own norm):
            // ||B||_2 >= ||B||_1 / sqrt(n)
            const double floor2 = std::fmax(l.h1, l.hinf) / std::sqrt((double)K);
            inv *= std::fmax(std::fmin(by_norm1, by_iteration), floor2);
        }
    }
    if (norm_out) *norm_out = nrm;
    if (inv_norm_out) *inv_norm_out = inv;
    return nrm * inv;
}

double FactorChain::condition_bound(double* norm_out, double* inv_norm_out) const {
    return chain_bound_from(K, R, chain_looks(K, R), norm_out, inv_norm_out);
}

bool FactorChain::certified(double rcond, double* norm_out, double* inv_norm_out, double* bound_out) const {
    HostProf hp_("certified");
    const double rc = rcond > 0.0 ? rcond : 0.0;
    const std::vector<ChainLook> L = chain_looks(K, R);
    // A quick look first: ||R||_2 <= sqrt(||R||_1 ||R||_inf) (exact) and, for ||R^-1||_2, the largest of three lower
    // estimates (three Lanczos steps on the inverse, Hager's 1- / inf-norm pair, the inverse's diagonal), times 10 -- inverse
    // iteration approaches ||R^-1||_2 from BELOW, and on a steeply graded factor three steps from a flat start can sit far below
    // it; the exact lower bound max_i 1 / |r_ii| and Hager's estimates (dlacon's iteration is exact on graded triangular factors
    // in all but contrived cases) keep the quick look honest.  When even that leaves FOUR orders of margin the sharper
    // combination below has nothing to add.
    {
        double nrm = 1.0, inv = 1.0;
        for (const ChainLook& l : L) {
            nrm *= std::sqrt(l.n1 * l.ninf);
            if (l.near_identity) inv *= 1.0 / (1.0 - l.enorm);
            else inv *= 10.0 * std::fmax(std::fmax(l.inv2_3, l.dmax), std::sqrt(l.h1 * l.hinf));
        }
        if (std::isfinite(nrm * inv) && nrm * inv * rc < 1.0e-4) {
            if (norm_out) *norm_out = nrm;
            if (inv_norm_out) *inv_norm_out = inv;
            if (bound_out) *bound_out = nrm * inv;
            return true;
        }
    }
    const double est = chain_bound_from(K, R, L, norm_out, inv_norm_out);
    if (bound_out) *bound_out = est;
    // two orders of margin over estimators that sit 2-8 x above the truth in the tests but are NOT bounds (power / inverse
    // iteration and Hager's estimator approach the norms from below).  Without that margin the caller multiplies the
    // factors out and FactorSolver::prepare decides with PROVABLE bounds (||T||_F ||T^-1||_F >= cond_2, explicit inverse)
    // before it falls back on the SVD.  (A cheap provable bound through the comparison matrix, |R^-1| <= M(R)^-1, was
    // tried: it overshoots dense triangular factors by 20-80 orders of magnitude and never certifies anything.)
    return est * rc < 1.0e-2;
}

void FactorChain::solve(const double* z, double* beta) const {
    for (int j = 0; j < K; ++j) beta[j] = active[j] ? z[j] : 0.0;
    TriTeam team(K, host_threads(K));
    for (size_t k = R.size(); k-- > 0;) team.solve_upper(R[k], beta);      // latest factor first
    for (int j = 0; j < K; ++j)
        if (!active[j]) beta[j] = 0.0;
}

void FactorChain::product(double* Rhat) const {
    HostProf hp_("product");
    if (R.empty()) {
        std::fill(Rhat, Rhat + (size_t)K * K, 0.0);
        return;
    }
    if (R.size() == 1) memcpy(Rhat, R[0], (size_t)K * K * sizeof(double));
    // R[k] ... R[1] R[0], 64 x 64 tiles on the host threads; the products alternate between Rhat and a second array so that the
    // last one lands in Rhat (upper_product writes every tile of its output, the zeros below the diagonal included)
    const size_t nprod = R.size() - 1;
    Scratch other(nprod > 1 ? (size_t)K * K : 0);              // (every tile of an output is written: no zeroing)
    const double* cur = R[0];
    for (size_t k = 1; k <= nprod; ++k) {
        double* out = ((nprod - k) & 1) ? other.data() : Rhat;
        upper_product(K, R[k], cur, out);
        cur = out;
    }
    for (int j = 0; j < K; ++j)
        if (!active[j]) {
            for (int c = 0; c < K; ++c) Rhat[(size_t)j * K + c] = 0.0;       // zero row, zero diagonal: "inactive"
        }
}

namespace {

// sqrt(||B||_1 ||B||_inf) >= ||B||_2 of a full n x n matrix
double one_inf_norm(int n, const double* B) {
    vec col((size_t)n, 0.0);
    double ninf = 0.0;
    for (int i = 0; i < n; ++i) {
        double rs = 0.0;
        for (int c = 0; c < n; ++c) {
            const double a = std::fabs(B[(size_t)i * n + c]);
            rs += a;
            col[c] += a;
        }
        ninf = std::fmax(ninf, rs);
    }
    double n1 = 0.0;
    for (int c = 0; c < n; ++c) n1 = std::fmax(n1, col[c]);
    return std::sqrt(n1 * ninf);
}

// ... and ||B||_F^2 beside it, on the host threads: the rows in 8 FIXED parts (partial column sums added in part order, the squares
// row by row): the values do not depend on the thread count
void full_norms(int n, const double* B, double* fro2_out, double* one_inf_out) {
    constexpr int NPART = 8;
    vec col((size_t)NPART * n, 0.0), rowsq((size_t)n, 0.0);
    double ninf_part[NPART] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int nt = std::min(threads_for(n, 2.0 * (double)n * n), NPART);
    run_threads(nt, [&](int t) {
        for (int pt = t; pt < NPART; pt += nt) {
            double* __restrict__ cs = col.data() + (size_t)pt * n;
            double ninf = 0.0;
            for (int i = pt; i < n; i += NPART) {
                const double* __restrict__ bi = B + (size_t)i * n;
                double rs = 0.0;
                for (int c = 0; c < n; ++c) {
                    const double a = std::fabs(bi[c]);
                    rs += a;
                    cs[c] += a;
                }
                ninf = std::fmax(ninf, rs);
                rowsq[i] = sum_squares(bi, (size_t)n);
            }
            ninf_part[pt] = ninf;
        }
    });
    double n1 = 0.0, ninf = 0.0, f2 = 0.0;
    for (int c = 0; c < n; ++c) {
        double tcol = 0.0;
        for (int pt = 0; pt < NPART; ++pt) tcol += col[(size_t)pt * n + c];
        n1 = std::fmax(n1, tcol);
    }
    for (int pt = 0; pt < NPART; ++pt) ninf = std::fmax(ninf, ninf_part[pt]);
    for (int i = 0; i < n; ++i) f2 += rowsq[i];
    *fro2_out = f2;
    *one_inf_out = std::sqrt(n1 * ninf);
}


// x <- x - sum_j (q_j . x) q_j for the orthonormal rows q_j of Q (nq x n), twice ("twice is enough")
inline void project_out(int n, int nq, const double* Q, double* x) {
    for (int rep = 0; rep < 2; ++rep)
        for (int j = 0; j < nq; ++j) {
            const double* q = Q + (size_t)j * n;
            const double t = dot_n(q, x, n);
            for (int k = 0; k < n; ++k) x[k] -= t * q[k];
        }
}

}  // namespace

namespace {

// DB = block width of the subspace iteration: the dropped directions + guard vectors (8: up to 4 dropped; 32: up to 24)

// Storage of an n x DB block of vectors: GROUPS of four columns (one AVX2 vector), each group an n x 4 row-major panel of its own --
// a substitution walks one panel top to bottom or back (51 KB at n = 1595: it stays in L1 / L2), instead of touching one 32-byte
// piece of every 256-byte row of an n x 32 array (19 ms per block solve at n = 1595, DB = 32: L2-bandwidth bound).
constexpr int BCW = 4;
inline size_t bix(int n, int i, int c) { return ((size_t)(c / BCW) * n + (size_t)i) * BCW + (size_t)(c % BCW); }

// W <- T^-1 W / W <- T^-T W for an n x DB block.  The right-hand sides are independent: the column groups are dealt to the host threads.
// (Round 6 also built the cooperative form -- blocks of 64 / 128 unknowns on a team, T read once for all columns --: 1.4 ms per solve
// at n = 1595 with 2 ... 12 threads against 2.0 with one; the barriers and the row-strided pieces of T cost what the team gains.)
template <int DB>
void solve_upper_block(int n, const double* T, double* W) {
    constexpr int CW = BCW, NG = DB / CW;
    static_assert(DB % CW == 0, "block width");
    const int nt = std::min(threads_for(n, 0.5 * (double)n * n * DB), NG);
    run_threads(nt, [&](int t) {
        for (int g = t; g < NG; g += nt) {
            double* Wg = W + (size_t)g * n * CW;
            for (int i = n - 1; i >= 0; --i) {
                const double* __restrict__ ti = T + (size_t)i * n;
                // four partial sums over k (one chain of dependent multiply-adds is 4 cycles per entry of T: 1.5 ms per solve at
                // n = 1595), added in a fixed order
                v4 s0 = {0, 0, 0, 0}, s1 = s0, s2 = s0, s3 = s0;
                int k = i + 1;
                for (; k + 4 <= n; k += 4) {
                    const double* wk = Wg + (size_t)k * CW;
                    s0 += v4{ti[k], ti[k], ti[k], ti[k]} * *(const v4u*)wk;
                    s1 += v4{ti[k + 1], ti[k + 1], ti[k + 1], ti[k + 1]} * *(const v4u*)(wk + CW);
                    s2 += v4{ti[k + 2], ti[k + 2], ti[k + 2], ti[k + 2]} * *(const v4u*)(wk + 2 * CW);
                    s3 += v4{ti[k + 3], ti[k + 3], ti[k + 3], ti[k + 3]} * *(const v4u*)(wk + 3 * CW);
                }
                for (; k < n; ++k) s0 += v4{ti[k], ti[k], ti[k], ti[k]} * *(const v4u*)(Wg + (size_t)k * CW);
                const double inv = 1.0 / ti[i];
                const v4 acc = *(const v4u*)(Wg + (size_t)i * CW) - ((s0 + s1) + (s2 + s3));
                *(v4u*)(Wg + (size_t)i * CW) = acc * v4{inv, inv, inv, inv};
            }
        }
    });
}

template <int DB>
void solve_upper_transposed_block(int n, const double* T, double* W) {
    constexpr int CW = BCW, NG = DB / CW;
    const int nt = std::min(threads_for(n, 0.5 * (double)n * n * DB), NG);
    run_threads(nt, [&](int t) {
        for (int g = t; g < NG; g += nt) {
            double* Wg = W + (size_t)g * n * CW;
            for (int i = 0; i < n; ++i) {
                const double* ti = T + (size_t)i * n;
                const double inv = 1.0 / ti[i];
                double zi[CW];
                for (int c = 0; c < CW; ++c) Wg[(size_t)i * CW + c] = zi[c] = Wg[(size_t)i * CW + c] * inv;
                for (int k = i + 1; k < n; ++k) {
                    const double f = ti[k];
                    double* wk = Wg + (size_t)k * CW;
                    for (int c = 0; c < CW; ++c) wk[c] -= f * zi[c];
                }
            }
        }
    });
}

// orthonormal columns by modified Gram-Schmidt, twice; a column that vanishes is replaced by a pseudo-random one.  On a
// column-major copy (contiguous dot products and axpys; the panel layout of the block walks a column with stride 4).
template <int DB>
void orthonormalise_block(int n, double* W) {
    vec C((size_t)DB * n);
    for (int c = 0; c < DB; ++c) {
        double* col = C.data() + (size_t)c * n;
        for (int i = 0; i < n; ++i) col[i] = W[bix(n, i, c)];
    }
    unsigned long long state = 0xD1B54A32D192ED03ull;
    for (int c = 0; c < DB; ++c) {
        double* __restrict__ col = C.data() + (size_t)c * n;
        for (int attempt = 0; attempt < 3; ++attempt) {
            const double before = dot_n(col, col, n);
            for (int rep = 0; rep < 2; ++rep)
                for (int p = 0; p < c; ++p) {
                    const double* __restrict__ prev = C.data() + (size_t)p * n;
                    const double t = dot_n(prev, col, n);
                    for (int i = 0; i < n; ++i) col[i] -= t * prev[i];
                }
            const double nn = dot_n(col, col, n);
            if (nn > 1.0e-24 * before && nn > 0.0 && std::isfinite(nn)) {
                const double f = 1.0 / std::sqrt(nn);
                for (int i = 0; i < n; ++i) col[i] *= f;
                break;
            }
            for (int i = 0; i < n; ++i) {          // (numerically) inside the span of the earlier columns: start over
                state = state * 6364136223846793005ull + 1442695040888963407ull;
                col[i] = ((double)(state >> 11) / 9007199254740992.0) - 0.5;
            }
        }
    }
    for (int c = 0; c < DB; ++c) {
        const double* col = C.data() + (size_t)c * n;
        for (int i = 0; i < n; ++i) W[bix(n, i, c)] = col[i];
    }
}

// SVD of a DB x DB matrix G (row-major) by one-sided Jacobi on its columns: G Q = P diag(sig), sorted descending.
// P and Q are returned row-major (columns = singular vectors).
template <int DB>
void small_svd(const double* G, double* P, double* sig, double* Q) {
    double M[DB][DB], R[DB][DB];
    for (int i = 0; i < DB; ++i)
        for (int j = 0; j < DB; ++j) {
            M[i][j] = G[i * DB + j];
            R[i][j] = i == j ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 40; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < DB - 1; ++p)
            for (int q = p + 1; q < DB; ++q) {
                double al = 0.0, be = 0.0, ga = 0.0;
                for (int i = 0; i < DB; ++i) {
                    al += M[i][p] * M[i][p];
                    be += M[i][q] * M[i][q];
                    ga += M[i][p] * M[i][q];
                }
                if (ga == 0.0 || std::fabs(ga) <= 4.0 * EPS * std::sqrt(al) * std::sqrt(be)) continue;
                ++rotated;
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
                for (int i = 0; i < DB; ++i) {
                    const double a = M[i][p], b = M[i][q];
                    M[i][p] = c * a - sn * b;
                    M[i][q] = sn * a + c * b;
                    const double ra = R[i][p], rb = R[i][q];
                    R[i][p] = c * ra - sn * rb;
                    R[i][q] = sn * ra + c * rb;
                }
            }
        if (!rotated) break;
    }
    int order[DB];
    double nrm[DB];
    for (int j = 0; j < DB; ++j) {
        double t = 0.0;
        for (int i = 0; i < DB; ++i) t += M[i][j] * M[i][j];
        nrm[j] = std::sqrt(t);
        order[j] = j;
    }
    std::sort(order, order + DB, [&](int a, int b) { return nrm[a] > nrm[b]; });
    for (int jj = 0; jj < DB; ++jj) {
        const int j = order[jj];
        sig[jj] = nrm[j];
        for (int i = 0; i < DB; ++i) {
            P[i * DB + jj] = nrm[j] > 0.0 ? M[i][j] / nrm[j] : 0.0;
            Q[i * DB + jj] = R[i][j];
        }
    }
}

}  // namespace

// The common ill-conditioned case: a few columns nearly dependent on the others -- a few singular values below rcond sigma_max
// and a comfortable gap above them.  dgelsd's answer is then x = sum_{kept} v_i (u_i . y) / sigma_i, and the kept part of T^-1
// is what is left of it after the dropped triplets are projected away:
//     x = (I - Vc Vc^T) T^-1 (I - Uc Uc^T) y.
// The dropped triplets come from subspace (block inverse) iteration on T with DB = 8 vectors (up to 4 dropped directions; when
// there are more, once again with 32 vectors: up to 24) -- two block back substitutions per step, the dropped directions
// converge at (sigma_dropped / sigma_(DB+1)th-smallest)^2 per step however they cluster among themselves -- started from the
// heaviest rows of T^-1, with a Rayleigh-Ritz step (SVD of the DB x DB matrix V^T T^-1 U) in every iteration.
// Ritz values of T^-1 never exceed its singular values, so a Ritz sigma below rcond x a LOWER bound of sigma_max (power
// iteration) is a dropped direction for certain; and the method is only used when the deflated inverse
//     X - sum_c v_c u_c^T / sigma_c
// passes the same norm certificate the triangular case uses (with margin 0.1): every remaining singular value is then above the
// cut.  Anything else -- more than 24 dropped values, a system narrower than 4 DB, a sigma too close to the cut to call, slow
// convergence, a certificate that does not close -- returns 0 and the Jacobi SVD decides.  n = 128, one dropped direction: ~0.25 ms
// against 1.8 ms of Jacobi sweeps.
template <int DB, int MAXCUT>
int FactorSolver::deflate_width(double rc, Scratch& X, double norm_bound, double lower) {
    constexpr int MAXIT = 10;
    if (n < 4 * DB) return 0;                                       // small systems: the Jacobi SVD costs little
    const double cut = rc * lower;
    // start: the DB heaviest rows of the inverse (row i of T^-1 is sum_k v_k[i] / sigma_k u_k^T)
    vec U((size_t)n * DB), W((size_t)n * DB), V((size_t)n * DB), Z((size_t)n * DB);
    {
        std::vector<std::pair<double, int>> heavy((size_t)n);
        const int nth = threads_for(n, (double)n * n);
        run_threads(nth, [&](int t) {
            const int nt = nth;
            for (int i = t; i < n; i += nt) heavy[i] = {dot_n(X.data() + (size_t)i * n, X.data() + (size_t)i * n, n), i};
        });
        std::partial_sort(heavy.begin(), heavy.begin() + DB, heavy.end(),
                          [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first > b.first; });
        for (int c = 0; c < DB; ++c) {
            if (!std::isfinite(heavy[c].first)) return 0;
            const double* row = X.data() + (size_t)heavy[c].second * n;
            for (int k = 0; k < n; ++k) U[bix(n, k, c)] = row[k];
        }
        orthonormalise_block<DB>(n, U.data());
    }
    HostProf hp5_("deflate: iteration + rest");
    HostProfSum ps_solve("deflate: block solves"), ps_ortho("deflate: orthonormalisations"), ps_ritz("deflate: Rayleigh-Ritz"),
        ps_check("deflate: subspace check");
    double G[DB * DB], P[DB * DB], Q[DB * DB], sig[DB];
    vec uc((size_t)MAXCUT * n), vc((size_t)MAXCUT * n);
    vec un((size_t)n);
    int k = 0;
    bool conv = false;
    double prev_worst = 1.0;
    for (int it = 0; it < MAXIT && !conv; ++it) {
        W = U;
        ps_solve.time([&] { solve_upper_block<DB>(n, T.data(), W.data()); });                   // W = T^-1 U
        V = W;
        ps_ortho.time([&] { orthonormalise_block<DB>(n, V.data()); });
        ps_ritz.time([&] {
            for (int a = 0; a < DB; ++a)
                for (int b = 0; b < DB; ++b) {
                    double t = 0.0;
                    for (int i = 0; i < n; ++i) t += V[bix(n, i, a)] * W[bix(n, i, b)];
                    G[a * DB + b] = t;                                  // G = V^T T^-1 U
                }
        });
        for (double g : G)
            if (!std::isfinite(g)) return 0;
        ps_ritz.time([&] { small_svd<DB>(G, P, sig, Q); });             // T^-1 (U Q) ~ (V P) diag(sig): Ritz triplets of the inverse
        k = 0;
        while (k < DB && sig[k] > 0.0 && 1.0 / sig[k] <= cut) ++k;
        if (k > MAXCUT) return -1;                                  // more dropped directions than this width takes on
        if (k == 0 && it >= 2) return 0;                        // nothing certainly below the cut: the SVD decides
        Z = V;
        ps_solve.time([&] { solve_upper_transposed_block<DB>(n, T.data(), Z.data()); });        // Z = T^-T V
        const auto t_check = std::chrono::steady_clock::now();
        // converged when the dropped Ritz vectors span a singular subspace: T^-T (their v's) lies inside the span of their u's.
        // (Not triplet by triplet: values at the rounding level of T mix freely among themselves from one solve to the next,
        // the SUBSPACE is what the projections of apply() need and what is well determined.)
        for (int c = 0; c < k; ++c) {
            double* u = uc.data() + (size_t)c * n;
            double* v = vc.data() + (size_t)c * n;
            for (int i = 0; i < n; ++i) {
                double tu = 0.0, tv = 0.0;
                for (int b = 0; b < DB; ++b) {
                    tu += U[bix(n, i, b)] * Q[b * DB + c];
                    tv += V[bix(n, i, b)] * P[b * DB + c];
                }
                u[i] = tu;
                v[i] = tv;
            }
        }
        double worst = 0.0;
        for (int c = 0; c < k; ++c) {
            for (int i = 0; i < n; ++i) {
                double tz = 0.0;
                for (int b = 0; b < DB; ++b) tz += Z[bix(n, i, b)] * P[b * DB + c];
                un[i] = tz;
            }
            const double before = std::sqrt(dot_n(un.data(), un.data(), n));
            project_out(n, k, uc.data(), un.data());
            const double after = std::sqrt(dot_n(un.data(), un.data(), n));
            if (!(before > 0.0) || !std::isfinite(before)) return 0;
            worst = std::fmax(worst, after / before);
        }
        // done at the rounding floor -- or where the iteration stops improving: the floor sits at ~eps x the condition of the KEPT part
        conv = k > 0 && (worst <= 64.0 * EPS || (it >= 1 && worst <= 1.0e-11 && worst > 0.5 * prev_worst));
        prev_worst = worst;
        ps_check.ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_check).count();
        ++ps_check.calls;
        if (!conv) {
            U = Z;
            ps_ortho.time([&] { orthonormalise_block<DB>(n, U.data()); });
        }
    }
    if (!conv) return 0;
    Uc.assign(uc.begin(), uc.begin() + (size_t)k * n);
    Vc.assign(vc.begin(), vc.begin() + (size_t)k * n);
    // what is left of the inverse: (I - Vc Vc^T) X (I - Uc Uc^T), the operator apply() uses.  (Subtracting v u^T / sigma instead
    // fails for values at the rounding level of T: X holds ITS OWN rounding-level values for those directions.)
    {
        HostProf hp4_("deflate: projections of X");
        const int nt = threads_for(n, 2.0 * (double)n * n * k);
        // X <- X (I - Uc^T Uc): row by row, the dropped u's one after the other on the row while it is in L1 (one pass over X)
        run_threads(nt, [&](int t) {
            for (int i = t; i < n; i += nt) {
                double* xi = X.data() + (size_t)i * n;
                for (int c = 0; c < k; ++c) {
                    const double* u = Uc.data() + (size_t)c * n;
                    const double f = dot_n(xi, u, n);
                    for (int j = 0; j < n; ++j) xi[j] -= f * u[j];
                }
            }
        });
        // X <- (I - Vc^T Vc) X: columns are independent -- blocks of 64 columns, COPIED into a contiguous n x 64 panel for the 2 k
        // passes (the rows of a block lie a page apart in X: twelve strided passes per block were latency-bound, 3.5 of the 4.4 ms
        // of these projections at n = 1595) and written back once; same arithmetic, element by element
        const int nblk = (n + 63) / 64;
        run_threads(nt, [&](int t) {
            double tb[64];
            vec panel((size_t)n * 64);
            for (int blk = t; blk < nblk; blk += nt) {
                const int c0 = blk * 64, cw = std::min(64, n - c0);
                for (int i = 0; i < n; ++i) memcpy(panel.data() + (size_t)i * 64, X.data() + (size_t)i * n + c0, (size_t)cw * sizeof(double));
                for (int c = 0; c < k; ++c) {
                    const double* v = Vc.data() + (size_t)c * n;
                    for (int j = 0; j < cw; ++j) tb[j] = 0.0;
                    for (int i = 0; i < n; ++i) {
                        const double* __restrict__ xi = panel.data() + (size_t)i * 64;
                        const double vi = v[i];
                        for (int j = 0; j < cw; ++j) tb[j] += vi * xi[j];
                    }
                    for (int i = 0; i < n; ++i) {
                        double* __restrict__ xi = panel.data() + (size_t)i * 64;
                        const double vi = v[i];
                        for (int j = 0; j < cw; ++j) xi[j] -= vi * tb[j];
                    }
                }
                for (int i = 0; i < n; ++i) memcpy(X.data() + (size_t)i * n + c0, panel.data() + (size_t)i * 64, (size_t)cw * sizeof(double));
            }
        });
    }
    double fr = 0.0, oi = 0.0;
    {
        HostProf hp6_("deflate: norms of the rest");
        full_norms(n, X.data(), &fr, &oi);
    }
    if (!std::isfinite(fr)) return 0;
    const double inv_norm = std::fmin(std::sqrt(fr), oi);
    if (!(norm_bound * inv_norm * rc < 0.1)) return 0;
    deflated = true;
    ncut = k;
    rank = n - k;
    smax = lower;                // a lower estimate of sigma_max (Lanczos) ...
    smin = 1.0 / inv_norm;       // ... and a lower bound of the smallest kept singular value
    return 1;
}


// 1 ... 4 dropped directions with 8 vectors; when there are more, once again with 16 (up to 10) and with 32 vectors (up to 24; X is
// only touched after the iteration has converged, and the first Rayleigh-Ritz step already tells that a width is too narrow).
// false = the SVD decides.
bool FactorSolver::deflate(double rc, Scratch& X, double norm_bound) {
    HostProf hp_("deflate");
    if (const char* e = getenv("FSNAP_ROWSPACE_DEFLATE"))           // A/B switch: 0 = always the Jacobi SVD
        if (e[0] == '0') return false;
    double lower;
    {
        HostProf hp3_("deflate: sigma_max estimate");
        lower = norm2_estimate(n, T.data(), 12);          // <= sigma_max
    }
    if (!(lower > 0.0) || !std::isfinite(lower)) return false;
    int r = deflate_width<8, 4>(rc, X, norm_bound, lower);
    if (r < 0) r = deflate_width<16, 10>(rc, X, norm_bound, lower);
    if (r < 0) r = deflate_width<32, 24>(rc, X, norm_bound, lower);
    return r == 1;
}

void FactorSolver::prepare(int K_, const double* Rhat, double rcond) {
    HostProf hp_("prepare (all)");
        K = K_;
        act.clear();
        for (int j = 0; j < K; ++j)
            if (Rhat[(size_t)j * K + j] != 0.0) act.push_back(j);
        n = (int)act.size();
        rank = 0;
        T.reset((size_t)n * n);
        if (n == 0) return;
        // Frobenius bounds: sigma_max <= ||T||_F, sigma_min >= 1 / ||T^-1||_F.  If even these cannot put a singular
        // value below rcond * sigma_max, dgelsd would not truncate either and its solution is T^-1 z.
        // (rows gathered and squared by the host threads; the sum of squares of row a is one number whatever the split, the
        // rows are added in order)
        vec rowsq((size_t)n);
        {
            const int ntg = threads_for(n, 2.0 * (double)n * n);
            run_threads(ntg, [&](int t) {
                for (int a = t; a < n; a += ntg) {
                    double* ta = T.data() + (size_t)a * n;
                    const double* ra = Rhat + (size_t)act[a] * K;
                    std::fill(ta, ta + a, 0.0);                          // (a Scratch block: nothing is zero by itself)
                    if (n == K) memcpy(ta + a, ra + a, (size_t)(n - a) * sizeof(double));
                    else
                        for (int b = a; b < n; ++b) ta[b] = ra[act[b]];
                    rowsq[a] = sum_squares(ta + a, (size_t)(n - a));
                }
            });
        }
        double fro = 0.0;
        for (double v : rowsq) fro += v;
        fro = std::sqrt(fro);
        double inv2 = 0.0, fro2 = fro, inv_norm = 0.0;
        bool ok = true;
        deflated = false;
        ncut = 0;
        // X = T^-1 by back substitution, row by row from the bottom: X_i = (e_i - sum_{k>i} T_ik X_k) / T_ii -- every
        // update is an axpy of contiguous rows (the column-by-column form walked X with stride n: 0.4 ms at n = 128)
        Scratch X((size_t)n * n);                                  // (upper_inverse writes all of it)
        {
            HostProf hp2_("prepare: inverse + norms");
            const int nt = threads_for(n, (double)n * n * n / 3.0);
            {
                HostProf hpi_("  prepare: inverse alone");
                upper_inverse(n, T.data(), X.data());                   // 64 x 64 tiles, column strips on the host threads
            }
            double bt = fro, bx = 0.0;
            std::vector<std::function<void()>> tasks;
            tasks.emplace_back([&] {
                double t2 = 0.0;
                for (int i = 0; i < n; ++i) t2 += sum_squares(X.data() + (size_t)i * n + i, (size_t)(n - i));      // (X is upper triangular)
                inv2 = t2;
            });
            tasks.emplace_back([&] { bt = one_inf_norm(n, T.data()); });
            tasks.emplace_back([&] { bx = one_inf_norm(n, X.data()); });
            run_tasks(nt, tasks);
            ok = std::isfinite(inv2);
            // a second provable pair, usually sharper on graded factors: ||B||_2 <= sqrt(||B||_1 ||B||_inf) for B = T and for
            // B = T^-1 (both matrices are at hand).  The Frobenius norm charges up to sqrt(n) per factor -- at n = 128 a system
            // with cond ~ 1e9 and rcond = 1e-13 missed the certificate by that margin and paid 1.8 ms of Jacobi sweeps for a
            // solution that back substitution gives in 10 us (profiles/r05_lstsq_rows_phases.txt)
            if (ok) {
                fro2 = std::fmin(fro, bt);
                inv_norm = std::fmin(std::sqrt(inv2), bx);
            }
        }
        const double rc = rcond > 0.0 ? rcond : 0.0;
        triangular = ok && (fro2 * inv_norm * rc < 0.5);
        if (triangular) {
            rank = n;
            smax = fro2;
            smin = 1.0 / inv_norm;
            return;
        }
        rcond_used = rc;
        use_external = false;
        if (ok && rc > 0.0 && deflate(rc, X, fro2)) return;
        if (external && n > 256) {
            // try the host language's dense kernel on a probe right-hand side; it also tells the rank
            vec y((size_t)n, 0.0), x((size_t)n, 0.0);
            y[0] = 1.0;
            int rk = 0;
            if (external(external_user, token, n, T.data(), rc, y.data(), x.data(), &rk) == 0 && rk >= 0 && rk <= n &&
                finite_all(x.data(), x.size())) {
                use_external = true;
                rank = rk;
                smax = fro;
                smin = 0.0;
                return;
            }
        }
        jacobi_svd(rc);
    }

    // one-sided Jacobi on the ROWS of W = T (left rotations): J T = diag(sigma) V^T with J orthogonal.  Rows instead of
    // columns because T is upper triangular (the preconditioned orientation of Drmac & Veselic) and rows are contiguous.
namespace {

// 8-wide fp64 vectors (two ymm operations under -mavx2): the Jacobi sweeps are dot products and plane rotations of
// contiguous rows
typedef double v8 __attribute__((vector_size(64)));
typedef double v8u __attribute__((vector_size(64), aligned(8)));

inline double dot_rows(const double* __restrict__ x, const double* __restrict__ y, int n) {
    v8 a0 = {0, 0, 0, 0, 0, 0, 0, 0}, a1 = a0;
    int k = 0;
    for (; k + 16 <= n; k += 16) {
        a0 += *(const v8u*)(x + k) * *(const v8u*)(y + k);
        a1 += *(const v8u*)(x + k + 8) * *(const v8u*)(y + k + 8);
    }
    const v8 a = a0 + a1;
    double t = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    for (; k < n; ++k) t += x[k] * y[k];
    return t;
}

// (p, q) <- (c p - s q, s p + c q) on two rows
inline void rotate_rows(double* __restrict__ p, double* __restrict__ q, double c, double s, int n) {
    const v8 cv = {c, c, c, c, c, c, c, c}, sv = {s, s, s, s, s, s, s, s};
    int k = 0;
    for (; k + 8 <= n; k += 8) {
        const v8 a = *(const v8u*)(p + k), b = *(const v8u*)(q + k);
        *(v8u*)(p + k) = cv * a - sv * b;
        *(v8u*)(q + k) = sv * a + cv * b;
    }
    for (; k < n; ++k) {
        const double a = p[k], b = q[k];
        p[k] = c * a - s * b;
        q[k] = s * a + c * b;
    }
}

}  // namespace

// one-sided Jacobi on the ROWS of W = T (left rotations): J T = diag(sigma) V^T with J orthogonal.  Rows instead of
// columns because T is upper triangular (the preconditioned orientation of Drmac & Veselic) and rows are contiguous.
// J is carried along in the same rows (W | J side by side: one rotation call covers both).
void FactorSolver::jacobi_svd(double rcond) {
    const int n2 = 2 * n;
    vec WJ((size_t)n * n2, 0.0);
    for (int i = 0; i < n; ++i) {
        memcpy(WJ.data() + (size_t)i * n2, T.data() + (size_t)i * n, (size_t)n * sizeof(double));
        WJ[(size_t)i * n2 + n + i] = 1.0;
    }
    const double tol = std::sqrt((double)n) * EPS;
    vec nrm(n);
    for (sweeps = 0; sweeps < 60; ++sweeps) {
        int rotated = 0;
        for (int i = 0; i < n; ++i) nrm[i] = dot_rows(WJ.data() + (size_t)i * n2, WJ.data() + (size_t)i * n2, n);
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                double* wp = WJ.data() + (size_t)p * n2;
                double* wq = WJ.data() + (size_t)q * n2;
                const double al = nrm[p], be = nrm[q];
                if (al == 0.0 || be == 0.0) continue;
                const double ga = dot_rows(wp, wq, n);
                if (std::fabs(ga) <= tol * std::sqrt(al) * std::sqrt(be)) continue;
                ++rotated;
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
                rotate_rows(wp, wq, c, s, n2);
                nrm[p] = dot_rows(wp, wp, n);       // recomputed, not updated: graded rows lose digits in the update formula
                nrm[q] = dot_rows(wq, wq, n);
            }
        if (!rotated) break;
    }
    W.assign((size_t)n * n, 0.0);
    J.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) {
        memcpy(W.data() + (size_t)i * n, WJ.data() + (size_t)i * n2, (size_t)n * sizeof(double));
        memcpy(J.data() + (size_t)i * n, WJ.data() + (size_t)i * n2 + n, (size_t)n * sizeof(double));
    }
    auto dot = [&](const double* x, const double* y) { return dot_rows(x, y, n); };
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
void FactorSolver::apply(const double* z, double* beta) const {
        for (int j = 0; j < K; ++j) beta[j] = 0.0;
        if (n == 0) return;
        vec y(n);
        for (int a = 0; a < n; ++a) y[a] = z[act[a]];
        if (triangular || deflated) {
            if (deflated) project_out(n, ncut, Uc.data(), y.data());
            solve_upper(n, T.data(), y.data());
            if (deflated) project_out(n, ncut, Vc.data(), y.data());
            for (int a = 0; a < n; ++a) beta[act[a]] = y[a];
            return;
        }
        vec x(n, 0.0);
        if (use_external) {
            int rk = 0;
            if (external(external_user, token, n, T.data(), rcond_used, y.data(), x.data(), &rk) == 0) {
                for (int a = 0; a < n; ++a) beta[act[a]] = x[a];
                return;
            }
            // the hook failed after having worked in prepare(): leave zeros (the caller's finiteness / rank checks see it)
            return;
        }
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

}  // namespace fsnap_rs

