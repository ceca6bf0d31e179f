// fsnap_solve.cpp — host-side K x K back-solve for the normal-equation statistics
// produced on the GPU.  Self-contained C++ (no LAPACK dependency).
//
// Replaces, on the K x K statistics instead of the m x K matrix:
//   * scipy.linalg.lstsq(aw, bw, 1.0e-13)              fitsnap3lib/solvers/svd.py:54
//   * sklearn Ridge(alpha, fit_intercept=False).fit     fitsnap3lib/solvers/ridge.py:47-57
//       (dense path = (X^T X + alpha I) coef = X^T y, Cholesky, SVD fallback)
//   * inv(xtx + alpha I) @ xty                          fitsnap3lib/lib/ridge_solver/regressor.py:10-16
//
// Numerics: entries of G span > 30 decades on the golden Ta
// matrices, so every factorisation works on the Jacobi-scaled matrix D^-1 G D^-1
// (unit diagonal), which lowers the condition number from kappa(A_w)^2 to
// kappa_equilibrated^2, followed by one step of iterative refinement with the
// residual accumulated in long double.  Exactly-zero columns (e.g. SNAP columns
// multiplied by blank2J = 0, lammps_snap.py:467-468) get beta = 0, which is what the
// minimum-norm solution of lstsq gives them.  A numerically rank-deficient system
// falls back to a cyclic-Jacobi eigendecomposition of G and a truncated pseudo-inverse
// (minimum-norm solution, the gelsd semantics).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "../../include/fsnap_hip.h"
#include "fsnap_condest.h"

// What the last K x K solve of this thread learned about the conditioning (fsnap_cond_info): smallest scaled pivot,
// lambda_min estimate from the factor (0 when none was taken), applications of S^-1, 0 = host factor / 1 = device factor.
static thread_local double g_cond_info[4] = {0.0, 0.0, 0.0, 0.0};

extern "C" __attribute__((visibility("hidden"))) void fsnap_cond_note(double min_pivot, double lambda_min, int steps, int where) {
    g_cond_info[0] = min_pivot;
    g_cond_info[1] = lambda_min;
    g_cond_info[2] = (double)steps;
    g_cond_info[3] = (double)where;
}

extern "C" int fsnap_cond_info(double* info) {
    if (!info) return FSNAP_E_ARG;
    for (int i = 0; i < 4; ++i) info[i] = g_cond_info[i];
    return FSNAP_OK;
}

namespace {

typedef std::vector<double> vec;

// the hot K x K loops are compiled twice (AVX-512 / AVX2+FMA) and resolved at load time
#define FSNAP_CLONES __attribute__((target_clones("avx512f", "default")))
#define FSNAP_INLINE static inline __attribute__((always_inline))

// 4-wide fp64 vectors (AVX2 + FMA on the host; this file is compiled with -mavx2 -mfma)
typedef double v4d __attribute__((vector_size(32)));
typedef double v4du __attribute__((vector_size(32), aligned(8)));

FSNAP_INLINE v4d ld4(const double* p) { return *reinterpret_cast<const v4du*>(p); }
FSNAP_INLINE double hsum(v4d v) { return (v[0] + v[1]) + (v[2] + v[3]); }

// dot product with 4 independent vector accumulators (fixed association order)
FSNAP_INLINE double dotv(const double* x, const double* y, int n) {
    v4d s0 = {0, 0, 0, 0}, s1 = s0, s2 = s0, s3 = s0;
    int k = 0;
    for (; k + 16 <= n; k += 16) {
        s0 += ld4(x + k) * ld4(y + k);
        s1 += ld4(x + k + 4) * ld4(y + k + 4);
        s2 += ld4(x + k + 8) * ld4(y + k + 8);
        s3 += ld4(x + k + 12) * ld4(y + k + 12);
    }
    for (; k + 4 <= n; k += 4) s0 += ld4(x + k) * ld4(y + k);
    double s = hsum((s0 + s1) + (s2 + s3));
    for (; k < n; ++k) s += x[k] * y[k];
    return s;
}

// 8-wide vectors: one zmm op under the avx512f clone, two ymm ops under the default (AVX2) clone
typedef double v8d __attribute__((vector_size(64)));
typedef double v8du __attribute__((vector_size(64), aligned(8)));

// y[0:n] -= alpha * x[0:n]
FSNAP_INLINE void axpy_neg(double* y, const double* x, double alpha, int n) {
    const v8d av = {alpha, alpha, alpha, alpha, alpha, alpha, alpha, alpha};
    int k = 0;
    for (; k + 16 <= n; k += 16) {
        v8d y0 = *reinterpret_cast<const v8du*>(y + k), y1 = *reinterpret_cast<const v8du*>(y + k + 8);
        y0 -= av * *reinterpret_cast<const v8du*>(x + k);
        y1 -= av * *reinterpret_cast<const v8du*>(x + k + 8);
        *reinterpret_cast<v8du*>(y + k) = y0;
        *reinterpret_cast<v8du*>(y + k + 8) = y1;
    }
    for (; k + 8 <= n; k += 8) {
        v8d yv = *reinterpret_cast<const v8du*>(y + k);
        yv -= av * *reinterpret_cast<const v8du*>(x + k);
        *reinterpret_cast<v8du*>(y + k) = yv;
    }
    for (; k < n; ++k) y[k] -= alpha * x[k];
}

// In-place Cholesky A = U^T U of the n x n row-major matrix a, UPPER triangle referenced and
// overwritten with U (right-looking: every inner loop is a contiguous axpy on a row tail,
// no horizontal sums).  Returns -1 on success or the index of the failing pivot.
// *min_piv2 receives the smallest pivot (before sqrt); for the Jacobi-scaled (unit
// diagonal) matrices this file factorises that is the pivot relative to the original
// diagonal entry.
FSNAP_CLONES int chol_upper(double* a, int n, double* min_piv2) {
    double mp = std::numeric_limits<double>::infinity();
    for (int j = 0; j < n; ++j) {
        double* uj = a + (size_t)j * n;
        const double d = uj[j];
        const double rel = d;   // callers pass Jacobi-scaled matrices: original diagonal = 1
        if (rel < mp) mp = rel;
        if (!(d > 0.0) || !std::isfinite(d)) {
            if (min_piv2) *min_piv2 = mp;
            return j;
        }
        const double r = std::sqrt(d), inv = 1.0 / r;
        uj[j] = r;
        for (int k = j + 1; k < n; ++k) uj[k] *= inv;
        // trailing update: row i (i > j) tail [i, n) -= u[j][i] * u[j][i:n]
        for (int i = j + 1; i < n; ++i) {
            const double f = uj[i];
            if (f != 0.0) axpy_neg(a + (size_t)i * n + i, uj + i, f, n - i);
        }
    }
    if (min_piv2) *min_piv2 = mp;
    return -1;
}

// ---- blocked variant for large n --------------------------------------------------------
// (the unblocked sweep streams the whole trailing matrix from L2/L3 once per column: 10 GB
// at n = 1595: measured 92 ms unblocked vs 22 ms blocked + threads on the MI355X host; below
// n ~ 768 the unblocked sweep is faster, 1.8 vs 4.0 ms at n = 480).  Panels of NBK pivot rows are factorised with the unblocked recurrence
// restricted to the panel's rows; every trailing row then receives the NBK rank-1 updates in
// ONE pass, 32-column register chunks at a time; trailing rows are independent, so they are
// dealt round-robin to a few host threads when the trailing matrix is large.

// factorise pivot rows [jb, je) (unblocked, updates restricted to rows < je); -1 or failing row
FSNAP_CLONES int chol_panel(double* a, int n, int jb, int je, double* mp) {
    for (int j = jb; j < je; ++j) {
        double* uj = a + (size_t)j * n;
        const double d = uj[j];
        if (d < *mp) *mp = d;
        if (!(d > 0.0) || !std::isfinite(d)) return j;
        const double r = std::sqrt(d), inv = 1.0 / r;
        uj[j] = r;
        for (int k = j + 1; k < n; ++k) uj[k] *= inv;
        for (int i = j + 1; i < je; ++i) {
            const double f = uj[i];
            if (f != 0.0) axpy_neg(a + (size_t)i * n + i, uj + i, f, n - i);
        }
    }
    return -1;
}

// rows i = i0, i0 + step, ... < n:  row_i[i:] -= sum_{k in [jb, je)} u[k][i] * u[k][i:]
FSNAP_CLONES void chol_trailing_rows(double* a, int n, int jb, int je, int i0, int step) {
    for (int i = i0; i < n; i += step) {
        double* ri = a + (size_t)i * n;
        int c = i;
        for (; c + 32 <= n; c += 32) {
            v8d y0 = *reinterpret_cast<const v8du*>(ri + c), y1 = *reinterpret_cast<const v8du*>(ri + c + 8);
            v8d y2 = *reinterpret_cast<const v8du*>(ri + c + 16), y3 = *reinterpret_cast<const v8du*>(ri + c + 24);
            for (int k = jb; k < je; ++k) {
                const double* uk = a + (size_t)k * n;
                const double f = uk[i];
                const v8d fv = {f, f, f, f, f, f, f, f};
                y0 -= fv * *reinterpret_cast<const v8du*>(uk + c);
                y1 -= fv * *reinterpret_cast<const v8du*>(uk + c + 8);
                y2 -= fv * *reinterpret_cast<const v8du*>(uk + c + 16);
                y3 -= fv * *reinterpret_cast<const v8du*>(uk + c + 24);
            }
            *reinterpret_cast<v8du*>(ri + c) = y0;
            *reinterpret_cast<v8du*>(ri + c + 8) = y1;
            *reinterpret_cast<v8du*>(ri + c + 16) = y2;
            *reinterpret_cast<v8du*>(ri + c + 24) = y3;
        }
        if (c < n) {
            for (int k = jb; k < je; ++k) {
                const double* uk = a + (size_t)k * n;
                axpy_neg(ri + c, uk + c, uk[i], n - c);
            }
        }
    }
}

int chol_upper_blocked(double* a, int n, double* min_piv2, int NBK) {
    double mp = std::numeric_limits<double>::infinity();
    unsigned hw = std::thread::hardware_concurrency();
    const int tmax = (int)(hw > 16 ? 16 : (hw < 1 ? 1 : hw));
    for (int jb = 0; jb < n; jb += NBK) {
        const int je = (jb + NBK < n) ? jb + NBK : n;
        const int fail = chol_panel(a, n, jb, je, &mp);
        if (fail >= 0) {
            if (min_piv2) *min_piv2 = mp;
            return fail;
        }
        const int rows = n - je;
        int nt = rows / 256;                    // >= 256 trailing rows per thread (thread start-up ~50-100 us)
        if (nt > tmax) nt = tmax;
        if (nt <= 1) {
            chol_trailing_rows(a, n, jb, je, je, 1);
        } else {
            std::vector<std::thread> th;
            th.reserve(nt - 1);
            for (int t = 1; t < nt; ++t) th.emplace_back(chol_trailing_rows, a, n, jb, je, je + t, nt);
            chol_trailing_rows(a, n, jb, je, je, nt);
            for (auto& x : th) x.join();
        }
    }
    if (min_piv2) *min_piv2 = mp;
    return -1;
}

// ---- register-blocked variant (all sizes >= 32) ----------------------------------------------
// Panels of NBK pivot rows (unblocked recurrence inside the panel), then ONE pass over the
// trailing matrix per panel: 32-column chunks OUTER (the panel's chunk, NBK x 32 doubles, stays
// in L1), trailing rows INNER, four rows at a time in registers (16 accumulator vectors, 4 panel
// vectors, 4 broadcasts: 16 FMAs per 4 panel loads).  Chunks are updated in full, i.e. a few
// entries left of the diagonal (never read; the caller zeroes the lower triangle) are touched too.
// chunks first_chunk, first_chunk + chunk_step, ... (column chunk index = c0 / 32): threads take disjoint chunks
FSNAP_CLONES void chol_trailing_chunks(double* a, int n, int jb, int je, int first_chunk, int chunk_step) {
    const int nk = je - jb;
    for (int c0 = ((je >> 5) + first_chunk) << 5; c0 < n; c0 += 32 * chunk_step) {
        const int cw = (n - c0 < 32) ? n - c0 : 32;
        const int i_end = (c0 + 32 < n) ? c0 + 32 : n;    // rows with entries in this chunk: i < c0 + 32
        int i = je;
        if (cw == 32) {
            for (; i + 4 <= i_end; i += 4) {
                double* r0 = a + (size_t)i * n + c0;
                double* r1 = r0 + n;
                double* r2 = r1 + n;
                double* r3 = r2 + n;
                v8d y00 = *(const v8du*)(r0), y01 = *(const v8du*)(r0 + 8), y02 = *(const v8du*)(r0 + 16), y03 = *(const v8du*)(r0 + 24);
                v8d y10 = *(const v8du*)(r1), y11 = *(const v8du*)(r1 + 8), y12 = *(const v8du*)(r1 + 16), y13 = *(const v8du*)(r1 + 24);
                v8d y20 = *(const v8du*)(r2), y21 = *(const v8du*)(r2 + 8), y22 = *(const v8du*)(r2 + 16), y23 = *(const v8du*)(r2 + 24);
                v8d y30 = *(const v8du*)(r3), y31 = *(const v8du*)(r3 + 8), y32 = *(const v8du*)(r3 + 16), y33 = *(const v8du*)(r3 + 24);
                const double* uk = a + (size_t)jb * n;
                for (int k = 0; k < nk; ++k, uk += n) {
                    const v8d u0 = *(const v8du*)(uk + c0), u1 = *(const v8du*)(uk + c0 + 8);
                    const v8d u2 = *(const v8du*)(uk + c0 + 16), u3 = *(const v8du*)(uk + c0 + 24);
                    const double f0 = uk[i], f1 = uk[i + 1], f2 = uk[i + 2], f3 = uk[i + 3];
                    const v8d b0 = {f0, f0, f0, f0, f0, f0, f0, f0}, b1 = {f1, f1, f1, f1, f1, f1, f1, f1};
                    const v8d b2 = {f2, f2, f2, f2, f2, f2, f2, f2}, b3 = {f3, f3, f3, f3, f3, f3, f3, f3};
                    y00 -= b0 * u0; y01 -= b0 * u1; y02 -= b0 * u2; y03 -= b0 * u3;
                    y10 -= b1 * u0; y11 -= b1 * u1; y12 -= b1 * u2; y13 -= b1 * u3;
                    y20 -= b2 * u0; y21 -= b2 * u1; y22 -= b2 * u2; y23 -= b2 * u3;
                    y30 -= b3 * u0; y31 -= b3 * u1; y32 -= b3 * u2; y33 -= b3 * u3;
                }
                *(v8du*)(r0) = y00; *(v8du*)(r0 + 8) = y01; *(v8du*)(r0 + 16) = y02; *(v8du*)(r0 + 24) = y03;
                *(v8du*)(r1) = y10; *(v8du*)(r1 + 8) = y11; *(v8du*)(r1 + 16) = y12; *(v8du*)(r1 + 24) = y13;
                *(v8du*)(r2) = y20; *(v8du*)(r2 + 8) = y21; *(v8du*)(r2 + 16) = y22; *(v8du*)(r2 + 24) = y23;
                *(v8du*)(r3) = y30; *(v8du*)(r3 + 8) = y31; *(v8du*)(r3 + 16) = y32; *(v8du*)(r3 + 24) = y33;
            }
        }
        if (cw == 16) {               // half chunk at the right edge (n = 16 mod 32: the ACE width 142 pads to 144, not 160)
            for (; i + 4 <= i_end; i += 4) {
                double* r0 = a + (size_t)i * n + c0;
                double* r1 = r0 + n;
                double* r2 = r1 + n;
                double* r3 = r2 + n;
                v8d y00 = *(const v8du*)(r0), y01 = *(const v8du*)(r0 + 8);
                v8d y10 = *(const v8du*)(r1), y11 = *(const v8du*)(r1 + 8);
                v8d y20 = *(const v8du*)(r2), y21 = *(const v8du*)(r2 + 8);
                v8d y30 = *(const v8du*)(r3), y31 = *(const v8du*)(r3 + 8);
                const double* uk = a + (size_t)jb * n;
                for (int k = 0; k < nk; ++k, uk += n) {
                    const v8d u0 = *(const v8du*)(uk + c0), u1 = *(const v8du*)(uk + c0 + 8);
                    const double f0 = uk[i], f1 = uk[i + 1], f2 = uk[i + 2], f3 = uk[i + 3];
                    const v8d b0 = {f0, f0, f0, f0, f0, f0, f0, f0}, b1 = {f1, f1, f1, f1, f1, f1, f1, f1};
                    const v8d b2 = {f2, f2, f2, f2, f2, f2, f2, f2}, b3 = {f3, f3, f3, f3, f3, f3, f3, f3};
                    y00 -= b0 * u0; y01 -= b0 * u1;
                    y10 -= b1 * u0; y11 -= b1 * u1;
                    y20 -= b2 * u0; y21 -= b2 * u1;
                    y30 -= b3 * u0; y31 -= b3 * u1;
                }
                *(v8du*)(r0) = y00; *(v8du*)(r0 + 8) = y01;
                *(v8du*)(r1) = y10; *(v8du*)(r1 + 8) = y11;
                *(v8du*)(r2) = y20; *(v8du*)(r2 + 8) = y21;
                *(v8du*)(r3) = y30; *(v8du*)(r3 + 8) = y31;
            }
        }
        for (; i < i_end; ++i) {      // leftover rows / ragged last chunk
            double* ri = a + (size_t)i * n + c0;
            const double* uk = a + (size_t)jb * n;
            for (int k = 0; k < nk; ++k, uk += n) axpy_neg(ri, uk + c0, uk[i], cw);
        }
    }
}

// rows i = i0, i0 + step, ... of the chunk columns: used by the threaded driver for large n
int chol_upper_rb(double* a, int n, double* min_piv2, int NBK) {
    double mp = std::numeric_limits<double>::infinity();
    for (int jb = 0; jb < n; jb += NBK) {
        const int je = (jb + NBK < n) ? jb + NBK : n;
        const int fail = chol_panel(a, n, jb, je, &mp);
        if (fail >= 0) {
            if (min_piv2) *min_piv2 = mp;
            return fail;
        }
        if (je < n) chol_trailing_chunks(a, n, jb, je, 0, 1);
    }
    if (min_piv2) *min_piv2 = mp;
    return -1;
}

// solve U^T U x = rhs in place (u upper, row-major): forward sweep in axpy form, backward
// sweep by dot products — both over contiguous row tails
FSNAP_CLONES void chol_solve_inv(const double* u, int n, double* x, double* inv) {
    // reciprocals of the diagonal first (independent, pipelined divisions): the two sweeps are dependency
    // chains of n steps each and a division on the chain costs more than the rest of a step
    for (int k = 0; k < n; ++k) inv[k] = 1.0 / u[(size_t)k * n + k];
    for (int k = 0; k < n; ++k) {
        const double* r = u + (size_t)k * n;
        x[k] *= inv[k];
        axpy_neg(x + k + 1, r + k + 1, x[k], n - 1 - k);
    }
    for (int i = n - 1; i >= 0; --i) {
        const double* r = u + (size_t)i * n;
        x[i] = (x[i] - dotv(r + i + 1, x + i + 1, n - 1 - i)) * inv[i];
    }
}
void chol_solve(const double* u, int n, double* x) {
    static thread_local vec inv;
    inv.resize(n);
    chol_solve_inv(u, n, x, inv.data());
}

// LU with partial pivoting, solve a x = b (a destroyed).  Returns false if singular.
bool lu_solve(double* a, int n, double* b) {
    std::vector<int> piv(n);
    for (int j = 0; j < n; ++j) {
        int p = j;
        double mx = std::fabs(a[(size_t)j * n + j]);
        for (int i = j + 1; i < n; ++i) {
            double v = std::fabs(a[(size_t)i * n + j]);
            if (v > mx) {
                mx = v;
                p = i;
            }
        }
        if (!(mx > 0.0) || !std::isfinite(mx)) return false;
        if (p != j) {
            for (int k = 0; k < n; ++k) std::swap(a[(size_t)j * n + k], a[(size_t)p * n + k]);
            std::swap(b[j], b[p]);
        }
        const double inv = 1.0 / a[(size_t)j * n + j];
        for (int i = j + 1; i < n; ++i) {
            double f = a[(size_t)i * n + j] * inv;
            if (f == 0.0) continue;
            a[(size_t)i * n + j] = f;
            for (int k = j + 1; k < n; ++k) a[(size_t)i * n + k] -= f * a[(size_t)j * n + k];
            b[i] -= f * b[j];
        }
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= a[(size_t)i * n + k] * b[k];
        b[i] = s / a[(size_t)i * n + i];
    }
    return true;
}

// Cyclic Jacobi eigendecomposition of the symmetric n x n matrix a (row-major, full
// storage, destroyed): on return eval[i] are the eigenvalues and the ROWS of v the
// eigenvectors.  Jacobi is used because it resolves small eigenvalues of graded
// matrices to high relative accuracy (Demmel & Veselic 1992), which is what the huge
// dynamic range of G needs.
void jacobi_eigh(double* a, int n, double* eval, double* v) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) v[(size_t)i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; ++i) {
            diag += a[(size_t)i * n + i] * a[(size_t)i * n + i];
            for (int j = i + 1; j < n; ++j) off += a[(size_t)i * n + j] * a[(size_t)i * n + j];
        }
        if (off == 0.0 || off <= 1e-34 * diag) break;
        for (int p = 0; p < n - 1; ++p) {
            for (int q = p + 1; q < n; ++q) {
                const double apq = a[(size_t)p * n + q];
                if (apq == 0.0) continue;
                const double app = a[(size_t)p * n + p], aqq = a[(size_t)q * n + q];
                // skip if negligible relative to the geometric mean of the diagonals
                if (std::fabs(apq) <= 1e-18 * std::sqrt(std::fabs(app) * std::fabs(aqq))) {
                    a[(size_t)p * n + q] = a[(size_t)q * n + p] = 0.0;
                    continue;
                }
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
                // rotate rows/cols p,q of a (symmetric update, full storage)
                for (int k = 0; k < n; ++k) {
                    const double akp = a[(size_t)k * n + p], akq = a[(size_t)k * n + q];
                    a[(size_t)k * n + p] = cs * akp - sn * akq;
                    a[(size_t)k * n + q] = sn * akp + cs * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = a[(size_t)p * n + k], aqk = a[(size_t)q * n + k];
                    a[(size_t)p * n + k] = cs * apk - sn * aqk;
                    a[(size_t)q * n + k] = sn * apk + cs * aqk;
                }
                a[(size_t)p * n + q] = a[(size_t)q * n + p] = 0.0;
                // accumulate eigenvectors as rows of v
                double* vp = v + (size_t)p * n;
                double* vq = v + (size_t)q * n;
                for (int k = 0; k < n; ++k) {
                    const double x = vp[k], y = vq[k];
                    vp[k] = cs * x - sn * y;
                    vq[k] = sn * x + cs * y;
                }
            }
        }
    }
    for (int i = 0; i < n; ++i) eval[i] = a[(size_t)i * n + i];
}

// residual r = rhs - M x, each row accumulated with the compensated dot product of
// Ogita, Rump & Oishi (Dot2: TwoProduct via FMA + TwoSum), i.e. as if in twice the
// working precision; 4 lanes wide.  M symmetric n x n row-major.
void residual_dot2(const double* M, int n, const double* x, const double* rhs, double* r) {
    for (int i = 0; i < n; ++i) {
        const double* mi = M + (size_t)i * n;
        v4d s = {0, 0, 0, 0}, c = {0, 0, 0, 0};
        int k = 0;
        for (; k + 4 <= n; k += 4) {
            const v4d a = ld4(mi + k), b = ld4(x + k);
            const v4d p = a * b;
            v4d e;
            for (int t = 0; t < 4; ++t) e[t] = __builtin_fma(a[t], b[t], -p[t]);
            const v4d sn = s + p;
            const v4d bp = sn - s;
            const v4d err = (s - (sn - bp)) + (p - bp);
            s = sn;
            c += err + e;
        }
        double ss = 0.0, cc = 0.0;
        for (int t = 0; t < 4; ++t) {  // fold lanes (TwoSum)
            const double sn = ss + s[t];
            const double bp = sn - ss;
            cc += (ss - (sn - bp)) + (s[t] - bp) + c[t];
            ss = sn;
        }
        for (; k < n; ++k) {
            const double p = mi[k] * x[k];
            const double e = __builtin_fma(mi[k], x[k], -p);
            const double sn = ss + p;
            const double bp = sn - ss;
            cc += (ss - (sn - bp)) + (p - bp) + e;
            ss = sn;
        }
        // rhs - (ss + cc)
        const double t1 = rhs[i] - ss;
        r[i] = t1 - cc;
    }
}

struct Reduced {
    int n;                  // active (non-zero) columns
    std::vector<int> idx;   // active -> original column
    vec M;                  // n x n active system (G + alpha I restricted)
    vec rhs;                // n
};

bool all_finite(const double* p, size_t n) {
    // x * 0 is 0 for finite x and NaN otherwise: four independent vector sums, no branches
    const v8d zero = {0, 0, 0, 0, 0, 0, 0, 0};
    v8d s0 = zero, s1 = zero, s2 = zero, s3 = zero;
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        s0 += *(const v8du*)(p + i) * zero;
        s1 += *(const v8du*)(p + i + 8) * zero;
        s2 += *(const v8du*)(p + i + 16) * zero;
        s3 += *(const v8du*)(p + i + 24) * zero;
    }
    const v8d s = (s0 + s1) + (s2 + s3);
    double t = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    for (; i < n; ++i) t += p[i] * 0.0;
    return t == 0.0;
}

// Jacobi-scaled Cholesky solve with one refinement step.  Returns -1 ok, else failing
// pivot.  min_piv2 = smallest relative squared pivot of the scaled matrix (a cheap
// lower-bound style estimate of 1/cond).
int scaled_chol_solve(const vec& M, const vec& rhs, int n, vec& x, double* min_piv2, fsnap::CondEstimate* cond = nullptr) {
    vec d(n), y(n);
    static thread_local vec S;   // scratch reused across calls (a fit loop calls this every step)
    S.resize((size_t)n * n);
    for (int i = 0; i < n; ++i) {
        const double g = M[(size_t)i * n + i];
        if (!(g > 0.0)) {
            if (min_piv2) *min_piv2 = 0.0;
            return i;
        }
        d[i] = 1.0 / std::sqrt(g);
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) S[(size_t)i * n + j] = M[(size_t)i * n + j] * d[i] * d[j];
    static thread_local vec L;
    L = S;
    const int fail = (n >= 768) ? chol_upper_blocked(L.data(), n, min_piv2, 64) : chol_upper(L.data(), n, min_piv2);
    if (fail >= 0) return fail;
    if (cond) {
        const double* Lp = L.data();
        *cond = fsnap::lanczos_lambda_min(n, [Lp, n](double* v) {
            chol_solve(Lp, n, v);
            return true;
        });
    }
    for (int i = 0; i < n; ++i) y[i] = rhs[i] * d[i];
    vec z(y);
    chol_solve(L.data(), n, z.data());
    // one step of iterative refinement on the scaled system, residual in ~2x precision
    vec r(n);
    residual_dot2(S.data(), n, z.data(), y.data(), r.data());
    chol_solve(L.data(), n, r.data());
    for (int i = 0; i < n; ++i) z[i] += r[i];
    x.resize(n);
    for (int i = 0; i < n; ++i) x[i] = z[i] * d[i];
    return -1;
}

// truncated pseudo-inverse solve via Jacobi eigendecomposition; returns rank
int eig_pinv_solve(const vec& M, const vec& rhs, int n, double rel_cut, double shift, vec& x) {
    vec a(M), ev(n), V((size_t)n * n);
    jacobi_eigh(a.data(), n, ev.data(), V.data());
    double lmax = 0.0;
    for (int i = 0; i < n; ++i) lmax = std::fmax(lmax, std::fabs(ev[i]));
    x.assign(n, 0.0);
    int rank = 0;
    for (int i = 0; i < n; ++i) {
        if (!(ev[i] > rel_cut * lmax)) continue;
        ++rank;
        const double* vi = V.data() + (size_t)i * n;
        long double s = 0.0L;
        for (int k = 0; k < n; ++k) s += (long double)vi[k] * rhs[k];
        const double coef = (double)(s / (long double)(ev[i] + shift));
        for (int k = 0; k < n; ++k) x[k] += coef * vi[k];
    }
    return rank;
}

// panel step split for the threaded driver: (1) the NBK x NBK diagonal block, columns [jb, je) only ...
FSNAP_CLONES int chol_panel_diag(double* a, int n, int jb, int je, double* mp) {
    for (int j = jb; j < je; ++j) {
        double* uj = a + (size_t)j * n;
        const double d = uj[j];
        if (d < *mp) *mp = d;
        if (!(d > 0.0) || !std::isfinite(d)) return j;
        const double r = std::sqrt(d), inv = 1.0 / r;
        uj[j] = r;
        for (int k = j + 1; k < je; ++k) uj[k] *= inv;
        for (int i = j + 1; i < je; ++i) {
            const double f = uj[i];
            double* ui = a + (size_t)i * n;
            for (int k = i; k < je; ++k) ui[k] -= f * uj[k];
        }
    }
    return -1;
}
// ... (2) the forward substitution of the panel's row tails, restricted to the caller's 32-column chunks
FSNAP_CLONES void chol_panel_tails(double* a, int n, int jb, int je, int first_chunk, int chunk_step) {
    for (int c0 = je + 32 * first_chunk; c0 < n; c0 += 32 * chunk_step) {
        const int cw = (n - c0 < 32) ? n - c0 : 32;
        for (int j = jb; j < je; ++j) {
            double* uj = a + (size_t)j * n;
            const double inv = 1.0 / uj[j];
            for (int k = 0; k < cw; ++k) uj[c0 + k] *= inv;
            for (int i = j + 1; i < je; ++i) axpy_neg(a + (size_t)i * n + c0, uj + c0, uj[i], cw);
        }
    }
}

// multi-threaded driver for large n: the threads live for the whole factorisation and meet at a spinning
// barrier after every panel (thread 0 factorises the panel) and after every trailing sweep
int chol_upper_rb_mt(double* a, int n, double* min_piv2, int NBK, int nt) {
    struct Shared {
        std::atomic<int> arrived{0};
        std::atomic<int> phase{0};
        std::atomic<int> fail{-1};
        double mp = std::numeric_limits<double>::infinity();
    } sh;
    auto barrier = [&](int& local_phase) {
        ++local_phase;
        if (sh.arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == nt) {
            sh.arrived.store(0, std::memory_order_relaxed);
            sh.phase.store(local_phase, std::memory_order_release);
        } else {
            int spins = 0;
            while (sh.phase.load(std::memory_order_acquire) != local_phase) {
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
                if (++spins > 4096) {          // oversubscribed host: let the thread we wait for run
                    std::this_thread::yield();
                    spins = 0;
                }
            }
        }
    };
    auto body = [&](int t) {
        int ph = 0;
        for (int jb = 0; jb < n; jb += NBK) {
            const int je = (jb + NBK < n) ? jb + NBK : n;
            if (t == 0) {
                const int f = chol_panel_diag(a, n, jb, je, &sh.mp);
                if (f >= 0) sh.fail.store(f, std::memory_order_relaxed);
            }
            barrier(ph);
            if (sh.fail.load(std::memory_order_relaxed) >= 0) return;
            if (je < n) {
                chol_panel_tails(a, n, jb, je, t, nt);
                barrier(ph);            // the trailing update reads panel entries of other threads' chunks
                chol_trailing_chunks(a, n, jb, je, t, nt);
            }
            barrier(ph);
        }
    };
    std::vector<std::thread> th;
    th.reserve(nt - 1);
    for (int t = 1; t < nt; ++t) th.emplace_back(body, t);
    body(0);
    for (auto& x : th) x.join();
    if (min_piv2) *min_piv2 = sh.mp;
    return sh.fail.load();
}

// factorisation used by the fast path; FSNAP_CHOL_VARIANT (environment, tests / tuning only) forces a variant:
// 0 = auto, 1 = unblocked, 2 = 64-row panels + threads, 3 = register-blocked chunks
int fast_chol(double* u, int n, double* mp2) {
    static const int forced = [] {
        const char* e = std::getenv("FSNAP_CHOL_VARIANT");
        return e ? std::atoi(e) : 0;
    }();
    static const int nbk_env = [] {
        const char* e = std::getenv("FSNAP_CHOL_NBK");
        return e ? std::atoi(e) : 0;
    }();
    if (forced == 1) return chol_upper(u, n, mp2);
    if (forced == 2) return chol_upper_blocked(u, n, mp2, 64);
    if (forced == 3 || n >= 48) {
        const int nbk = nbk_env > 0 ? nbk_env : (n <= 256 ? 8 : 32);
        static const int nt_env = [] {
            const char* e = std::getenv("FSNAP_CHOL_THREADS");
            return e ? std::atoi(e) : 0;
        }();
        // Host threads: FSNAP_CHOL_THREADS when set; else min(8, cores) -- unless the process runs under a CPU quota
        // (cgroup cpu.max / cfs_quota_us): there the spinning barriers of the threaded driver made the factorisation
        // SLOWER on the MI355X boxes (K = 1595: 18 -> 23-41 ms, K = 480: 0.4 -> 2.3 ms with 4 threads), so a quota'd
        // container stays on one thread.  Large systems of a fit are factorised on the GPU anyway (fsnap_solve_device);
        // this is the factorisation of the row-space chain and of callers that hold the statistics on the host.
        static const int nt_auto = [] {
            auto quota = [](const char* path, bool v2) -> bool {
                FILE* f = std::fopen(path, "r");
                if (!f) return false;
                char buf[64] = {0};
                const bool got = std::fgets(buf, sizeof buf, f) != nullptr;
                std::fclose(f);
                if (!got) return false;
                return v2 ? std::strncmp(buf, "max", 3) != 0 : std::atol(buf) > 0;
            };
            if (quota("/sys/fs/cgroup/cpu.max", true) || quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", false)) return 1;
            const unsigned hc = std::thread::hardware_concurrency();
            return (int)(hc >= 8 ? 8 : (hc > 0 ? hc : 1));
        }();
        int nt = nt_env > 0 ? nt_env : nt_auto;
        if (nt > n / 160) nt = n / 160;          // >= 5 column chunks per thread
        if (nt > 1) return chol_upper_rb_mt(u, n, mp2, nbk, nt);
        return chol_upper_rb(u, n, mp2, nbk);
    }
    return chol_upper(u, n, mp2);
}

}  // namespace

namespace {
struct PhaseTimer {   // FSNAP_SOLVE_TIMING=1: print the phases of the fast path to stderr (tuning aid)
    bool on;
    std::chrono::steady_clock::time_point t0;
    PhaseTimer() : on(std::getenv("FSNAP_SOLVE_TIMING") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void lap(const char* what) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[fsnap_solve] %-14s %9.1f us\n", what, std::chrono::duration<double, std::micro>(t1 - t0).count());
        t0 = t1;
    }
};
}  // namespace

extern "C" int fsnap_solve_diag(int kind, double param, int64_t K64, const double* G, const double* c, const double* diag,
                                double* beta, int* rank_out, double* rcond_est);
extern "C" int fsnap_solve_diag_upper(int kind, double param, int64_t K64, const double* G, const double* c, const double* diag,
                                      double* beta, int* rank_out, double* rcond_est);
namespace {
int solve_impl(int kind, double param, int64_t K64, const double* G, const double* c, const double* diag, double* beta,
               int* rank_out, double* rcond_est, bool upper_only, const void* owner = nullptr, unsigned long long generation = 0);
}

// In-place Cholesky U^T U of an n x n row-major matrix for the other translation units of the library (the row-space
// passes, fsnap_rowspace.cpp): upper triangle in, U out, lower triangle must be zero on entry; n a multiple of 32 runs
// the register-blocked variant at full speed.  Returns -1 or the failing pivot; *min_piv = smallest pivot before sqrt.
extern "C" __attribute__((visibility("hidden"))) int fsnap_host_chol_upper(double* a, int n, double* min_piv) {
    return fast_chol(a, n, min_piv);
}

// Largest theta with M x = theta N x for symmetric k x k matrices (row-major), N positive definite: Cholesky of N, the
// congruence C = U^-T M U^-1, Lanczos on C.  The Rayleigh-Ritz step of the device factor's condition estimate (k = 31:
// fsnap_chol_probe_gram_k, fsnap_solve_device_rhs).  Returns -1.0 when N is not positive definite or a value is not finite.
extern "C" __attribute__((visibility("hidden"))) double fsnap_gen_eig_max(const double* M, const double* N, int k) {
    if (k <= 0 || !all_finite(M, (size_t)k * k) || !all_finite(N, (size_t)k * k)) return -1.0;
    vec U(N, N + (size_t)k * k), C((size_t)k * k);
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < i; ++j) U[(size_t)i * k + j] = 0.0;
    double mp = 0.0;
    if (chol_upper(U.data(), k, &mp) >= 0) return -1.0;
    // T = U^-T M: forward substitution down every column of M; then C^T = U^-T T^T the same way (C is symmetric)
    vec T(M, M + (size_t)k * k);
    auto forward_columns = [&](vec& X) {
        for (int i = 0; i < k; ++i) {
            const double inv = 1.0 / U[(size_t)i * k + i];
            for (int c = 0; c < k; ++c) {
                double sacc = X[(size_t)i * k + c];
                for (int l = 0; l < i; ++l) sacc -= U[(size_t)l * k + i] * X[(size_t)l * k + c];
                X[(size_t)i * k + c] = sacc * inv;
            }
        }
    };
    forward_columns(T);
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j) C[(size_t)i * k + j] = T[(size_t)j * k + i];
    forward_columns(C);
    for (int i = 0; i < k; ++i)
        for (int j = i + 1; j < k; ++j) C[(size_t)i * k + j] = C[(size_t)j * k + i] = 0.5 * (C[(size_t)i * k + j] + C[(size_t)j * k + i]);
    // largest eigenvalue of the k x k matrix C: Lanczos with full re-orthogonalisation (a dozen matrix-vector products of k^2
    // flops; the cyclic Jacobi sweeps this replaced took 0.25 ms at k = 31 -- more than the launch they follow)
    const double* Cp = C.data();
    vec tmp((size_t)k);
    const fsnap::CondEstimate top = fsnap::lanczos_lambda_min(k, [Cp, k, &tmp](double* x) {
        for (int i = 0; i < k; ++i) {
            double acc = 0.0;
            for (int j = 0; j < k; ++j) acc += Cp[(size_t)i * k + j] * x[j];
            tmp[i] = acc;
        }
        for (int i = 0; i < k; ++i) x[i] = tmp[i];
        return true;
    }, 4, 16);
    return top.lambda_min > 0.0 && std::isfinite(top.lambda_min) ? 1.0 / top.lambda_min : -1.0;
}

extern "C" int fsnap_solve(int kind, double param, int64_t K64, const double* G, const double* c, double* beta,
                           int* rank_out, double* rcond_est) {
    return fsnap_solve_diag(kind, param, K64, G, c, nullptr, beta, rank_out, rcond_est);
}

// diag (optional): contiguous copy of diag(G) -- the device reduction writes one next to its host mirror so that
// the scaling pass does not start with K cold cache lines
extern "C" __attribute__((visibility("hidden"))) int fsnap_solve_diag(int kind, double param, int64_t K64, const double* G,
                                                                      const double* c, const double* diag, double* beta,
                                                                      int* rank_out, double* rcond_est) {
    return solve_impl(kind, param, K64, G, c, diag, beta, rank_out, rcond_est, false);
}

// the same on a matrix of which only the UPPER triangle (row i: columns >= i) is meaningful -- the host mirror the
// reduction kernel fills with one PCIe write per element; whatever sits below the diagonal is never read
extern "C" __attribute__((visibility("hidden"))) int fsnap_solve_diag_upper(int kind, double param, int64_t K64, const double* G,
                                                                            const double* c, const double* diag, double* beta,
                                                                            int* rank_out, double* rcond_est) {
    return solve_impl(kind, param, K64, G, c, diag, beta, rank_out, rcond_est, true);
}

// The same with a tag of the matrix: (owner, generation) names ONE content of G (the context's host mirror after its
// n-th fill).  A second solve with the same tag, kind and parameter -- the refinement steps of the SVD solver solve
// G delta = s two more times per fit -- reuses the Cholesky factor the first one left in this thread's workspace and
// runs the two triangular sweeps only (3.5 instead of 23 us at K = 128).  upper != 0: fsnap_solve_diag_upper semantics.
extern "C" __attribute__((visibility("hidden"))) int fsnap_solve_diag_tagged(int kind, double param, int64_t K64, const double* G,
                                                                             const double* c, const double* diag, double* beta,
                                                                             int* rank_out, double* rcond_est, int upper,
                                                                             const void* owner, unsigned long long generation) {
    return solve_impl(kind, param, K64, G, c, diag, beta, rank_out, rcond_est, upper != 0, owner, generation);
}

namespace {
int solve_impl(int kind, double param, int64_t K64, const double* G, const double* c, const double* diag, double* beta,
               int* rank_out, double* rcond_est, bool upper_only, const void* owner, unsigned long long generation) {
    PhaseTimer timer;
    if (!G || !c || !beta || K64 <= 0 || K64 > (1 << 20)) return FSNAP_E_ARG;
    if (kind < FSNAP_SOLVE_CHOL || kind > FSNAP_SOLVE_RIDGE_INV_PROBE) return FSNAP_E_ARG;
    const bool probe = kind >= FSNAP_SOLVE_LSTSQ_PROBE;                     // no eigen / LU fallback
    if (kind == FSNAP_SOLVE_LSTSQ_PROBE) kind = FSNAP_SOLVE_LSTSQ;
    if (kind == FSNAP_SOLVE_RIDGE_PROBE) kind = FSNAP_SOLVE_RIDGE;
    if (kind == FSNAP_SOLVE_RIDGE_INV_PROBE) kind = FSNAP_SOLVE_RIDGE_INV;
    const int K = (int)K64;
    if (!std::isfinite(param)) return FSNAP_NUM_NONFINITE;
    const double alpha = (kind == FSNAP_SOLVE_RIDGE || kind == FSNAP_SOLVE_RIDGE_INV) ? param : 0.0;
    const double eps = std::numeric_limits<double>::epsilon();
    timer.lap("finite check");

    // ---- fast path (the common case of a fit loop): no zero column, well conditioned -----
    // One contiguous pass over the UPPER triangle of G builds the Jacobi-scaled matrix (G is
    // already exactly symmetric: the GPU reduction mirrors the triangle), checks finiteness
    // on the fly, then Cholesky + two triangular sweeps.  Falls through to the general path
    // when a diagonal entry is not positive, a pivot is small, or a value is not finite.
    {
        // The register-blocked factorisation works on 32-column chunks; a partial last chunk runs a much slower
        // remainder path (K = 184: 0.30 ms, K = 192: 0.12 ms).  So the scaled matrix is padded to a multiple of 32 with
        // an identity block (pivots 1, solution components 0): Kp = leading dimension and order of the padded system.
        // (a final HALF chunk of 16 columns has a register-blocked path of its own: K = 142 pads to 144, not 160)
        int Kp = (K >= 48 && (K & 15) != 0) ? ((K + 15) & ~15) : K;
        // a row stride that is a multiple of 1 KiB maps the rows of a column onto a few cache sets: on the MI355X
        // hosts K = 512 took 0.9 or 8.9 ms depending on the physical pages of the run -- one more chunk of padding
        if (Kp >= 384 && (Kp & 127) == 0) Kp += 32;
        static thread_local vec U, dsc, z;
        // the factor this thread's workspace holds: valid for one tagged content of G (see fsnap_solve_diag_tagged)
        static thread_local struct { const void* owner; unsigned long long gen; int K; double alpha, mp2, piv, lam; } held = {nullptr, 0, 0, 0.0, 0.0, 0.0, 0.0};
        if (owner && held.owner == owner && held.gen == generation && held.K == K && held.alpha == alpha &&
            U.size() == (size_t)Kp * Kp) {
            bool fin = true;
            for (int i = 0; i < K; ++i) {
                z[i] = c[i] * dsc[i];
                fin = fin && (c[i] - c[i] == 0.0);
            }
            if (fin) {
                for (int i = K; i < Kp; ++i) z[i] = 0.0;
                chol_solve(U.data(), Kp, z.data());
                for (int i = 0; i < K; ++i) beta[i] = z[i] * dsc[i];
                timer.lap("tri solves (factor reused)");
                if (all_finite(beta, K)) {
                    if (rank_out) *rank_out = K;
                    if (rcond_est) *rcond_est = held.mp2;
                    fsnap_cond_note(held.piv, held.lam, 0, 0);
                    return FSNAP_OK;
                }
            }
        }
        held.owner = nullptr;                 // the workspace is about to be overwritten
        U.resize((size_t)Kp * Kp);
        dsc.resize(Kp);
        z.resize(Kp);
        bool ok = true;
        double chk = 0.0;
        for (int i = 0; i < K && ok; ++i) {
            const double g = (diag ? diag[i] : G[(size_t)i * K + i]) + alpha;
            if (!(g > 0.0) || !std::isfinite(g)) ok = false;
            else dsc[i] = 1.0 / std::sqrt(g);
            chk += c[i] * 0.0;
        }
        timer.lap("diag scale");
        if (ok) {
            // rows are built in 8-wide vectors from the vector that holds the diagonal: the < 8 entries left of the
            // diagonal receive (valid, unused) scaled values, everything further left is zeroed -- the blocked
            // factorisation updates whole 32-column chunks.  The finiteness of everything read is folded into chkv.
            const v8d zero = {0, 0, 0, 0, 0, 0, 0, 0};
            v8d chk0 = zero, chk1 = zero;
            for (int i = 0; i < K; ++i) {
                const double* gi = G + (size_t)i * K;
                double* ui = U.data() + (size_t)i * Kp;
                const double di = dsc[i];
                const v8d dv = {di, di, di, di, di, di, di, di};
                const int j0 = i & ~7;
                if (i + 2 < K) {     // the statistics were just written by the device: fetch the row after next early
                    const double* gn = G + (size_t)(i + 2) * K;
                    for (int jp = (i + 2) & ~7; jp < K; jp += 8) __builtin_prefetch(gn + jp, 0, 0);
                }
                int j = 0;
                for (; j + 8 <= j0; j += 8) *(v8du*)(ui + j) = zero;
                if (upper_only && j + 8 <= K) {
                    // the vector that holds the diagonal: what lies left of it was never written by the device
                    typedef long long v8l __attribute__((vector_size(64)));
                    const v8l lanes = {0, 1, 2, 3, 4, 5, 6, 7};
                    const v8l keep = (lanes + (long long)j) >= (long long)i;
                    const v8d g0 = (v8d)((v8l)(*(const v8du*)(gi + j)) & keep);
                    chk0 += g0 * zero;
                    *(v8du*)(ui + j) = g0 * dv * *(const v8du*)(dsc.data() + j);
                    j += 8;
                }
                for (; j + 16 <= K; j += 16) {
                    const v8d g0 = *(const v8du*)(gi + j), g1 = *(const v8du*)(gi + j + 8);
                    chk0 += g0 * zero;
                    chk1 += g1 * zero;
                    *(v8du*)(ui + j) = g0 * dv * *(const v8du*)(dsc.data() + j);
                    *(v8du*)(ui + j + 8) = g1 * dv * *(const v8du*)(dsc.data() + j + 8);
                }
                for (; j + 8 <= K; j += 8) {
                    const v8d g0 = *(const v8du*)(gi + j);
                    chk0 += g0 * zero;
                    *(v8du*)(ui + j) = g0 * dv * *(const v8du*)(dsc.data() + j);
                }
                for (; j < K; ++j) {
                    const double gij = (upper_only && j < i) ? 0.0 : gi[j];
                    ui[j] = gij * di * dsc[j];
                    chk += gij * 0.0;
                }
                for (; j < Kp; ++j) ui[j] = 0.0;
                ui[i] = (gi[i] + alpha) * di * di;
            }
            for (int i = K; i < Kp; ++i) {          // identity block of the padding
                double* ui = U.data() + (size_t)i * Kp;
                for (int j = 0; j < Kp; ++j) ui[j] = 0.0;
                ui[i] = 1.0;
            }
            const v8d cs = chk0 + chk1;
            chk += ((cs[0] + cs[1]) + (cs[2] + cs[3])) + ((cs[4] + cs[5]) + (cs[6] + cs[7]));
            double mp2 = 0.0;
            timer.lap("scale/build");
            const bool fact_ok = (chk == 0.0) && fast_chol(U.data(), Kp, &mp2) < 0;
            timer.lap("cholesky");
            // LSTSQ stands in for an SVD of the rows (svd.py:54), which knows the conditioning: the smallest pivot does not
            // (it bounds lambda_min from above only), so the factor is asked -- a few S^-1 applications, fsnap_condest.h.
            // A factor whose lambda_min is at the rounding level of the statistics is no solution: the general path decides.
            fsnap::CondEstimate ce;
            bool cond_ok = true;
            if (fact_ok && mp2 > 1.0e-3 && kind == FSNAP_SOLVE_LSTSQ) {
                const double* Up = U.data();
                ce = fsnap::lanczos_lambda_min(Kp, [Up, Kp](double* v) {
                    chol_solve(Up, Kp, v);
                    return true;
                });
                cond_ok = ce.lambda_min > 64.0 * K * eps;
                timer.lap("cond estimate");
            }
            if (fact_ok && mp2 > 1.0e-3 && cond_ok) {
                const double piv = mp2;
                if (ce.steps && ce.lambda_min < mp2) mp2 = ce.lambda_min;
                fsnap_cond_note(piv, ce.lambda_min, ce.steps, 0);
                for (int i = 0; i < K; ++i) z[i] = c[i] * dsc[i];
                for (int i = K; i < Kp; ++i) z[i] = 0.0;
                chol_solve(U.data(), Kp, z.data());
                for (int i = 0; i < K; ++i) beta[i] = z[i] * dsc[i];
                timer.lap("tri solves");
                if (all_finite(beta, K)) {
                    if (rank_out) *rank_out = K;
                    if (rcond_est) *rcond_est = mp2;
                    if (owner) {
                        held.owner = owner;
                        held.gen = generation;
                        held.K = K;
                        held.alpha = alpha;
                        held.mp2 = mp2;
                        held.piv = piv;
                        held.lam = ce.lambda_min;
                    }
                    return FSNAP_OK;
                }
            }
        }
    }

    static thread_local vec Gfull;
    if (upper_only) {                    // the general path reads both triangles: mirror the meaningful one
        Gfull.resize((size_t)K * K);
        for (int i = 0; i < K; ++i)
            for (int j = i; j < K; ++j) Gfull[(size_t)i * K + j] = Gfull[(size_t)j * K + i] = G[(size_t)i * K + j];
        G = Gfull.data();
    }
    if (!all_finite(G, (size_t)K * K) || !all_finite(c, K)) return FSNAP_NUM_NONFINITE;
    // active columns: drop exactly-zero columns when there is no ridge shift
    static thread_local Reduced R;
    R.idx.clear();
    R.idx.reserve(K);
    for (int j = 0; j < K; ++j) {
        const double gjj = G[(size_t)j * K + j] + alpha;
        if (gjj == 0.0 && kind != FSNAP_SOLVE_CHOL && kind != FSNAP_SOLVE_RIDGE_INV) continue;  // beta_j = 0
        R.idx.push_back(j);
    }
    const int n = R.n = (int)R.idx.size();
    for (int j = 0; j < K; ++j) beta[j] = 0.0;
    if (rank_out) *rank_out = 0;
    if (rcond_est) *rcond_est = 0.0;
    if (n == 0) return FSNAP_OK;
    R.M.resize((size_t)n * n);
    R.rhs.resize(n);
    for (int i = 0; i < n; ++i) {
        R.rhs[i] = c[R.idx[i]];
        for (int j = 0; j < n; ++j) {
            // symmetrise defensively (the GPU path already mirrors the triangle)
            const double gij = 0.5 * (G[(size_t)R.idx[i] * K + R.idx[j]] + G[(size_t)R.idx[j] * K + R.idx[i]]);
            R.M[(size_t)i * n + j] = gij + ((i == j) ? alpha : 0.0);
        }
    }

    vec x;
    double mp2 = 0.0;
    fsnap::CondEstimate ce;
    const int fail = scaled_chol_solve(R.M, R.rhs, n, x, &mp2, kind == FSNAP_SOLVE_LSTSQ ? &ce : nullptr);
    const double piv = mp2;
    if (fail < 0 && ce.steps && ce.lambda_min < mp2) mp2 = ce.lambda_min;     // (see the fast path)
    fsnap_cond_note(piv, ce.lambda_min, ce.steps, 0);
    if (rcond_est) *rcond_est = mp2;
    // a Jacobi-scaled SPD matrix with relative pivot -- or smallest eigenvalue -- below ~n*eps has lost all digits
    const double piv_tol = 64.0 * n * eps;
    const bool chol_ok = (fail < 0) && (mp2 > piv_tol);

    if (chol_ok) {
        for (int i = 0; i < n; ++i) beta[R.idx[i]] = x[i];
        if (rank_out) *rank_out = n;
        return all_finite(beta, K) ? FSNAP_OK : FSNAP_NUM_NONFINITE;
    }

    if (probe) {        // unresolved: the caller falls back on the rows / on LAPACK (rcond_est holds the smallest pivot)
        if (rank_out) *rank_out = -1;
        return FSNAP_OK;
    }
    switch (kind) {
        case FSNAP_SOLVE_CHOL:
            return FSNAP_NUM_NOT_SPD;
        case FSNAP_SOLVE_LSTSQ: {
            // gelsd semantics: drop singular values sigma < rcond * sigma_max, i.e.
            // eigenvalues of G below rcond^2 * lambda_max; eigenvalues of a computed G are
            // not resolved below ~n*eps*lambda_max, so that is the floor of the cut.
            double cut = param > 0 ? param * param : 0.0;
            const double floor_cut = 4.0 * n * eps;
            if (cut < floor_cut) cut = floor_cut;
            const int rk = eig_pinv_solve(R.M, R.rhs, n, cut, 0.0, x);
            for (int i = 0; i < n; ++i) beta[R.idx[i]] = x[i];
            if (rank_out) *rank_out = rk;
            return all_finite(beta, K) ? FSNAP_OK : FSNAP_NUM_NONFINITE;
        }
        case FSNAP_SOLVE_RIDGE: {
            // sklearn falls back from Cholesky to an SVD solve (linear_model/_ridge.py,
            // _ridge_regression: except LinAlgError -> solver = 'svd'); on the statistics
            // that is the eigen form coef = V diag(1/(lambda + alpha)) V^T c over
            // lambda > 1e-15-ish.  R.M already contains the alpha shift.
            const int rk = eig_pinv_solve(R.M, R.rhs, n, 4.0 * n * eps, 0.0, x);
            for (int i = 0; i < n; ++i) beta[R.idx[i]] = x[i];
            if (rank_out) *rank_out = rk;
            return all_finite(beta, K) ? FSNAP_OK : FSNAP_NUM_NONFINITE;
        }
        case FSNAP_SOLVE_RIDGE_INV: {
            // np.linalg.inv semantics: LU with partial pivoting; singular -> LinAlgError
            vec a(R.M), b(R.rhs);
            if (!lu_solve(a.data(), n, b.data())) return FSNAP_NUM_SINGULAR;
            for (int i = 0; i < n; ++i) beta[R.idx[i]] = b[i];
            if (rank_out) *rank_out = n;
            return all_finite(beta, K) ? FSNAP_OK : FSNAP_NUM_NONFINITE;
        }
    }
    return FSNAP_E_ARG;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// LASSO on the statistics (fitsnap3lib/solvers/lasso.py:17-29: sklearn Lasso(alpha, fit_intercept=False, max_iter) on
// (aw, bw)).  scikit-learn minimises (1 / 2n) |y - X w|^2 + alpha |w|_1 by cyclic coordinate descent; every update
// touches the data only through Q = X^T X, q = X^T y and |y|^2 (its own Gram variant, linear_model/_cd_fast.pyx
// enet_coordinate_descent_gram, is this algorithm with an elastic-net term this path does not use): one sweep updates
// w_i <- soft(q_i - (Q w)_i + Q_ii w_i, l1_reg) / Q_ii with H = Q w kept current by two axpys of a row of Q, and the
// stopping rule is the duality gap < tol |y|^2, checked once the largest update of a sweep falls below tol x the largest
// coefficient (or in the last sweep).  l1_reg = alpha * n_samples.  Same iterates as scikit-learn's row-space loop in
// exact arithmetic; on the Ta golden rows the coefficients agree to 1e-11 with identical sweep counts.
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int fsnap_lasso_gram(int64_t K64, const double* Q, const double* q, double y_norm2, double l1_reg, int64_t max_iter,
                                double tol, double* w, int64_t* n_iter_out, double* gap_out) {
    if (!Q || !q || !w || K64 <= 0 || K64 > (1 << 20) || max_iter < 1 || !(l1_reg >= 0.0) || !(tol >= 0.0)) return FSNAP_E_ARG;
    const int n = (int)K64;
    if (!all_finite(Q, (size_t)n * n) || !all_finite(q, n) || !std::isfinite(y_norm2)) return FSNAP_NUM_NONFINITE;
    vec H((size_t)n, 0.0);
    for (int i = 0; i < n; ++i) {          // H = Q w for the caller's start vector (zeros in the reference)
        if (w[i] == 0.0) continue;
        const double* Qi = Q + (size_t)i * n;
        for (int j = 0; j < n; ++j) H[j] += w[i] * Qi[j];
    }
    const double d_w_tol = tol;
    const double gap_tol = tol * y_norm2;
    double gap = gap_tol + 1.0;
    int64_t it = 0;
    for (it = 0; it < max_iter; ++it) {
        double w_max = 0.0, d_w_max = 0.0;
        for (int i = 0; i < n; ++i) {
            const double* __restrict__ Qi = Q + (size_t)i * n;
            const double Qii = Qi[i];
            if (Qii == 0.0) continue;
            const double w_old = w[i];
            double* __restrict__ h = H.data();
            if (w_old != 0.0)
                for (int j = 0; j < n; ++j) h[j] -= w_old * Qi[j];
            const double t = q[i] - h[i];
            const double mag = std::fabs(t) - l1_reg;
            const double w_new = mag > 0.0 ? std::copysign(mag, t) / Qii : 0.0;
            w[i] = w_new;
            if (w_new != 0.0)
                for (int j = 0; j < n; ++j) h[j] += w_new * Qi[j];
            d_w_max = std::fmax(d_w_max, std::fabs(w_new - w_old));
            w_max = std::fmax(w_max, std::fabs(w_new));
        }
        if (w_max == 0.0 || d_w_max / w_max < d_w_tol || it == max_iter - 1) {
            double q_dot_w = 0.0, wHw = 0.0, l1 = 0.0, dual = 0.0;
            for (int i = 0; i < n; ++i) {
                q_dot_w += w[i] * q[i];
                wHw += w[i] * H[i];
                l1 += std::fabs(w[i]);
                dual = std::fmax(dual, std::fabs(q[i] - H[i]));
            }
            const double R_norm2 = y_norm2 + wHw - 2.0 * q_dot_w;
            double c = 1.0;
            if (dual > l1_reg) {
                c = l1_reg / dual;
                gap = 0.5 * (R_norm2 + R_norm2 * c * c);
            } else {
                gap = R_norm2;
            }
            gap += l1_reg * l1 - c * y_norm2 + c * q_dot_w;
            if (gap < gap_tol) {
                ++it;
                break;
            }
        }
    }
    if (n_iter_out) *n_iter_out = it > max_iter ? max_iter : it;
    if (gap_out) *gap_out = gap;
    return all_finite(w, n) ? FSNAP_OK : FSNAP_NUM_NONFINITE;
}
