// fsnap_chol.hip — K x K Cholesky solves of the packed statistics on the GPU (gfx950 only):
//   6      fsnap_chol_solve_k   one workgroup, K <= 128 (optional: the host factorisation is faster)
//   8a-8e  blocked solve for K >= 768 (scaling, 64-row panels, MFMA trailing update, blocked sweeps)
// Same algorithm as the host fast path in fsnap_solve.cpp (Jacobi-scaled Cholesky, pivots checked by the caller).
#include "fsnap_device_common.h"
#include "fsnap_kernels.h"

// ---------------------------------------------------------------------------------
// Kernel 6: K x K solve on the device for K <= 128 (the latency path of a fit: avoids the
// D2H of G and the host factorisation).  ONE workgroup of 1024 threads; the Jacobi-scaled
// matrix S = D (G + alpha I) D, D = diag(G + alpha I)^-1/2, is held IN REGISTERS in a 32 x 32
// block-cyclic distribution (thread (ti, tk) owns S[ti + 32a][tk + 32b], a, b < 4), so the
// right-looking upper Cholesky S = U^T U does no LDS read-modify-write: per column the owners
// of the pivot row publish it (unscaled) through a double-buffered 1 KB LDS row, ONE barrier,
// then every thread updates its 16 elements.  Finished rows of U are parked in LDS (row
// stride K + 1: row and column access conflict-free) for the forward / backward sweeps,
// which one wave runs with x in registers and pre-inverted diagonals.
// Same arithmetic as the host fast path (fsnap_solve.cpp): no refinement; the host falls
// back to the full host solver when the kernel reports a small pivot, a non-positive
// diagonal or a non-finite value.
//   in : packed statistics [G (K*K) | c (K) | ...]
//   out: [beta (K) | min relative pivot | status (0 ok, 1 = fall back)]
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void fsnap_chol_solve_k(const double* __restrict__ packed, int K, double alpha,
                                                           double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int LD = K + 1;
    double* U = sm;                        // K x LD, final (scaled) rows of U
    double* dsc = sm + (size_t)K * LD;     // K
    double* rowbuf = dsc + K;              // 2 x 128, unscaled pivot rows
    __shared__ int bad;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ti = tid >> 5, tk = tid & 31;
    const double* G = packed;
    const double* c = packed + (size_t)K * K;
    if (tid == 0) bad = 0;
    __syncthreads();
    for (int i = tid; i < K; i += 1024) {
        const double g = G[(size_t)i * K + i] + alpha;
        if (!(g > 0.0) || !(g < 1.0e300)) {
            bad = 1;
            dsc[i] = 0.0;
        } else {
            dsc[i] = 1.0 / sqrt(g);
        }
    }
    __syncthreads();
    double e[4][4];
    double chk = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int i = ti + 32 * a;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int k = tk + 32 * b;
            double v = 0.0;
            if (i < K && k < K && k >= i) {
                const double g = G[(size_t)i * K + k];
                chk += g * 0.0;
                v = ((i == k) ? g + alpha : g) * dsc[i] * dsc[k];
            }
            e[a][b] = v;
        }
    }
    if (tid < K) chk += c[tid] * 0.0;
    if (chk != 0.0) bad = 1;   // NaN: some entry was not finite
    __syncthreads();
    double minp = 1.0e300;
    if (!bad) {
        for (int j = 0; j < K; ++j) {
            double* rb = rowbuf + (j & 1) * 128;
            const int aj = j >> 5;
            if (ti == (j & 31)) {   // owners of row j publish it (unscaled)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    if (a == aj) {
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const int k = tk + 32 * b;
                            if (k >= j && k < K) rb[k] = e[a][b];
                        }
                    }
                }
            }
            __syncthreads();
            const double d = rb[j];
            if (d < minp) minp = d;
            if (!(d > 0.0)) {   // uniform: every thread reads the same value
                minp = 0.0;
                break;
            }
            const double r = sqrt(d), inv = 1.0 / r;
            double fi[4], gk[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int i = ti + 32 * a;
                fi[a] = (i > j && i < K) ? rb[i] * inv : 0.0;
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int k = tk + 32 * b;
                gk[b] = (k > j && k < K) ? rb[k] * inv : 0.0;
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) e[a][b] -= fi[a] * gk[b];   // rows i <= j get fi = 0
            // park the final row j of U for the sweeps
            if (ti == (j & 31)) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int k = tk + 32 * b;
                    if (k > j && k < K) U[j * LD + k] = gk[b];
                    if (k == j) U[j * LD + j] = r;
                }
            }
        }
    }
    __syncthreads();
    const bool fail = bad || !(minp > 0.0);
    if (wv == 0) {
        if (!fail) {
            // pre-inverted diagonal
            const double id0 = (lane < K) ? 1.0 / U[lane * LD + lane] : 0.0;
            const double id1 = (lane + 64 < K) ? 1.0 / U[(lane + 64) * LD + lane + 64] : 0.0;
            // forward: U^T y = D c   (axpy form over contiguous rows), x in registers
            double x0 = (lane < K) ? c[lane] * dsc[lane] : 0.0;
            double x1 = (lane + 64 < K) ? c[lane + 64] * dsc[lane + 64] : 0.0;
            for (int k = 0; k < K; ++k) {
                const double yk = (k < 64) ? __shfl(x0 * id0, k, 64) : __shfl(x1 * id1, k - 64, 64);
                if (lane == (k & 63)) {
                    if (k < 64) x0 = yk;
                    else x1 = yk;
                }
                if (lane > k && lane < K) x0 -= U[k * LD + lane] * yk;
                if (lane + 64 > k && lane + 64 < K) x1 -= U[k * LD + lane + 64] * yk;
            }
            // backward: U x = y   (column access; LD = K + 1 keeps it conflict free)
            for (int i = K - 1; i >= 0; --i) {
                const double xi = (i < 64) ? __shfl(x0 * id0, i, 64) : __shfl(x1 * id1, i - 64, 64);
                if (lane == (i & 63)) {
                    if (i < 64) x0 = xi;
                    else x1 = xi;
                }
                if (lane < i) x0 -= U[lane * LD + i] * xi;
                if (lane + 64 < i) x1 -= U[(lane + 64) * LD + i] * xi;
            }
            if (lane < K) out[lane] = x0 * dsc[lane];
            if (lane + 64 < K) out[lane + 64] = x1 * dsc[lane + 64];
        }
        if (lane == 0) {
            out[K] = minp;
            out[K + 1] = fail ? 1.0 : 0.0;
        }
    }
}

// ---------------------------------------------------------------------------------
// Kernels 8a-8e: blocked Cholesky solve of the K x K statistics on the GPU for LARGE K (ACE / quadratic-SNAP
// widths; the host factorisation takes 17-22 ms at K = 1595).  Same algorithm as the host fast path
// (fsnap_solve.cpp): Jacobi scaling S = D (G + alpha I) D with D = diag(G + alpha I)^-1/2, S = U^T U,
// two triangular sweeps, beta = D x; accepted by the caller only if every pivot of the scaled matrix stays
// above 1e-3 (otherwise the general host path runs).
// The work matrix is padded to a multiple of 64 with an identity block, so no kernel has edge cases, and carries
// an extra 32-column strip whose first column is the scaled right-hand side z = D c: the factorisation transforms
// it along with the trailing columns, which IS the forward sweep U^T y = z.
//   8a prepare   d, z = D c, status; S (upper and lower) and the strip into the padded work matrix (row stride np + 32)
//   per 64-row panel [jb, je):
//   8b diag      ONE wave factorises the 64 x 64 diagonal block in registers and inverts its four 16 x 16
//                diagonal sub-blocks (Y_b = U_bb^-1)
//   8c tails     U12 = U11^-T S12 as a blocked substitution ON THE MATRIX PIPE: one wave per 16-column strip,
//                four stages X_b = Y_b^T (S_b - sum_{b'<b} L_bb' X_b'), 40 MFMAs; an accumulator tile in the D layout
//                (row 4r + k, column e) is exactly the B operand of k-step r, so the stages chain in registers
//   8d update    S22 -= U12^T U12 on the matrix pipe: one wave per 32 x 32 block pair (2 x 2 MFMA tiles),
//                k = 64 rows in 16 MFMA steps -- a 64-row SYRK, the same operand trick as kernel 1; the strip is one
//                more block column
//   8d+8b fused  the launch that updates the trailing matrix of panel j also factorises the diagonal block of panel
//                j + 1 (workgroup 0: its three block pairs first, then kernel 8b's body in its first wave): look-ahead
//   8e backsolve U x = y from the bottom in macro-blocks of 256 rows: one workgroup solves the 256 x 256 diagonal block
//                (wave 0 runs the dependency chain, the other waves apply the previous panel's x inside the block),
//                then the whole chip applies the 256 new x to all rows above (a 256-column GEMV); beta = D x
// status[0]: bit 0 = non-positive / non-finite diagonal of G + alpha I, bit 1 = failed pivot;
// minpiv[p] = smallest pivot of panel p.
// ---------------------------------------------------------------------------------
constexpr int CHOL_NB = 64;
constexpr int CHOL_XS = 32;   // width of the right-hand-side strip (one block column of kernel 8d)

__global__ __launch_bounds__(256) void fsnap_chol_prepare_d_k(const double* __restrict__ packed,
                                                             const double* __restrict__ cvec, int n, int np,
                                                             double alpha, double* __restrict__ dsc,
                                                             double* __restrict__ z, int* __restrict__ status,
                                                             double* __restrict__ minpiv, int npanel) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < npanel) minpiv[i] = 1.0e300;
    if (i >= np) return;
    if (i >= n) {
        dsc[i] = 1.0;
        z[i] = 0.0;
        return;
    }
    const double g = packed[(size_t)i * n + i] + alpha;
    const double c = cvec[i];
    const bool ok = (g > 0.0) && __builtin_isfinite(g) && __builtin_isfinite(c);
    const double d = ok ? 1.0 / sqrt(g) : 0.0;
    dsc[i] = d;
    z[i] = ok ? c * d : 0.0;
    if (!ok) atomicOr(status, 1);
}

__global__ __launch_bounds__(256) void fsnap_chol_prepare_s_k(const double* __restrict__ packed, int n, int np,
                                                             double alpha, const double* __restrict__ dsc,
                                                             const double* __restrict__ z, double* __restrict__ S,
                                                             int* __restrict__ status) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    const int ld = np + CHOL_XS;
    if (j >= ld) return;
    double v;
    if (j >= np) {
        v = (j == np) ? z[i] : 0.0;
    } else if (i < n && j < n) {
        const double g = packed[(size_t)i * n + j] + ((i == j) ? alpha : 0.0);
        v = g * dsc[i] * dsc[j];
        // (not `v - v == 0`: with fp contraction that becomes fma(g d_i, d_j, -v), the rounding error of the product)
        if (!__builtin_isfinite(v)) atomicOr(status, 1);
    } else {
        v = (i == j) ? 1.0 : 0.0;
    }
    S[(size_t)i * ld + j] = v;
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(d) to full precision: v_rsq_f64 + three Newton steps (the IEEE sqrt + divide pair the compiler emits for
// sqrt(d), 1.0 / r is a ~60-instruction dependent chain, and this sits on the critical path of every column)
__device__ __forceinline__ double rsqrt_newton(double d) {
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const double e = __builtin_fma(-h * y, y, 0.5);
        y = __builtin_fma(y, e, y);
    }
    return y;
}

// the same with TWO Newton steps: v_rsq_f64 is good to ~2^-23 relative, one step squares that (x 1.5): 2^-45, 2^-90 -- the
// third step of rsqrt_newton changes nothing but the chain (tools/lat_bench: 60 -> 50 cycles per pivot)
__device__ __forceinline__ double rsqrt_newton2(double d) {
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double e = __builtin_fma(-h * y, y, 0.5);
        y = __builtin_fma(y, e, y);
    }
    return y;
}

// 8b: ONE wave factorises the 64 x 64 diagonal block as a 4 x 4 grid of 16 x 16 blocks held in registers in the
// MFMA accumulator layout (lane (k, e): rows 4r + k, column e of a block).  Per block step a:
//   * the 16 x 16 diagonal block is factorised WHERE IT IS: pivot j of the block sits in register j / 4 of the lanes
//     with k = j % 4, and so does the whole pivot row -- which is exactly where an MFMA reads the k-slot j % 4 of BOTH
//     operands.  The rank-1 update of a pivot is therefore ONE MFMA whose operands are the wave's own registers
//     (A: the scaled row for the columns right of the pivot, zero elsewhere; B: the scaled row; the other three
//     k-slots zero): no transpose through LDS, no broadcast of multipliers (the scalar recurrence it replaces took
//     2 v_readlane + 1 FMA per multiplier: 360 such groups per block).  Only the pivot itself is broadcast
//     (v_readlane), then v_rsq_f64 + Newton instead of IEEE sqrt / divide.
//   * the same row operations applied to an identity block give Z = U_aa^-T (forward substitution on I), one more
//     MFMA per pivot: the inverse Y_a = Z^T that kernels 8c and 8e need comes out of the same loop (it used to be a
//     second 16-step recurrence, a third of the kernel's time);
//   * U_ab = Z S_ab for the blocks right of it and S_bc -= U_ab^T U_ac for the blocks below: MFMAs whose operands
//     are the accumulator registers themselves (the layout of a D tile is the layout of the B operand of k-step r
//     and, transposed, of the A operand).
// History: scalar recurrence over all 64 columns 22 us per block; 16-column recurrences + MFMA 16.4 us; this form:
// see DESIGN 4.
// single-wave synchronisation of LDS traffic (kernel 8b runs in ONE wave, also when that wave is part of a larger
// workgroup -- the fused update + diagonal kernel -- where s_barrier would wait for waves that are not coming)
__device__ __forceinline__ void chol_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

template <int a, int V>
__device__ __forceinline__ void chol_diag_step(d4 (&B)[4][4], double (*T)[17], double (*UT)[16], double* __restrict__ S,
                                               int ld, int jb, double* __restrict__ Y, int e, int kr, double& pmin,
                                               double& psum) {
    (void)UT;
    d4& D = B[a][a];
    d4 Z;
#pragma unroll
    for (int r = 0; r < 4; ++r) Z[r] = (4 * r + kr == e) ? 1.0 : 0.0;
    // The pivot chain, three forms (V; FSNAP_CHOL_DIAG selects, A/B):
    //   0  pivot j read back from the rank-1 MFMA of step j - 1: MFMA -> v_readlane -> rsq + Newton -> scale -> MFMA;
    //   1  pivot j + 1 formed on the side from two entries of the block as step j found it,
    //      d_{j+1} = D[j+1][j+1] - (D[j][j+1] / sqrt(d_j))^2 (one multiply and one FMA behind 1/sqrt(d_j); the same value
    //      as the MFMA's own: one fused multiply-add on the same operands), so that the 16 passes of the MFMA leave the
    //      chain; instruction order left to the compiler;
    //   2  the same with the order fixed by hand: D's MFMA, the next pivot and the first Newton step, Z's MFMA (the matrix
    //      pipe takes one fp64 MFMA per 64 cycles: the second one waits in front of whatever follows it), the rest of Newton.
    if constexpr (V == 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int q = j >> 2, k = j & 3;
            const double d = readlane_f64(D[q], k * 16 + j);   // pivot (the same value in every lane)
            pmin = d < pmin ? d : pmin;                        // (a NaN pivot is caught by the sum)
            psum += d;
            const double inv = rsqrt_newton(d);
            const bool own = (kr == k);                        // the lanes that hold row j
            const double ud = D[q] * inv;                      // U[j][e] (e >= j; e == j: d / sqrt(d))
            const double zd = Z[q] * inv;                      // (U^-T)[j][e]
            const bool keep = own && e >= j;
            D[q] = keep ? ud : D[q];                           // the strictly lower part keeps its (finite) input values
            Z[q] = own ? zd : Z[q];
            if (j < 15) {
                const double aop = (own && e > j) ? -ud : 0.0;  // A[i][k] = -U[j][i] for the rows i > j
                const double bop = keep ? ud : 0.0;             // B[k][e] = U[j][e]
                const double zop = own ? zd : 0.0;
                D = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, D, 0, 0, 0);     // S[i][e] -= U[j][i] U[j][e]
                Z = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, zop, Z, 0, 0, 0);     // Z[i][e] -= U[j][i] Z[j][e]
            }
        }
    } else if constexpr (V == 1) {
        double dcur = readlane_f64(D[0], 0);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int q = j >> 2, k = j & 3;
            const double d = dcur;
            pmin = d < pmin ? d : pmin;
            psum += d;
            const double inv = rsqrt_newton(d);
            if (j < 15) {
                const int q1 = (j + 1) >> 2, k1 = (j + 1) & 3;
                const double t = readlane_f64(D[q], k * 16 + j + 1);        // D[j][j + 1] before this step's scaling
                const double pn = readlane_f64(D[q1], k1 * 16 + j + 1);     // D[j + 1][j + 1] before this step's update
                const double u = t * inv;
                dcur = __builtin_fma(-u, u, pn);
            }
            const bool own = (kr == k);
            const double ud = D[q] * inv;
            const double zd = Z[q] * inv;
            const bool keep = own && e >= j;
            D[q] = keep ? ud : D[q];
            Z[q] = own ? zd : Z[q];
            if (j < 15) {
                const double aop = (own && e > j) ? -ud : 0.0;
                const double bop = keep ? ud : 0.0;
                const double zop = own ? zd : 0.0;
                D = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, D, 0, 0, 0);
                Z = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, zop, Z, 0, 0, 0);
            }
        }
    } else {
        double dcur = readlane_f64(D[0], 0);
        double inv = rsqrt_newton(dcur);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int q = j >> 2, k = j & 3;
            const double d = dcur;
            pmin = d < pmin ? d : pmin;
            psum += d;
            double t = 0.0, pn = 0.0;
            if (j < 15) {
                const int q1 = (j + 1) >> 2, k1 = (j + 1) & 3;
                t = readlane_f64(D[q], k * 16 + j + 1);
                pn = readlane_f64(D[q1], k1 * 16 + j + 1);
            }
            const bool own = (kr == k);
            const double ud = D[q] * inv;
            const double zd = Z[q] * inv;
            const bool keep = own && e >= j;
            D[q] = keep ? ud : D[q];
            Z[q] = own ? zd : Z[q];
            if (j < 15) {
                const double aop = (own && e > j) ? -ud : 0.0;
                const double bop = keep ? ud : 0.0;
                const double zop = own ? zd : 0.0;
                D = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, D, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const double u = t * inv;
                dcur = __builtin_fma(-u, u, pn);
                double y = __builtin_amdgcn_rsq(dcur);
                const double h = 0.5 * dcur;
                {
                    const double e1 = __builtin_fma(-h * y, y, 0.5);
                    y = __builtin_fma(y, e1, y);
                }
                __builtin_amdgcn_sched_barrier(0);
                Z = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, zop, Z, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const double e1 = __builtin_fma(-h * y, y, 0.5);
                    y = __builtin_fma(y, e1, y);
                }
                inv = y;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (e >= 4 * r + kr) S[(size_t)(jb + 16 * a + 4 * r + kr) * ld + jb + 16 * a + e] = D[r];
    // Y_a = Z^T in the accumulator layout = the A operand of Z S_ab: through LDS
    chol_wave_sync();
#pragma unroll
    for (int r = 0; r < 4; ++r) T[4 * r + kr][e] = Z[r];
    chol_wave_sync();
    double yt[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) yt[s] = T[e][4 * s + kr];               // Z[e][4 s + kr] = Y_a[4 s + kr][e]
#pragma unroll
    for (int s = 0; s < 4; ++s) Y[(a * 16 + 4 * s + kr) * 16 + e] = yt[s];
    if constexpr (a < 3) {
#pragma unroll
        for (int b = a + 1; b < 4; ++b) {
            d4 u = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 4; ++s) u = __builtin_amdgcn_mfma_f64_16x16x4f64(yt[s], B[a][b][s], u, 0, 0, 0);
            B[a][b] = u;
        }
#pragma unroll
        for (int b = a + 1; b < 4; ++b)
#pragma unroll
            for (int c = b; c < 4; ++c)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    B[b][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(-B[a][b][s], B[a][c][s], B[b][c], 0, 0, 0);
    }
}

// the whole of kernel 8b for the panel at jb, executed by ONE wave (lane = threadIdx.x & 63); T: 16 x 17 doubles of LDS
template <int V>
__device__ __forceinline__ void chol_diag_body_v(double* __restrict__ S, int ld, int jb, double* __restrict__ Y,
                                               int* __restrict__ status, double* __restrict__ minpiv, double (*T)[17],
                                               double (*UT)[16], int lane) {
    const int e = lane & 15, kr = lane >> 4;
    d4 B[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = a; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) B[a][b][r] = S[(size_t)(jb + 16 * a + 4 * r + kr) * ld + jb + 16 * b + e];
    double pmin = 1.0e300, psum = 0.0;
    chol_diag_step<0, V>(B, T, UT, S, ld, jb, Y, e, kr, pmin, psum);
    chol_diag_step<1, V>(B, T, UT, S, ld, jb, Y, e, kr, pmin, psum);
    chol_diag_step<2, V>(B, T, UT, S, ld, jb, Y, e, kr, pmin, psum);
    chol_diag_step<3, V>(B, T, UT, S, ld, jb, Y, e, kr, pmin, psum);
    if (!(pmin > 0.0) || !__builtin_isfinite(psum)) {      // non-positive, NaN or infinite pivot
        if (lane == 0) atomicOr(status, 2);
        return;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = a + 1; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) S[(size_t)(jb + 16 * a + 4 * r + kr) * ld + jb + 16 * b + e] = B[a][b][r];
    if (lane == 0) minpiv[jb / CHOL_NB] = pmin;
}

// variant: wave-uniform (a kernel argument; FSNAP_CHOL_DIAG on the host side)
__device__ __forceinline__ void chol_diag_body(double* __restrict__ S, int ld, int jb, double* __restrict__ Y,
                                               int* __restrict__ status, double* __restrict__ minpiv, double (*T)[17],
                                               double (*UT)[16], int lane, int variant) {
    if (variant == 0) chol_diag_body_v<0>(S, ld, jb, Y, status, minpiv, T, UT, lane);
    else if (variant == 1) chol_diag_body_v<1>(S, ld, jb, Y, status, minpiv, T, UT, lane);
    else chol_diag_body_v<2>(S, ld, jb, Y, status, minpiv, T, UT, lane);
}

__global__ __launch_bounds__(64) void fsnap_chol_diag_k(double* __restrict__ S, int ld, int jb, double* __restrict__ Y,
                                                       int* __restrict__ status, double* __restrict__ minpiv, int variant,
                                                       int* __restrict__ flag) {
    __shared__ double T[16][17];
    __shared__ __attribute__((aligned(16))) double UT[16][16];
    if (threadIdx.x == 0 && flag) *flag = 0;          // hand-off word of the fused panel launches: cleared once per factorisation
    chol_diag_body(S, ld, jb, Y, status, minpiv, T, UT, (int)threadIdx.x, variant);
}

// ---------------------------------------------------------------------------------
// 8b4 (round 5): the 64 x 64 diagonal block on FOUR waves (one per SIMD of the CU), the default form.
//
// The single-wave form above spends 12.5 us per block: 64 pivots x (two fp64 MFMAs at 64 cycles each on ONE matrix pipe
// + ~25 VALU issues), and the 25 blocks of a K = 1595 solve are a serial chain.  Here wave w OWNS the 16-column strip w of
// the block -- the tiles T[0][w] ... T[w][w] in the accumulator layout -- and the block becomes a pipeline down the
// diagonal:
//   * block step a: wave a (the owner) factorises T[a][a] where it is, one rank-1 MFMA per pivot as before, but WITHOUT
//     the second MFMA on the inverse; per pivot it publishes the sixteen multipliers -U[j][i] (the A operand of that MFMA)
//     and 1/sqrt(d_j) in LDS (slot p = 16 a + j: two ds_write, no barrier);
//   * the waves c > a apply the same sixteen row operations to THEIR tile T[a][c] (scale row j, one MFMA with the
//     published A operand and their own scaled row as B operand): after pivot 15 the tile IS U_ac -- no inverse, no
//     triangular solve on the chain.  A consumer's chain is one dependent MFMA per pivot (65 cycles), shorter than the
//     owner's (pivot -> rsq + Newton -> scale -> MFMA), so it follows the owner one LDS round trip behind;
//   * Schur updates T[b][c] -= U_ab^T U_ac (b = a + 1 ... c) run in wave c: U_ac are its own registers (B operand), U_ab
//     for b < c comes from wave b through LDS lane for lane (an accumulator tile is, as it stands, the A operand of
//     U_ab^T X); the NEXT owner (c = a + 1) needs only its own registers: 4 MFMAs between the last pivot of block a and the
//     first pivot of block a + 1;
//   * the inverses Y_a = U_aa^-1 that kernels 8c / 8e need leave the chain: a wave that has nothing else to do replays the
//     published row operations on an identity tile (wave a for a < 3 once its block is done, wave 0 for block 3, following
//     wave 3's pivots as a consumer would).
// Hand-offs are polled in LDS (the owner never waits for anybody, so nothing can deadlock; every poll is bounded and a wave
// that gives up sets status bit 2: the host then factorises itself).  LDS executes the DS instructions of a wave in
// order: the writer stores payload then marker, the reader loads marker then payload in ONE round trip and retries when
// the marker was not there yet.
// ---------------------------------------------------------------------------------
// LDS hand-off accesses: `volatile` through a generic pointer compiles to FLAT loads / stores (the aperture path: several hundred
// cycles per access and both wait counters) -- the address space has to be spelled out for ds_read / ds_write
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) int lds_i32;
__device__ __forceinline__ double lds_ld(const double* p) { return *(const volatile lds_f64*)p; }
__device__ __forceinline__ void lds_st(double* p, double v) { *(volatile lds_f64*)p = v; }
__device__ __forceinline__ int lds_ld(const int* p) { return *(const volatile lds_i32*)p; }
__device__ __forceinline__ void lds_st(int* p, int v) { *(volatile lds_i32*)p = v; }

// A/B switches of kernel 8b4 (tools/chol_pipeline_check.hip builds the variants)
#ifndef FSNAP_D4_NEWTON
#define FSNAP_D4_NEWTON 2      // Newton steps behind v_rsq_f64 on the owner's chain
#endif
#ifndef FSNAP_D4_SLEEP
#define FSNAP_D4_SLEEP 1       // s_sleep between two looks at a marker that is not there yet
#endif
#ifndef FSNAP_D4_DEFER
#define FSNAP_D4_DEFER 1       // stores of the consumed tiles behind the owner phase
#endif
#ifndef FSNAP_D4_HALF
#define FSNAP_D4_HALF 1        // the next owner's diagonal update as two MFMA chains
#endif
#ifndef FSNAP_D4_RANK4
#define FSNAP_D4_RANK4 0       // 1: pivots in groups of four (the 4 x 4 block factorised on uniform values, TWO MFMAs per group).
                               // Built, parity-tested (tools/chol_pipeline_check -DFSNAP_D4_RANK4=1) and NOT faster: tools/lat_bench
                               // puts a group at 1087 cycles = 272 per pivot (ten v_readlane pairs 200, the 4 x 4 factorisation + W on
                               // uniform values 390, two MFMAs + masks 353, selects the rest) against 225 for the one-MFMA-per-pivot
                               // step; K = 1595 534 us either way (profiles/r05_chol_forms.txt)
#endif

#ifndef FSNAP_D4_REDUNDANT
#define FSNAP_D4_REDUNDANT 0   // 1: every wave right of a block step's owner factorises ITS OWN copy of the step's diagonal tile (handed over
                               //    once, before the first pivot) instead of replaying the owner's pivots one LDS hand-off at a time.  Bit-identical
                               //    results, no LDS traffic inside a block step -- and SLOWER: K = 1595 597 against 537 us, 256: 101 / 89, 480: 190 / 165
                               //    (tools/chol_pipeline_check, profiles/r05_chol_forms.txt).  A consumer's step on its private copy takes as long as
                               //    the owner's (~4 600-5 100 cycles for 16 pivots), and it now sits ON the critical path in front of the next owner's
                               //    step (+ the tile hand-over, ~6 750 cycles from one owner's start to the next) where the replay form overlaps all
                               //    but the last pivot with the owner (~6 100-6 600).  So the ~300 cycles per pivot in situ are the chain itself, not
                               //    the polling.  Kept as an A/B build switch.
#endif

#ifdef FSNAP_CHOL_TRACE
// tools/chol_diag4_trace.hip: shader-clock stamps of the four waves (entry, end of each consumer step, owner start / end, exit)
__device__ long long chol_trace_buf[4][8];
__device__ long long chol_trace_step[4][4];
__device__ long long chol_trace_last[4];         // when a consumer had block step A's LAST pivot in hand      // kernel 8s, workgroup 0: entry, strip substituted, tiles formed
#define CHOL_STAMP(w, i) do { if (lane == 0) chol_trace_buf[w][i] = (long long)__builtin_readcyclecounter(); } while (0)
#define CHOL_STAMP_STEP(w, i) do { if (lane == 0) chol_trace_step[w][i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define CHOL_STAMP(w, i) do { } while (0)
#define CHOL_STAMP_STEP(w, i) do { } while (0)
#endif

struct __attribute__((aligned(16))) Diag4Lds {
    double slot[64][16];       // multipliers of pivot p: -U[j][i] for i > j, 0 elsewhere
    double inv[64];            // 1/sqrt(d_p); 0.0 = not published yet
    double dummy[64];          // where the lanes that have nothing to publish store
    double gw[16][64];         // rank-4 form: per pivot group (block, q) the A operand of the group's row transformation ...
    double gu[16][64];         // ... and of its rank-4 update, lane for lane
    int gflag[16];             // group published
    double tile[3][4][64];     // U_01, U_02, U_12 handed from wave b to the waves right of it (accumulator layout, lane for lane)
    int tflag[4];              // [0] U_01, [1] U_02, [2] U_12 published
    double dtile[4][4][64];    // redundant form: the diagonal tile of block step a as its owner holds it BEFORE the first pivot (accumulator layout)
    int dflag[4];              // ... published
    double T[4][16][17];       // per-wave transpose scratch of the inverses
};

constexpr int CHOL_D4_SPINS = 1 << 17;        // polls of one hand-off before a wave gives up (~10 ms)

__device__ __forceinline__ int diag4_tile_index(int a, int b) { return a == 0 ? b - 1 : 2; }     // (0,1) (0,2) (1,2)

// clears the hand-off markers; every wave of the workgroup calls it, then ONE __syncthreads() before the first use
__device__ __forceinline__ void diag4_lds_reset(Diag4Lds& L, int tid) {
    if (tid < 64) L.inv[tid] = 0.0;
    if (tid < 4) L.tflag[tid] = 0;
    if (tid < 4) L.dflag[tid] = 0;
    if (tid < 16) L.gflag[tid] = 0;
}

// the sixteen multipliers and 1/sqrt(d) of pivot p, as published by the owner: one look (marker first, then payload)
__device__ __forceinline__ void diag4_peek(const Diag4Lds& L, int p, int e, double& inv, double& mult) {
    inv = lds_ld(&L.inv[p]);
    mult = lds_ld(&L.slot[p][e]);
}

// ... looked at again until the marker is there; *ok cleared when the wait ran out
__device__ __forceinline__ void diag4_wait(const Diag4Lds& L, int p, int e, double& inv, double& mult, bool& ok) {
    int n = 0;
    while (inv == 0.0 && ++n < CHOL_D4_SPINS) {
#if FSNAP_D4_SLEEP
        __builtin_amdgcn_s_sleep(1);                       // the owner shares the LDS queue with three pollers
#endif
        diag4_peek(L, p, e, inv, mult);
    }
    if (inv == 0.0) ok = false;
}

// row operations of block step A applied to the tile X = T[A][c] of a wave c > A: X becomes U_Ac.  The look at pivot j + 1
// is issued BEFORE the MFMA of pivot j, so that its LDS round trip runs beside the matrix pipe instead of behind it.
template <int A>
__device__ __forceinline__ void diag4_consume(d4& X, const Diag4Lds& L, int e, int kr, bool& ok) {
    double inv, mult;
    diag4_peek(L, 16 * A, e, inv, mult);
    diag4_wait(L, 16 * A, e, inv, mult, ok);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int q = j >> 2, k = j & 3;
        double inv1 = 1.0, mult1 = 0.0;
        if (j < 15) diag4_peek(L, 16 * A + j + 1, e, inv1, mult1);
        const bool own = (kr == k);
        const double xs = X[q] * inv;                      // row j of U_Ac
        X[q] = own ? xs : X[q];
        if (j < 15) {
            const double aop = own ? mult : 0.0;           // A[i][k] = -U[j][i], rows i > j
            const double bop = own ? xs : 0.0;             // B[k][e] = U_Ac[j][e]
            X = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, X, 0, 0, 0);
            diag4_wait(L, 16 * A + j + 1, e, inv1, mult1, ok);
            inv = inv1;
            mult = mult1;
#ifdef FSNAP_CHOL_TRACE
            if (j == 14 && kr == 0 && e == 0) chol_trace_last[A] = (long long)__builtin_readcyclecounter();   // (last writer wins: the slowest consumer)
#endif
        }
    }
}

// ---- rank-4 form (FSNAP_D4_RANK4) -------------------------------------------------------------------------------------
// Four pivots at a time.  Rows 4 q .. 4 q + 3 of a tile are register q of the four lane groups -- the four k-slots of ONE
// MFMA.  The owner reads the 4 x 4 diagonal block of the group (10 values, v_readlane), factorises it on uniform values (four
// rsq + Newton chains, a dozen FMAs), forms W = U4^-T, and then
//     rows' = W rows            one MFMA: A operand = W scattered to the lanes (i = e, k = kr), B operand = register q as it is;
//                               the result's register q ARE the four finished rows of U
//     D    -= U_cols^T U_rows   one MFMA, rank 4: A operand = -rows' masked to the columns right of the group
// Two MFMAs and one accumulator round trip per four pivots instead of four of each -- on paper; measured it is a wash (see the
// switch FSNAP_D4_RANK4 above), so the default stays one MFMA per pivot.  The consumers and the inverses replay a group with the two published A
// operands: 8 dependent MFMAs per block step instead of 16.
struct Rank4 {
    double aW, aU;
};

template <int Q>
__device__ __forceinline__ Rank4 diag4_group_factor(d4& D, int e, int kr, double& pmin, double& psum) {
    const double b00 = readlane_f64(D[Q], 4 * Q), b01 = readlane_f64(D[Q], 4 * Q + 1), b02 = readlane_f64(D[Q], 4 * Q + 2),
                 b03 = readlane_f64(D[Q], 4 * Q + 3);
    const double b11 = readlane_f64(D[Q], 16 + 4 * Q + 1), b12 = readlane_f64(D[Q], 16 + 4 * Q + 2),
                 b13 = readlane_f64(D[Q], 16 + 4 * Q + 3);
    const double b22 = readlane_f64(D[Q], 32 + 4 * Q + 2), b23 = readlane_f64(D[Q], 32 + 4 * Q + 3);
    const double b33 = readlane_f64(D[Q], 48 + 4 * Q + 3);
    const double i0 = rsqrt_newton2(b00);
    const double u01 = b01 * i0, u02 = b02 * i0, u03 = b03 * i0;
    const double d1 = __builtin_fma(-u01, u01, b11);
    const double i1 = rsqrt_newton2(d1);
    const double u12 = __builtin_fma(-u01, u02, b12) * i1, u13 = __builtin_fma(-u01, u03, b13) * i1;
    const double d2 = __builtin_fma(-u12, u12, __builtin_fma(-u02, u02, b22));
    const double i2 = rsqrt_newton2(d2);
    const double u23 = __builtin_fma(-u12, u13, __builtin_fma(-u02, u03, b23)) * i2;
    const double d3 = __builtin_fma(-u23, u23, __builtin_fma(-u13, u13, __builtin_fma(-u03, u03, b33)));
    const double i3 = rsqrt_newton2(d3);
    const double m01 = b00 < d1 ? b00 : d1, m23 = d2 < d3 ? d2 : d3, m = m01 < m23 ? m01 : m23;
    pmin = m < pmin ? m : pmin;                         // (a NaN pivot is caught by the sum)
    psum += (b00 + d1) + (d2 + d3);
    // W = U4^-T, row by row: W_k = i_k (e_k - sum_{m < k} u_mk W_m)
    const double w00 = i0;
    const double w10 = -i1 * u01 * w00, w11 = i1;
    const double w20 = i2 * (-u02 * w00 - u12 * w10), w21 = -i2 * u12 * w11, w22 = i2;
    const double w30 = i3 * (-u03 * w00 - u13 * w10 - u23 * w20), w31 = i3 * (-u13 * w11 - u23 * w21), w32 = -i3 * u23 * w22,
                 w33 = i3;
    // A operand of rows' = W rows: lane (k = kr, e) holds W[e - 4 Q][k] (lower triangular), zero outside the group's columns
    const int i = e - 4 * Q;
    const double c0 = i == 0 ? w00 : i == 1 ? w10 : i == 2 ? w20 : i == 3 ? w30 : 0.0;
    const double c1 = i == 1 ? w11 : i == 2 ? w21 : i == 3 ? w31 : 0.0;
    const double c2 = i == 2 ? w22 : i == 3 ? w32 : 0.0;
    const double c3 = i == 3 ? w33 : 0.0;
    Rank4 g;
    g.aW = kr == 0 ? c0 : kr == 1 ? c1 : kr == 2 ? c2 : c3;
    const d4 zero = {0.0, 0.0, 0.0, 0.0};
    const d4 Z0 = __builtin_amdgcn_mfma_f64_16x16x4f64(g.aW, D[Q], zero, 0, 0, 0);
    const double un = Z0[Q];                            // the four finished rows of U (lane group kr: row 4 Q + kr)
    D[Q] = un;
    g.aU = (e > 4 * Q + 3) ? -un : 0.0;
    return g;
}

__device__ __forceinline__ void diag4_group_peek(const Diag4Lds& L, int grp, int lane, int& flag, double& aW, double& aU) {
    flag = lds_ld(&L.gflag[grp]);                       // marker first, then payload: one round trip
    aW = lds_ld(&L.gw[grp][lane]);
    aU = lds_ld(&L.gu[grp][lane]);
}

__device__ __forceinline__ void diag4_group_wait(const Diag4Lds& L, int grp, int lane, int& flag, double& aW, double& aU, bool& ok) {
    int n = 0;
    while (flag == 0 && ++n < CHOL_D4_SPINS) {
#if FSNAP_D4_SLEEP
        __builtin_amdgcn_s_sleep(1);
#endif
        diag4_group_peek(L, grp, lane, flag, aW, aU);
    }
    if (flag == 0) ok = false;
}

// the four groups of block step A replayed on a tile X (a consumer's T[A][c], or the identity for the inverse)
template <int A>
__device__ __forceinline__ void diag4_replay(d4& X, const Diag4Lds& L, int lane, bool& ok) {
    int flag;
    double aW, aU;
    diag4_group_peek(L, 4 * A, lane, flag, aW, aU);
    diag4_group_wait(L, 4 * A, lane, flag, aW, aU, ok);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int flag1 = 1;
        double aW1 = 0.0, aU1 = 0.0;
        if (q < 3) diag4_group_peek(L, 4 * A + q + 1, lane, flag1, aW1, aU1);      // its LDS round trip runs beside the MFMAs
        const d4 zero = {0.0, 0.0, 0.0, 0.0};
        const d4 Z0 = __builtin_amdgcn_mfma_f64_16x16x4f64(aW, X[q], zero, 0, 0, 0);
        const double xn = Z0[q];
        X[q] = xn;
        if (q < 3) {
            X = __builtin_amdgcn_mfma_f64_16x16x4f64(aU, xn, X, 0, 0, 0);
            diag4_group_wait(L, 4 * A + q + 1, lane, flag1, aW1, aU1, ok);
            aW = aW1;
            aU = aU1;
        }
    }
}

// Y_A = U_AA^-1: the published row operations of block A replayed on an identity tile (Z = U_AA^-T), transposed through
// this wave's scratch, stored for kernels 8c / 8e
template <int A>
__device__ __forceinline__ void diag4_inverse(Diag4Lds& L, double (*T)[17], double* __restrict__ Y, int e, int kr, bool& ok) {
    d4 Z;
#pragma unroll
    for (int r = 0; r < 4; ++r) Z[r] = (4 * r + kr == e) ? 1.0 : 0.0;
#if FSNAP_D4_RANK4
    diag4_replay<A>(Z, L, 16 * kr + e, ok);
#else
    double inv, mult;
    diag4_peek(L, 16 * A, e, inv, mult);
    diag4_wait(L, 16 * A, e, inv, mult, ok);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int q = j >> 2, k = j & 3;
        double inv1 = 1.0, mult1 = 0.0;
        if (j < 15) diag4_peek(L, 16 * A + j + 1, e, inv1, mult1);
        const bool own = (kr == k);
        const double zd = Z[q] * inv;
        Z[q] = own ? zd : Z[q];
        if (j < 15) {
            const double aop = own ? mult : 0.0;
            const double zop = own ? zd : 0.0;
            Z = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, zop, Z, 0, 0, 0);
            diag4_wait(L, 16 * A + j + 1, e, inv1, mult1, ok);
            inv = inv1;
            mult = mult1;
        }
    }
#endif
    chol_wave_sync();
#pragma unroll
    for (int r = 0; r < 4; ++r) T[4 * r + kr][e] = Z[r];
    chol_wave_sync();
#pragma unroll
    for (int s = 0; s < 4; ++s) Y[(A * 16 + 4 * s + kr) * 16 + e] = T[e][4 * s + kr];      // Y_A[4 s + kr][e] = Z[e][4 s + kr]
}

// Wave W of the four: Tl[a] = T[a][W] (a <= W) on entry, already holding everything the panels left of this block
// contributed.  On exit the strip of U is in S, the inverses this wave is responsible for are in Y.
template <int W>
__device__ __forceinline__ void chol_diag4_wave(d4 (&Tl)[4], Diag4Lds& L, double* S, int ld, int jb,
                                                double* __restrict__ Y, int* __restrict__ status,
                                                double* __restrict__ minpiv, int lane) {
    const int e = lane & 15, kr = lane >> 4;
    bool ok = true;
    d4 held = {0.0, 0.0, 0.0, 0.0};
    CHOL_STAMP(W, 0);
    // ---- block steps left of the own one: this wave is a consumer -----------------------------------------------
#pragma unroll
    for (int a = 0; a < W; ++a) {
        d4& X = Tl[a];
#if FSNAP_D4_RANK4
        if (a == 0) diag4_replay<0>(X, L, lane, ok);
        else if (a == 1) diag4_replay<1>(X, L, lane, ok);
        else diag4_replay<2>(X, L, lane, ok);
#else
        if (a == 0) diag4_consume<0>(X, L, e, kr, ok);
        else if (a == 1) diag4_consume<1>(X, L, e, kr, ok);
        else diag4_consume<2>(X, L, e, kr, ok);
#endif
        if (a + 1 == W) {
            // next owner: its diagonal tile needs nothing but its own registers -- first thing after the last pivot, as two
            // chains of two MFMAs (a chain of four on one accumulator is 4 x 65 cycles on the hand-over)
#if FSNAP_D4_HALF
            d4 half = {0.0, 0.0, 0.0, 0.0};
            Tl[W] = __builtin_amdgcn_mfma_f64_16x16x4f64(-X[0], X[0], Tl[W], 0, 0, 0);
            half = __builtin_amdgcn_mfma_f64_16x16x4f64(-X[1], X[1], half, 0, 0, 0);
            Tl[W] = __builtin_amdgcn_mfma_f64_16x16x4f64(-X[2], X[2], Tl[W], 0, 0, 0);
            half = __builtin_amdgcn_mfma_f64_16x16x4f64(-X[3], X[3], half, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) Tl[W][r] += half[r];
#else
#pragma unroll
            for (int s = 0; s < 4; ++s) Tl[W] = __builtin_amdgcn_mfma_f64_16x16x4f64(-X[s], X[s], Tl[W], 0, 0, 0);
#endif
        }
        if (W < 3) {
            // U_aW for the waves right of this one (payload, then marker)
            const int ti = diag4_tile_index(a, W);
#pragma unroll
            for (int s = 0; s < 4; ++s) lds_st(&L.tile[ti][s][lane], X[s]);
            lds_st(&L.tflag[ti], 1);
        }
        if (a + 1 < W) {
            // T[b][W] -= U_ab^T U_aW, b = a + 1 ... W - 1 with U_ab from wave b, then b = W from the own registers
#pragma unroll
            for (int b = a + 1; b < W; ++b) {
                const int ti = diag4_tile_index(a, b);
                int n = 0;
                while (lds_ld(&L.tflag[ti]) == 0 && ++n < CHOL_D4_SPINS) {
                }
                if (n >= CHOL_D4_SPINS) ok = false;
                double at[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) at[s] = lds_ld(&L.tile[ti][s][lane]);
#pragma unroll
                for (int s = 0; s < 4; ++s) Tl[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(-at[s], X[s], Tl[b], 0, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) Tl[W] = __builtin_amdgcn_mfma_f64_16x16x4f64(-X[s], X[s], Tl[W], 0, 0, 0);
        }
        // the finished tile U_aW goes out at once -- except on the hand-over to the own block step (a + 1 == W), where the
        // stores wait behind the owner phase (FSNAP_D4_DEFER; `held` is a copy: see the note at the deferred store)
        if (!FSNAP_D4_DEFER || a + 1 < W) {
#pragma unroll
            for (int r = 0; r < 4; ++r) S[(size_t)(jb + 16 * a + 4 * r + kr) * ld + jb + 16 * W + e] = X[r];
        } else {
            held = X;
        }
        CHOL_STAMP(W, 1 + a);
    }
    CHOL_STAMP(W, 4);
    // ---- the own block step: owner ------------------------------------------------------------------------------
    d4& D = Tl[W];
    double pmin = 1.0e300, psum = 0.0;
#if FSNAP_D4_RANK4
    {
        auto group = [&](auto qc) {
            constexpr int Q = decltype(qc)::value;
            const Rank4 g = diag4_group_factor<Q>(D, e, kr, pmin, psum);
            // publish the two A operands of the group (payload, then marker), then the rank-4 update of the rows below
            lds_st(&L.gw[4 * W + Q][lane], g.aW);
            lds_st(&L.gu[4 * W + Q][lane], g.aU);
            lds_st(&L.gflag[4 * W + Q], 1);
            if constexpr (Q < 3) D = __builtin_amdgcn_mfma_f64_16x16x4f64(g.aU, D[Q], D, 0, 0, 0);
        };
        group(std::integral_constant<int, 0>{});
        group(std::integral_constant<int, 1>{});
        group(std::integral_constant<int, 2>{});
        group(std::integral_constant<int, 3>{});
    }
#else
    double dcur = readlane_f64(D[0], 0);
    double inv = FSNAP_D4_NEWTON == 2 ? rsqrt_newton2(dcur) : rsqrt_newton(dcur);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int q = j >> 2, k = j & 3;
        pmin = dcur < pmin ? dcur : pmin;                  // (a NaN pivot is caught by the sum)
        psum += dcur;
        double t = 0.0, pn = 0.0;
        if (j < 15) {
            const int q1 = (j + 1) >> 2, k1 = (j + 1) & 3;
            t = readlane_f64(D[q], k * 16 + j + 1);        // D[j][j + 1] before this step's scaling
            pn = readlane_f64(D[q1], k1 * 16 + j + 1);     // D[j + 1][j + 1] before this step's update
        }
        const bool own = (kr == k);
        const double ud = D[q] * inv;                      // U[j][e]
        const bool keep = own && e >= j;
        D[q] = keep ? ud : D[q];
        const double aop = (own && e > j) ? -ud : 0.0;
        // publish: the multipliers from the lanes that hold row j, then the marker (= 1/sqrt(d), never 0 for a finite pivot)
        lds_st(own ? &L.slot[16 * W + j][e] : &L.dummy[lane], aop);
        lds_st(&L.inv[16 * W + j], inv);
        if (j < 15) {
            const double bop = keep ? ud : 0.0;
            D = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, D, 0, 0, 0);
            const double u = t * inv;
            dcur = __builtin_fma(-u, u, pn);               // the MFMA's own value for D[j + 1][j + 1], one FMA behind 1/sqrt(d_j)
            inv = FSNAP_D4_NEWTON == 2 ? rsqrt_newton2(dcur) : rsqrt_newton(dcur);
        }
    }
#endif
    CHOL_STAMP(W, 5);
    // the strip of U: the tiles above the diagonal (kept in registers until here: their stores are off the hand-over), then
    // the diagonal tile's upper triangle
    // (storing Tl[a] itself here, for every a < W, produced stale tiles in the fused launches -- the values before the row
    // operations -- although the same code was right in the stand-alone kernel: tools/chol_pipeline_check.hip, round 5)
    if (FSNAP_D4_DEFER && W > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(size_t)(jb + 16 * (W - 1) + 4 * r + kr) * ld + jb + 16 * W + e] = held[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (e >= 4 * r + kr) S[(size_t)(jb + 16 * W + 4 * r + kr) * ld + jb + 16 * W + e] = D[r];
    if (!(pmin > 0.0) || !__builtin_isfinite(psum)) {      // non-positive, NaN or infinite pivot
        if (lane == 0) atomicOr(status, 2);
    } else if (lane == 0) {
        // positive doubles order like their bit patterns: the panel's smallest pivot over the four owners
        atomicMin(reinterpret_cast<unsigned long long*>(minpiv + jb / CHOL_NB), (unsigned long long)__double_as_longlong(pmin));
    }
    // ---- inverses -----------------------------------------------------------------------------------------------
    double(*T)[17] = L.T[W];
    if (W == 0) {
        diag4_inverse<0>(L, T, Y, e, kr, ok);
        diag4_inverse<3>(L, T, Y, e, kr, ok);              // follows wave 3's pivots
    } else if (W == 1) {
        diag4_inverse<1>(L, T, Y, e, kr, ok);
    } else if (W == 2) {
        diag4_inverse<2>(L, T, Y, e, kr, ok);
    }
    if (!ok && lane == 0) atomicOr(status, 2);
    CHOL_STAMP(W, 6);
}

// ---- redundant form (FSNAP_D4_REDUNDANT = 1; measured slower, see the switch) ----------------------------------------------------
// In the form above the owner of a block step publishes every pivot (16 multipliers + 1/sqrt(d)) through LDS and three consumers
// poll for it: 312 cycles per pivot in situ against 225 for the chain alone (tools/lat_bench), and a hand-over of ~1 150 cycles
// from one owner to the next.  Here the owner hands its diagonal tile over ONCE, before its first pivot (2 KB, payload then
// marker), and every wave right of it runs the owner's own chain on a private copy -- the same instructions on the same values,
// so every wave holds bit-identical multipliers -- with the row operations on its own tile T[a][W] as one more MFMA per pivot
// that issues inside the chain's VALU stretch (readlane, rsqrt + Newton, scaling: the matrix pipe is idle there).  No LDS
// traffic inside a block step, nobody polls while somebody computes, all four waves run the same code (instruction cache).
// The owner carries the identity tile in the same slot: Z = U_aa^-T comes out of its own pass (no replay afterwards).
//   X: the wave's tile of this block step (consumer) / the identity (owner); D: the diagonal tile, factorised in place
template <bool OWNER>
__device__ __forceinline__ void diag4r_step(d4& D, d4& X, int e, int kr, double& pmin, double& psum) {
    double dcur = readlane_f64(D[0], 0);
    double inv = FSNAP_D4_NEWTON == 2 ? rsqrt_newton2(dcur) : rsqrt_newton(dcur);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int q = j >> 2, k = j & 3;
        if (OWNER) {
            pmin = dcur < pmin ? dcur : pmin;              // (a NaN pivot is caught by the sum)
            psum += dcur;
        }
        double t = 0.0, pn = 0.0;
        if (j < 15) {
            const int q1 = (j + 1) >> 2, k1 = (j + 1) & 3;
            t = readlane_f64(D[q], k * 16 + j + 1);        // D[j][j + 1] before this step's scaling
            pn = readlane_f64(D[q1], k1 * 16 + j + 1);     // D[j + 1][j + 1] before this step's update
        }
        const bool own = (kr == k);
        const double ud = D[q] * inv;                      // U[j][e]
        const bool keep = own && e >= j;
        D[q] = keep ? ud : D[q];
        const double aop = (own && e > j) ? -ud : 0.0;     // A[i][k] = -U[j][i], rows i > j
        const double xs = X[q] * inv;                      // row j of U_aW (of Z)
        X[q] = own ? xs : X[q];
        if (j < 15) {
            const double bop = keep ? ud : 0.0;
            D = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, D, 0, 0, 0);
            X = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, own ? xs : 0.0, X, 0, 0, 0);
            const double u = t * inv;
            dcur = __builtin_fma(-u, u, pn);               // the MFMA's own value for D[j + 1][j + 1], one FMA behind 1/sqrt(d_j)
            inv = FSNAP_D4_NEWTON == 2 ? rsqrt_newton2(dcur) : rsqrt_newton(dcur);
        }
    }
}

template <int W>
__device__ __forceinline__ void chol_diag4r_wave(d4 (&Tl)[4], Diag4Lds& L, double* S, int ld, int jb,
                                                 double* __restrict__ Y, int* __restrict__ status,
                                                 double* __restrict__ minpiv, int lane) {
    const int e = lane & 15, kr = lane >> 4;
    bool ok = true;
    d4 held = {0.0, 0.0, 0.0, 0.0};
    double pmin = 1.0e300, psum = 0.0;
    CHOL_STAMP(W, 0);
    // ---- block steps left of the own one: the step's diagonal tile from its owner, then its pivots on the own tile ------------
#pragma unroll
    for (int a = 0; a < W; ++a) {
        d4& X = Tl[a];
        d4 Dc;
        {
            int n = 0;
            while (lds_ld(&L.dflag[a]) == 0 && ++n < CHOL_D4_SPINS) {
#if FSNAP_D4_SLEEP
                __builtin_amdgcn_s_sleep(1);
#endif
            }
            if (n >= CHOL_D4_SPINS) ok = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) Dc[r] = lds_ld(&L.dtile[a][r][lane]);
        }
        double unused0 = 0.0, unused1 = 0.0;
        diag4r_step<false>(Dc, X, e, kr, unused0, unused1);
        if (a + 1 == W) {
            // next owner: its diagonal tile needs nothing but its own registers -- first thing after the last pivot, as two
            // chains of two MFMAs, then straight to the waves right of it
            d4 half = {0.0, 0.0, 0.0, 0.0};
            Tl[W] = __builtin_amdgcn_mfma_f64_16x16x4f64(-X[0], X[0], Tl[W], 0, 0, 0);
            half = __builtin_amdgcn_mfma_f64_16x16x4f64(-X[1], X[1], half, 0, 0, 0);
            Tl[W] = __builtin_amdgcn_mfma_f64_16x16x4f64(-X[2], X[2], Tl[W], 0, 0, 0);
            half = __builtin_amdgcn_mfma_f64_16x16x4f64(-X[3], X[3], half, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) Tl[W][r] += half[r];
            if (W < 3) {
#pragma unroll
                for (int r = 0; r < 4; ++r) lds_st(&L.dtile[W][r][lane], Tl[W][r]);
                lds_st(&L.dflag[W], 1);
            }
        }
        if (W < 3) {
            // U_aW for the waves right of this one (payload, then marker)
            const int ti = diag4_tile_index(a, W);
#pragma unroll
            for (int s = 0; s < 4; ++s) lds_st(&L.tile[ti][s][lane], X[s]);
            lds_st(&L.tflag[ti], 1);
        }
        if (a + 1 < W) {
            // T[b][W] -= U_ab^T U_aW, b = a + 1 ... W - 1 with U_ab from wave b, then b = W from the own registers
#pragma unroll
            for (int b = a + 1; b < W; ++b) {
                const int ti = diag4_tile_index(a, b);
                int n = 0;
                while (lds_ld(&L.tflag[ti]) == 0 && ++n < CHOL_D4_SPINS) {
                }
                if (n >= CHOL_D4_SPINS) ok = false;
                double at[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) at[s] = lds_ld(&L.tile[ti][s][lane]);
#pragma unroll
                for (int s = 0; s < 4; ++s) Tl[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(-at[s], X[s], Tl[b], 0, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) Tl[W] = __builtin_amdgcn_mfma_f64_16x16x4f64(-X[s], X[s], Tl[W], 0, 0, 0);
        }
        // the finished tile U_aW goes out at once -- except on the hand-over to the own block step, where the stores wait behind
        // the owner phase (`held` is a copy: see the note in chol_diag4_wave)
        if (a + 1 < W) {
#pragma unroll
            for (int r = 0; r < 4; ++r) S[(size_t)(jb + 16 * a + 4 * r + kr) * ld + jb + 16 * W + e] = X[r];
        } else {
            held = X;
        }
        CHOL_STAMP(W, 1 + a);
    }
    CHOL_STAMP(W, 4);
    // ---- the own block step -------------------------------------------------------------------------------------------------------
    d4& D = Tl[W];
    if (W == 0 && W < 3) {
        // (the later owners published theirs right behind their diagonal update above)
#pragma unroll
        for (int r = 0; r < 4; ++r) lds_st(&L.dtile[0][r][lane], D[r]);
        lds_st(&L.dflag[0], 1);
    }
    d4 Z;
#pragma unroll
    for (int r = 0; r < 4; ++r) Z[r] = (4 * r + kr == e) ? 1.0 : 0.0;
    diag4r_step<true>(D, Z, e, kr, pmin, psum);
    CHOL_STAMP(W, 5);
    if (W > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(size_t)(jb + 16 * (W - 1) + 4 * r + kr) * ld + jb + 16 * W + e] = held[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (e >= 4 * r + kr) S[(size_t)(jb + 16 * W + 4 * r + kr) * ld + jb + 16 * W + e] = D[r];
    if (!(pmin > 0.0) || !__builtin_isfinite(psum)) {      // non-positive, NaN or infinite pivot
        if (lane == 0) atomicOr(status, 2);
    } else if (lane == 0) {
        atomicMin(reinterpret_cast<unsigned long long*>(minpiv + jb / CHOL_NB), (unsigned long long)__double_as_longlong(pmin));
    }
    // Y_W = U_WW^-1 = Z^T, through this wave's transpose scratch
    double(*T)[17] = L.T[W];
    chol_wave_sync();
#pragma unroll
    for (int r = 0; r < 4; ++r) T[4 * r + kr][e] = Z[r];
    chol_wave_sync();
#pragma unroll
    for (int s = 0; s < 4; ++s) Y[(W * 16 + 4 * s + kr) * 16 + e] = T[e][4 * s + kr];
    if (!ok && lane == 0) atomicOr(status, 2);
    CHOL_STAMP(W, 6);
}

// (S here is where the FACTOR goes: the work matrix itself in the in-place forms, the second matrix in the one-launch form)
__device__ __forceinline__ void chol_diag4_dispatch(d4 (&Tl)[4], Diag4Lds& L, double* S, int ld, int jb,
                                                    double* __restrict__ Y, int* __restrict__ status,
                                                    double* __restrict__ minpiv, int wave, int lane) {
#if FSNAP_D4_REDUNDANT
    if (wave == 0) chol_diag4r_wave<0>(Tl, L, S, ld, jb, Y, status, minpiv, lane);
    else if (wave == 1) chol_diag4r_wave<1>(Tl, L, S, ld, jb, Y, status, minpiv, lane);
    else if (wave == 2) chol_diag4r_wave<2>(Tl, L, S, ld, jb, Y, status, minpiv, lane);
    else chol_diag4r_wave<3>(Tl, L, S, ld, jb, Y, status, minpiv, lane);
    return;
#endif
    if (wave == 0) chol_diag4_wave<0>(Tl, L, S, ld, jb, Y, status, minpiv, lane);
    else if (wave == 1) chol_diag4_wave<1>(Tl, L, S, ld, jb, Y, status, minpiv, lane);
    else if (wave == 2) chol_diag4_wave<2>(Tl, L, S, ld, jb, Y, status, minpiv, lane);
    else chol_diag4_wave<3>(Tl, L, S, ld, jb, Y, status, minpiv, lane);
}

// the first diagonal block of a factorisation: tiles straight from the work matrix
__global__ __launch_bounds__(256) void fsnap_chol_diag4_k(const double* S, double* Uf, int ld, int jb, double* __restrict__ Y,
                                                         int* __restrict__ status, double* __restrict__ minpiv,
                                                         int* __restrict__ flag) {
    __shared__ Diag4Lds L;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, e = lane & 15, kr = lane >> 4;
    if (threadIdx.x == 0 && flag) *flag = 0;
    diag4_lds_reset(L, (int)threadIdx.x);
    d4 Tl[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            Tl[a][r] = (a <= wave) ? S[(size_t)(jb + 16 * a + 4 * r + kr) * ld + jb + 16 * wave + e] : 0.0;
    __syncthreads();
    chol_diag4_dispatch(Tl, L, Uf, ld, jb, Y, status, minpiv, wave, lane);
}

// 8a + 8b4 fused (round 5, one-launch-per-panel form): scaling of the whole matrix AND the first diagonal block in ONE launch.
// Workgroup 0 is the four-wave pipeline on the first 64 x 64 block, which it scales itself straight from the packed statistics
// (the long pole of the launch starts at once); workgroup b > 0 writes a 256-column piece of one row of the scaled work matrix
// (+ the right-hand-side strip), computing 1/sqrt(G_jj + alpha) of its columns on the fly from the diagonal instead of waiting for
// a launch that fills `dsc` -- the three launches prepare_d | prepare_s | diag of a solve (16-20 us of a K = 256 solve's 95) become one.
__global__ __launch_bounds__(256) void fsnap_chol_prepare_diag4_k(const double* __restrict__ packed, const double* __restrict__ cvec,
                                                                 int n, int np, double alpha, double* __restrict__ dsc,
                                                                 double* __restrict__ S, double* Uf, double* __restrict__ Y,
                                                                 int* __restrict__ status, double* __restrict__ minpiv,
                                                                 int npanel, int* __restrict__ flag) {
    __shared__ Diag4Lds L;
    __shared__ double d0[CHOL_NB];
    const int ld = np + CHOL_XS;
    const int nx = (ld + 255) / 256;
    auto scale_of = [&](int i, bool& ok) {
        if (i >= n) return 1.0;
        const double g = packed[(size_t)i * n + i] + alpha;
        ok = (g > 0.0) && __builtin_isfinite(g);
        return ok ? 1.0 / sqrt(g) : 0.0;
    };
    if (blockIdx.x == 0) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, e = lane & 15, kr = lane >> 4;
        if (threadIdx.x == 0 && flag) *flag = 0;
        if ((int)threadIdx.x < npanel) minpiv[threadIdx.x] = 1.0e300;
        for (int p = 256 + (int)threadIdx.x; p < npanel; p += 256) minpiv[p] = 1.0e300;
        diag4_lds_reset(L, (int)threadIdx.x);
        if (threadIdx.x < CHOL_NB) {
            bool ok = true;
            d0[threadIdx.x] = scale_of((int)threadIdx.x, ok);
        }
        __syncthreads();
        d4 Tl[4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * a + 4 * r + kr, j = 16 * wave + e;
                double v = 0.0;
                if (a <= wave) {
                    if (i < n && j < n) v = (packed[(size_t)i * n + j] + ((i == j) ? alpha : 0.0)) * d0[i] * d0[j];
                    else v = (i == j) ? 1.0 : 0.0;
                }
                Tl[a][r] = v;
            }
        chol_diag4_dispatch(Tl, L, Uf, ld, 0, Y, status, minpiv, wave, lane);
        return;
    }
    const int b = (int)blockIdx.x - 1;
    const int i = b / nx, j = (b % nx) * 256 + (int)threadIdx.x;
    if (j >= ld) return;
    bool oki = true;
    const double di = scale_of(i, oki);
    const double ci = (i < n) ? cvec[i] : 0.0;
    const bool finite_c = __builtin_isfinite(ci);
    if (j == 0) {
        dsc[i] = (i < n) ? (oki && finite_c ? di : 0.0) : 1.0;
        if (i < n && !(oki && finite_c)) atomicOr(status, 1);
    }
    double v;
    if (j >= np) {
        v = (j == np && i < n && oki && finite_c) ? ci * di : 0.0;
    } else if (i < n && j < n) {
        bool okj = true;
        const double dj = scale_of(j, okj);
        const double g = packed[(size_t)i * n + j] + ((i == j) ? alpha : 0.0);
        v = g * (oki && finite_c ? di : 0.0) * dj;
        if (!__builtin_isfinite(v)) atomicOr(status, 1);
    } else {
        v = (i == j) ? 1.0 : 0.0;
    }
    S[(size_t)i * ld + j] = v;
}

#ifdef FSNAP_CHOL_TRACE
// the block factorised TWICE in one launch (tools/chol_diag4_trace.hip): the stamps are those of the second pass, whose code
// is in the instruction cache -- every wave of kernel 8b4 otherwise runs ITS instantiation of the pipeline exactly once
__global__ __launch_bounds__(256) void fsnap_chol_diag4_twice_k(const double* S, double* Uf, int ld, int jb, double* __restrict__ Y,
                                                               int* __restrict__ status, double* __restrict__ minpiv) {
    __shared__ Diag4Lds L;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, e = lane & 15, kr = lane >> 4;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
        diag4_lds_reset(L, (int)threadIdx.x);
        d4 Tl[4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Tl[a][r] = (a <= wave) ? S[(size_t)(jb + 16 * a + 4 * r + kr) * ld + jb + 16 * wave + e] : 0.0;
        __syncthreads();
        chol_diag4_dispatch(Tl, L, Uf, ld, jb, Y, status, minpiv, wave, lane);
    }
}
#endif

// 8c: blocked forward substitution on the matrix pipe.  One wave per 16-column strip of the columns right of the
// panel (trailing columns + the right-hand-side strip).
__device__ __forceinline__ void chol_tails_strip(double* __restrict__ S, int ld, int jb, int strip,
                                                 const double* __restrict__ Y, int lane) {
    const int e = lane & 15, kr = lane >> 4;
    const int c0 = jb + CHOL_NB + 16 * strip;
    double* base = S + (size_t)(jb + kr) * ld;     // row jb + kr
    d4 X[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int s = 0; s < 4; ++s) X[b][s] = base[(size_t)(16 * b + 4 * s) * ld + c0 + e];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        d4 acc = X[b];
#pragma unroll
        for (int bp = 0; bp < b; ++bp)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double a = -base[(size_t)(16 * bp + 4 * s) * ld + jb + 16 * b + e];   // -L[16b+e][16bp+4s+kr]
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, X[bp][s], acc, 0, 0, 0);
            }
        d4 xb = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const double t = Y[(b * 16 + 4 * s + kr) * 16 + e];   // (Y_b^T)[e][4s+kr]
            xb = __builtin_amdgcn_mfma_f64_16x16x4f64(t, acc[s], xb, 0, 0, 0);
        }
        X[b] = xb;
    }
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) base[(size_t)(16 * b + 4 * r) * ld + c0 + e] = X[b][r];
}

__global__ __launch_bounds__(256) void fsnap_chol_tails_k(double* __restrict__ S, int ld, int jb, int nstrip,
                                                         const double* __restrict__ Y, const int* __restrict__ status) {
    if (*status) return;
    const int strip = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (strip >= nstrip) return;
    chol_tails_strip(S, ld, jb, strip, Y, (int)(threadIdx.x & 63));
}

// one 32 x 32 block pair of the trailing update (a 2 x 2 group of MFMA tiles), executed by one wave
__device__ __forceinline__ void chol_update_pair_ij(double* __restrict__ S, int ld, int jb, int I, int J, int lane);

__device__ __forceinline__ void chol_update_pair(double* __restrict__ S, int ld, int jb, int nblk, int pair, int lane) {
    const int ntri = nblk * (nblk + 1) / 2;
    // pair -> (I, J), I <= J, row-major packed triangle over nblk 32-column blocks; then (I, strip) for every I
    int I, J;
    if (pair < ntri) {
        I = 0;
        int rem = pair;
        while (rem >= nblk - I) {
            rem -= nblk - I;
            ++I;
        }
        J = I + rem;
    } else {
        I = pair - ntri;
        J = nblk;          // je + 32 nblk = np: the right-hand-side strip
    }
    chol_update_pair_ij(S, ld, jb, I, J, lane);
}

// block pair (I, J) of the trailing matrix behind panel jb (32-column blocks counted from the end of the panel; J = number of
// blocks: the right-hand-side strip)
__device__ __forceinline__ void chol_update_pair_ij(double* __restrict__ S, int ld, int jb, int I, int J, int lane) {
    const int e = lane & 15, kr = lane >> 4;
    const int je = jb + CHOL_NB;
    const int cI = je + 32 * I, cJ = je + 32 * J;
    const double* base = S + (size_t)(jb + kr) * ld;
    // every load of this wave is independent: issue the 64 operand values and the 16 values of the target tiles
    // together (one memory round trip; with a partially unrolled loop the wave paid four, plus one for the
    // read-modify-write at the end)
    double x0[16], x1[16], y0[16], y1[16];
#pragma unroll
    for (int s4 = 0; s4 < CHOL_NB / 4; ++s4) {
        const double* r = base + (size_t)(4 * s4) * ld;
        x0[s4] = r[cI + e];
        x1[s4] = r[cI + 16 + e];
        y0[s4] = r[cJ + e];
        y1[s4] = r[cJ + 16 + e];
    }
    d4 a00, a01, a10, a11;
    // D tile layout: element (row = kr + 4 r, col = e)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const int row = kr + 4 * r4;
        const double* p0 = S + (size_t)(cI + row) * ld;
        const double* p1 = S + (size_t)(cI + 16 + row) * ld;
        a00[r4] = p0[cJ + e];
        a01[r4] = p0[cJ + 16 + e];
        a10[r4] = p1[cJ + e];
        a11[r4] = p1[cJ + 16 + e];
    }
#pragma unroll
    for (int s4 = 0; s4 < CHOL_NB / 4; ++s4) {      // C - U12^T U12: the A operand is negated
        a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(-x0[s4], y0[s4], a00, 0, 0, 0);
        a01 = __builtin_amdgcn_mfma_f64_16x16x4f64(-x0[s4], y1[s4], a01, 0, 0, 0);
        a10 = __builtin_amdgcn_mfma_f64_16x16x4f64(-x1[s4], y0[s4], a10, 0, 0, 0);
        a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(-x1[s4], y1[s4], a11, 0, 0, 0);
    }
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const int row = kr + 4 * r4;
        double* p0 = S + (size_t)(cI + row) * ld;
        double* p1 = S + (size_t)(cI + 16 + row) * ld;
        p0[cJ + e] = a00[r4];
        p0[cJ + 16 + e] = a01[r4];
        p1[cJ + e] = a10[r4];
        p1[cJ + 16 + e] = a11[r4];
    }
}

// 8d + 8b fused ("look-ahead"): the trailing update of panel jb AND the factorisation of the NEXT diagonal block in one
// launch.  Workgroup 0 updates the three block pairs that make up the next 64 x 64 diagonal block -- (0,0), (0,1),
// (1,1) -- and its first wave then factorises that block (kernel 8b's body) while all other workgroups are still busy
// with the rest of the trailing matrix: the 16 us single-wave recurrence leaves the serial chain of the panel loop
// (was: diagonal 16 us -> tails 5 us -> update 7 us per panel, one after the other).
__global__ __launch_bounds__(256) void fsnap_chol_update_diag_k(double* __restrict__ S, int ld, int jb, int nblk,
                                                               int* __restrict__ status, double* __restrict__ Ynext,
                                                               double* __restrict__ minpiv, int variant) {
    __shared__ double T[16][17];
    __shared__ __attribute__((aligned(16))) double UT[16][16];
    if (*status) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x == 0) {
        if (wave < 3) chol_update_pair(S, ld, jb, nblk, wave == 2 ? nblk : wave, lane);
        __syncthreads();
        if (wave == 0) chol_diag_body(S, ld, jb + CHOL_NB, Ynext, status, minpiv, T, UT, lane, variant);
        return;
    }
    const int total = nblk * (nblk + 1) / 2 + nblk;
    const int q = ((int)blockIdx.x - 1) * 4 + wave;        // the remaining pairs: all but 0, 1 and nblk
    const int pair = q < nblk - 2 ? q + 2 : q + 3;
    if (pair >= total) return;
    chol_update_pair(S, ld, jb, nblk, pair, lane);
}

// 8d + 8b4 fused ("look-ahead"): workgroup 0 forms the tiles of the NEXT diagonal block in registers -- wave w the tiles
// (0..w, w): target values minus the 64-row product of the current panel's row tails -- and factorises them with the
// four-wave pipeline without a trip through memory; all other workgroups update the rest of the trailing matrix
__global__ __launch_bounds__(256) void fsnap_chol_update_diag4_k(double* __restrict__ S, int ld, int jb, int nblk,
                                                                int* __restrict__ status, double* __restrict__ Ynext,
                                                                double* __restrict__ minpiv) {
    __shared__ Diag4Lds L;
    if (*status) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x == 0) {
        diag4_lds_reset(L, (int)threadIdx.x);
        __syncthreads();
        const int e = lane & 15, kr = lane >> 4;
        const int nb = jb + CHOL_NB;
        const double* base = S + (size_t)(jb + kr) * ld + nb;          // row jb + kr of the panel, first column of the next block
        double y[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) y[s] = base[(size_t)(4 * s) * ld + 16 * wave + e];
        d4 Tl[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (a <= wave) {                                            // (wave-uniform)
                double x[16];
#pragma unroll
                for (int s = 0; s < 16; ++s) x[s] = (a == wave) ? y[s] : base[(size_t)(4 * s) * ld + 16 * a + e];
                d4 acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = S[(size_t)(nb + 16 * a + 4 * r + kr) * ld + nb + 16 * wave + e];
#pragma unroll
                for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-x[s], y[s], acc, 0, 0, 0);
                Tl[a] = acc;
            } else {
                Tl[a] = d4{0.0, 0.0, 0.0, 0.0};
            }
        }
        chol_diag4_dispatch(Tl, L, S, ld, nb, Ynext, status, minpiv, wave, lane);
        return;
    }
    const int total = nblk * (nblk + 1) / 2 + nblk;
    const int q = ((int)blockIdx.x - 1) * 4 + wave;        // the remaining pairs: all but 0, 1 and nblk
    const int pair = q < nblk - 2 ? q + 2 : q + 3;
    if (pair >= total) return;
    chol_update_pair(S, ld, jb, nblk, pair, lane);
}

// ---------------------------------------------------------------------------------
// 8s (round 5, the default): ONE launch per panel.  The launch behind panel jb computes the panel's row tails, updates the
// trailing matrix with them AND factorises the next diagonal block -- without a hand-off between workgroups, because every
// wave computes the row tails it needs ITSELF (the matrix pipe has the time: a K = 1595 solve is 17 us of MFMA work in a
// 0.5 ms chain).  The two-launch form paid per panel: tails launch 4.8 us + boundary + update / diagonal launch + boundary.
//   * the factor goes to a SECOND matrix Uf (same layout as the work matrix S): S keeps the Schur complements, so a wave
//     that reads the raw rows of the panel never races with the wave that writes their substituted form;
//   * a bulk wave owns one 32 x 32 block pair (I, J) as before: it substitutes the four 16-column strips of the panel's
//     rows in its block columns I and J (kernel 8c's blocked substitution, in registers: 4 x 40 MFMAs), multiplies them
//     (64 MFMAs: an accumulator tile is already the A / B operand) and updates its tiles of S in place; the waves of the
//     pairs (J, J), and the pair (nblk - 1, nblk) for the right-hand-side strip, also store their strips into Uf;
//   * workgroup 0 is the critical chain: wave w substitutes strip w of the NEXT diagonal block's columns, the four waves
//     exchange the strips through LDS (an accumulator tile is the A operand lane for lane), wave w forms the tiles
//     (0..w, w) of the next diagonal block in registers and the four-wave pipeline of kernel 8b4 factorises it.
// ---------------------------------------------------------------------------------
struct __attribute__((aligned(16))) Step4Lds {
    Diag4Lds d;
    double xs[3][16][64];      // substituted strips of waves 0-2 (4 tiles x 4 registers, lane for lane)
    int xflag[4];
};

// The operands of a panel's blocked substitution that do not depend on the strip: the off-diagonal tiles of the diagonal block's
// factor (as A operands, negated) and its inverted 16 x 16 diagonal blocks.  Loaded ONCE per wave, up front: fetched inside the
// substitution they cost a round trip to L2 / HBM per block step on the launch's critical chain (tools/chol_pipeline_check with
// -DFSNAP_CHOL_TRACE: workgroup 0 had its strip substituted 5 400 cycles into the launch, 2 600 of them MFMAs).
struct TailOps {
    double l[6][4];        // pairs (b, bp), bp < b: index b (b - 1) / 2 + bp; l[.][s] = -L[16 b + e][16 bp + 4 s + kr]
    double y[4][4];        // y[b][s] = (Y_b^T)[e][4 s + kr]
};

__device__ __forceinline__ void chol_tail_ops_load(const double* Uf, int ld, int jb, const double* __restrict__ Y, int lane,
                                                   TailOps& o) {
    const int e = lane & 15, kr = lane >> 4;
    const double* ubase = Uf + (size_t)(jb + kr) * ld;
#pragma unroll
    for (int b = 1; b < 4; ++b)
#pragma unroll
        for (int bp = 0; bp < b; ++bp)
#pragma unroll
            for (int s = 0; s < 4; ++s) o.l[b * (b - 1) / 2 + bp][s] = -ubase[(size_t)(16 * bp + 4 * s) * ld + jb + 16 * b + e];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int s = 0; s < 4; ++s) o.y[b][s] = Y[(b * 16 + 4 * s + kr) * 16 + e];
}

// raw rows of ONE 16-column strip (first column c0) of the panel at jb: X[b][s] = row 16 b + 4 s + kr, column e
__device__ __forceinline__ void chol_strip_load(const double* S, int ld, int jb, int c0, int lane, d4 (&X)[4]) {
    const int e = lane & 15, kr = lane >> 4;
    const double* base = S + (size_t)(jb + kr) * ld;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int s = 0; s < 4; ++s) X[b][s] = base[(size_t)(16 * b + 4 * s) * ld + c0 + e];
}

// the blocked substitution U12 = U11^-T S12 of a strip held in registers (kernel 8c's, 40 MFMAs): X in, row tails out
__device__ __forceinline__ void chol_tails_compute(d4 (&X)[4], const TailOps& o) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        d4 acc = X[b];
#pragma unroll
        for (int bp = 0; bp < b; ++bp)
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(o.l[b * (b - 1) / 2 + bp][s], X[bp][s], acc, 0, 0, 0);
        d4 xb = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) xb = __builtin_amdgcn_mfma_f64_16x16x4f64(o.y[b][s], acc[s], xb, 0, 0, 0);
        X[b] = xb;
    }
}

__device__ __forceinline__ void chol_strip_store(double* Uf, int ld, int jb, int c0, const d4 (&X)[4], int lane) {
    const int e = lane & 15, kr = lane >> 4;
    double* base = Uf + (size_t)(jb + kr) * ld;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) base[(size_t)(16 * b + 4 * r) * ld + c0 + e] = X[b][r];
}

__global__ __launch_bounds__(256, 2) void fsnap_chol_step4_k(double* S, double* Uf, int ld, int jb, int nblk,
                                                         int* __restrict__ status, const double* __restrict__ Y,
                                                         double* __restrict__ Ynext, double* __restrict__ minpiv) {
    __shared__ Step4Lds L;
    if (*status) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = lane & 15, kr = lane >> 4;
    const int je = jb + CHOL_NB;
    if (blockIdx.x == 0) {
        diag4_lds_reset(L.d, (int)threadIdx.x);
        if (threadIdx.x < 4) L.xflag[threadIdx.x] = 0;
        CHOL_STAMP_STEP(wave, 0);
        __syncthreads();
        // every load of the prologue goes out first: the strip's raw rows, the substitution's operands, the target tiles
        d4 X[4];
        TailOps ops;
        chol_strip_load(S, ld, jb, je + 16 * wave, lane, X);
        chol_tail_ops_load(Uf, ld, jb, Y, lane, ops);
        d4 Tl[4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Tl[a][r] = (a <= wave) ? S[(size_t)(je + 16 * a + 4 * r + kr) * ld + je + 16 * wave + e] : 0.0;
        chol_tails_compute(X, ops);
        CHOL_STAMP_STEP(wave, 1);
        if (wave < 3) {                                        // (wave-uniform) payload, then marker
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) lds_st(&L.xs[wave][4 * b + r][lane], X[b][r]);
            lds_st(&L.xflag[wave], 1);
        }
        bool ok = true;
        // own tile first (own registers: wave 0 starts its pivots from here), then the tiles above it with the strips of the
        // waves left of this one
#pragma unroll
        for (int a = 3; a >= 0; --a) {
            if (a <= wave) {                                    // (wave-uniform)
                d4 acc = Tl[a];
                if (a == wave) {
#pragma unroll
                    for (int b = 0; b < 4; ++b)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-X[b][r], X[b][r], acc, 0, 0, 0);
                } else {
                    int n = 0;
                    while (lds_ld(&L.xflag[a]) == 0 && ++n < CHOL_D4_SPINS) {
                    }
                    if (n >= CHOL_D4_SPINS) ok = false;
                    double xa[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) xa[q] = lds_ld(&L.xs[a][q][lane]);
#pragma unroll
                    for (int b = 0; b < 4; ++b)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa[4 * b + r], X[b][r], acc, 0, 0, 0);
                }
                Tl[a] = acc;
            }
        }
        chol_strip_store(Uf, ld, jb, je + 16 * wave, X, lane);  // (off the chain: behind the tiles)
        if (!ok && lane == 0) atomicOr(status, 2);
        CHOL_STAMP_STEP(wave, 2);
        chol_diag4_dispatch(Tl, L.d, Uf, ld, je, Ynext, status, minpiv, wave, lane);
        CHOL_STAMP_STEP(wave, 3);
        return;
    }
    // bulk: one 32 x 32 block pair per wave, all pairs but (0,0), (0,1), (1,1)
    const int total = nblk * (nblk + 1) / 2 + nblk;
    const int q = ((int)blockIdx.x - 1) * 4 + wave;
    const int pair = q < nblk - 2 ? q + 2 : q + 3;
    if (pair >= total) return;
    const int ntri = nblk * (nblk + 1) / 2;
    int I, J;
    if (pair < ntri) {
        I = 0;
        int rem = pair;
        while (rem >= nblk - I) {
            rem -= nblk - I;
            ++I;
        }
        J = I + rem;
    } else {
        I = pair - ntri;
        J = nblk;          // the right-hand-side strip
    }
    const int cI = je + 32 * I, cJ = je + 32 * J;
    // the two strips of block column J stay in registers; the strips of block column I are substituted one at a time (register
    // budget: two waves per SIMD)
    d4 XJ0[4], XJ1[4];
    TailOps ops;
    chol_strip_load(S, ld, jb, cJ, lane, XJ0);
    chol_strip_load(S, ld, jb, cJ + 16, lane, XJ1);
    chol_tail_ops_load(Uf, ld, jb, Y, lane, ops);
    chol_tails_compute(XJ0, ops);
    chol_tails_compute(XJ1, ops);
    if (I == J || (J == nblk && I == nblk - 1)) {              // this wave's J strips are the factor's rows: store them
        chol_strip_store(Uf, ld, jb, cJ, XJ0, lane);
        chol_strip_store(Uf, ld, jb, cJ + 16, XJ1, lane);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        d4 XI[4];
        if (I == J) {                                           // (wave-uniform)
#pragma unroll
            for (int b = 0; b < 4; ++b) XI[b] = h ? XJ1[b] : XJ0[b];
        } else {
            chol_strip_load(S, ld, jb, cI + 16 * h, lane, XI);
            chol_tails_compute(XI, ops);
        }
        d4 a0, a1;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const double* p = S + (size_t)(cI + 16 * h + kr + 4 * r4) * ld;
            a0[r4] = p[cJ + e];
            a1[r4] = p[cJ + 16 + e];
        }
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-XI[b][r], XJ0[b][r], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-XI[b][r], XJ1[b][r], a1, 0, 0, 0);
            }
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            double* p = S + (size_t)(cI + 16 * h + kr + 4 * r4) * ld;
            p[cJ + e] = a0[r4];
            p[cJ + 16 + e] = a1[r4];
        }
    }
}

// 8d + 8b + 8c fused ("one launch per panel", FSNAP_CHOL_FUSED = 1; NOT the default: it measured slower, see chol_fused()): the launch that updates the trailing
// matrix behind panel jb and factorises the next diagonal block ALSO runs the row tails of that next panel, so that a
// panel costs one launch instead of two (a launch that does almost nothing still holds the stream for 4-5 us on this part:
// 25 of them were ~110 of the 700 us of a K = 1595 solve).  Inside the launch the tails wait for the diagonal block behind
// a flag:
//   * workgroup 0, waves 0-2: the three block pairs of the next diagonal block; wave 0 then factorises it (kernel 8b's
//     body), releases its stores (agent scope) and publishes `gen` in *flag -- also when a pivot failed (status is set then
//     and nobody uses the result);
//   * one "column wave" per 32-column block J >= 2 of the trailing matrix (and one for the right-hand-side strip): it
//     updates the two block pairs (0, J), (1, J) -- the rows of the NEXT panel in its own columns, which nobody else
//     touches --, polls the flag (relaxed agent-scope loads with s_sleep, BOUNDED: a wave that gives up sets status bit 2
//     and the host falls back on its own factorisation; nothing can hang), acquires, and runs kernel 8c's substitution on
//     its two 16-column strips;
//   * the remaining waves: the block pairs (I, J) with I >= 2, as before.
// Column waves need workgroup 0 to be running while they wait: workgroups are dispatched in index order on this part and
// the grid of the shapes this path serves (K <= ~2000: <= 504 workgroups at two per CU) is resident as a whole; the bound
// on the wait is the answer to everything else.
constexpr int CHOL_FLAG_SPINS = 1 << 18;      // x s_sleep 16 (~1024 cycles): ~0.1 s

__device__ __forceinline__ void chol_publish(int* flag, int gen) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the compiler may drop the wait behind the write-back (guide, G16)
    __hip_atomic_store(flag, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ bool chol_wait_flag(const int* flag, int gen) {
    int n = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gen) {
        __builtin_amdgcn_s_sleep(16);
        if (++n > CHOL_FLAG_SPINS) return false;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return true;
}

__global__ __launch_bounds__(256) void fsnap_chol_panel_k(double* __restrict__ S, int ld, int jb, int nblk,
                                                         int* __restrict__ status, double* __restrict__ Ynext,
                                                         double* __restrict__ minpiv, int variant, int* __restrict__ flag,
                                                         int gen) {
    __shared__ double T[16][17];
    __shared__ __attribute__((aligned(16))) double UT[16][16];
    if (*status) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = (int)blockIdx.x * 4 + wave;                 // global wave index
    // waves 0-2: diagonal pairs (+ wave 0: the factorisation); waves 3 ... 3 + ncol - 1: column waves J = 2 ... nblk;
    // then the rest pairs
    const int ncol = nblk - 1;
    if (blockIdx.x == 0) {
        if (wave < 3) chol_update_pair_ij(S, ld, jb, wave == 2 ? 1 : 0, wave == 0 ? 0 : 1, lane);
        // (the barrier below is reached by all four waves of workgroup 0: wave 3, a column wave, does its pairs first)
        int J3 = 0;
        if (wave == 3 && ncol > 0) {
            J3 = 2;
            chol_update_pair_ij(S, ld, jb, 0, J3, lane);
            chol_update_pair_ij(S, ld, jb, 1, J3, lane);
        }
        __syncthreads();
        if (wave == 0) {
            chol_diag_body(S, ld, jb + CHOL_NB, Ynext, status, minpiv, T, UT, lane, variant);
            chol_publish(flag, gen);
        } else if (wave == 3 && ncol > 0) {
            if (!chol_wait_flag(flag, gen)) {
                if (lane == 0) atomicOr(status, 4);
                return;
            }
            if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
            chol_tails_strip(S, ld, jb + CHOL_NB, 2 * J3 - 4, Ynext, lane);
            chol_tails_strip(S, ld, jb + CHOL_NB, 2 * J3 - 3, Ynext, lane);
        }
        return;
    }
    const int cw = gw - 4;                                     // 0 ...: column waves J = 3 ..., then rest pairs
    if (cw < ncol - 1) {
        const int J = 3 + cw;                                  // J == nblk: the right-hand-side strip
        chol_update_pair_ij(S, ld, jb, 0, J, lane);
        chol_update_pair_ij(S, ld, jb, 1, J, lane);
        if (!chol_wait_flag(flag, gen)) {
            if (lane == 0) atomicOr(status, 4);
            return;
        }
        if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        chol_tails_strip(S, ld, jb + CHOL_NB, 2 * J - 4, Ynext, lane);
        chol_tails_strip(S, ld, jb + CHOL_NB, 2 * J - 3, Ynext, lane);
        return;
    }
    // rest: pairs (I, J), 2 <= I <= J < nblk (packed triangle over nblk - 2 blocks), then (I, nblk) for 2 <= I < nblk
    const int q = cw - (ncol > 1 ? ncol - 1 : 0);
    const int nb2 = nblk - 2;
    if (nb2 <= 0) return;
    const int ntri2 = nb2 * (nb2 + 1) / 2;
    int I, J;
    if (q < ntri2) {
        I = 0;
        int rem = q;
        while (rem >= nb2 - I) {
            rem -= nb2 - I;
            ++I;
        }
        J = I + rem + 2;
        I += 2;
    } else if (q < ntri2 + nb2) {
        I = q - ntri2 + 2;
        J = nblk;
    } else {
        return;
    }
    chol_update_pair_ij(S, ld, jb, I, J, lane);
}

// 8e: the strip's first column now holds y = U^-T z; solve U x = y from the bottom in MACRO-BLOCKS of four panels
// (256 rows).  Per macro-block two launches:
//   fsnap_chol_backsolve_k   ONE workgroup solves the 256 x 256 diagonal block panel by panel, the two halves of a panel
//     step running side by side (they used to alternate over ALL rows above: 13 us per panel, 326 us at K = 1595):
//       wave 0 -- the dependency chain: y_pb minus U[pb, pb + 1] x_{pb+1} (a 64 x 64 block staged in LDS), then the
//         64 x 64 triangular system as a blocked substitution with the inverted 16 x 16 diagonal blocks of kernel 8b
//         (7 small matrix-vector products, multipliers broadcast with v_readlane);
//       waves 1-15 -- one panel behind: y_r -= U[r, panel pb + 1] x_{pb+1} for the rows of the macro-block above panel pb
//         (64 contiguous doubles per row); the finished values of the rows of panel pb - 1 go to LDS for the next step.
//     The diagonal block, the block right of it and the inverses of the NEXT panel are fetched into registers meanwhile.
//   fsnap_chol_backupdate_k  the whole chip applies the 256 new x to every row above the macro-block (one wave per row,
//     a 256-column GEMV): one workgroup streaming those rows was bound by its own outstanding loads (~150 GB/s).
// Dynamic LDS: two buffers of [U11 64 x 65 | Uoff 64 x 65 | Y 4 x 16 x 17] + 2 x x_panel (64) + 2 x y_panel (64).
constexpr int CHOL_BS_BUF = 2 * CHOL_NB * (CHOL_NB + 1) + 4 * 16 * 17;
constexpr size_t CHOL_BS_LDS = (size_t)(2 * CHOL_BS_BUF + 4 * CHOL_NB) * sizeof(double);
constexpr int CHOL_BS_MACRO = 4;             // panels per macro-block (8 measured slower: K = 1595 0.569 against 0.543 ms, profiles/r05_chol_large_k_sweep.txt)

// y_r -= U[r, c0 : c1] x[c0 : c1] for the rows r < nrows; one wave per row, c1 - c0 a multiple of 128
__global__ __launch_bounds__(256) void fsnap_chol_backupdate_k(const double* __restrict__ S, int ld, double* zv, int c0, int c1,
                                                              int nrows, const int* __restrict__ status) {
    if (*status) return;
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const double* u = S + (size_t)r * ld;
    double a0 = 0.0, a1 = 0.0;
    for (int c = c0 + 2 * lane; c < c1; c += 128) {
        const d2 uv = *reinterpret_cast<const d2*>(u + c);        // 16-byte aligned: ld, c0 multiples of 32
        const d2 xv = *reinterpret_cast<const d2*>(zv + c);
        a0 = __builtin_fma(uv[0], xv[0], a0);
        a1 = __builtin_fma(uv[1], xv[1], a1);
    }
    double acc = a0 + a1;
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) acc += __shfl_xor(acc, sft, 64);
    if (lane == 0) zv[r] -= acc;
}

// first = 1 (the launch of the bottom macro-block): also copies the strip's first column y = U^-T z into zv (was a launch of
// its own).  host_out != null and p_lo == 0 (the last launch): beta, the panel pivots and the status word also go to
// page-locked host memory -- [beta n | min pivots np / 64 | status] -- and the status word is cleared for the next
// solve (was: a 4-byte memset launch in front of every solve and a D2H copy behind it).
__global__ __launch_bounds__(1024) void fsnap_chol_backsolve_k(const double* __restrict__ S, int ld, int np, int n,
                                                              const double* __restrict__ Yall, double* zv,
                                                              const double* __restrict__ dsc, double* __restrict__ beta,
                                                              int* status, int p_lo, int p_hi, int first,
                                                              const double* __restrict__ minpiv, double* host_out,
                                                              const double* __restrict__ Sraw) {
    extern __shared__ __attribute__((aligned(16))) double bs_lds[];
    const int st_in = *status;
    __syncthreads();                                   // every thread has read the status before thread 0 may clear it
    if (st_in) {
        if (p_lo == 0 && host_out && threadIdx.x == 0) {
            reinterpret_cast<int*>(host_out + n + np / CHOL_NB)[0] = st_in;
            *status = 0;
        }
        return;
    }
    if (first) {
        // Sraw != null (one-launch-per-panel form): the strip rows of the LAST panel were never substituted (there is no launch
        // behind the last panel) -- they come raw from the work matrix and wave 0 runs their forward substitution below
        for (int i = threadIdx.x; i < np; i += 1024)
            zv[i] = (Sraw && i >= np - CHOL_NB) ? Sraw[(size_t)i * ld + np] : S[(size_t)i * ld + np];
        __syncthreads();
    }
    double* xbuf = bs_lds + 2 * CHOL_BS_BUF;   // x of the panel solved last / being solved (two slots)
    double* ybuf = xbuf + 2 * CHOL_NB;         // y of the panel being solved / of the next one (two slots)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int row_lo = p_lo * CHOL_NB;         // first row of the macro-block
    if (tid < CHOL_NB) ybuf[tid] = zv[(p_hi - 1) * CHOL_NB + tid];      // slot 0: the macro-block's last panel
    // thread t stages elements t, t + 1024, ... of U11 and of the block right of it, and Y element t of a panel
    double pu[4], po[4], py;
    auto fetch = [&](int pb) {
        const int jb = pb * CHOL_NB;
        const bool off = pb + 1 < p_hi;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = tid + 1024 * q;
            const double* row = S + (size_t)(jb + (t >> 6)) * ld + jb + (t & 63);
            pu[q] = row[0];
            po[q] = off ? row[CHOL_NB] : 0.0;
        }
        py = Yall[(size_t)pb * 1024 + tid];
    };
    auto park = [&](int buf) {
        double* U11 = bs_lds + buf * CHOL_BS_BUF;
        double* Uoff = U11 + CHOL_NB * (CHOL_NB + 1);
        double* Ysm = Uoff + CHOL_NB * (CHOL_NB + 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = tid + 1024 * q;
            U11[(t >> 6) * (CHOL_NB + 1) + (t & 63)] = pu[q];
            Uoff[(t >> 6) * (CHOL_NB + 1) + (t & 63)] = po[q];
        }
        Ysm[(tid >> 4) * 17 + (tid & 15)] = py;      // (block, row) = tid >> 4, column = tid & 15
    };
    fetch(p_hi - 1);
    park(0);
    __syncthreads();
    if (first && Sraw) {
        // y_last = U11^-T z_last for the last panel: 64 steps, one wave, U11 from the LDS copy just parked (row k is read by
        // the lanes right of k: conflict-free), multipliers broadcast with v_readlane
        if (wv == 0) {
            const double* U11 = bs_lds;
            double v = ybuf[lane];
            const double invd = 1.0 / U11[lane * (CHOL_NB + 1) + lane];
#pragma unroll 8
            for (int k = 0; k < CHOL_NB; ++k) {
                const double yk = readlane_f64(v, k) * readlane_f64(invd, k);
                if (lane == k) v = yk;
                if (lane > k) v = __builtin_fma(-U11[k * (CHOL_NB + 1) + lane], yk, v);
            }
            ybuf[lane] = v;
        }
        __syncthreads();
    }
    for (int pb = p_hi - 1; pb >= p_lo; --pb) {
        const int jb = pb * CHOL_NB;
        const int cur = (p_hi - 1 - pb) & 1;            // LDS buffer / x slot / y slot of this panel
        const double* U11 = bs_lds + cur * CHOL_BS_BUF;
        const double* Uoff = U11 + CHOL_NB * (CHOL_NB + 1);
        const double* Ysm = Uoff + CHOL_NB * (CHOL_NB + 1);
        const double* xprev = xbuf + (cur ^ 1) * CHOL_NB;      // x of panel pb + 1
        const bool have_prev = pb + 1 < p_hi;
        if (pb > p_lo) fetch(pb - 1);
        if (wv == 0) {
            double v = ybuf[cur * CHOL_NB + lane];
            if (have_prev) {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
                for (int k = 0; k < CHOL_NB; k += 2) {
                    s0 = __builtin_fma(Uoff[lane * (CHOL_NB + 1) + k], xprev[k], s0);
                    s1 = __builtin_fma(Uoff[lane * (CHOL_NB + 1) + k + 1], xprev[k + 1], s1);
                }
                v -= s0 + s1;
            }
            const int blk = lane >> 4, i = lane & 15;
#pragma unroll
            for (int b = 3; b >= 0; --b) {
                // x_b = Y_b v_b (lanes of block b)
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < 16; ++k) acc = __builtin_fma(Ysm[(b * 16 + i) * 17 + k], readlane_f64(v, 16 * b + k), acc);
                if (blk == b) v = acc;
                // v_b' -= U_b'b x_b for the blocks above (rows < 16 b)
                if (b > 0) {
                    double sub = 0.0;
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        sub = __builtin_fma(U11[lane * (CHOL_NB + 1) + 16 * b + k], readlane_f64(v, 16 * b + k), sub);
                    if (blk < b) v -= sub;
                }
            }
            xbuf[cur * CHOL_NB + lane] = v;
            zv[jb + lane] = v;
        } else if (have_prev) {
            // rows of the macro-block above panel pb: y_r -= U[r, panel pb + 1] x_{pb+1}; the rows of panel pb - 1 are then
            // complete up to the contribution of x_pb, which wave 0 adds in the next step
            const int cb = jb + CHOL_NB;                       // first column of panel pb + 1
            // FOUR lanes per row, 16 columns each: all eight 16-byte loads of a lane in flight at once (one thread per row walked
            // its 64 columns in two dependent rounds of sixteen loads: the round trips, not the arithmetic, set a panel step)
            const int q4 = (tid - 64) & 3;
            for (int r = row_lo + ((tid - 64) >> 2); r < jb; r += 240) {
                const d2* u = reinterpret_cast<const d2*>(S + (size_t)r * ld + cb + 16 * q4);   // 16-byte aligned: ld, cb multiples of 32
                d2 uv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) uv[q] = u[q];
                const double z0 = zv[r];
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    a0 = __builtin_fma(uv[q][0], xprev[16 * q4 + 2 * q], a0);
                    a1 = __builtin_fma(uv[q][1], xprev[16 * q4 + 2 * q + 1], a1);
                }
                double acc = a0 + a1;
                acc += __shfl_xor(acc, 1, 64);                 // the four lanes of a row are neighbours (tid - 64 is a multiple of 4 apart)
                acc += __shfl_xor(acc, 2, 64);
                if (q4 == 0) {
                    const double zn = z0 - acc;
                    zv[r] = zn;
                    if (r >= jb - CHOL_NB) ybuf[(cur ^ 1) * CHOL_NB + r - (jb - CHOL_NB)] = zn;
                }
            }
        } else if (pb > p_lo) {
            // first step of the macro-block: nothing to apply yet; the rows of the next panel come as they are
            if (tid - 64 < CHOL_NB) ybuf[(cur ^ 1) * CHOL_NB + tid - 64] = zv[jb - CHOL_NB + (tid - 64)];
        }
        if (pb > p_lo) park(cur ^ 1);
        __syncthreads();
    }
    if (p_lo == 0) {
        for (int i = tid; i < n; i += 1024) {
            const double v = zv[i] * dsc[i];
            beta[i] = v;
            if (host_out) host_out[i] = v;
        }
        if (host_out) {
            for (int p = tid; p < np / CHOL_NB; p += 1024) host_out[n + p] = minpiv[p];
            if (tid == 0) reinterpret_cast<int*>(host_out + n + np / CHOL_NB)[0] = 0;
        }
    }
}
// ---------------------------------------------------------------------------------
// Kernel 8f: y = U^-T (D r) with the factor a solve left behind -- the forward sweep for ONE MORE right-hand side (the
// refinement steps of the SVD solver solve G delta = s with the G of the fit: the factorisation itself carries the first
// right-hand side along as a strip, a second one used to pay the whole factorisation again: 0.54 ms at K = 1595, 0.17 at 480).
// One workgroup: per 64-row panel the diagonal block goes through LDS, wave 0 runs its 64-step substitution (multipliers
// broadcast with v_readlane), then every thread takes columns right of the panel, z_k -= sum_i U[i][k] y_i (64 independent
// loads per column, coalesced over the threads).  The factor is read once: 10 MB at K = 1595 through one CU.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void fsnap_chol_forward_k(const double* __restrict__ Uf, int ld, int np, int n,
                                                            const double* __restrict__ rhs, const double* __restrict__ dsc,
                                                            double* __restrict__ zv, const int* __restrict__ status) {
    __shared__ double U11[CHOL_NB * (CHOL_NB + 1)];
    __shared__ double ysm[CHOL_NB];
    if (*status) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < np; i += 1024) zv[i] = (i < n) ? rhs[i] * dsc[i] : 0.0;
    __syncthreads();
    for (int jb = 0; jb < np; jb += CHOL_NB) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = tid + 1024 * q;
            U11[(t >> 6) * (CHOL_NB + 1) + (t & 63)] = Uf[(size_t)(jb + (t >> 6)) * ld + jb + (t & 63)];
        }
        if (tid < CHOL_NB) ysm[tid] = zv[jb + tid];
        __syncthreads();
        if (wv == 0) {
            double v = ysm[lane];
            const double invd = 1.0 / U11[lane * (CHOL_NB + 1) + lane];
#pragma unroll 8
            for (int k = 0; k < CHOL_NB; ++k) {
                const double yk = readlane_f64(v, k) * readlane_f64(invd, k);
                if (lane == k) v = yk;
                if (lane > k) v = __builtin_fma(-U11[k * (CHOL_NB + 1) + lane], yk, v);
            }
            ysm[lane] = v;
            zv[jb + lane] = v;
        }
        __syncthreads();
        for (int k = jb + CHOL_NB + tid; k < np; k += 1024) {
            const double* u = Uf + (size_t)jb * ld + k;
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll 4
            for (int i = 0; i < CHOL_NB; i += 4) {
                a0 = __builtin_fma(u[(size_t)i * ld], ysm[i], a0);
                a1 = __builtin_fma(u[(size_t)(i + 1) * ld], ysm[i + 1], a1);
                a2 = __builtin_fma(u[(size_t)(i + 2) * ld], ysm[i + 2], a2);
                a3 = __builtin_fma(u[(size_t)(i + 3) * ld], ysm[i + 3], a3);
            }
            zv[k] -= (a0 + a1) + (a2 + a3);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------
// Factor-only use of kernels 8b-8d: the pass factor of the row-space solve (fsnap_rowspace.cpp, shifted CholeskyQR)
//   R_p = chol(D^-1 G D^-1 + s I) D,   D = diag(sqrt(G_jj)),
// for K >= 384, where the host factorisation (one core: 12 ms at K = 1595, 22 ms with the scaling passes around it, twice
// per call) was the largest item of an ill-conditioned fit.  The reduced Gram matrix is already in HBM; the factor goes
// straight into the layout the row-space pass reads (kernel 13B: K16 x K16 padded R + the inverses of its 16 x 16 diagonal
// blocks, which are the Y blocks of kernel 8b scaled by 1 / D) and comes back to the host once, for the K x K end.
// Columns with G_jj <= 0 (zero columns of A_w) are inactive: unit row / column, as in fsnap_rs::factor_pass.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fsnap_chol_factor_prepare_d_k(const double* __restrict__ G, int n, int np,
                                                                    double* __restrict__ dsc, int* __restrict__ status,
                                                                    double* __restrict__ minpiv, int npanel) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < npanel) minpiv[i] = 1.0e300;
    if (i >= np) return;
    double d = 0.0;                                        // 0 marks an inactive / padding column
    if (i < n) {
        const double g = G[(size_t)i * n + i];
        if (!__builtin_isfinite(g)) atomicOr(status, 1);
        else if (g > 0.0) d = 1.0 / sqrt(g);
    }
    dsc[i] = d;
}

__global__ __launch_bounds__(256) void fsnap_chol_factor_prepare_s_k(const double* __restrict__ G, int n, int np, double shift,
                                                                    const double* __restrict__ dsc, double* __restrict__ S,
                                                                    int* __restrict__ status) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    const int ld = np + CHOL_XS;
    if (j >= ld) return;
    double v = 0.0;
    if (j < np) {
        const double di = dsc[i], dj = dsc[j];
        if (i < n && j < n && di > 0.0 && dj > 0.0) {
            // symmetrised like the host pass (the reduction mirrors the triangle exactly; defensive)
            const double g = 0.5 * (G[(size_t)i * n + j] + G[(size_t)j * n + i]);
            v = (i == j) ? 1.0 + shift : g * di * dj;
            if (!__builtin_isfinite(v)) atomicOr(status, 1);
        } else {
            v = (i == j) ? 1.0 : 0.0;
        }
    }
    S[(size_t)i * ld + j] = v;                             // (the right-hand-side strip stays zero: nothing to carry)
}

// One sweep over the Gram matrix on the device, a workgroup per row: out[a] = max_b |G_ab - delta_ab| and out[n + a] = sum_b of
// the squared Jacobi-scaled entries (unit diagonal), both over the columns with a positive diagonal entry; a NaN / Inf anywhere
// in the row makes out[a] NaN.  What the host needs to steer a pass (convergence test, shift) without the K x K matrix itself:
// the host sweep over 20 MB cost 3 ms per pass at K = 1595.  Fixed-order reductions.
__global__ __launch_bounds__(256) void fsnap_gram_scan_k(const double* __restrict__ G, int n, double* __restrict__ out) {
    __shared__ double smax[256], ssum[256];
    const int a = blockIdx.x, tid = threadIdx.x;
    const double ga = G[(size_t)a * n + a];
    const bool act_a = ga > 0.0;
    const double ia = act_a ? 1.0 / sqrt(ga) : 0.0;
    double dmax = 0.0, f = 0.0, bad = 0.0;
    for (int b = tid; b < n; b += 256) {
        const double g = G[(size_t)a * n + b];
        bad += g * 0.0;                                   // NaN for NaN / Inf
        const double gb = G[(size_t)b * n + b];
        if (act_a && gb > 0.0) {
            const double dv = fabs(g - (a == b ? 1.0 : 0.0));
            dmax = dv > dmax ? dv : dmax;
            const double sc = (a == b) ? 1.0 : g * ia * (1.0 / sqrt(gb));
            f += sc * sc;
        }
    }
    smax[tid] = (bad == 0.0) ? dmax : __builtin_nan("");
    ssum[tid] = f;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) {
            const double x = smax[tid], y = smax[tid + w];
            smax[tid] = (x != x || y != y) ? __builtin_nan("") : (x > y ? x : y);
            ssum[tid] += ssum[tid + w];
        }
        __syncthreads();
    }
    if (tid == 0) {
        out[a] = smax[0];
        out[n + a] = ssum[0];
    }
}

// R[i][j] = U[i][j] sqrt(G_jj) into the K16 x K16 layout of the row-space pass, then the inverse blocks
// T_J^-1 = diag(1 / sqrt(G_jj)) Y_J (Y_J = U_JJ^-1 from kernel 8b).  One thread per element.
__global__ __launch_bounds__(256) void fsnap_chol_extract_factor_k(const double* __restrict__ S, const double* __restrict__ Yall,
                                                                  const double* __restrict__ dsc, int n, int np, int K16,
                                                                  double* __restrict__ Rout) {
    const int ld = np + CHOL_XS;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t nR = (int64_t)K16 * K16;
    if (idx < nR) {
        const int i = (int)(idx / K16), j = (int)(idx % K16);
        double v = 0.0;
        if (j >= i) {
            const double dj = dsc[j];
            v = S[(size_t)i * ld + j] * (dj > 0.0 ? 1.0 / dj : 1.0);
        }
        Rout[idx] = v;
    } else if (idx < nR + (int64_t)K16 * 16) {
        const int64_t q = idx - nR;
        const int jb = (int)(q >> 8), k = (int)((q >> 4) & 15), c = (int)(q & 15);
        const int row = jb * 16 + k;
        const double dk = dsc[row];
        const double y = Yall[(size_t)(row / CHOL_NB) * 1024 + (size_t)((row % CHOL_NB) / 16) * 256 + k * 16 + c];
        Rout[idx] = (dk > 0.0 ? dk : 1.0) * y;
    }
}

// ---------------------------------------------------------------------------------
// host-side launchers (C++ linkage, used by fsnap_capi.cpp)
// ---------------------------------------------------------------------------------
namespace fsnap {

size_t chol_large_work_doubles(int n) {
    const size_t np = (size_t)(n + CHOL_NB - 1) / CHOL_NB * CHOL_NB;
    // work matrix + strip, Y blocks of every panel, hand-off word, the factor matrix of the one-launch-per-panel form
    return 2 * np * (np + CHOL_XS) + (np / CHOL_NB) * 1024 + 8;
}

// where the factor ends up: the work matrix itself (in-place forms), or the second matrix behind Y blocks and hand-off word
static double* chol_factor_matrix(double* work, int np, int form);

// form of the panel loop, FSNAP_CHOL_DIAG =
//   5 (default)  ONE launch per panel (kernel 8s: every wave substitutes the row tails it needs itself; factor in a second
//                matrix), diagonal block on four waves (kernel 8b4);
//   4            two launches per panel (tails; update + next diagonal block), diagonal block on four waves;
//   0 | 1 | 2    two launches per panel, the single-wave kernel 8b with the pivot chains of chol_diag_step (rounds 2-4; what
//                the flag-synchronised panel launch FSNAP_CHOL_FUSED = 1 uses)
static int chol_diag_variant() {
    static const int v = [] {
        const char* e = getenv("FSNAP_CHOL_DIAG");
        const int x = e ? atoi(e) : 5;
        return (x < 0 || x > 5 || x == 3) ? 5 : x;
    }();
    return v;
}

int chol_default_form() { return chol_diag_variant(); }

static int chol_resolve_form(int form) { return (form < 0 || form > 5 || form == 3) ? chol_diag_variant() : form; }

static void launch_first_diag(double* S, double* Uf, int ld, double* Yall, int* status, double* minpiv, int* flag, int form,
                              hipStream_t st) {
    if (form >= 4)
        hipLaunchKernelGGL(fsnap_chol_diag4_k, dim3(1), dim3(256), 0, st, (const double*)S, Uf, ld, 0, Yall, status, minpiv, flag);
    else
        hipLaunchKernelGGL(fsnap_chol_diag_k, dim3(1), dim3(64), 0, st, S, ld, 0, Yall, status, minpiv, form, flag);
}

// FSNAP_CHOL_FUSED = 1: one launch per panel (kernel fsnap_chol_panel_k); default 0: two launches per panel (tails; update +
// next diagonal block), the form of rounds 2-3 -- which measured FASTER: K = 1595 0.692 ms against 0.840 ms fused, K = 480
// 0.222 / 0.261 (profiles/r04_chol_fused_ab.txt): the column waves of the fused launch run their two strips one after the
// other behind the flag (the tails launch gives every strip a wave of its own), ~50 polling waves sit beside the single wave
// that factorises the diagonal block, and the hand-off costs a release + an acquire per panel -- together more than the
// ~4.5 us launch they replace.
static bool chol_fused() {
    static const bool v = [] {
        const char* e = getenv("FSNAP_CHOL_FUSED");
        return e && atoi(e) != 0;
    }();
    return v;
}

static double* chol_factor_matrix(double* work, int np, int form) {
    if (form != 5 || chol_fused()) return work;
    const size_t ld = (size_t)np + CHOL_XS;
    return work + (size_t)np * ld + (size_t)(np / CHOL_NB) * 1024 + 8;
}

// the panel loop behind the first diagonal block (kernel 8b on panel 0 has run): row tails of panel 0, then per panel ONE
// launch (update behind panel pb + diagonal block and row tails of panel pb + 1) -- or the two-launch form
static void launch_chol_panels(double* S, double* Uf, int ld, int np, double* Yall, int* status, double* minpiv, int* flag,
                               int form, hipStream_t st) {
    const int npanel = np / CHOL_NB;
    const bool fused = chol_fused();
    const int form1 = form >= 4 ? 2 : form;                 // the flag-synchronised panel launch keeps the single-wave block
    for (int pb = 0; pb < npanel; ++pb) {
        const int jb = pb * CHOL_NB;
        double* Y = Yall + (size_t)pb * 1024;
        const int ntail = np - jb - CHOL_NB;
        const int nstrip = (ntail + CHOL_XS) / 16;
        const int nblk = ntail / 32;
        if (form == 5 && !fused) {
            // ONE launch per panel: row tails (every wave its own), trailing update, next diagonal block; S -> Uf
            if (ntail > 0) {
                const int nrest = nblk * (nblk + 1) / 2 + nblk - 3;
                hipLaunchKernelGGL(fsnap_chol_step4_k, dim3(1 + (nrest + 3) / 4), dim3(256), 0, st, S, Uf, ld, jb, nblk, status,
                                   (const double*)Y, Y + 1024, minpiv);
            }
            // (behind the LAST panel there is nothing to launch: the forward substitution of its right-hand-side rows is the
            // first thing the back substitution does, fsnap_chol_backsolve_k with Sraw; the factor-only use has no strip)
            continue;
        }
        if (pb == 0 || !fused)
            hipLaunchKernelGGL(fsnap_chol_tails_k, dim3((nstrip + 3) / 4), dim3(256), 0, st, S, ld, jb, nstrip, Y, status);
        if (ntail > 0) {
            if (fused) {
                // waves: 3 diagonal pairs + (nblk - 1) column waves + the pairs with I >= 2
                const int nb2 = nblk - 2;
                const int nwaves = 4 + (nblk - 2 > 0 ? nblk - 2 : 0) + (nb2 > 0 ? nb2 * (nb2 + 1) / 2 + nb2 : 0);
                hipLaunchKernelGGL(fsnap_chol_panel_k, dim3((nwaves + 3) / 4), dim3(256), 0, st, S, ld, jb, nblk, status, Y + 1024,
                                   minpiv, form1, flag, pb + 1);
            } else {
                // trailing update of this panel + factorisation of the next diagonal block (look-ahead), one launch
                const int nrest = nblk * (nblk + 1) / 2 + nblk - 3;
                if (form == 4)
                    hipLaunchKernelGGL(fsnap_chol_update_diag4_k, dim3(1 + (nrest + 3) / 4), dim3(256), 0, st, S, ld, jb, nblk,
                                       status, Y + 1024, minpiv);
                else
                    hipLaunchKernelGGL(fsnap_chol_update_diag_k, dim3(1 + (nrest + 3) / 4), dim3(256), 0, st, S, ld, jb, nblk,
                                       status, Y + 1024, minpiv, form);
            }
        }
    }
}

hipError_t launch_chol_large(const double* packed, const double* cvec, int n, double alpha, double* work, double* dsc, double* z,
                             double* beta, int* status, double* minpiv, double* host_out, bool clear_status, int form,
                             hipStream_t st) {
    form = chol_resolve_form(form);
    if (!cvec) cvec = packed + (size_t)n * n;
    const int np = (n + CHOL_NB - 1) / CHOL_NB * CHOL_NB, npanel = np / CHOL_NB, ld = np + CHOL_XS;
    double* S = work;
    double* Yall = work + (size_t)np * ld;
    int* flag = (int*)(Yall + (size_t)npanel * 1024);      // hand-off word of the fused panel launches
    hipError_t e;
    // the status word is cleared by the last launch of the previous solve (host_out path); a launch of its own only the
    // first time this buffer is used, or when the results still travel by D2H copy
    if (clear_status || !host_out) {
        e = hipMemsetAsync(status, 0, sizeof(int), st);
        if (e != hipSuccess) return e;
    }
    double* Uf = chol_factor_matrix(work, np, form);       // where the factor ends up (S itself in the in-place forms)
    const bool one_launch = form == 5 && Uf != S;
    if (one_launch) {
        // scaling + first diagonal block in one launch (workgroup 0: the four-wave pipeline)
        const int nx = (ld + 255) / 256;
        hipLaunchKernelGGL(fsnap_chol_prepare_diag4_k, dim3((unsigned)(1 + np * nx)), dim3(256), 0, st, packed, cvec, n, np, alpha, dsc,
                           S, Uf, Yall, status, minpiv, npanel, flag);
    } else {
        hipLaunchKernelGGL(fsnap_chol_prepare_d_k, dim3((np + 255) / 256), dim3(256), 0, st, packed, cvec, n, np, alpha, dsc, z,
                           status, minpiv, npanel);
        hipLaunchKernelGGL(fsnap_chol_prepare_s_k, dim3((ld + 255) / 256, np), dim3(256), 0, st, packed, n, np, alpha, dsc, z, S,
                           status);
        launch_first_diag(S, Uf, ld, Yall, status, minpiv, flag, form, st);
    }
    launch_chol_panels(S, Uf, ld, np, Yall, status, minpiv, flag, form, st);
    static bool bs_attr_set = false;
    if (!bs_attr_set) {
        e = hipFuncSetAttribute((const void*)fsnap_chol_backsolve_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CHOL_BS_LDS);
        if (e != hipSuccess) return e;
        bs_attr_set = true;
    }
    for (int hi = npanel; hi > 0; hi -= CHOL_BS_MACRO) {
        const int lo = hi > CHOL_BS_MACRO ? hi - CHOL_BS_MACRO : 0;
        hipLaunchKernelGGL(fsnap_chol_backsolve_k, dim3(1), dim3(1024), CHOL_BS_LDS, st, (const double*)Uf, ld, np, n, Yall, z, dsc,
                           beta, status, lo, hi, hi == npanel ? 1 : 0, minpiv, host_out, one_launch ? (const double*)S : (const double*)nullptr);
        if (lo > 0) {
            const int nrows = lo * CHOL_NB;
            hipLaunchKernelGGL(fsnap_chol_backupdate_k, dim3((nrows + 3) / 4), dim3(256), 0, st, (const double*)Uf, ld, z, lo * CHOL_NB,
                               hi * CHOL_NB, nrows, status);
        }
    }
    return hipGetLastError();
}

// One more right-hand side for the factor the last launch_chol_large(n, form) left in `work` (same work, dsc, z, minpiv buffers):
// kernel 8f, then the backward sweep as in launch_chol_large.  d_rhs: n doubles in device memory (unscaled).
hipError_t launch_chol_resolve(const double* d_rhs, int n, double* work, const double* dsc, double* z, double* beta, int* status,
                               const double* minpiv, double* host_out, int form, hipStream_t st) {
    form = chol_resolve_form(form);
    const int np = (n + CHOL_NB - 1) / CHOL_NB * CHOL_NB, npanel = np / CHOL_NB, ld = np + CHOL_XS;
    double* Yall = work + (size_t)np * ld;
    const double* Uf = chol_factor_matrix(work, np, form);
    hipError_t e;
    if (!host_out) {
        e = hipMemsetAsync(status, 0, sizeof(int), st);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(fsnap_chol_forward_k, dim3(1), dim3(1024), 0, st, Uf, ld, np, n, d_rhs, dsc, z, (const int*)status);
    static bool bs_attr_set = false;
    if (!bs_attr_set) {
        e = hipFuncSetAttribute((const void*)fsnap_chol_backsolve_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CHOL_BS_LDS);
        if (e != hipSuccess) return e;
        bs_attr_set = true;
    }
    for (int hi = npanel; hi > 0; hi -= CHOL_BS_MACRO) {
        const int lo = hi > CHOL_BS_MACRO ? hi - CHOL_BS_MACRO : 0;
        hipLaunchKernelGGL(fsnap_chol_backsolve_k, dim3(1), dim3(1024), CHOL_BS_LDS, st, Uf, ld, np, n, (const double*)Yall, z, dsc, beta,
                           status, lo, hi, 0, minpiv, host_out, (const double*)nullptr);
        if (lo > 0) {
            const int nrows = lo * CHOL_NB;
            hipLaunchKernelGGL(fsnap_chol_backupdate_k, dim3((nrows + 3) / 4), dim3(256), 0, st, Uf, ld, z, lo * CHOL_NB, hi * CHOL_NB,
                               nrows, (const int*)status);
        }
    }
    return hipGetLastError();
}

hipError_t launch_chol_factor(const double* G, int n, double shift, double* work, double* dsc, int* status, double* minpiv,
                              int K16, double* Rout, int form, hipStream_t st) {
    form = chol_resolve_form(form);
    const int np = (n + CHOL_NB - 1) / CHOL_NB * CHOL_NB, npanel = np / CHOL_NB, ld = np + CHOL_XS;
    double* S = work;
    double* Yall = work + (size_t)np * ld;
    int* flag = (int*)(Yall + (size_t)npanel * 1024);      // hand-off word of the fused panel launches
    hipError_t e = hipMemsetAsync(status, 0, sizeof(int), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fsnap_chol_factor_prepare_d_k, dim3((np + 255) / 256), dim3(256), 0, st, G, n, np, dsc, status, minpiv, npanel);
    hipLaunchKernelGGL(fsnap_chol_factor_prepare_s_k, dim3((ld + 255) / 256, np), dim3(256), 0, st, G, n, np, shift, dsc, S, status);
    double* Uf = chol_factor_matrix(work, np, form);
    launch_first_diag(S, Uf, ld, Yall, status, minpiv, flag, form, st);
    launch_chol_panels(S, Uf, ld, np, Yall, status, minpiv, flag, form, st);  // (the strip is carried along as in the solve: zero here)
    const int64_t total = (int64_t)K16 * K16 + (int64_t)K16 * 16;
    hipLaunchKernelGGL(fsnap_chol_extract_factor_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const double*)Uf, Yall,
                       dsc, n, np, K16, Rout);
    return hipGetLastError();
}

hipError_t launch_gram_scan(const double* G, int n, double* out, hipStream_t st) {
    hipLaunchKernelGGL(fsnap_gram_scan_k, dim3((unsigned)n), dim3(256), 0, st, G, n, out);
    return hipGetLastError();
}

hipError_t launch_chol_solve(const double* packed, int K, double alpha, double* out, hipStream_t st) {
    const size_t lds = ((size_t)K * (K + 1) + (size_t)K + 256) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)fsnap_chol_solve_k, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024 - 64);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(fsnap_chol_solve_k, dim3(1), dim3(1024), lds, st, packed, K, alpha, out);
    return hipGetLastError();
}

}  // namespace fsnap
