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
// widths; the host factorisation takes 20-40 ms at K = 1595, this path ~1 ms).  Same algorithm as the host fast
// path (fsnap_solve.cpp): Jacobi scaling S = D (G + alpha I) D with D = diag(G + alpha I)^-1/2, S = U^T U,
// two triangular sweeps, beta = D x; accepted by the caller only if every pivot of the scaled matrix stays
// above 1e-3 (otherwise the general host path runs).
// The work matrix is padded to a multiple of 64 with an identity block, so no kernel has edge cases:
//   8a prepare   d, z = D c, status; S (upper and lower) into the padded work matrix
//   per 64-row panel [jb, je):
//   8b diag      one workgroup factorises the 64 x 64 diagonal block in LDS (64-step recurrence)
//   8c tails     one thread per trailing column: forward substitution U12 = U11^-T S12
//   8d update    S22 -= U12^T U12 on the matrix pipe: one wave per 32 x 32 block pair (2 x 2 MFMA tiles),
//                k = 64 rows in 16 MFMA steps -- a 64-row SYRK, the same operand trick as kernel 1
//   8e sweeps    one workgroup: blocked forward / backward substitution and the un-scaling
// status[0]: bit 0 = non-positive / non-finite diagonal of G + alpha I, bit 1 = failed pivot;
// minpiv[p] = smallest pivot of panel p.
// ---------------------------------------------------------------------------------
constexpr int CHOL_NB = 64;

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
                                                             double* __restrict__ S, int* __restrict__ status) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= np) return;
    double v;
    if (i < n && j < n) {
        const double g = packed[(size_t)i * n + j] + ((i == j) ? alpha : 0.0);
        v = g * dsc[i] * dsc[j];
        // (not `v - v == 0`: with fp contraction that becomes fma(g d_i, d_j, -v), the rounding error of the product)
        if (!__builtin_isfinite(v)) atomicOr(status, 1);
    } else {
        v = (i == j) ? 1.0 : 0.0;
    }
    S[(size_t)i * np + j] = v;
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// 8b: ONE wave, lane c holds column c of the 64 x 64 block in 64 registers; the recurrence is fully unrolled, the
// pivot row entry B[j][i] reaches all lanes through v_readlane (scalar broadcast): no LDS, no barrier.  Entries
// below the diagonal are updated along (never read, not written back).  (Measured 58 us per block; an LDS version
// with 256 threads took 95 us, one with 4 x 4 register sub-blocks and two barriers per step 168 us: these
// single-workgroup kernels run while the chip is almost idle, at whatever clock it then holds.)
__global__ __launch_bounds__(64) void fsnap_chol_diag_k(double* __restrict__ S, int np, int jb, int* __restrict__ status,
                                                       double* __restrict__ minpiv) {
    const int c = threadIdx.x;
    double col[CHOL_NB];
#pragma unroll
    for (int r = 0; r < CHOL_NB; ++r) col[r] = S[(size_t)(jb + r) * np + jb + c];
    double pmin = 1.0e300;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < CHOL_NB; ++j) {
        const double d = readlane_f64(col[j], j);          // pivot (the same value in every lane)
        bad = bad || !(d > 0.0) || !__builtin_isfinite(d);
        pmin = d < pmin ? d : pmin;
        const double r = sqrt(d), inv = 1.0 / r;
        col[j] = (c == j) ? r : col[j] * inv;
#pragma unroll
        for (int i = j + 1; i < CHOL_NB; ++i) {
            const double f = readlane_f64(col[j], i);      // B[j][i]
            col[i] = __builtin_fma(-f, col[j], col[i]);    // B[i][c] -= B[j][i] * B[j][c]
        }
    }
    if (bad) {
        if (c == 0) atomicOr(status, 2);
        return;
    }
#pragma unroll
    for (int r = 0; r < CHOL_NB; ++r)
        if (c >= r) S[(size_t)(jb + r) * np + jb + c] = col[r];
    if (c == 0) minpiv[jb / CHOL_NB] = pmin;
}

// 8c: one thread per trailing column, forward substitution over the 64 panel rows with U11 in LDS (uniform reads)
__global__ __launch_bounds__(256) void fsnap_chol_tails_k(double* __restrict__ S, int np, int jb,
                                                         const int* __restrict__ status) {
    __shared__ double U11[CHOL_NB][CHOL_NB + 1];
    __shared__ double rinv[CHOL_NB];
    if (*status) return;
    const int tid = threadIdx.x;
    for (int t = tid; t < CHOL_NB * CHOL_NB; t += 256) {
        const int i = t >> 6, k = t & 63;
        U11[i][k] = S[(size_t)(jb + i) * np + jb + k];
    }
    __syncthreads();
    if (tid < CHOL_NB) rinv[tid] = 1.0 / U11[tid][tid];
    __syncthreads();
    const int c = jb + CHOL_NB + blockIdx.x * 256 + tid;
    if (c >= np) return;
    double x[CHOL_NB];
#pragma unroll
    for (int k = 0; k < CHOL_NB; ++k) x[k] = S[(size_t)(jb + k) * np + c];
#pragma unroll
    for (int k = 0; k < CHOL_NB; ++k) {
        double v = x[k];
#pragma unroll
        for (int p = 0; p < k; ++p) v -= U11[p][k] * x[p];
        x[k] = v * rinv[k];
    }
#pragma unroll
    for (int k = 0; k < CHOL_NB; ++k) S[(size_t)(jb + k) * np + c] = x[k];
}

__global__ __launch_bounds__(256) void fsnap_chol_update_k(double* __restrict__ S, int np, int jb, int nblk,
                                                          const int* __restrict__ status) {
    if (*status) return;
    const int lane = threadIdx.x & 63, e = lane & 15, kr = lane >> 4;
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair >= nblk * (nblk + 1) / 2) return;
    // pair -> (I, J), I <= J, row-major packed triangle over nblk 32-column blocks
    int I = 0, rem = pair;
    while (rem >= nblk - I) {
        rem -= nblk - I;
        ++I;
    }
    const int J = I + rem;
    const int je = jb + CHOL_NB;
    const int cI = je + 32 * I, cJ = je + 32 * J;
    d4 a00 = {0, 0, 0, 0}, a01 = a00, a10 = a00, a11 = a00;
    const double* base = S + (size_t)(jb + kr) * np;
#pragma unroll 4
    for (int s4 = 0; s4 < CHOL_NB / 4; ++s4) {
        const double* r = base + (size_t)(4 * s4) * np;
        const double x0 = r[cI + e], x1 = r[cI + 16 + e];
        const double y0 = r[cJ + e], y1 = r[cJ + 16 + e];
        a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, y0, a00, 0, 0, 0);
        a01 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, y1, a01, 0, 0, 0);
        a10 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, y0, a10, 0, 0, 0);
        a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, y1, a11, 0, 0, 0);
    }
    // D tile layout: element (row = kr + 4 r, col = e)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const int row = kr + 4 * r4;
        double* p0 = S + (size_t)(cI + row) * np;
        double* p1 = S + (size_t)(cI + 16 + row) * np;
        p0[cJ + e] -= a00[r4];
        p0[cJ + 16 + e] -= a01[r4];
        p1[cJ + e] -= a10[r4];
        p1[cJ + 16 + e] -= a11[r4];
    }
}

__global__ __launch_bounds__(1024) void fsnap_chol_sweeps_k(const double* __restrict__ S, int np, int n,
                                                           double* __restrict__ z, const double* __restrict__ dsc,
                                                           double* __restrict__ beta, const int* __restrict__ status) {
    // z (length np, global) is solved in place: forward U^T y = z, backward U x = y; beta = D x
    __shared__ double U11[CHOL_NB][CHOL_NB + 1];
    __shared__ double xb[CHOL_NB];
    __shared__ double red[1024];
    if (*status) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int npanel = np / CHOL_NB;
    for (int pb = 0; pb < npanel; ++pb) {
        const int jb = pb * CHOL_NB, je = jb + CHOL_NB;
        for (int t = tid; t < CHOL_NB * CHOL_NB; t += 1024) U11[t >> 6][t & 63] = S[(size_t)(jb + (t >> 6)) * np + jb + (t & 63)];
        __syncthreads();
        if (wv == 0) {      // 64 x 64 lower-triangular solve (U11^T y = z_b) inside one wave
            double v = z[jb + lane];
            const double rdiag = 1.0 / U11[lane][lane];
            for (int k = 0; k < CHOL_NB; ++k) {
                const double yk = readlane_f64(v, k) * readlane_f64(rdiag, k);
                if (lane == k) v = yk;
                else if (lane > k) v -= U11[k][lane] * yk;
            }
            xb[lane] = v;
            z[jb + lane] = v;
        }
        __syncthreads();
        for (int c = je + tid; c < np; c += 1024) {
            double acc = z[c];
#pragma unroll 8
            for (int k = 0; k < CHOL_NB; ++k) acc -= S[(size_t)(jb + k) * np + c] * xb[k];
            z[c] = acc;
        }
        __syncthreads();
    }
    for (int pb = npanel - 1; pb >= 0; --pb) {
        const int jb = pb * CHOL_NB, je = jb + CHOL_NB;
        for (int t = tid; t < CHOL_NB * CHOL_NB; t += 1024) U11[t >> 6][t & 63] = S[(size_t)(jb + (t >> 6)) * np + jb + (t & 63)];
        // row k of the panel: z_k -= sum_{c >= je} U[k][c] x_c, 16 threads per row
        {
            const int k = tid >> 4, q = tid & 15;
            double acc = 0.0;
            const double* r = S + (size_t)(jb + k) * np;
            for (int c = je + q; c < np; c += 16) acc += r[c] * z[c];
            red[tid] = acc;
        }
        __syncthreads();
        if (wv == 0) {
            double v = z[jb + lane];
            double sub = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) sub += red[lane * 16 + q];
            v -= sub;
            const double rdiag = 1.0 / U11[lane][lane];
            for (int k = CHOL_NB - 1; k >= 0; --k) {
                const double xk = readlane_f64(v, k) * readlane_f64(rdiag, k);
                if (lane == k) v = xk;
                else if (lane < k) v -= U11[lane][k] * xk;
            }
            z[jb + lane] = v;
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += 1024) beta[i] = z[i] * dsc[i];
}
// ---------------------------------------------------------------------------------
// host-side launchers (C++ linkage, used by fsnap_capi.cpp)
// ---------------------------------------------------------------------------------
namespace fsnap {

hipError_t launch_chol_large(const double* packed, const double* cvec, int n, double alpha, double* S, double* dsc, double* z,
                             double* beta, int* status, double* minpiv, hipStream_t st) {
    if (!cvec) cvec = packed + (size_t)n * n;
    const int np = (n + CHOL_NB - 1) / CHOL_NB * CHOL_NB, npanel = np / CHOL_NB;
    hipError_t e = hipMemsetAsync(status, 0, sizeof(int), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fsnap_chol_prepare_d_k, dim3((np + 255) / 256), dim3(256), 0, st, packed, cvec, n, np, alpha, dsc, z, status,
                       minpiv, npanel);
    hipLaunchKernelGGL(fsnap_chol_prepare_s_k, dim3((np + 255) / 256, np), dim3(256), 0, st, packed, n, np, alpha, dsc, S,
                       status);
    for (int pb = 0; pb < npanel; ++pb) {
        const int jb = pb * CHOL_NB;
        hipLaunchKernelGGL(fsnap_chol_diag_k, dim3(1), dim3(64), 0, st, S, np, jb, status, minpiv);
        const int ntail = np - jb - CHOL_NB;
        if (ntail > 0) {
            hipLaunchKernelGGL(fsnap_chol_tails_k, dim3((ntail + 255) / 256), dim3(256), 0, st, S, np, jb, status);
            const int nblk = ntail / 32, npair = nblk * (nblk + 1) / 2;
            hipLaunchKernelGGL(fsnap_chol_update_k, dim3((npair + 3) / 4), dim3(256), 0, st, S, np, jb, nblk, status);
        }
    }
    hipLaunchKernelGGL(fsnap_chol_sweeps_k, dim3(1), dim3(1024), 0, st, S, np, n, z, dsc, beta, status);
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
