// fsnap_trsm.hip — row-space orthogonalisation pass of the least-squares path (gfx950 only).
//
// The reference's SVD solver works on the ROWS: scipy.linalg.lstsq(aw, bw, 1.0e-13) (fitsnap3lib/solvers/svd.py:54,
// LAPACK dgelsd: QR of aw, then the SVD of the K x K factor with the 1e-13 cut).  The normal equations square the
// condition number, so for kappa(A_w) beyond ~1e7 the statistics alone cannot reproduce that answer.  This file holds
// the one kernel the row-space path adds to the SYRK / GEMV kernels: Q <- X R^-1 for an upper triangular K x K factor R
// from the (shifted) Cholesky factorisation of X^T X -- a "CholeskyQR" pass.  Two or three passes give a Q with
// orthonormal columns and A_w = Q R_hat to working precision for kappa up to ~1/eps (shifted CholeskyQR3, Fukaya et
// al. 2020); the host then treats the K x K factor R_hat exactly as dgelsd treats its R (fsnap_rowspace.cpp).
//
// Kernel 13  fsnap_trsm_rows_k   one wave per 64 rows; column blocks of 16 in ascending order:
//     S_J = X_J - sum_{I<J} Q_I R_IJ      fp64 MFMA (v_mfma_f64_16x16x4_f64), A operand = the solved blocks of Q,
//                                         read back from global memory (L1 / L2), B operand = R (L2-resident)
//     Q_J = S_J R_JJ^-1                   true substitution (not a multiplication by an inverse: the backward error
//                                         has to stay ~eps |R|, that is what makes A_w = Q R_hat hold), one row per
//                                         lane after a transpose of the 64 x 16 block through LDS; R_JJ is staged
//                                         in LDS too (all lanes read the same entry: a broadcast)
//   m K^2 flop per pass (K = 128: as much as the SYRK), launched m / 64 workgroups wide.  In-place safe in the later
//   passes: block J of a row is read before it is overwritten and never again.
#include "fsnap_device_common.h"
#include "fsnap_kernels.h"

template <bool FIRST>
__global__ __launch_bounds__(64, FIRST ? 2 : 3) void fsnap_trsm_rows_k(const double* __restrict__ src, int64_t lds_,
                                                        const double* __restrict__ wpack, double* Q, int64_t ldq,
                                                        int64_t m, int K, const double* __restrict__ R, int K16) {
    __shared__ double X[64][17];                  // 64 rows x 16 columns of the current block (+1: no bank conflicts)
    __shared__ double Rs[16][32];                 // diagonal block R_JJ, zero-padded to 32 columns (read by all lanes at the same
                                                  // address: a broadcast)
    const int lane = threadIdx.x, e = lane & 15, g = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    const int NB = K16 >> 4;
    // first pass: X = diag(w_eff) A; a zero weight (masked row, zero weight) gives a zero row of Q whatever the row
    // of A holds (NaN in masked rows is legal input).  w_eff is re-read per column block (cached; keeps registers free).
    for (int jb = 0; jb < NB; ++jb) {
        const int col = jb * 16 + e;
        d4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
        // S_J accumulation over the solved blocks: MFMA operands A[i = lane & 15][k = lane >> 4], B[k][j = lane & 15].
        // The k index of an MFMA is free as long as both operands agree: k-step s of lane group g takes column 4 g + s
        // of the block (not 4 s + g), so a lane needs FOUR ADJACENT doubles of its row -- two 16-byte loads per row
        // tile instead of four 8-byte ones, whole 32-byte sectors.  (A solved block kb < jb <= NB - 1 is never the
        // partial last block: no column guard.)
        for (int kb = 0; kb < jb; ++kb) {
            d2u qa[4], qb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int64_t r = row0 + t * 16 + e;
                const double* p = Q + (r < m ? r : 0) * ldq + kb * 16 + 4 * g;
                qa[t] = *reinterpret_cast<const d2u*>(p);
                qb[t] = *reinterpret_cast<const d2u*>(p + 2);
                if (r >= m) {
                    qa[t] = (d2u){0.0, 0.0};
                    qb[t] = (d2u){0.0, 0.0};
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double bf = R[(size_t)(kb * 16 + 4 * g + s) * K16 + col];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const double af = (s < 2) ? qa[t][s & 1] : qb[t][s & 1];
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc[t], 0, 0, 0);
                }
            }
        }
        // X_J - S_J in the accumulator layout (row = (lane >> 4) + 4 v, column = lane & 15) -> LDS
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int lr = t * 16 + g + 4 * v;
                const int64_t r = row0 + lr;
                double x = 0.0;
                if (r < m && col < K) {
                    if (FIRST) {
                        const double wgt = wpack[2 * r];
                        const double a = src[r * lds_ + col];
                        x = (wgt != 0.0) ? wgt * a : 0.0;
                    } else {
                        x = Q[r * ldq + col];        // in place: read through the same (non-restrict) pointer that writes
                    }
                }
                X[lr][e] = x - acc[t][v];
            }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            Rs[g + 4 * v][e] = R[(size_t)(jb * 16 + g + 4 * v) * K16 + col];
            Rs[g + 4 * v][16 + e] = 0.0;
        }
        __syncthreads();
        // one row per lane, right-looking: q_i = x_i / R[i][i], then x_j -= q_i R[i][j] for the columns right of it -- the
        // 15 updates of a step are independent FMAs, only the division sits on the chain (a left-looking loop with its
        // operands in LDS was a chain of 120 dependent FMAs, each behind two LDS reads: ~7 us per block).  The row stays
        // in 16 registers; each step shifts it left by one while updating (x[t] = x[t+1] - q R[i][i+1+t]), so the pivot is
        // always x[0] and the loop can stay rolled (fully unrolled, the compiler hoists all 120 broadcasts of R_JJ and
        // spills); the zero padding of Rs makes the reads past column 15 harmless.
        {
            double x[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) x[j] = X[lane][j];
#pragma unroll 1
            for (int i = 0; i < 16; ++i) {
                const double q = x[0] / Rs[i][i];
                X[lane][i] = q;
                const double* rr = &Rs[i][i + 1];
#pragma unroll
                for (int t = 0; t < 15; ++t) x[t] = __builtin_fma(-q, rr[t], x[t + 1]);
            }
        }
        __syncthreads();
        // store the block from LDS in the accumulator layout: 16 lanes write 16 adjacent doubles of a row
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int lr = t * 16 + g + 4 * v;
                const int64_t r = row0 + lr;
                if (r < m && col < K) Q[r * ldq + col] = X[lr][e];
            }
        __syncthreads();     // the block just written is an MFMA operand of the next column blocks; X is reused
    }
}

// per-row pairs for the SYRK kernels on Q: first component = "row takes part" (zero rows are skipped by the loads),
// second = w_eff b, so that the kernels' c output is Q^T (w b)
__global__ __launch_bounds__(256) void fsnap_qpack_k(const double* __restrict__ wpack, int64_t m, double* __restrict__ qpack) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < m) {
        const d2 p = *reinterpret_cast<const d2*>(wpack + 2 * i);
        d2 o;
        o[0] = (p[0] != 0.0) ? 1.0 : 0.0;
        o[1] = (p[0] != 0.0) ? p[1] : 0.0;
        *reinterpret_cast<d2*>(qpack + 2 * i) = o;
    }
}

namespace fsnap {

hipError_t launch_trsm_rows(const double* src, int64_t lds, const double* wpack, double* Q, int64_t ldq, int64_t m, int K,
                            const double* R, int K16, hipStream_t st) {
    const int64_t nb = (m + 63) / 64;
    if (nb > 0x7FFFFFFF) return hipErrorInvalidValue;
    if (wpack)
        hipLaunchKernelGGL(fsnap_trsm_rows_k<true>, dim3((unsigned)nb), dim3(64), 0, st, src, lds, wpack, Q, ldq, m, K, R, K16);
    else
        hipLaunchKernelGGL(fsnap_trsm_rows_k<false>, dim3((unsigned)nb), dim3(64), 0, st, src, lds, wpack, Q, ldq, m, K, R, K16);
    return hipGetLastError();
}

hipError_t launch_qpack(const double* wpack, int64_t m, double* qpack, hipStream_t st) {
    const int64_t nb = (m + 255) / 256;
    if (nb > 0x7FFFFFFF) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fsnap_qpack_k, dim3((unsigned)nb), dim3(256), 0, st, wpack, m, qpack);
    return hipGetLastError();
}

}  // namespace fsnap
