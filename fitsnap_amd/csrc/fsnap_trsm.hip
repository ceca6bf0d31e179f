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
// Kernel 13  fsnap_trsm_rows_k   (K > 128) one wave per 64 rows; column blocks of 16 in ascending order, left-looking:
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

// TR = 16-row tiles per wave (4: 64 rows per wave; 2, 1: shorter tiles for short matrices -- the quadratic-SNAP shape
// 15 213 x 1 595 gives only 238 waves of 64 rows for 1024 SIMDs, each with 79 200 MFMAs = 2.1 ms of matrix pipe to itself)
template <bool FIRST, int TR>
__global__ __launch_bounds__(64, (FIRST || TR == 4) ? 2 : 3) void fsnap_trsm_rows_k(const double* __restrict__ src, int64_t lds_,
                                                        const double* __restrict__ wpack, double* Q, int64_t ldq,
                                                        int64_t m, int K, const double* __restrict__ R, int K16) {
    constexpr int ROWS = 16 * TR;
    __shared__ double X[ROWS][17];                // ROWS rows x 16 columns of the current block (+1: no bank conflicts)
    __shared__ double Rs[16][32];                 // diagonal block R_JJ, zero-padded to 32 columns (read by all lanes at the same
                                                  // address: a broadcast)
    const int lane = threadIdx.x, e = lane & 15, g = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * ROWS;
    const int NB = K16 >> 4;
    // first pass: X = diag(w_eff) A; a zero weight (masked row, zero weight) gives a zero row of Q whatever the row
    // of A holds (NaN in masked rows is legal input).  w_eff is re-read per column block (cached; keeps registers free).
    for (int jb = 0; jb < NB; ++jb) {
        const int col = jb * 16 + e;
        d4 acc[TR];
#pragma unroll
        for (int t = 0; t < TR; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
        // S_J accumulation over the solved blocks: MFMA operands A[i = lane & 15][k = lane >> 4], B[k][j = lane & 15].
        // The k index of an MFMA is free as long as both operands agree: k-step s of lane group g takes column 4 g + s
        // of the block (not 4 s + g), so a lane needs FOUR ADJACENT doubles of its row -- two 16-byte loads per row
        // tile instead of four 8-byte ones, whole 32-byte sectors.  (A solved block kb < jb <= NB - 1 is never the
        // partial last block: no column guard.)
        // software pipeline: the operands of block kb + 1 are requested before the MFMAs of block kb are issued (as a
        // plain loop every trip paid an L2 round trip in front of its 4 TR MFMAs)
        d2u qa[TR], qb[TR];
        double bf[4];
        auto fetch = [&](int kb, d2u (&a)[TR], d2u (&b)[TR], double (&f)[4]) {
#pragma unroll
            for (int t = 0; t < TR; ++t) {
                const int64_t r = row0 + t * 16 + e;
                const double* p = Q + (r < m ? r : 0) * ldq + kb * 16 + 4 * g;
                a[t] = *reinterpret_cast<const d2u*>(p);
                b[t] = *reinterpret_cast<const d2u*>(p + 2);
                if (r >= m) {
                    a[t] = (d2u){0.0, 0.0};
                    b[t] = (d2u){0.0, 0.0};
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) f[s] = R[(size_t)(kb * 16 + 4 * g + s) * K16 + col];
        };
        if (jb > 0) fetch(0, qa, qb, bf);
        for (int kb = 0; kb < jb; ++kb) {
            d2u na[TR], nb2[TR];
            double nf[4];
            const int kn = kb + 1 < jb ? kb + 1 : kb;          // (the last trip re-reads its own block: in range, unused)
            fetch(kn, na, nb2, nf);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int t = 0; t < TR; ++t) {
                    const double af = (s < 2) ? qa[t][s & 1] : qb[t][s & 1];
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf[s], acc[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < TR; ++t) {
                qa[t] = na[t];
                qb[t] = nb2[t];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) bf[s] = nf[s];
        }
        // X_J - S_J in the accumulator layout (row = (lane >> 4) + 4 v, column = lane & 15) -> LDS
#pragma unroll
        for (int t = 0; t < TR; ++t)
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
        if (lane < ROWS) {
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
        for (int t = 0; t < TR; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int lr = t * 16 + g + 4 * v;
                const int64_t r = row0 + lr;
                if (r < m && col < K) Q[r * ldq + col] = X[lr][e];
            }
        __syncthreads();     // the block just written is an MFMA operand of the next column blocks; X is reused
    }
}

// Kernel 13A (K <= 128, the default there): the same pass RIGHT-LOOKING with the wave's whole 64 x K row tile in the
// accumulation registers (NB x 4 tiles of 16 x 16 = up to 256 registers, one wave per SIMD -- kernel 1A's register plan).
// Kernel 13 above is left-looking: block J gathers the contributions of all solved blocks, which it re-reads from global
// memory -- a wave's 64 KB of solved blocks once per later block, and the 256 resident waves of an XCD hold 16 MB of
// them, four times its L2: 1.75-2.0 ms per pass at 10^6 x 128.  Here a solved block Q_J is applied to ALL later blocks
// at once, X_L -= Q_J R_JL, straight from the LDS copy that the substitution produced (16 operand reads per block,
// reused for every L); a row of A is read from HBM once, a row of Q written once, nothing is re-read.
// LDS hand-off inside ONE wave (the workgroup of kernel 13A is a single wave): the LDS queue of a wave is in order, so a
// read sees the wave's earlier writes -- only the compiler must not reorder them.  __syncthreads() would also wait for
// the block's global STORES (s_waitcnt vmcnt(0)): ~4 us per column block with nothing else on the SIMD to run.
__device__ __forceinline__ void trsm_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int NB, bool FIRST>
__global__ __launch_bounds__(64, 1) void fsnap_trsm_acc_k(const double* __restrict__ src, int64_t lds_,
                                                          const double* __restrict__ wpack, double* Q, int64_t ldq,
                                                          int64_t m, int K, const double* __restrict__ R) {
    constexpr int K16 = 16 * NB;
    constexpr int RLD = K16 + 36;                 // row stride of the staged row block of R: even (16-byte row bases), >= 31
                                                  // readable entries behind the last block's 16 columns, and the four k rows
                                                  // of a B operand land on different banks
    __shared__ double X[64][17];
    __shared__ __attribute__((aligned(16))) double Rblk[16][RLD];   // rows 16 J .. 16 J + 15 of R from the diagonal block on
    __shared__ double Rinv[16];                   // reciprocals of the diagonal of R_JJ: the division leaves the 16-step chain
    const int lane = threadIdx.x, e = lane & 15, g = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    d4 acc[NB][4];
    // the whole row tile, in the accumulator layout (tile rows g + 4 v, column e): 16 lanes read 128 contiguous bytes
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int J = 0; J < NB; ++J)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int64_t r = row0 + t * 16 + g + 4 * v;
                const int col = J * 16 + e;
                double x = 0.0;
                if (r < m && col < K) x = FIRST ? src[r * lds_ + col] : Q[r * ldq + col];
                acc[J][t][v] = x;
            }
    // first pass: the row weights commute with the solve, diag(w) (A R^-1) = (diag(w) A) R^-1 -- the rows are solved as
    // they come and scaled when they are stored (no second copy of the tile while the loads are in flight); a zero weight
    // stores a zero row whatever A holds (NaN in masked rows is legal input)
    double wgt[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int64_t r = row0 + t * 16 + g + 4 * v;
            wgt[t][v] = (FIRST && r < m) ? wpack[2 * r] : 1.0;
        }
    double qst[4][4];                 // the block solved last, waiting for its stores
    auto store_block = [&](int Jb) {
        const int col = Jb * 16 + e;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int64_t r = row0 + t * 16 + g + 4 * v;
                if (r < m && col < K) Q[r * ldq + col] = FIRST ? ((wgt[t][v] != 0.0) ? wgt[t][v] * qst[t][v] : 0.0) : qst[t][v];
            }
    };
#pragma unroll
    for (int J = 0; J < NB; ++J) {
        // block J -> LDS (row-per-lane layout for the substitution), with its diagonal block of R
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) X[t * 16 + g + 4 * v][e] = acc[J][t][v];
        trsm_wave_sync();
        // rows 16 J .. 16 J + 15 of R, from the diagonal block to the last column, straight into LDS: one asynchronous
        // global_load_lds per row (lane l carries 16 bytes to row base + 16 l), no registers, all 16 in flight together.
        // (Staged through registers, the compiler -- at 360+ live registers -- issued these ~20 loads one at a time,
        // each behind a full s_waitcnt: ~12 us per column block, 60 % of the kernel's 1.5 ms.)
        {
            const int ncol = K16 - J * 16;
            if (2 * lane < ncol) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(R + (size_t)(J * 16 + i) * K16 + J * 16 + 2 * lane),
                        (__attribute__((address_space(3))) void*)&Rblk[i][0], 16, 0, 0);
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): the row block (and, in the first round, the row tile) has landed
        if (J > 0) store_block(J - 1);
        trsm_wave_sync();
        if (lane < 16) Rinv[lane] = 1.0 / Rblk[lane][lane];
        trsm_wave_sync();
        {
            // two steps per trip, the multipliers of a step read from LDS one step ahead (the wave is alone on its SIMD:
            // an LDS round trip inside the 16-step chain is paid in full)
            double x[16], ra[15], rb[15];
#pragma unroll
            for (int j = 0; j < 16; ++j) x[j] = X[lane][j];
#pragma unroll
            for (int t = 0; t < 15; ++t) ra[t] = Rblk[0][1 + t];
            double ia = Rinv[0];
#pragma unroll 1
            for (int i = 0; i < 16; i += 2) {
                const double ib = Rinv[i + 1];
#pragma unroll
                for (int t = 0; t < 15; ++t) rb[t] = Rblk[i + 1][i + 2 + t];
                double q = x[0] * ia;
                X[lane][i] = q;
#pragma unroll
                for (int t = 0; t < 15; ++t) x[t] = __builtin_fma(-q, ra[t], x[t + 1]);
                const int in = (i + 2) & 15;          // (the last trip reads row 0 again: in range, unused)
                ia = Rinv[in];
#pragma unroll
                for (int t = 0; t < 15; ++t) ra[t] = Rblk[in][in + 1 + t];
                q = x[0] * ib;
                X[lane][i + 1] = q;
#pragma unroll
                for (int t = 0; t < 15; ++t) x[t] = __builtin_fma(-q, rb[t], x[t + 1]);
            }
        }
        trsm_wave_sync();
        // the solved block back in the accumulator layout; its stores are issued one block LATE (right after the next
        // block's staging wait): s_waitcnt vmcnt(0) also waits for stores, and with the stores issued just before it
        // every block paid a full write round trip (PMC: 47 k of 182 k cycles per wave in s_waitcnt)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) qst[t][v] = X[t * 16 + g + 4 * v][e];
        if (J + 1 < NB) {
            // X_L -= Q_J R_JL for every later block: A operand (Q_J)[i = e][k = 4 s + g] from LDS, B operand R[k][j = e]
            double af[4][4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int sk = 0; sk < 4; ++sk) af[t][sk] = -X[t * 16 + e][4 * sk + g];
#pragma unroll
            for (int L = J + 1; L < NB; ++L)
#pragma unroll
                for (int sk = 0; sk < 4; ++sk) {
                    const double bf = Rblk[4 * sk + g][(L - J) * 16 + e];
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[L][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[t][sk], bf, acc[L][t], 0, 0, 0);
                }
        }
        trsm_wave_sync();    // X and Rs are reused by the next block
    }
    store_block(NB - 1);
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
    if (K16 <= 128) {
        // kernel 13A: whole row tile in the accumulation registers
        const dim3 grid((unsigned)nb), block(64);
#define FSNAP_TRSM_ACC(NBV)                                                                                                  \
    case NBV:                                                                                                                \
        if (wpack) hipLaunchKernelGGL((fsnap_trsm_acc_k<NBV, true>), grid, block, 0, st, src, lds, wpack, Q, ldq, m, K, R);    \
        else hipLaunchKernelGGL((fsnap_trsm_acc_k<NBV, false>), grid, block, 0, st, src, lds, wpack, Q, ldq, m, K, R);         \
        break;
        switch (K16 / 16) {
            FSNAP_TRSM_ACC(1) FSNAP_TRSM_ACC(2) FSNAP_TRSM_ACC(3) FSNAP_TRSM_ACC(4)
            FSNAP_TRSM_ACC(5) FSNAP_TRSM_ACC(6) FSNAP_TRSM_ACC(7) FSNAP_TRSM_ACC(8)
            default: return hipErrorInvalidValue;
        }
#undef FSNAP_TRSM_ACC
        return hipGetLastError();
    }
    // kernel 13: 64-row tiles when there is a wave of them for every SIMD, shorter tiles for short matrices
#define FSNAP_TRSM_ROWS(TRV)                                                                                                 \
    {                                                                                                                        \
        const dim3 grid((unsigned)((m + 16 * TRV - 1) / (16 * TRV))), block(64);                                             \
        if (wpack) hipLaunchKernelGGL((fsnap_trsm_rows_k<true, TRV>), grid, block, 0, st, src, lds, wpack, Q, ldq, m, K, R, K16); \
        else hipLaunchKernelGGL((fsnap_trsm_rows_k<false, TRV>), grid, block, 0, st, src, lds, wpack, Q, ldq, m, K, R, K16);  \
    }
    // (measured: 100 000 x 256 0.45 ms with 64-row tiles, 0.62 ms with 32-row tiles; 15 213 x 1 595 6.0 / 3.5 ms with 64- /
    // 16-row tiles)
    if (m >= 64 * 1024) FSNAP_TRSM_ROWS(4)
    else if (m >= 32 * 1024) FSNAP_TRSM_ROWS(2)
    else FSNAP_TRSM_ROWS(1)
#undef FSNAP_TRSM_ROWS
    return hipGetLastError();
}

hipError_t launch_qpack(const double* wpack, int64_t m, double* qpack, hipStream_t st) {
    const int64_t nb = (m + 255) / 256;
    if (nb > 0x7FFFFFFF) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fsnap_qpack_k, dim3((unsigned)nb), dim3(256), 0, st, wpack, m, qpack);
    return hipGetLastError();
}

}  // namespace fsnap
