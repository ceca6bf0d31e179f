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
// Every pass: Q_J = S_J R_JJ^-1 by TRUE substitution (not a multiplication by an inverse: the backward error has to stay
// ~eps |R|, that is what makes A_w = Q R_hat hold), the updates S_L -= Q_J R_JL on the fp64 matrix pipe; m K^2 flop per pass
// (K = 128: as much as the SYRK).  Kernel 13C (K <= 128; 129 ... 144 columns from 32 768 rows on) and kernel 13B (K > 128) below; kernel 13 (left-looking, round 2)
// and kernel 13A (one wave per SIMD with the whole 64 x K tile, rounds 3-4) are gone -- profiles/r04_trsm_13a_vs_13b.txt,
// r05_bench_kernel_stats.csv hold their numbers.
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>

#include "fsnap_device_common.h"
#include "fsnap_kernels.h"

// LDS hand-off inside ONE wave: the LDS queue of a wave is in order, so a read sees the wave's earlier writes -- only the
// compiler must not reorder them.  __syncthreads() would also wait for the block's global STORES (s_waitcnt vmcnt(0)): ~4 us
// per column block with nothing else on the SIMD to run.
__device__ __forceinline__ void trsm_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int... I, class F>
__device__ __forceinline__ void trsm_static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void trsm_static_for(F&& f) {
    trsm_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

#ifdef FSNAP_TRSM_TRACE
// tools/trsm_trace.hip only: wall-clock stamps (100 MHz) per row tile of kernel 13C: 0 entry, 1 + 5 P panel P loaded / weighted
// and its left-looking updates done, 2 + 5 P + J block J of panel P done, 15 exit
__device__ unsigned long long fsnap_trsm_trace[8192 * 16];
#define FSNAP_TRSM_STAMP(i)                                                                                  \
    do {                                                                                                     \
        if (threadIdx.x == 0 && blockIdx.x < 8192) fsnap_trsm_trace[blockIdx.x * 16 + (i)] = wall_clock64(); \
    } while (0)
#else
#define FSNAP_TRSM_STAMP(i)
#endif

// Kernel 13C (round 5; K <= 128, round 6: K <= 144): kernel 13A's pass with TWO waves per SIMD.  Kernel 13A keeps a wave's whole 64 x K tile in
// the accumulation registers -- one wave per SIMD, and of the 48 us a tile takes only 12 are matrix pipe and 13 the VALU
// substitution: the rest are LDS hand-overs, the staging of R and the tile load with nothing else on the SIMD to run
// (profiles/r04_trsm_13a_vs_13b.txt).  Here a wave holds a 64 x 64 PANEL of its rows (128 accumulation registers), so two
// waves fit a SIMD and one wave's substitution runs beside the other's MFMAs and waits:
//   panel P of a row tile:  S_P = sum_{I < P} Q_I R_IP   left-looking: the solved panels of the wave's OWN rows come back from
//                                                        L2 (just written; 32 KiB per tile and earlier panel), R from L2
//                           Q_P = (X_P - S_P) R_PP^-1    right-looking inside the panel, block by block as kernel 13A: the
//                                                        16-step substitution one row per lane on the VALU (all 64 lanes),
//                                                        updates of the later blocks of the panel on the matrix pipe with the
//                                                        B operands (rows of R) straight from L2 into registers
// The same MFMA count as kernel 13A (nothing is computed twice); LDS per wave 11 KiB (X 64 x 17, the diagonal block).  The row
// weights of the first pass are applied when a panel is LOADED (the solved panels a later panel reads back must be the
// weighted ones); a zero weight makes a zero row whatever A holds.  In place in the later passes: a panel of X is read in full
// before any of it is overwritten, earlier panels are read back solved.
template <int NB, bool FIRST>
__global__ __launch_bounds__(64, 2) void fsnap_trsm_acc2_k(const double* __restrict__ src, int64_t lds_,
                                                           const double* __restrict__ wpack, double* Q, int64_t ldq,
                                                           int64_t m, int K, const double* __restrict__ R) {
    constexpr int K16 = 16 * NB;
    constexpr int NP = (NB + 3) / 4;
    __shared__ double X[64][17];
    __shared__ double Rd[16][18];                 // diagonal block R_JJ (every lane reads the same entry: a broadcast)
    __shared__ double Rinv[16];
    __shared__ double Wl[64];                     // first pass: the row weights of the tile
    const int lane = threadIdx.x, e = lane & 15, g = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * 64;
    FSNAP_TRSM_STAMP(0);
    if constexpr (FIRST) {
        Wl[lane] = (row0 + lane < m) ? wpack[2 * (row0 + lane)] : 0.0;
        trsm_wave_sync();
    }
    trsm_static_for<NP>([&](auto pc) {
        constexpr int P = decltype(pc)::value;
        constexpr int NBP = (NB - 4 * P) < 4 ? (NB - 4 * P) : 4;       // blocks of this panel
        d4 acc[NBP][4];
        // the panel of the row tile in the accumulator layout (tile rows g + 4 v, column e): straight into the accumulators,
        // through a bounds-checked buffer descriptor over the tile's rows -- ONE 32-bit offset per row of the lane (16 registers)
        // + immediates for the column blocks instead of 64 pointer pairs (which spilled in the first-pass form), and rows past
        // the end of the matrix read zeros in hardware
        {
            const double* tbase = FIRST ? src + row0 * lds_ : Q + row0 * ldq;
            const int64_t tld = FIRST ? lds_ : ldq;
            const int64_t trows = (m - row0) < 64 ? (m - row0) : 64;
            const __amdgpu_buffer_rsrc_t rs =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(tbase), 0, (int)(trows * tld * 8), 0x00020000);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const unsigned voff = (unsigned)(((int64_t)(t * 16 + g + 4 * v) * tld + e) * 8);
#pragma unroll
                    for (int jb = 0; jb < NBP; ++jb) {
                        const int col = (4 * P + jb) * 16 + e;
                        const u2 raw = __builtin_amdgcn_raw_buffer_load_b64(rs, voff + (unsigned)((4 * P + jb) * 128), 0, 0);
                        const double x = __builtin_bit_cast(double, raw);
                        // (a column >= K of the LAST block lies in the NEXT row when lda == K: selected away, not masked by the
                        // descriptor; every other block is inside the K columns by the definition of NB)
                        if (4 * P + jb == NB - 1) acc[jb][t][v] = (col < K) ? x : 0.0;      // (compile-time after unrolling)
                        else acc[jb][t][v] = x;
                    }
                }
        }
        // ... and weighted in place in the first pass (a zero weight makes a zero row whatever A holds); the 64 weights of the
        // tile wait in LDS (kept in registers next to the 64 values of a lane they spilled), four at a time
        if constexpr (FIRST) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const double wgt = Wl[t * 16 + g + 4 * v];
#pragma unroll
                    for (int jb = 0; jb < NBP; ++jb) acc[jb][t][v] = (wgt != 0.0) ? wgt * acc[jb][t][v] : 0.0;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // left-looking: X_P -= Q_I R_IP over the solved panels (all of them complete: 4 blocks of 16 columns)
        if constexpr (P > 0) {
#pragma unroll 1
            for (int kb = 0; kb < 4 * P; ++kb) {
                // A operand (Q)[i = e][k]: k-step s of lane group g takes column 4 g + s of the block (kernel 13's pairing:
                // four adjacent doubles of the lane's row, two 16-byte loads); B operand R[16 kb + 4 g + s][column]
                d2u qa[4], qb[4];
                if (row0 + 64 <= m) {              // (a tile inside the matrix: no row guards -- wave-uniform)
                    const double* p0 = Q + (row0 + e) * ldq + kb * 16 + 4 * g;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const double* p = p0 + (int64_t)(t * 16) * ldq;
                        qa[t] = *reinterpret_cast<const d2u*>(p);
                        qb[t] = *reinterpret_cast<const d2u*>(p + 2);
                    }
                } else {
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
                }
                // every B operand of the block row requested before the first product: taken one k-step at a time (4 loads, 16
                // products, 4 loads, ...) the loop waited for an L2 round trip per k-step (tools/trsm_trace.hip: 30 -> 26 us of a
                // tile's 92 in this loop at K = 128; the pass 0.604 -> 0.595 ms in place, 0.740 -> 0.702 ms first pass)
                double bf[4][NBP];
#pragma unroll
                for (int sk = 0; sk < 4; ++sk)
#pragma unroll
                    for (int jb = 0; jb < NBP; ++jb) bf[sk][jb] = R[(size_t)(kb * 16 + 4 * g + sk) * K16 + (4 * P + jb) * 16 + e];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int sk = 0; sk < 4; ++sk) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const double af = -((sk < 2) ? qa[t][sk & 1] : qb[t][sk & 1]);
#pragma unroll
                        for (int jb = 0; jb < NBP; ++jb)
                            acc[jb][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf[sk][jb], acc[jb][t], 0, 0, 0);
                    }
                }
            }
        }
        FSNAP_TRSM_STAMP(1 + 5 * P);
        // right-looking inside the panel
        trsm_static_for<NBP>([&](auto jc) {
            constexpr int J = decltype(jc)::value;
            constexpr int JG = 4 * P + J;                                // block index in the matrix
            // B operands of this block's updates, requested before the substitution: rows 16 JG + 4 sk + g of R
            double bfu[(NBP - J - 1) > 0 ? (NBP - J - 1) : 1][4];
#pragma unroll
            for (int L = J + 1; L < NBP; ++L)
#pragma unroll
                for (int sk = 0; sk < 4; ++sk) bfu[L - J - 1][sk] = R[(size_t)(JG * 16 + 4 * sk + g) * K16 + (4 * P + L) * 16 + e];
            // block J -> LDS in the row-per-lane layout, with its diagonal block of R
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int v = 0; v < 4; ++v) X[t * 16 + g + 4 * v][e] = acc[J][t][v];
#pragma unroll
            for (int v = 0; v < 4; ++v) Rd[g + 4 * v][e] = R[(size_t)(JG * 16 + g + 4 * v) * K16 + JG * 16 + e];
            trsm_wave_sync();
            if constexpr (P == 0 && J == 1) FSNAP_TRSM_STAMP(11);
            if (lane < 16) Rinv[lane] = 1.0 / Rd[lane][lane];      // (the reciprocals the host left on the diagonal of the block's
                                                                   // inverse, loaded instead: one more L2 round trip per block, 3 % slower)
            trsm_wave_sync();
            if constexpr (P == 0 && J == 1) FSNAP_TRSM_STAMP(12);
            const bool fast = row0 + 64 <= m && (JG < NB - 1 || JG * 16 + 16 <= K);      // wave-uniform
            {
                double x[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) x[j] = X[lane][j];
                // (R_JJ and the reciprocals from uniform addresses -- scalar loads, SGPR operands, no LDS read in the chain of the 16
                // steps -- made this block's substitution 3.3 -> 2.4 us and the pass 7 % SLOWER: the scalar loads' waits drain the LDS
                // queue and R does not fit the scalar cache; profiles/r06_trsm_trace.txt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const double q = x[i] * Rinv[i];
                    x[i] = q;
#pragma unroll
                    for (int j = i + 1; j < 16; ++j) x[j] = __builtin_fma(-q, Rd[i][j], x[j]);
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) X[lane][j] = x[j];
            }
            trsm_wave_sync();
            if constexpr (P == 0 && J == 1) FSNAP_TRSM_STAMP(13);
            // the solved block: out to Q in the accumulator layout (16 lanes = 128 contiguous bytes of a row; the lane's own row
            // straight from the registers -- eight 16-byte stores to 64 different rows per instruction -- measured 16 % slower),
            // and as the A operand of the updates inside the panel
            {
                const int col = JG * 16 + e;
                // (a tile inside the matrix and a block inside its K columns -- every tile but the last, every block but maybe
                // the last: one wave-uniform branch -- stores without the 16 row / column guards and their exec-mask branches:
                // the pass 0.595 -> 0.55 ms in place, 0.70 -> 0.56 ms first pass at 10^6 x 128)
                if (fast) {
                    double* qb0 = Q + (row0 + g) * ldq + col;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int v = 0; v < 4; ++v) qb0[(int64_t)(t * 16 + 4 * v) * ldq] = X[t * 16 + g + 4 * v][e];
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const int64_t r = row0 + t * 16 + g + 4 * v;
                            if (r < m && col < K) Q[r * ldq + col] = X[t * 16 + g + 4 * v][e];
                        }
                }
            }
            if constexpr (J + 1 < NBP) {
                double af[4][4];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int sk = 0; sk < 4; ++sk) af[t][sk] = -X[t * 16 + e][4 * sk + g];
                if constexpr (P == 0 && J == 1) FSNAP_TRSM_STAMP(14);
#pragma unroll
                for (int L = J + 1; L < NBP; ++L)
#pragma unroll
                    for (int sk = 0; sk < 4; ++sk)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            acc[L][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[t][sk], bfu[L - J - 1][sk], acc[L][t], 0, 0, 0);
            }
            trsm_wave_sync();    // X and Rd are reused by the next block
            FSNAP_TRSM_STAMP(2 + 5 * P + J);
        });
        // the panel's stores must have landed before the next panel reads them back as operands (same wave, same addresses)
        if constexpr (P + 1 < NP) __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0)
    });
    FSNAP_TRSM_STAMP(15);
}

// Kernel 13B (K > 128, the default there): the pass in PANELS of 128 columns with a 16 TR x 128 row tile per wave in the
// accumulators -- kernel 13 above re-reads every solved 16-column block once per LATER 16-column block (K / 32 times per
// row on average: 50 x at K = 1595) with four MFMAs between two L2 round trips, and reaches 14-23 % of the matrix peak.
// Here panel P (columns 128 P ..) of a row tile is
//     S_P   = sum_{I < P} Q_I R_IP          a GEMM: the solved panels of the wave's OWN rows stream from L2 / HBM once per
//                                           later panel (K / 256 times per row on average), 8 TR MFMAs per k-step against
//                                           a slab of KG rows of R that the four waves of the workgroup share through LDS
//                                           (direct-to-LDS loads, double buffered, one barrier per slab)
//     Q_P   = (X_P - S_P) R_PP^-1           right-looking inside the panel, block J of 16 columns at a time:
//                                             q0 = x T^-1,  r = x - q0 T,  q = q0 + r T^-1      (T = R_JJ, 12 MFMAs per tile)
//                                           with the 16 x 16 inverses of the diagonal blocks formed by the host next to R --
//                                           one step of iterative refinement on top of the multiplication by the inverse
//                                           brings the residual x - q T back to the eps |q| |T| of a substitution (the
//                                           diagonal blocks of a Jacobi-scaled Cholesky factor are benign: on the test
//                                           problems Q R_hat = A_w holds to 0.03 K eps either way, tests/test_rowspace_cpu.py).
//                                           The substitution it replaces (one row per lane, 16 dependent steps of 17 VALU
//                                           instructions) cost ~7 k cycles per block however few rows a wave holds: 60 % of
//                                           the kernel at both large-K shapes.  Then S_L += Q_J R_JL for the blocks right of
//                                           J in the panel
// Waves never exchange row data (each owns its rows from the first panel to the last); the barriers only order the shared
// slabs of R.  TR = 1 (16 rows per wave, 64 per workgroup) for short matrices -- 15 213 rows are 238 workgroups for 256
// CUs -- TR = 2 otherwise.  In place in the later passes: a panel of X is read in full before any of it is overwritten.
template <int TR, int KG, bool FIRST>
__global__ __launch_bounds__(256, 1) void fsnap_trsm_panel_k(const double* __restrict__ src, int64_t lds_,
                                                             const double* __restrict__ wpack, double* Q, int64_t ldq, int64_t m,
                                                             int K, const double* __restrict__ R, int K16) {
    constexpr int PW = 128, RS = PW + 4;          // row stride of a staged slab: the four k rows of a B operand (4 g + s) land on
                                                  // different bank groups (4 * 2 * RS = 32 mod 64 dwords), rows stay 16-byte aligned
    constexpr int ROWS = 16 * TR;
    constexpr int BPS = KG / 16;                  // 16-row blocks per slab
    constexpr int NKS = KG / 4;                   // k-steps per slab
    // two slabs of R in flight -- two SEPARATE arrays on purpose: the compiler orders LDS reads behind direct-to-LDS loads
    // per LDS variable.  With one array [2][KG][RS] every read of the current slab waited for the load of the NEXT one
    // (s_waitcnt vmcnt(0) in front of the first operand read of every slab and of every substitution step, which also
    // drains the operand prefetch and the stores of the solved block: the double buffering did nothing)
    __shared__ __attribute__((aligned(16))) double Rst0[KG][RS];
    __shared__ __attribute__((aligned(16))) double Rst1[KG][RS];
    __shared__ double Xs[4][ROWS][17];
    __shared__ double Ws[4][ROWS];                // first pass: the row weights of the wave's rows
    const int lane = threadIdx.x & 63, e = lane & 15, g = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wv) * ROWS;
    const int NBK = K16 >> 4;                     // 16-column blocks
    const int NP = (NBK + 7) >> 3;                // panels
    double (*X)[17] = Xs[wv];
    const double* __restrict__ Tinv = R + (size_t)K16 * K16;      // [NBK][16][16]: inverses of the diagonal blocks

    // one slab = rows k0 .. k0 + KG - 1 of R, columns c0 .. c0 + 16 nb - 1, into Rst[buf]: a direct-to-LDS load carries a
    // whole row (lane l: 16 bytes to row base + 16 l); the KG rows are dealt to the four waves
    auto stage = [&](auto bufc, int k0, int c0, int nb) {
        double (*dst)[RS] = decltype(bufc)::value ? Rst1 : Rst0;
        if (2 * lane < 16 * nb) {
#pragma unroll
            for (int i = 0; i < KG / 4; ++i) {
                const int kr = wv * (KG / 4) + i;
                if (k0 + kr < K16)                 // (a short last panel has fewer rows than a slab)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)(R + (size_t)(k0 + kr) * K16 + c0 + 2 * lane),
                        (__attribute__((address_space(3))) void*)&dst[kr][0], 16, 0, 0);
            }
        }
    };
    // A operands of one slab: for every 16 columns of Q a lane holds four adjacent doubles of its row (k-step s of lane
    // group g is column 4 g + s: both operands agree on that order), two 16-byte loads
    struct AOp {
        d2u lo[TR][BPS], hi[TR][BPS];
    };
    auto fetch_a = [&](AOp& a, int k0) {
#pragma unroll
        for (int t = 0; t < TR; ++t) {
            const int64_t r = row0 + t * 16 + e;
            const double* p = Q + (r < m ? r : 0) * ldq + k0 + 4 * g;
#pragma unroll
            for (int u = 0; u < BPS; ++u) {
                a.lo[t][u] = *reinterpret_cast<const d2u*>(p + 16 * u);
                a.hi[t][u] = *reinterpret_cast<const d2u*>(p + 16 * u + 2);
                if (r >= m) {
                    a.lo[t][u] = (d2u){0.0, 0.0};
                    a.hi[t][u] = (d2u){0.0, 0.0};
                }
            }
        }
    };
    // the same two requests one instruction at a time, for the slots between the MFMAs of a slab (issued as one block at the
    // top of a slab -- KG / 4 direct-to-LDS loads of ~100 cycles of issue each plus the row operands -- they kept the matrix
    // pipe idle for a quarter of the slab)
    auto stage_piece = [&](auto bufc, int k0, int c0, int nb, int i) {
        double (*dst)[RS] = decltype(bufc)::value ? Rst1 : Rst0;
        const int kr = wv * (KG / 4) + i;
        if (2 * lane < 16 * nb && k0 + kr < K16)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(R + (size_t)(k0 + kr) * K16 + c0 + 2 * lane),
                (__attribute__((address_space(3))) void*)&dst[kr][0], 16, 0, 0);
    };
    auto fetch_a_piece = [&](AOp& a, int k0, int t, int u) {
        const int64_t r = row0 + t * 16 + e;
        const double* p = Q + (r < m ? r : 0) * ldq + k0 + 4 * g + 16 * u;
        a.lo[t][u] = *reinterpret_cast<const d2u*>(p);
        a.hi[t][u] = *reinterpret_cast<const d2u*>(p + 2);
        if (r >= m) {
            a.lo[t][u] = (d2u){0.0, 0.0};
            a.hi[t][u] = (d2u){0.0, 0.0};
        }
    };
    // first pass: X = diag(w_eff) A; a zero weight (masked row, zero weight) gives a zero row of Q whatever the row of A
    // holds (NaN in masked rows is legal input).  The rows of a block are requested a block ahead and must not be TOUCHED
    // before they are used (an arithmetic instruction on them sits in the basic block of the request and waits for the
    // memory round trip there: the weighting inside the request cost 1.9 ms of a 4.6 ms pass at 367 900 x 480), so the
    // request brings the raw values and the weights wait in LDS (the wave's rows never change)
    if (FIRST) {
        if (lane < ROWS) {
            const int64_t r = row0 + lane;
            Ws[wv][lane] = r < m ? wpack[2 * r] : 0.0;
        }
        trsm_wave_sync();
    }
    // block (columns col0 .. col0 + 15) of the wave's rows in the accumulator layout (row g + 4 v of a tile, column e)
    auto load_x = [&](d4 (&xt)[TR], int col0) {
        const int col = col0 + e;
#pragma unroll
        for (int t = 0; t < TR; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int64_t r = row0 + t * 16 + g + 4 * v;
                double x = 0.0;
                if (r < m && col < K) x = FIRST ? src[r * lds_ + col] : Q[r * ldq + col];   // (in place: the pointer that writes)
                xt[t][v] = x;
            }
    };
    // B operand of the inverse of diagonal block jb: k-step s of lane group g is row 4 s + g
    auto load_inv = [&](double (&bi)[4], int jb) {
#pragma unroll
        for (int sk = 0; sk < 4; ++sk) bi[sk] = Tinv[(size_t)jb * 256 + (4 * sk + g) * 16 + e];
    };
    // a 16 TR x 16 block from the accumulator layout (row g + 4 v, column e) to the A-operand layout (row e, column
    // 4 s + g) through the wave's own LDS tile; sign = -1 hands back the negated block
    auto to_a_operand = [&](const d4 (&d)[TR], double (&a)[TR][4], double sign) {
        trsm_wave_sync();                         // earlier reads of the tile are done
#pragma unroll
        for (int t = 0; t < TR; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) X[t * 16 + g + 4 * v][e] = d[t][v];
        trsm_wave_sync();
#pragma unroll
        for (int t = 0; t < TR; ++t)
#pragma unroll
            for (int sk = 0; sk < 4; ++sk) a[t][sk] = sign * X[t * 16 + e][4 * sk + g];
    };
    // Results of a chain of fp64 MFMAs that are read right away.  Seen on gfx950 / ROCm 7.2: the compiler left too few wait
    // states between the last v_mfma_f64_16x16x4 of a chain and the read of its result when little else separated them (its
    // table seems to carry the 8-pass timing of the older parts; the instruction takes 16 passes here) -- K16 = 208: the last
    // block of the last panel came out without the update of the block before it, error 3e-2, and any change of the
    // surrounding schedule hid it.  The tile passes through an asm statement in its accumulation registers: whatever copies
    // it to VGPRs cannot be placed in front of the 32 extra wait states.
    auto mfma_settle = [&](d4 (&d)[TR]) {
#pragma unroll
        for (int t = 0; t < TR; ++t) asm volatile("s_nop 15\n\ts_nop 15" : "+a"(d[t]));
    };
    auto store_q = [&](const d4 (&q)[TR], int col0) {
        const int col = col0 + e;
#pragma unroll
        for (int t = 0; t < TR; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int64_t r = row0 + t * 16 + g + 4 * v;
                if (r < m && col < K) Q[r * ldq + col] = q[t][v];
            }
    };

    for (int P = 0; P < NP; ++P) {
        const int c0 = P * PW;
        const int nb = (NBK - 8 * P) < 8 ? (NBK - 8 * P) : 8;          // blocks of this panel (the last one may be short)
        d4 acc[TR][8];
#pragma unroll
        for (int t = 0; t < TR; ++t)
#pragma unroll
            for (int L = 0; L < 8; ++L) acc[t][L] = (d4){0.0, 0.0, 0.0, 0.0};
        d4 xt[TR];
        double bi[4];                                     // inverse of the next diagonal block, one block ahead like xt
        // ---- S_P: all solved panels against their rows of R, slab by slab ------------------------------------------------
        const int nslab = c0 / KG;                        // (even: 128 / KG slabs per panel)
        using B0 = std::integral_constant<int, 0>;
        using B1 = std::integral_constant<int, 1>;
        if (P > 0) __syncthreads();                       // every wave has left the slabs of the previous panel
        if (nslab == 0) {
            stage(B0{}, c0, c0, nb);
            load_x(xt, c0);
            load_inv(bi, 8 * P);
        } else {
            AOp a0, a1;
            stage(B0{}, 0, c0, nb);
            fetch_a(a0, 0);
            auto slab = [&](auto full, auto par, int sidx, AOp& cur, AOp& nxt) {
                constexpr bool FULL = decltype(full)::value;        // all 8 blocks: no per-block predicate in the MFMA loop
                constexpr int PAR = decltype(par)::value;           // slab sidx lives in Rst<PAR>, the next one goes to the other
                using OTHER = std::integral_constant<int, 1 - PAR>;
                // the slab's rows of R (all four waves' pieces) and this wave's operands have landed
                __builtin_amdgcn_s_waitcnt(0x0f70);       // vmcnt(0)
                __syncthreads();
                // what the next slab needs is requested piece by piece between the MFMAs below: its rows of R into the other
                // buffer and this wave's row operands -- or, behind the last slab of the GEMM (an odd one: it lives in Rst1),
                // the first slab of R_PP, the first block of the panel's own rows and its inverse block
                const bool more = sidx + 1 < nslab;
                const int next_k0 = more ? (sidx + 1) * KG : c0;
                if (!more) {
                    load_x(xt, c0);
                    load_inv(bi, 8 * P);
                }
                const double (*Rb)[RS] = PAR ? Rst1 : Rst0;
                // B operands one k-step ahead of the MFMAs that use them (the wave may be alone on its SIMD: an LDS round
                // trip in front of every eight MFMAs is paid in full)
                double bc[8], bn[8];
                auto loadb = [&](double (&bb)[8], int ks) {
#pragma unroll
                    for (int L = 0; L < 8; ++L)
                        if (FULL || L < nb) bb[L] = Rb[16 * (ks >> 2) + 4 * g + (ks & 3)][16 * L + e];
                };
                constexpr bool AHEAD = TR <= 2;          // (TR = 4: a B operand feeds four MFMAs, 256 cycles of matrix pipe: the
                                                         // next one arrives meanwhile; no registers for a second set)
                loadb(bc, 0);
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    if (AHEAD) {
                        if (ks + 1 < NKS) loadb(bn, ks + 1);
                    } else if (ks > 0) {
                        loadb(bc, ks);
                    }
                    // the pieces of the requests go out in the FIRST half of the slab, two rows of R per k-step (NKS = KG / 4
                    // k-steps, KG / 4 rows per wave): the wait at the top of the next slab then finds them at least half a
                    // slab old
                    if (2 * ks < NKS) {
                        stage_piece(OTHER{}, next_k0, c0, nb, 2 * ks);
                        stage_piece(OTHER{}, next_k0, c0, nb, 2 * ks + 1);
                    }
                    if (more && ks < TR * BPS) fetch_a_piece(nxt, next_k0, ks / BPS, ks % BPS);
                    if (AHEAD) __builtin_amdgcn_sched_barrier(0);
                    const int u = ks >> 2, sk = ks & 3;
#pragma unroll
                    for (int L = 0; L < 8; ++L) {
                        if (FULL || L < nb) {
#pragma unroll
                            for (int t = 0; t < TR; ++t) {
                                const double af = (sk < 2) ? cur.lo[t][u][sk & 1] : cur.hi[t][u][sk & 1];
                                acc[t][L] = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bc[L], acc[t][L], 0, 0, 0);
                            }
                        }
                    }
                    if (AHEAD) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int L = 0; L < 8; ++L) bc[L] = bn[L];
                    }
                }
            };
            if (nb == 8) {
                for (int sidx = 0; sidx < nslab; sidx += 2) {
                    slab(std::true_type{}, B0{}, sidx, a0, a1);
                    slab(std::true_type{}, B1{}, sidx + 1, a1, a0);
                }
            } else {
                for (int sidx = 0; sidx < nslab; sidx += 2) {
                    slab(std::false_type{}, B0{}, sidx, a0, a1);
                    slab(std::false_type{}, B1{}, sidx + 1, a1, a0);
                }
            }
        }
        // ---- the panel's own triangle, block by block --------------------------------------------------------------------
        // rows 16 J .. of R_PP come in slabs of KG rows as well (BPS blocks per slab).  Per block: x = X_J - S_J, the stores of
        // block J - 1 and the loads of block J + 1 (rows and inverse block) go out -- both are waited for a whole block
        // later, at the next slab boundary or first use: nothing in this loop waits for a memory round trip it has just
        // started --, q = x T^-1 refined once, S_L += Q_J R_JL for the blocks right of J.  Every hand-over between the
        // accumulator layout of a result and the A-operand layout of the next product goes through the wave's own LDS tile.
        d4 qprev[TR];
        trsm_static_for<8>([&](auto Jc) {          // (compile-time J: accumulator tiles and slab buffers are named statically)
            constexpr int J = decltype(Jc)::value;
            if (J >= nb) return;
            constexpr int sl = J / BPS, jr = (J % BPS) * 16;  // slab of this block (slab sl of R_PP lives in Rst<sl & 1>), its first row inside the slab
            using NEXTBUF = std::integral_constant<int, (sl + 1) & 1>;
            // x = X_J - S_J (the last MFMAs of the previous block wrote tile J: settle before the read)
            d4 xd[TR];
#pragma unroll
            for (int t = 0; t < TR; ++t) asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[t][J]));
#pragma unroll
            for (int t = 0; t < TR; ++t) {
                if (FIRST) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const double wgt = Ws[wv][t * 16 + g + 4 * v];
                        xd[t][v] = ((wgt != 0.0) ? wgt * xt[t][v] : 0.0) - acc[t][J][v];
                    }
                } else {
                    xd[t] = xt[t] - acc[t][J];
                }
            }
            double xa[TR][4];
            to_a_operand(xd, xa, 1.0);
            if (jr == 0) {
                __builtin_amdgcn_s_waitcnt(0x0f70);       // this slab's rows of R (the other waits are a block old)
                __syncthreads();
                if ((sl + 1) * BPS < nb) stage(NEXTBUF{}, c0 + (sl + 1) * KG, c0, nb);
            }
            if (J > 0) store_q(qprev, c0 + 16 * (J - 1));
            double binv[4];
#pragma unroll
            for (int sk = 0; sk < 4; ++sk) binv[sk] = bi[sk];
            if (J + 1 < nb) {
                load_x(xt, c0 + 16 * (J + 1));
                load_inv(bi, 8 * P + J + 1);
            }
            const double (*Rb)[RS] = (sl & 1) ? Rst1 : Rst0;
            double bt[4];
#pragma unroll
            for (int sk = 0; sk < 4; ++sk) bt[sk] = Rb[jr + 4 * sk + g][16 * J + e];
            // q0 = x T^-1
            d4 q0[TR];
#pragma unroll
            for (int t = 0; t < TR; ++t) q0[t] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int sk = 0; sk < 4; ++sk)
#pragma unroll
                for (int t = 0; t < TR; ++t) q0[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[t][sk], binv[sk], q0[t], 0, 0, 0);
            mfma_settle(q0);
            // r = x - q0 T
            double qa[TR][4];
            to_a_operand(q0, qa, -1.0);
            d4 rr[TR];
#pragma unroll
            for (int t = 0; t < TR; ++t) rr[t] = xd[t];
#pragma unroll
            for (int sk = 0; sk < 4; ++sk)
#pragma unroll
                for (int t = 0; t < TR; ++t) rr[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[t][sk], bt[sk], rr[t], 0, 0, 0);
            mfma_settle(rr);
            // q = q0 + r T^-1
            double ra[TR][4];
            to_a_operand(rr, ra, 1.0);
#pragma unroll
            for (int sk = 0; sk < 4; ++sk)
#pragma unroll
                for (int t = 0; t < TR; ++t) q0[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[t][sk], binv[sk], q0[t], 0, 0, 0);
            mfma_settle(q0);
            // the solved block: kept in the accumulator layout for its (late) store, and as the A operand of the blocks right
            // of it
#pragma unroll
            for (int t = 0; t < TR; ++t) qprev[t] = q0[t];
            if (J + 1 < nb) {
                double af[TR][4];
                to_a_operand(q0, af, 1.0);
#pragma unroll
                for (int L = 1; L < 8; ++L) {
                    if (L > J && L < nb) {
#pragma unroll
                        for (int sk = 0; sk < 4; ++sk) {
                            const double bf = Rb[jr + 4 * sk + g][16 * L + e];
#pragma unroll
                            for (int t = 0; t < TR; ++t) acc[t][L] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[t][sk], bf, acc[t][L], 0, 0, 0);
                        }
                    }
                }
            }
        });
        store_q(qprev, c0 + 16 * (nb - 1));
        // the panel just stored is an MFMA operand of the next panels -- of this wave's own rows only.  Its blocks are
        // fetched again in the LAST slabs of the next panel's GEMM, behind at least one vmcnt(0) (a slab is at most 32
        // columns, a panel 128: every panel has >= 4 slabs), so no wait is needed here.
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
    // K <= 128: kernel 13C (64-column panels, two waves per SIMD) for every pass.  tools/trsm_check, 10^6 rows, round 5 (kernel 13A,
    // one wave per SIMD with the whole 64 x K tile -- gone since -- / 13C, ms): in-place passes K = 128 0.737 / 0.590, 110 0.642 / 0.535,
    // 96 0.517 / 0.383, 64 0.249 / 0.201, 31 0.135 / 0.108; first pass 128 0.765 / 0.709, 110 0.690 / 0.701, 96 0.546 / 0.479
    // Round 6 (unguarded loads / stores of tiles inside the matrix, B operands of a left-looking block row at once): in place 128
    // 0.524, 110 0.52 ... 0.53, 96 0.385; first pass 128 0.54 ... 0.55, 96 0.38 ... 0.40 (profiles/r06_trsm_trace.txt)
    // Nine blocks (129 ... 144 columns: the ACE width 142 of examples/Ta_PACE_RIDGE) since round 6, from 32 768 rows on: 10^6 x 142
    // 1.31 -> 0.86 ms in place, 1.45 -> 0.92 ms first pass, 10^6 x 144 1.31 -> 0.70 ms (rows of 142 doubles straddle the 128-byte
    // lines); a short system has too few 64-row tiles for one wave each (13 035 x 142: 0.047 against kernel 13B's 0.032 ms)
    if (K16 <= 128 || (K16 == 144 && m >= 64 * 512)) {
        const dim3 grid((unsigned)nb), block(64);
#define FSNAP_TRSM_ACC2(NBV)                                                                                                 \
    case NBV:                                                                                                                \
        if (wpack) hipLaunchKernelGGL((fsnap_trsm_acc2_k<NBV, true>), grid, block, 0, st, src, lds, wpack, Q, ldq, m, K, R);   \
        else hipLaunchKernelGGL((fsnap_trsm_acc2_k<NBV, false>), grid, block, 0, st, src, lds, wpack, Q, ldq, m, K, R);        \
        break;
        switch (K16 / 16) {
            FSNAP_TRSM_ACC2(1) FSNAP_TRSM_ACC2(2) FSNAP_TRSM_ACC2(3) FSNAP_TRSM_ACC2(4)
            FSNAP_TRSM_ACC2(5) FSNAP_TRSM_ACC2(6) FSNAP_TRSM_ACC2(7) FSNAP_TRSM_ACC2(8) FSNAP_TRSM_ACC2(9)
            default: return hipErrorInvalidValue;
        }
#undef FSNAP_TRSM_ACC2
        return hipGetLastError();
    }
    // kernel 13B (panels of 128 columns)
    {

        // one wave per SIMD (the row tile in the accumulation registers, everything else hand-pipelined): 16 rows per wave
        // while that still leaves fewer than two workgroups per CU, 32 rows otherwise (a 64-row tile -- 256 accumulation registers -- does not leave the 256 VGPRs the rest needs)
        const int tr = m < 64 * 512 ? 1 : 2;
#define FSNAP_TRSM_PANEL(TRV, KGV)                                                                                               \
    {                                                                                                                            \
        const dim3 grid((unsigned)((m + 64 * TRV - 1) / (64 * TRV))), block(256);                                                \
        if (wpack) hipLaunchKernelGGL((fsnap_trsm_panel_k<TRV, KGV, true>), grid, block, 0, st, src, lds, wpack, Q, ldq, m, K, R, K16); \
        else hipLaunchKernelGGL((fsnap_trsm_panel_k<TRV, KGV, false>), grid, block, 0, st, src, lds, wpack, Q, ldq, m, K, R, K16);      \
    }
        // slab height KG (rows of R staged per barrier): 64 for the 16-row tiles, 16 for the 32-row tiles (measured: 15 213 x
        // 1 595 1.13 ms with (1, 64), 1.16 with (1, 32); 367 900 x 480 2.58 ms with (2, 16), 2.69 with (2, 32), whose first-pass
        // variant also spills)
        if (tr == 1) FSNAP_TRSM_PANEL(1, 64)
        else FSNAP_TRSM_PANEL(2, 16)
#undef FSNAP_TRSM_PANEL
        return hipGetLastError();
    }
}

hipError_t launch_qpack(const double* wpack, int64_t m, double* qpack, hipStream_t st) {
    const int64_t nb = (m + 255) / 256;
    if (nb > 0x7FFFFFFF) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fsnap_qpack_k, dim3((unsigned)nb), dim3(256), 0, st, wpack, m, qpack);
    return hipGetLastError();
}

}  // namespace fsnap
