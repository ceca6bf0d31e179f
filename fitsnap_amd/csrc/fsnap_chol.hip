// fsnap_chol.hip — K x K Cholesky solves of the packed statistics on the GPU (gfx950 only):
//   8a-8f  blocked solve for K >= 232 (scaling, 64-row panels, four-wave diagonal block, MFMA trailing update, blocked sweeps)
// Same algorithm as the host fast path in fsnap_solve.cpp (Jacobi-scaled Cholesky, pivots checked by the caller).
#include "fsnap_device_common.h"
#include "fsnap_kernels.h"

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

// probe p (1 .. 31) of the condition estimate at a row: see fsnap_chol_probe_gram_k
namespace fsnap {
__host__ __device__ inline double chol_probe_value(int row, int p) {
    unsigned x = (unsigned)row * 0x9E3779B1u + (unsigned)p * 0x85EBCA77u + 0x27D4EB2Fu;
    x ^= x >> 15;
    x *= 0x2C1B3C6Du;
    x ^= x >> 12;
    x *= 0x297A2D39u;
    x ^= x >> 15;
    const double mag = 0.25 + 0.75 * ((double)(x & 0xFFFFu) / 65536.0);       // exact in fp64: the host forms B^T B from the same values
    return (x & 0x10000u) ? -mag : mag;
}
}  // namespace fsnap


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
    double tile[3][4][64];     // U_01, U_02, U_12 handed from wave b to the waves right of it (accumulator layout, lane for lane)
    int tflag[4];              // [0] U_01, [1] U_02, [2] U_12 published
    double T[4][16][17];       // per-wave transpose scratch of the inverses
};

constexpr int CHOL_D4_SPINS = 1 << 17;        // polls of one hand-off before a wave gives up (~10 ms)

__device__ __forceinline__ int diag4_tile_index(int a, int b) { return a == 0 ? b - 1 : 2; }     // (0,1) (0,2) (1,2)

// clears the hand-off markers; every wave of the workgroup calls it, then ONE __syncthreads() before the first use
__device__ __forceinline__ void diag4_lds_reset(Diag4Lds& L, int tid) {
    if (tid < 64) L.inv[tid] = 0.0;
    if (tid < 4) L.tflag[tid] = 0;
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
        __builtin_amdgcn_s_sleep(1);                       // the owner shares the LDS queue with three pollers
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

// Y_A = U_AA^-1: the published row operations of block A replayed on an identity tile (Z = U_AA^-T), transposed through
// this wave's scratch, stored for kernels 8c / 8e
template <int A>
__device__ __forceinline__ void diag4_inverse(Diag4Lds& L, double (*T)[17], double* __restrict__ Y, int e, int kr, bool& ok) {
    d4 Z;
#pragma unroll
    for (int r = 0; r < 4; ++r) Z[r] = (4 * r + kr == e) ? 1.0 : 0.0;
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
        if (a == 0) diag4_consume<0>(X, L, e, kr, ok);
        else if (a == 1) diag4_consume<1>(X, L, e, kr, ok);
        else diag4_consume<2>(X, L, e, kr, ok);
        if (a + 1 == W) {
            // next owner: its diagonal tile needs nothing but its own registers -- first thing after the last pivot, as two
            // chains of two MFMAs (a chain of four on one accumulator is 4 x 65 cycles on the hand-over)
            d4 half = {0.0, 0.0, 0.0, 0.0};
            Tl[W] = __builtin_amdgcn_mfma_f64_16x16x4f64(-X[0], X[0], Tl[W], 0, 0, 0);
            half = __builtin_amdgcn_mfma_f64_16x16x4f64(-X[1], X[1], half, 0, 0, 0);
            Tl[W] = __builtin_amdgcn_mfma_f64_16x16x4f64(-X[2], X[2], Tl[W], 0, 0, 0);
            half = __builtin_amdgcn_mfma_f64_16x16x4f64(-X[3], X[3], half, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) Tl[W][r] += half[r];
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
        if (a + 1 < W) {
#pragma unroll
            for (int r = 0; r < 4; ++r) S[(size_t)(jb + 16 * a + 4 * r + kr) * ld + jb + 16 * W + e] = X[r];
        } else {
            held = X;
        }
        CHOL_STAMP(W, 1 + a);
    }
    CHOL_STAMP(W, 4);
    // ---- the own block step: owner ------------------------------------------------------------------------------
    // (Round 6 tried the tile on the VALU instead, one COLUMN per lane -- sixteen registers per lane, pivot j = one multiply,
    // the update of row i = a v_readlane pair for U[j][i] + one FMA, the chain multiply -> readlane -> FMA -> readlane -> 1 / sqrt:
    // bit-for-bit as accurate and NOT faster, 5 100-5 400 cycles per owner phase against 5 000-5 300 for the form below
    // (tools/chol_pipeline_check -DFSNAP_CHOL_TRACE, profiles/r06_chol_valu_diag.txt): the 120 broadcast + FMA pairs of a
    // 16 x 16 tile cost ~18 cycles each in issue slots and SGPR wait states, and a wave issues them in order with the chain.)
    d4& D = Tl[W];
    double pmin = 1.0e300, psum = 0.0;
    double dcur = readlane_f64(D[0], 0);
    double inv = rsqrt_newton2(dcur);
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
            inv = rsqrt_newton2(dcur);
        }
    }
    CHOL_STAMP(W, 5);
    // the strip of U: the tiles above the diagonal (kept in registers until here: their stores are off the hand-over), then
    // the diagonal tile's upper triangle
    // (storing Tl[a] itself here, for every a < W, produced stale tiles in the fused launches -- the values before the row
    // operations -- although the same code was right in the stand-alone kernel: tools/chol_pipeline_check.hip, round 5)
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

// (S here is where the FACTOR goes: the work matrix itself in the in-place forms, the second matrix in the one-launch form)
__device__ __forceinline__ void chol_diag4_dispatch(d4 (&Tl)[4], Diag4Lds& L, double* S, int ld, int jb,
                                                    double* __restrict__ Y, int* __restrict__ status,
                                                    double* __restrict__ minpiv, int wave, int lane) {
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
        // strip: column 0 the scaled right-hand side; columns 1 .. 31 the PROBE vectors of the condition estimate (fixed
        // pseudo-random values in +-[0.25, 1]): the factorisation carries them along for nothing, and what comes out --
        // Z = U^-T B -- gives B^T S^-1 B = Z^T Z without a single extra sweep (fsnap_chol_probe_gram_k)
        v = (i < n && oki && finite_c) ? (j == np ? ci * di : fsnap::chol_probe_value(i, j - np)) : 0.0;
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
// Condition estimate of the device factor without a sweep (round 6).  The right-hand-side strip of the factorisation is 32
// columns wide and only its first column is the right-hand side: columns 1 .. 31 carry fixed pseudo-random probe vectors B
// (fsnap_chol_prepare_diag4_k), the panel loop transforms them with everything else, and at the end the strip holds
// Z = U^-T B.  Z^T Z = B^T S^-1 B is the Rayleigh-Ritz matrix of S^-1 on the 31-dimensional random subspace span(B): its
// largest generalised eigenvalue theta (against B^T B; host, 31 x 31) never exceeds 1 / lambda_min(S) and captures at least
// ~0.4 x 31 / K of it (the share of ANY fixed direction in a random 31-subspace of K dimensions: Beta(31 / 2, (K - 31) / 2),
// mean 31 / K, three standard deviations below it ~0.4 of that).  So 1 / theta is an estimate of lambda_min from ABOVE that is
// at most ~K / 12 too large -- the caller scales it by that (fsnap_solve_device_rhs) -- at the price of ONE small launch, where
// the Lanczos sweeps with the device factor that this replaces cost 0.06 / 0.10 / 0.40 ms EACH at K = 256 / 480 / 1 595 and a
// well-conditioned 15 213 x 1 595 SVD fit went from 1.3 to 3.5 ms.
// The strip rows of the LAST panel are never substituted by the panel loop (there is no launch behind the last panel): this
// kernel runs their forward substitution itself, like fsnap_chol_backsolve_k does for the right-hand side; then Z^T Z on the
// matrix pipe (a 4-row chunk of the strip = one register per 16-column half, three MFMAs), eight waves over the rows.
// out[(i - 1) * 31 + (j - 1)] = sum_r Z[r][i] Z[r][j], i, j = 1 .. 31; fixed summation order.
// ---------------------------------------------------------------------------------
constexpr int CHOL_NPROBE = CHOL_XS - 1;

__global__ __launch_bounds__(512) void fsnap_chol_probe_gram_k(const double* __restrict__ Uf, const double* __restrict__ Sraw, int ld,
                                                              int np, double* __restrict__ out, const int* __restrict__ status) {
    __shared__ double U11[CHOL_NB * (CHOL_NB + 1)];
    __shared__ double Zl[CHOL_NB][CHOL_XS];
    __shared__ double red[8][3][4][64];                // the waves' partial tiles (0,0), (0,1), (1,1) in the accumulator layout
    if (*status) return;                       // the factorisation failed: there is no factor to ask (the host does not look)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, e = lane & 15, kr = lane >> 4;
    const int r0 = np - CHOL_NB;
    for (int t = tid; t < CHOL_NB * CHOL_NB; t += 512) U11[(t >> 6) * (CHOL_NB + 1) + (t & 63)] = Uf[(size_t)(r0 + (t >> 6)) * ld + r0 + (t & 63)];
    __syncthreads();
    // y = U11^-T s for the last panel's raw strip rows (all 32 columns: column 0 is the right-hand side, whose products land in
    // row / column 0 of the Gram matrix and are not handed out): one wave per column (wave wv: columns wv, wv + 8, ...), 64 steps,
    // multipliers broadcast with v_readlane
    for (int c = wv; c < CHOL_XS; c += 8) {
        double v = Sraw[(size_t)(r0 + lane) * ld + np + c];
        const double invd = 1.0 / U11[lane * (CHOL_NB + 1) + lane];
#pragma unroll 8
        for (int k = 0; k < CHOL_NB; ++k) {
            const double yk = readlane_f64(v, k) * readlane_f64(invd, k);
            if (lane == k) v = yk;
            if (lane > k) v = __builtin_fma(-U11[k * (CHOL_NB + 1) + lane], yk, v);
        }
        Zl[lane][c] = v;
    }
    __syncthreads();
    // Z^T Z on the matrix pipe: a 4-row chunk of the strip is one register per 16-column half -- A operand (column e, row kr) and B
    // operand (row kr, column e) at once, kernel 1's operand trick --, three MFMAs per chunk for the tiles (0,0), (0,1), (1,1);
    // wave wv takes the chunks wv, wv + 8, ...; eight loads in flight per lane
    d4 t00 = {0.0, 0.0, 0.0, 0.0}, t01 = t00, t11 = t00;
    const int nchunk = np / 4;
    for (int c0 = wv; c0 < nchunk; c0 += 8 * 4) {
        double h0[4], h1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = 4 * (c0 + 8 * u) + kr;
            if (row < r0) {
                h0[u] = Uf[(size_t)row * ld + np + e];
                h1[u] = Uf[(size_t)row * ld + np + 16 + e];
            } else if (row < np) {
                h0[u] = Zl[row - r0][e];
                h1[u] = Zl[row - r0][16 + e];
            } else {
                h0[u] = h1[u] = 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            t00 = __builtin_amdgcn_mfma_f64_16x16x4f64(h0[u], h0[u], t00, 0, 0, 0);
            t01 = __builtin_amdgcn_mfma_f64_16x16x4f64(h0[u], h1[u], t01, 0, 0, 0);
            t11 = __builtin_amdgcn_mfma_f64_16x16x4f64(h1[u], h1[u], t11, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[wv][0][r][lane] = t00[r];
        red[wv][1][r][lane] = t01[r];
        red[wv][2][r][lane] = t11[r];
    }
    __syncthreads();
    // fixed-order sum over the eight waves; element (tile, r, lane) = entry (16 a + 4 r + kr, 16 b + e) of the 32 x 32 Gram matrix
    for (int t = tid; t < 3 * 256; t += 512) {
        const int tile = t >> 8, r = (t >> 6) & 3, ln = t & 63;
        double acc = red[0][tile][r][ln];
#pragma unroll
        for (int w = 1; w < 8; ++w) acc += red[w][tile][r][ln];
        const int i = (tile == 2 ? 16 : 0) + 4 * r + (ln >> 4), j = (tile == 0 ? 0 : 16) + (ln & 15);
        if (i >= 1 && j >= 1) {
            out[(i - 1) * CHOL_NPROBE + (j - 1)] = acc;
            if (tile == 1) out[(j - 1) * CHOL_NPROBE + (i - 1)] = acc;      // the (1,0) tile is the transpose of (0,1)
        }
    }
}

// ---------------------------------------------------------------------------------
// host-side launchers (C++ linkage, used by fsnap_capi.cpp)
// ---------------------------------------------------------------------------------
namespace fsnap {

double chol_probe(int row, int p) { return chol_probe_value(row, p); }

size_t chol_large_work_doubles(int n) {
    const size_t np = (size_t)(n + CHOL_NB - 1) / CHOL_NB * CHOL_NB;
    // work matrix + strip, Y blocks of every panel, hand-off word, the factor matrix of the one-launch-per-panel form
    return 2 * np * (np + CHOL_XS) + (np / CHOL_NB) * 1024 + 8;
}

// The panel loop has ONE form (the forms of rounds 2-4 -- two launches per panel, the single-wave diagonal block with three
// pivot chains, the flag-synchronised fused launch -- and round 5's rank-4 / redundant diagonal blocks lost their A/B runs:
// profiles/r04_chol_fused_ab.txt, r05_chol_forms.txt): scaling + first diagonal block in one launch, then ONE launch per panel
// (kernel 8s: every wave substitutes the row tails it needs itself, diagonal block on four waves, kernel 8b4), the factor in a
// second matrix behind the Y blocks and the hand-off word.
static double* chol_factor_matrix(double* work, int np) {
    const size_t ld = (size_t)np + CHOL_XS;
    return work + (size_t)np * ld + (size_t)(np / CHOL_NB) * 1024 + 8;
}

// the panel loop behind the first diagonal block: per panel ONE launch (row tails of panel pb, update behind it, diagonal
// block of panel pb + 1); S -> Uf
static void launch_chol_panels(double* S, double* Uf, int ld, int np, double* Yall, int* status, double* minpiv, hipStream_t st) {
    const int npanel = np / CHOL_NB;
    for (int pb = 0; pb < npanel; ++pb) {
        const int jb = pb * CHOL_NB;
        double* Y = Yall + (size_t)pb * 1024;
        const int ntail = np - jb - CHOL_NB;
        const int nblk = ntail / 32;
        if (ntail > 0) {
            const int nrest = nblk * (nblk + 1) / 2 + nblk - 3;
            hipLaunchKernelGGL(fsnap_chol_step4_k, dim3(1 + (nrest + 3) / 4), dim3(256), 0, st, S, Uf, ld, jb, nblk, status,
                               (const double*)Y, Y + 1024, minpiv);
        }
        // (behind the LAST panel there is nothing to launch: the forward substitution of its right-hand-side rows is the
        // first thing the back substitution does, fsnap_chol_backsolve_k with Sraw; the factor-only use has no strip)
    }
}

hipError_t launch_chol_large(const double* packed, const double* cvec, int n, double alpha, double* work, double* dsc, double* z,
                             double* beta, int* status, double* minpiv, double* host_out, bool clear_status, double* probe_out,
                             hipStream_t st) {
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
    double* Uf = chol_factor_matrix(work, np);
    {
        // scaling + first diagonal block in one launch (workgroup 0: the four-wave pipeline)
        const int nx = (ld + 255) / 256;
        hipLaunchKernelGGL(fsnap_chol_prepare_diag4_k, dim3((unsigned)(1 + np * nx)), dim3(256), 0, st, packed, cvec, n, np, alpha, dsc,
                           S, Uf, Yall, status, minpiv, npanel, flag);
    }
    launch_chol_panels(S, Uf, ld, np, Yall, status, minpiv, st);
    // the probes' Rayleigh-Ritz matrix (condition estimate), before the back substitution's last launch hands the results over
    if (probe_out)
        hipLaunchKernelGGL(fsnap_chol_probe_gram_k, dim3(1), dim3(512), 0, st, (const double*)Uf, (const double*)S, ld, np, probe_out,
                           (const int*)status);
    static bool bs_attr_set = false;
    if (!bs_attr_set) {
        e = hipFuncSetAttribute((const void*)fsnap_chol_backsolve_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CHOL_BS_LDS);
        if (e != hipSuccess) return e;
        bs_attr_set = true;
    }
    for (int hi = npanel; hi > 0; hi -= CHOL_BS_MACRO) {
        const int lo = hi > CHOL_BS_MACRO ? hi - CHOL_BS_MACRO : 0;
        hipLaunchKernelGGL(fsnap_chol_backsolve_k, dim3(1), dim3(1024), CHOL_BS_LDS, st, (const double*)Uf, ld, np, n, Yall, z, dsc,
                           beta, status, lo, hi, hi == npanel ? 1 : 0, minpiv, host_out, (const double*)S);
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
                               const double* minpiv, double* host_out, hipStream_t st) {
    const int np = (n + CHOL_NB - 1) / CHOL_NB * CHOL_NB, npanel = np / CHOL_NB, ld = np + CHOL_XS;
    double* Yall = work + (size_t)np * ld;
    const double* Uf = chol_factor_matrix(work, np);
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
                              int K16, double* Rout, hipStream_t st) {
    const int np = (n + CHOL_NB - 1) / CHOL_NB * CHOL_NB, npanel = np / CHOL_NB, ld = np + CHOL_XS;
    double* S = work;
    double* Yall = work + (size_t)np * ld;
    int* flag = (int*)(Yall + (size_t)npanel * 1024);      // hand-off word of the fused panel launches
    hipError_t e = hipMemsetAsync(status, 0, sizeof(int), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fsnap_chol_factor_prepare_d_k, dim3((np + 255) / 256), dim3(256), 0, st, G, n, np, dsc, status, minpiv, npanel);
    hipLaunchKernelGGL(fsnap_chol_factor_prepare_s_k, dim3((ld + 255) / 256, np), dim3(256), 0, st, G, n, np, shift, dsc, S, status);
    double* Uf = chol_factor_matrix(work, np);
    hipLaunchKernelGGL(fsnap_chol_diag4_k, dim3(1), dim3(256), 0, st, (const double*)S, Uf, ld, 0, Yall, status, minpiv, flag);
    launch_chol_panels(S, Uf, ld, np, Yall, status, minpiv, st);  // (the strip is carried along as in the solve: zero here)
    const int64_t total = (int64_t)K16 * K16 + (int64_t)K16 * 16;
    hipLaunchKernelGGL(fsnap_chol_extract_factor_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const double*)Uf, Yall,
                       dsc, n, np, K16, Rout);
    return hipGetLastError();
}

hipError_t launch_gram_scan(const double* G, int n, double* out, hipStream_t st) {
    hipLaunchKernelGGL(fsnap_gram_scan_k, dim3((unsigned)n), dim3(256), 0, st, G, n, out);
    return hipGetLastError();
}

}  // namespace fsnap
