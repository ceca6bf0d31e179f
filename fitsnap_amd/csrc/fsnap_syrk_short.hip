// fsnap_syrk_short.hip — kernel 1S, `fsnap_syrk_short<NB>`: the statistics kernel of SHORT systems of 81 ... 144 columns
// (BASELINE configs[3]'s own shape, examples/Ta_PACE_RIDGE: 13 035 x 142).  gfx950 only (wave64, v_mfma_f64_16x16x4_f64).
//
// What it replaces in the reference is what every SYRK kernel of this library replaces (fsnap_syrk.hip's header):
//   fitsnap3lib/solvers/svd.py:35-51, ridge.py:28-43, lib/ridge_solver/regressor.py:11-12   aw = w * a[train], G = aw^T aw, c = aw^T bw
//
// Why another kernel.  Kernel 1A gives every wave the WHOLE tile triangle (45 tiles at 142 columns) and a row range of
// its own, and needs >= 12 chunks per wave to pay for the 92 KiB partial triangle of its workgroup: 13 035 rows fill 68
// workgroups = 272 of the chip's 1 024 SIMDs, each with 14 us of matrix-pipe work, and the launch takes 32 us for 15 MB
// that HBM delivers in 3 (profiles/r06_short_systems.txt).  Here the TRIANGLE is dealt over the waves instead of the
// rows:
//   * a chunk of <= 128 rows is staged ONCE through LDS, weighted on the way (x = w_eff a; rows with w_eff = 0 and
//     columns >= K become zeros), by the 512 threads of a workgroup, every load of the chunk in flight at once;
//   * the chunk's triangle is cut into MACRO-TILES over 32-column groups -- (P, Q): 4 tiles, (P, P): 3, with the last
//     16-column block of an odd NB: 2 / 1 -- at most 15 of them, dealt heaviest-first over the 16 waves of the TWO
//     workgroups that share the chunk (both stage the same rows; they sit on the same XCD, so the second read is an
//     L2 hit).  A wave reads its two operand groups from LDS with one 16-byte read each per 4-row step (a lane's two
//     adjacent columns go to the even / odd blocks of the group: the column interleave of kernel 1A, undone by the
//     reduction kernel 2b) and owns its tiles outright: no fold through LDS, every tile is stored once;
//   * c = (wA)^T (wb) rides on the staging: the thread that weights an element also multiplies it by (w b) of its row;
//     b^T W^2 b, sum(w b) and the row count come from the threads that form the per-row pairs.
// Partials: part[chunk][NT][4][64] | cpart[chunk][NB][16] | spart[chunk][4] -- kernel 1A's layout with one entry per
// CHUNK (~126 x 92 KiB at 13 035 x 142), summed in chunk order by kernel 2b: run-to-run bit-identical, no atomics.
// Systems longer than 128 rows x (CUs / 2) take several phases per workgroup (stage, barrier, multiply, barrier);
// the planner (fsnap_capi.cpp: plan_geometry) hands over to kernel 1A where that stops paying.
#include <type_traits>

#include "fsnap_device_common.h"
#include "fsnap_kernels.h"

namespace {

// rows of a phase: what fits the LDS beside the pairs (an odd NB stages a zero block behind its last one: 160 doubles per row at NB = 9)
__host__ __device__ constexpr int short_rp(int NB) { return NB == 9 ? 112 : 128; }
constexpr int SHORT_THREADS = 512;     // 8 waves: two per SIMD

__host__ __device__ constexpr int tri_index(int p, int q, int NB) { return p * NB - (p * (p - 1)) / 2 + (q - p); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// Work of wave slot s = 8 g + wave (g = which of the chunk's two workgroups): macro-tile (P, Q) and the MASK of its four
// products the wave owns -- bit 0: a0 b0 = tile (2P, 2Q), bit 1: a0 b1 = (2P, 2Q + 1), bit 2: a1 b0 = (2P + 1, 2Q), bit 3:
// a1 b1 = (2P + 1, 2Q + 1); an off-diagonal macro-tile is 1111, a diagonal one 1011 (not the mirror a1 b0), with the zero
// block of an odd NB 0101 / 0001.  P = -1: none.  The items are sorted by their tile count and dealt over the 8 (workgroup,
// SIMD) bins in a snake: the first eight to the waves 0 ... 3 of the bins 0 ... 7, the rest to the waves 4 ... 7 of the bins
// 7 ... 0 (waves w and w + 4 share a SIMD).  NB = 9: 15 items, at most 6 of the 45 tiles on a SIMD (5.6 would be even).
// When that leaves a bin with two diagonal macro-tiles (NB = 8: ten items, 3 + 3 against 4 everywhere else) the diagonal ones
// are split into 0011 and 1000 -- 14 items, at most 5 of the 36 tiles on a SIMD; make_slots takes whichever deal has the
// smaller maximum.
struct SlotTable {
    int P[16], Q[16], M[16];
    int max_load;
};
constexpr int popcount4(int m) { return (m & 1) + ((m >> 1) & 1) + ((m >> 2) & 1) + ((m >> 3) & 1); }
constexpr SlotTable deal_slots(int NB, bool split_diag) {
    SlotTable t{};
    for (int s = 0; s < 16; ++s) {
        t.P[s] = -1;
        t.Q[s] = -1;
        t.M[s] = 0;
    }
    t.max_load = 1 << 20;
    const int np = (NB + 1) / 2;
    int lp[24] = {}, lq[24] = {}, lm[24] = {};
    int n = 0;
    auto add = [&](int P, int Q, int mask) {        // insertion by tile count, descending
        int k = n;
        while (k > 0 && popcount4(lm[k - 1]) < popcount4(mask)) {
            lp[k] = lp[k - 1];
            lq[k] = lq[k - 1];
            lm[k] = lm[k - 1];
            --k;
        }
        lp[k] = P;
        lq[k] = Q;
        lm[k] = mask;
        ++n;
    };
    for (int P = 0; P < np; ++P)
        for (int Q = P; Q < np; ++Q) {
            const bool qfull = 2 * Q + 1 < NB;
            if (P != Q) {
                add(P, Q, qfull ? 15 : 5);
            } else if (!qfull) {
                add(P, Q, 1);
            } else if (split_diag) {
                add(P, Q, 3);
                add(P, Q, 8);
            } else {
                add(P, Q, 11);
            }
        }
    if (n > 16) return t;
    int load[8] = {};
    for (int i = 0; i < n; ++i) {
        const int bin = i < 8 ? i : 15 - i;
        const int g = bin & 1, simd = bin >> 1;
        const int slot = g * 8 + (i < 8 ? simd : 4 + simd);
        t.P[slot] = lp[i];
        t.Q[slot] = lq[i];
        t.M[slot] = lm[i];
        load[bin] += popcount4(lm[i]);
    }
    t.max_load = 0;
    for (int b = 0; b < 8; ++b)
        if (load[b] > t.max_load) t.max_load = load[b];
    return t;
}
constexpr SlotTable make_slots(int NB) {
    const SlotTable a = deal_slots(NB, false), b = deal_slots(NB, true);
    return b.max_load < a.max_load ? b : a;
}

template <int NB>
struct ShortSlots {
    static_assert(NB >= 1 && NB <= 9, "at most 15 macro-tiles for the 16 wave slots of a chunk");
    static constexpr SlotTable tab = make_slots(NB);
    static_assert(tab.max_load <= 6, "a SIMD carries at most 6 tiles");
};
static_assert(ShortSlots<9>::tab.max_load == 6 && ShortSlots<8>::tab.max_load == 5 && ShortSlots<7>::tab.max_load == 4 &&
                  ShortSlots<6>::tab.max_load == 4,
              "the deals DESIGN 3.1b quotes");

}  // namespace

#ifdef FSNAP_SHORT_TRACE
// tools/short_trace.hip only: per-workgroup wall-clock stamps (100 MHz) {entry, pairs in LDS, rows staged, products done, tiles stored}
__device__ unsigned long long fsnap_short_trace[1024 * 8];
#define FSNAP_SHORT_STAMP(i)                                                                             \
    do {                                                                                                 \
        if (threadIdx.x == 0 && blockIdx.x < 1024) fsnap_short_trace[blockIdx.x * 8 + (i)] = wall_clock64(); \
    } while (0)
#else
#define FSNAP_SHORT_STAMP(i)
#endif

// grid: 16 x ceil(nchunk / 8) workgroups; workgroup bid serves chunk (bid & 7) + 8 (bid >> 4) as member g = (bid >> 3) & 1:
// consecutive workgroup ids go round the 8 XCDs, so both members of a chunk share an L2
// WPACK: the per-row pairs come from `wpack` (fsnap_pack_weights_k's array, or the pairs of a row-space pass) instead of being
// formed from b, w and the mask -- a compile-time switch: a run-time one put a join (and a wait for the loads in flight) between
// the loads of the next phase and the products of this one
template <int NB, bool WPACK>
__global__ __launch_bounds__(SHORT_THREADS) void fsnap_syrk_short(const double* __restrict__ A, int64_t lda,
                                                                   const double* __restrict__ wpack,
                                                                   const double* __restrict__ bvec,
                                                                   const double* __restrict__ wvec,
                                                                   const unsigned char* __restrict__ mask, int64_t m, int K,
                                                                   int rows_per_chunk, int nchunk, double* __restrict__ part,
                                                                   double* __restrict__ cpart, double* __restrict__ spart) {
    constexpr int SHORT_RP = short_rp(NB);
    constexpr int NBP = NB + (NB & 1);              // an odd NB is staged with a zero block behind its last block: every 32-column
                                                    // group is then read the same way, and the last block's columns sit at the
                                                    // EVEN places of their group (element e of block NB - 1 = column 16 (NB - 1) + e,
                                                    // the plain order kernel 2b expects of an odd last block)
    constexpr int LDW = 16 * NBP;                   // doubles per staged row
    constexpr int UP = 8 * NB;                      // 16-byte units per row of A
    constexpr int RG = SHORT_THREADS / UP;          // rows staged side by side
    constexpr int ITER = (SHORT_RP + RG - 1) / RG;  // loads in flight per thread
    constexpr int NTILE = NB * (NB + 1) / 2;
    static_assert((ITER * RG * 2 + SHORT_RP * LDW) * 8 <= 160 * 1024, "LDS of a CU");
    // per-row pairs first (their addresses stay inside the LDS instructions' 16-bit offsets), then the staged rows; the pair
    // slots past SHORT_RP (the last trip of the staging loop looks at them) hold zeros for good
    constexpr int PKR = ITER * RG;
    __shared__ __attribute__((aligned(16))) double lds[PKR * 2 + SHORT_RP * LDW];
    double* const PK = lds;
    double* const X = lds + PKR * 2;

    const int tid = threadIdx.x, lane = tid & 63, e = lane & 15, kr = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    const int g = (bid >> 3) & 1;
    const int chunk = (bid & 7) + 8 * (bid >> 4);
    if (chunk >= nchunk) return;
    FSNAP_SHORT_STAMP(0);
    const int64_t row0 = (int64_t)chunk * rows_per_chunk;
    int64_t row1 = row0 + rows_per_chunk;
    if (row1 > m) row1 = m;

    // staging role: 16-byte unit u of the rows rg, rg + RG, ...
    const bool stager = tid < RG * UP;
    const int u = tid % UP, rg = tid / UP;
    // columns 2u, 2u + 1 -> which of them exist
    const bool col0 = 2 * u < K, col1 = 2 * u + 1 < K;
    // multiply role
    const int slot = g * 8 + wave;
    const int P = ShortSlots<NB>::tab.P[slot], Q = ShortSlots<NB>::tab.Q[slot];
    const int MK = __builtin_amdgcn_readfirstlane(ShortSlots<NB>::tab.M[slot]);      // which of the four products are this wave's
    const double* xp = X + kr * LDW + 32 * (P < 0 ? 0 : P) + 2 * e;
    const double* xq = X + kr * LDW + 32 * (Q < 0 ? 0 : Q) + 2 * e;
    // where the unit lands in the staged row
    const bool last_odd = (NB & 1) && u >= 8 * (NB - 1);
    const int xcol = last_odd ? 16 * (NB - 1) + 4 * (u - 8 * (NB - 1)) : 2 * u;
    const int xdx = last_odd ? 2 : 1;

    if (tid >= SHORT_RP && tid < PKR) {
        const d2 z = {0.0, 0.0};
        *reinterpret_cast<d2*>(PK + 2 * tid) = z;
    }
    if constexpr (NB & 1) {     // the zero block of an odd NB: the odd places of the last group, written here and never again
        for (int i = tid; i < SHORT_RP * 16; i += SHORT_THREADS) X[(i >> 4) * LDW + 16 * (NB - 1) + 2 * (i & 15) + 1] = 0.0;
    }
    d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    d2 cacc = {0.0, 0.0};
    double bb = 0.0, sb = 0.0, cnt = 0.0;

    // The loads of a phase: its per-row pairs (w_eff, w_eff b) FIRST (vmcnt retires in order: behind the rows they would be
    // seen only when every row has arrived), the three of a row side by side, not behind the mask's branch; then every row
    // of the phase (16 bytes past the last row may be read: fsnap_hip.h's padding rule).  Every thread loads -- the ones
    // without a row of their own row 0 of the phase, the non-stagers out of range -- so that the code up to the barrier is one
    // straight line and the compiler's wait for the pairs is vmcnt(ITER), not vmcnt(0).  The row step is the instruction's
    // scalar offset; rows past nr are out of the descriptor's range and read zeros (the first 16 bytes behind it excepted --
    // those meet a (0, 0) pair).
    // The loads of phase p + 1 go out BEFORE the products of phase p (the rows wait in the 64 registers of `raw`), so a
    // chunk of several phases pays the HBM round trip once: stage | load next, multiply | stage | ...
    unsigned char mk = 0;
    double wr = 0.0, br = 0.0;
    d2 pv = {0.0, 0.0};
    u4 raw[ITER];
    auto issue_loads = [&](int64_t ph0) {
        const int nr = (int)(row1 - ph0 < SHORT_RP ? row1 - ph0 : SHORT_RP);
        const int64_t row = ph0 + (tid < nr ? tid : 0);
        if constexpr (WPACK) {
            pv = *reinterpret_cast<const d2*>(wpack + 2 * row);
        } else {
            mk = mask[row];
            wr = wvec[row];
            br = bvec[row];
        }
        const __amdgpu_buffer_rsrc_t ra = make_rsrc(A + ph0 * lda, (unsigned)((int64_t)nr * lda * 8 + 16));
        const unsigned voff = stager ? (unsigned)(((int64_t)rg * lda + 2 * u) * 8) : 0xFFFFF000u;
        const unsigned rstep = (unsigned)(lda * 8 * RG);
#pragma unroll
        for (int it = 0; it < ITER; ++it) raw[it] = __builtin_amdgcn_raw_buffer_load_b128(ra, voff, (unsigned)it * rstep, 0);
    };
    issue_loads(row0);

    for (int64_t ph0 = row0; ph0 < row1; ph0 += SHORT_RP) {
        const int nr = (int)(row1 - ph0 < SHORT_RP ? row1 - ph0 : SHORT_RP);
        const int nr4 = (nr + 7) & ~7;         // whole pairs of 4-row steps: the rows behind the last one are staged as zeros
        // the pairs into LDS while the rows are still in flight
        if (tid < SHORT_RP) {
            d2 pr = {0.0, 0.0};
            if constexpr (WPACK) {
                if (tid < nr) pr = pv;
            } else {
                const bool keep = tid < nr && mk != 0;
                const double wv = keep ? wr : 0.0;
                const double wbv = keep ? wv * br : 0.0;
                pr[0] = wv;
                pr[1] = wbv;
                bb = __builtin_fma(wbv, wbv, bb);
                sb += wbv;
                cnt += keep ? 1.0 : 0.0;
            }
            *reinterpret_cast<d2*>(PK + 2 * tid) = pr;
        }
        // barrier on the LDS writes only: __syncthreads() would also wait for every row load in flight (s_waitcnt vmcnt(0))
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        FSNAP_SHORT_STAMP(1);
        // 3. weight, zero what does not exist, park in LDS; c rides along (member 0 only: both members see the same rows).
        // Every row slot of the phase is written (the ones past nr as zeros: their loads were out of range, their pairs are
        // (0, 0)), so the loop has no branch but the compile-time one of its last trip.
        if (stager) {
            constexpr int B = 5;           // pairs read from LDS ahead of their use
#pragma unroll
            for (int it0 = 0; it0 < ITER; it0 += B) {
                d2 prs[B];
#pragma unroll
                for (int j = 0; j < B; ++j) {
                    const int row = (it0 + j) * RG + rg;
                    prs[j] = *reinterpret_cast<const d2*>(PK + 2 * (row < PKR ? row : 0));
                }
#pragma unroll
                for (int j = 0; j < B; ++j) {
                    const int it = it0 + j;
                    if (it < ITER) {
                        const int row = it * RG + rg;
                        const d2 pr = prs[j];
                        const d2 a = __builtin_bit_cast(d2, raw[it]);
                        const bool keep = pr[0] != 0.0;
                        d2 x;
                        x[0] = (keep && col0) ? pr[0] * a[0] : 0.0;
                        x[1] = (keep && col1) ? pr[0] * a[1] : 0.0;
                        if ((it + 1) * RG <= SHORT_RP || row < SHORT_RP) {
                            double* xr = X + row * LDW + xcol;
                            if constexpr (NB & 1) {
                                // two 8-byte stores, xdx apart (1; 2 in the last block of an odd NB): no branch
                                xr[0] = x[0];
                                xr[xdx] = x[1];
                            } else {
                                *reinterpret_cast<d2*>(xr) = x;      // (8-byte stores 16 bytes apart meet two to a bank)
                            }
                        }
                        if (g == 0) {
                            cacc[0] = __builtin_fma(x[0], pr[1], cacc[0]);
                            cacc[1] = __builtin_fma(x[1], pr[1], cacc[1]);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);      // (left alone the scheduler hoists every read of the pairs: 76 registers)
            }
        }
        __syncthreads();
        FSNAP_SHORT_STAMP(2);
        if (ph0 + SHORT_RP < row1) issue_loads(ph0 + SHORT_RP);       // the next phase's rows travel while this one is multiplied
        // 4. the wave's products over the phase's 4-row steps: a = group P, b = group Q; one v_mfma_f64_16x16x4_f64 is 64
        // cycles of the SIMD's matrix pipe, so only the tiles that exist are multiplied -- one loop per mask over the same
        // accumulators (acc0 = a0 b0, acc1 = a0 b1, acc2 = a1 b0, acc3 = a1 b1)
        const int nstep = nr4 >> 2;
        auto products = [&](auto mask_tag, auto diag_tag) {
            constexpr int MASK = decltype(mask_tag)::value;
            constexpr bool DIAG = decltype(diag_tag)::value;
            for (int s0 = 0; s0 < nstep; s0 += 2) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int off = (s0 + s) * 4 * LDW;
                    const d2 a = *reinterpret_cast<const d2*>(xp + off);
                    d2 b = a;
                    if constexpr (!DIAG) b = *reinterpret_cast<const d2*>(xq + off);
                    if constexpr (MASK & 1) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], acc0, 0, 0, 0);
                    if constexpr (MASK & 2) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[1], acc1, 0, 0, 0);
                    if constexpr (MASK & 4) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[0], acc2, 0, 0, 0);
                    if constexpr (MASK & 8) acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], acc3, 0, 0, 0);
                }
            }
        };
        using std::integral_constant;
        using std::false_type;
        using std::true_type;
        switch (P >= 0 ? MK : 0) {
            case 15: products(integral_constant<int, 15>{}, false_type{}); break;
            case 5: products(integral_constant<int, 5>{}, false_type{}); break;
            case 11: products(integral_constant<int, 11>{}, true_type{}); break;
            case 3: products(integral_constant<int, 3>{}, true_type{}); break;
            case 8: products(integral_constant<int, 8>{}, true_type{}); break;
            case 1: products(integral_constant<int, 1>{}, true_type{}); break;
            default: break;
        }
        __syncthreads();        // the next phase (or the c fold) overwrites X and PK
        FSNAP_SHORT_STAMP(3);
    }

    // the wave's tiles, each stored once
    if (P >= 0) {
        double* pw = part + (int64_t)chunk * (NTILE * 256);
        auto store = [&](int p, int q, const d4& v) {
            double* t = pw + tri_index(p, q, NB) * 256 + lane;
#pragma unroll
            // (plain stores: nontemporal ones take 1 us off this launch -- 12.7 -> 11.7 us back to back, nothing dirty in L2 at
            // the kernel's end -- and put 1.5 ... 2.4 us on kernel 2b, which then reads the partial tiles from memory)
            for (int i = 0; i < 4; ++i) t[i * 64] = v[i];
        };
        if (MK & 1) store(2 * P, 2 * Q, acc0);
        if (MK & 2) store(2 * P, 2 * Q + 1, acc1);
        if (MK & 4) store(2 * P + 1, 2 * Q, acc2);
        if (MK & 8) store(2 * P + 1, 2 * Q + 1, acc3);
    }
    FSNAP_SHORT_STAMP(4);
    if (g != 0) return;

    // c: the RG row groups in a fixed order; column 2u + h of the matrix is element (p, e) of kernel 1A's interleave
    constexpr int LDC = 16 * NB;
    if (stager) *reinterpret_cast<d2*>(X + rg * LDC + 2 * u) = cacc;
    if (!WPACK && tid < 128) {                     // waves 0 and 1 in full (the threads past SHORT_RP bring zeros)
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) {       // fixed butterfly: deterministic
            bb += __shfl_xor(bb, sh, 64);
            sb += __shfl_xor(sb, sh, 64);
            cnt += __shfl_xor(cnt, sh, 64);
        }
        if (tid == 64) {
            PK[0] = bb;
            PK[1] = sb;
            PK[2] = cnt;
        }
    }
    __syncthreads();
    if (tid < LDC) {
        double s = X[tid];
#pragma unroll
        for (int k = 1; k < RG; ++k) s += X[k * LDC + tid];
        int p, el;
        if ((NB & 1) && tid >= 16 * (NB - 1)) {
            p = NB - 1;
            el = tid - 16 * (NB - 1);
        } else {
            p = 2 * (tid >> 5) + (tid & 1);
            el = (tid & 31) >> 1;
        }
        cpart[(int64_t)chunk * (NB * 16) + p * 16 + el] = s;
    }
    if (!WPACK && tid == 0) {
        double* so = spart + (int64_t)chunk * 4;
        so[0] = bb + PK[0];
        so[1] = sb + PK[1];
        so[2] = cnt + PK[2];
        so[3] = 0.0;
    }
}

namespace fsnap {

bool syrk_short_takes(int K) { return K > 80 && K <= 144; }
int syrk_short_phase_rows(int K) { return short_rp(syrk_num_blocks(K)); }

template <int NB>
static hipError_t launch_syrk_short_nb(const SyrkArgs& a, hipStream_t st) {
    const int nchunk = a.nblocks;
    dim3 grid((unsigned)(16 * ((nchunk + 7) / 8))), block(SHORT_THREADS);
    if (a.fused_pack)
        hipLaunchKernelGGL((fsnap_syrk_short<NB, false>), grid, block, 0, st, a.A, a.lda, nullptr, a.b, a.w, a.mask, a.m, a.K,
                           (int)a.chunks_per_wave, nchunk, a.part, a.cpart, a.spart);
    else
        hipLaunchKernelGGL((fsnap_syrk_short<NB, true>), grid, block, 0, st, a.A, a.lda, a.wpack, a.b, a.w, a.mask, a.m, a.K,
                           (int)a.chunks_per_wave, nchunk, a.part, a.cpart, a.spart);
    return hipGetLastError();
}

// kernel 1S: a.nblocks chunks of a.chunks_per_wave ROWS (a multiple of 4), two workgroups each; pairs from a.wpack or
// (a.fused_pack) formed from b, w, mask, with the b-only scalars in a.spart[chunk][4]
hipError_t launch_syrk_short(const SyrkArgs& a, hipStream_t st) {
    if (a.fused_pack ? (!a.b || !a.w || !a.mask || !a.spart) : !a.wpack) return hipErrorInvalidValue;
    if (a.nblocks < 1 || a.chunks_per_wave < 4 || (a.chunks_per_wave & 3) || a.chunks_per_wave > 0x7FFFFFF0) return hipErrorInvalidValue;
    if (a.lda * 8 * (int64_t)136 + 16 > (int64_t)0xFFFFF000) return hipErrorInvalidValue;      // 32-bit offsets inside a phase (the staging loop looks at up to 135 row slots)
    switch (syrk_num_blocks(a.K)) {
        case 6: return launch_syrk_short_nb<6>(a, st);
        case 7: return launch_syrk_short_nb<7>(a, st);
        case 8: return launch_syrk_short_nb<8>(a, st);
        case 9: return launch_syrk_short_nb<9>(a, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace fsnap
