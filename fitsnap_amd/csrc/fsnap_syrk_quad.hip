// fsnap_syrk_quad.hip — kernel 1Q: fused mask x weight x SYRK for 144 < K <= 288 (gfx950 only).
//
// The statistics of the linear fit, G = (wA)^T (wA), c = (wA)^T (wb) and the three scalars
// (fitsnap3lib/solvers/ridge.py:40-49, svd.py:35-54 after the transpose trick), for the widths between kernel 1A
// (one wave holds the whole tile triangle: NB <= 9 column blocks) and the widths where the tiled kernel 1T has
// enough equal work items (K >~ 300): three- / four- / five-element SNAP (168, 220 + offsets, 275), short ACE bases.
//
// Kernel 1A's plan with the triangle DEALT TO THE FOUR WAVES OF A WORKGROUP: NB = 10 ... 18 column blocks are 55 ...
// 171 accumulator tiles, 14 ... 43 per wave (32 in the accumulation registers a[0:255], up to eleven more in VGPRs).
// The four waves sweep the SAME rows -- the workgroup's contiguous row range -- each with its own loads: a wave
// reads the column blocks its tile rows touch (all of them for the wave that owns tile row 0) straight into
// MFMA-fragment registers, three chunks ahead (two from 17 blocks on), like kernel 1A; the first wave to ask brings
// a row in from HBM, the other three are merged with it in the L1 or find the row in L2 (the workgroup lives on one
// CU).  No LDS traffic for the rows, one wave per SIMD; one bare s_barrier per six-chunk trip keeps the four waves
// on the same rows.  HBM sees every row once: 32 KiB per 4-row chunk and 2200 cycles of matrix pipe at K = 256.
// Whole tile ROWS (tiles (p, p .. NB-1), A operand block p) are dealt so that the four tile counts differ by at
// most two (QUAD_ROWS below); inside a wave the rows run in ascending order, so column block j of the NEXT chunk
// can overwrite V[j] as soon as the last tile row p <= j of this wave has been issued (kernel 1A's single operand
// set).  c = (wA)^T (wb) rides along on the VALU, its column blocks dealt to the waves that hold them, fewest
// multiplies first.  The per-row pairs (w_eff, w_eff b) of the workgroup's rows are formed in LDS by the prologue
// (all 256 threads), like kernel 1A with fused packing; the b-only scalars leave from there.  PACK = false reads
// the pairs from HBM instead (fsnap_pack_weights_k, or the pairs a row-space pass brings; workgroups whose rows'
// pairs do not fit the LDS).
// Partials: part[workgroup][NT][4][64] -- every tile written by the one wave that owns it, no fold --,
// cpart[workgroup * 4 + wave][NB][16] (zeros for the blocks another wave owns), spart[workgroup * 4 + wave][4]:
// the layout of kernel 1A, reduced by the same kernel 2b.  No floating-point atomics; bit-identical run to run.
#include <utility>

#include "fsnap_device_common.h"
#include "fsnap_kernels.h"

namespace {

__host__ __device__ constexpr int quad_tri_index(int p, int q, int NB) { return p * NB - (p * (p - 1)) / 2 + (q - p); }

// tile rows of wave w for NB column blocks (-1 ends the list): the lengths NB - p of a wave's rows add up to
// ceil or floor of NB (NB + 1) / 8 -- checked by the static_asserts below
constexpr int QUAD_ROWS[9][4][9] = {
    /* NB = 10: 14 14 14 13 */ {{0, 6, -1}, {1, 5, -1}, {2, 4, -1}, {3, 7, 8, 9, -1}},
    /* NB = 11: 17 17 17 15 */ {{0, 5, -1}, {1, 4, -1}, {2, 3, -1}, {6, 7, 8, 9, 10, -1}},
    /* NB = 12: 20 20 20 18 */ {{0, 4, -1}, {1, 3, -1}, {2, 5, 9, -1}, {6, 7, 8, 10, 11, -1}},
    /* NB = 13: 23 23 23 22 */ {{0, 3, -1}, {1, 2, -1}, {4, 5, 7, -1}, {6, 8, 9, 10, 11, 12, -1}},
    /* NB = 14: 27 26 26 26 */ {{0, 1, -1}, {2, 3, 11, -1}, {4, 5, 7, -1}, {6, 8, 9, 10, 12, 13, -1}},
    /* NB = 15: 30 30 30 30 */ {{0, 1, 14, -1}, {2, 3, 10, -1}, {4, 5, 6, -1}, {7, 8, 9, 11, 12, 13, -1}},
    /* NB = 16: 34 34 34 34 */ {{0, 1, 13, -1}, {2, 3, 9, -1}, {4, 5, 6, 15, -1}, {7, 8, 10, 11, 12, 14, -1}},
    /* NB = 17: 38 38 38 39 */ {{0, 1, 12, -1}, {2, 3, 8, -1}, {4, 5, 6, 15, -1}, {7, 9, 10, 11, 13, 14, 16, -1}},
    /* NB = 18: 43 43 43 42 */ {{0, 1, 10, -1}, {2, 3, 6, -1}, {4, 5, 7, 13, -1}, {8, 9, 11, 12, 14, 15, 16, 17, -1}},
};

constexpr int QUAD_MAXT = 43;
constexpr int QUAD_MAXNB = 32;
constexpr int QUAD_MAXW = 16;

// Kernel 1QC (round 5): 289 ... 512 columns (NB = 19 ... 32: 190 ... 528 tiles) do not fit the accumulators of one
// workgroup.  The same plan on a CLUSTER of workgroups placed on one XCD -- 2 workgroups (8 waves) up to 23 column blocks (at most 38 tiles per wave),
// 4 workgroups (16 waves) beyond: whole tile rows dealt to the 4 C waves, the longest row first to the least loaded wave.
__host__ __device__ constexpr int quad_cluster(int NB) { return NB <= 18 ? 1 : (NB <= 23 ? 2 : 4); }
__host__ __device__ constexpr int quad_waves(int NB) { return 4 * quad_cluster(NB); }

struct QuadDeal {
    int r[QUAD_MAXW][9];     // tile rows of every wave, ascending, -1 ends a list
};

constexpr QuadDeal quad_deal(int NB) {
    QuadDeal D{};
    const int NW = quad_waves(NB);
    for (int v = 0; v < QUAD_MAXW; ++v)
        for (int i = 0; i < 9; ++i) D.r[v][i] = -1;
    if (NB <= 18) {
        for (int v = 0; v < 4; ++v)
            for (int i = 0; i < 9; ++i) D.r[v][i] = QUAD_ROWS[NB - 10][v][i];
        return D;
    }
    int load[QUAD_MAXW] = {};
    int cnt[QUAD_MAXW] = {};
    for (int p = 0; p < NB; ++p) {          // row lengths NB - p descend: longest processing time first
        int best = 0;
        for (int v = 1; v < NW; ++v)
            if (load[v] < load[best]) best = v;
        if (cnt[best] < 9) D.r[best][cnt[best]] = p;
        cnt[best] += 1;
        load[best] += NB - p;
    }
    return D;
}

struct QuadPlan {
    int n;                  // tiles of this wave
    int jmin;               // first column block it reads
    int tp[QUAD_MAXT], tq[QUAD_MAXT];
    int rf_lo[QUAD_MAXT], rf_hi[QUAD_MAXT];   // after tile i: refresh V[rf_lo .. rf_hi) with the next chunk (empty unless i ends a tile row)
    int cown[QUAD_MAXNB];   // 1 = this wave accumulates c for column block j
};

constexpr QuadPlan quad_plan(int NB, int w) {
    QuadPlan P{};
    const QuadDeal DL = quad_deal(NB);
    const int(&rows)[9] = DL.r[w];
    int n = 0;
    P.jmin = rows[0];
    for (int i = 0; i < 9 && rows[i] >= 0; ++i) {
        const int p = rows[i];
        const int pnext = (i + 1 < 9 && rows[i + 1] >= 0) ? rows[i + 1] : NB;
        for (int q = p; q < NB; ++q) {
            if (n < QUAD_MAXT) {
                P.tp[n] = p;
                P.tq[n] = q;
                P.rf_lo[n] = 0;
                P.rf_hi[n] = 0;
            }
            ++n;
        }
        if (n <= QUAD_MAXT) {
            P.rf_lo[n - 1] = p;
            P.rf_hi[n - 1] = pnext;
        }
    }
    P.n = n;
    // c: column block j goes to the wave with the least VALU work so far among those that hold it (jmin <= j)
    const int NW = quad_waves(NB);
    int load[QUAD_MAXW] = {};
    int jm[QUAD_MAXW] = {};
    for (int v = 0; v < NW; ++v) {
        jm[v] = DL.r[v][0];
        load[v] = NB - jm[v];
    }
    for (int j = NB - 1; j >= 0; --j) {
        int best = -1;
        for (int v = NW - 1; v >= 0; --v)
            if (jm[v] <= j && (best < 0 || load[v] < load[best])) best = v;
        load[best] += 1;
        P.cown[j] = (best == w) ? 1 : 0;
    }
    return P;
}

constexpr bool quad_plans_cover(int NB) {
    int seen[QUAD_MAXNB * (QUAD_MAXNB + 1) / 2] = {};
    int mx = 0, mn = 1000;
    const int NW = quad_waves(NB);
    for (int w = 0; w < NW; ++w) {
        const QuadPlan P = quad_plan(NB, w);
        if (P.n > QUAD_MAXT || P.n < 1) return false;
        mx = P.n > mx ? P.n : mx;
        mn = P.n < mn ? P.n : mn;
        for (int i = 0; i < P.n; ++i) seen[quad_tri_index(P.tp[i], P.tq[i], NB)] += 1;
        for (int i = 1; i < P.n; ++i)
            if (P.tp[i] < P.tp[i - 1]) return false;          // ascending tile rows
    }
    for (int t = 0; t < NB * (NB + 1) / 2; ++t)
        if (seen[t] != 1) return false;
    // one workgroup: the four counts within two of each other; a cluster: no wave above the accumulators + 8 tiles in VGPRs
    return NW == 4 ? mx - mn <= 2 : mx <= 40;
}
static_assert(quad_plans_cover(10) && quad_plans_cover(11) && quad_plans_cover(12) && quad_plans_cover(13) &&
                  quad_plans_cover(14) && quad_plans_cover(15) && quad_plans_cover(16) && quad_plans_cover(17) && quad_plans_cover(18),
              "every tile of the triangle belongs to exactly one wave");
static_assert(quad_plans_cover(19) && quad_plans_cover(20) && quad_plans_cover(21) && quad_plans_cover(22) &&
                  quad_plans_cover(23) && quad_plans_cover(24) && quad_plans_cover(25) && quad_plans_cover(26) && quad_plans_cover(27) &&
                  quad_plans_cover(28) && quad_plans_cover(29) && quad_plans_cover(30) && quad_plans_cover(31) && quad_plans_cover(32),
              "every tile of the triangle belongs to exactly one wave of the cluster");

template <int NB, int W>
struct QuadPlanOf {
    static constexpr QuadPlan P = quad_plan(NB, W);
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t quad_rsrc(const void* p, unsigned bytes) {
    // dword3 0x00020000: raw buffer, 32-bit data format (gfx9-family encoding)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

constexpr unsigned QUAD_OOB_VOFF = 0xFFFFF000u;    // beyond any workgroup's buffer (plan: < 0xFFF00010): reads back zeros

// accumulator tile T of a wave: a[8T : 8T + 7] for T < 32, VGPRs beyond
template <int T, int NV>
__device__ __forceinline__ void quad_mfma(double a, double b, d4 (&vt)[NV]) {
    if constexpr (T < 32) {
        asm volatile("v_mfma_f64_16x16x4_f64 a[%2:%3], %0, %1, a[%2:%3]" : : "v"(a), "v"(b), "n"(8 * T), "n"(8 * T + 7));
    } else {
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(vt[T - 32]) : "v"(a), "v"(b));
    }
}
template <int R>
__device__ __forceinline__ void quad_zero_reg() {
    asm volatile("v_accvgpr_write_b32 a[%0], 0" : : "n"(R));
}
template <int... R>
__device__ __forceinline__ void quad_zero_all(std::integer_sequence<int, R...>) {
    (quad_zero_reg<R>(), ...);
}
template <int T, int NV>
__device__ __forceinline__ d4 quad_read(const d4 (&vt)[NV]) {
    if constexpr (T >= 32) {
        return vt[T - 32];
    } else {
        unsigned r0, r1, r2, r3, r4, r5, r6, r7;
        asm volatile(
            "v_accvgpr_read_b32 %0, a[%8]\n\tv_accvgpr_read_b32 %1, a[%9]\n\t"
            "v_accvgpr_read_b32 %2, a[%10]\n\tv_accvgpr_read_b32 %3, a[%11]\n\t"
            "v_accvgpr_read_b32 %4, a[%12]\n\tv_accvgpr_read_b32 %5, a[%13]\n\t"
            "v_accvgpr_read_b32 %6, a[%14]\n\tv_accvgpr_read_b32 %7, a[%15]"
            : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7)
            : "n"(8 * T), "n"(8 * T + 1), "n"(8 * T + 2), "n"(8 * T + 3), "n"(8 * T + 4), "n"(8 * T + 5), "n"(8 * T + 6),
              "n"(8 * T + 7));
        d4 x;
        x[0] = __builtin_bit_cast(double, ((unsigned long long)r1 << 32) | r0);
        x[1] = __builtin_bit_cast(double, ((unsigned long long)r3 << 32) | r2);
        x[2] = __builtin_bit_cast(double, ((unsigned long long)r5 << 32) | r4);
        x[3] = __builtin_bit_cast(double, ((unsigned long long)r7 << 32) | r6);
        return x;
    }
}

template <class F, int... I>
__device__ __forceinline__ void quad_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void quad_for(F&& f) {
    quad_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Raw loads of one 4-row chunk for a wave that reads column blocks 2 PR0 .. NB - 1: 16-byte loads give a lane two
// ADJACENT columns, which go to two column blocks (even / odd columns of a 32-column group, fsnap_syrk.hip); an odd
// NB ends with a plain 16-column block.
template <int NB, int PR0>
struct QuadRaw {
    static constexpr int NPR = NB / 2 - PR0;
    u4 pr[NPR > 0 ? NPR : 1];
    u2 tail;
};

struct QuadBufs {
    __amdgpu_buffer_rsrc_t A, wp;
    unsigned voffA;        // (kr * lda + 2 e) * 8 + 256 PR0
    unsigned voffT;        // (kr * lda + 16 (NB - 1) + e) * 8
    unsigned voffP;        // kr * 16 (pairs in HBM)
    unsigned chunk_bytes;  // 4 * lda * 8
};

// The pair (w_eff, w_eff b) of the lane's row of chunk cl.  PACK: from the workgroup's LDS region, filled by the
// prologue (lpk = region + 2 kr doubles).  Otherwise from the wpack array in HBM (fsnap_pack_weights_k, or the
// per-row pairs of a row-space pass) through a bounds-checked descriptor: chunks past the range read (0, 0).
template <bool PACK>
__device__ __forceinline__ u4 quad_pair(const QuadBufs& wb, const double* lpk, unsigned cl) {
    if constexpr (PACK) return __builtin_bit_cast(u4, *reinterpret_cast<const d2*>(lpk + (size_t)cl * 8));
    else return __builtin_amdgcn_raw_buffer_load_b128(wb.wp, wb.voffP, cl * 64u, 0);
}

__device__ __forceinline__ bool quad_keep(const u4& wp) {
    return ((wp[0] | (wp[1] & 0x7FFFFFFFu)) != 0u);      // w_eff != +-0
}

template <int NB, int PR0, int P>
__device__ __forceinline__ void quad_issue_piece(QuadRaw<NB, PR0>& r, const QuadBufs& wb, unsigned cl, unsigned va, unsigned vtl) {
    constexpr int NPR = QuadRaw<NB, PR0>::NPR;
    const unsigned soff = cl * wb.chunk_bytes;
    if constexpr (P < NPR) r.pr[P] = __builtin_amdgcn_raw_buffer_load_b128(wb.A, va + 256u * P, soff, 0);
    if constexpr ((NB & 1) && P == NPR) r.tail = __builtin_amdgcn_raw_buffer_load_b64(wb.A, vtl, soff, 0);
}

template <int NB, int PR0>
__device__ __forceinline__ void quad_issue_rows(QuadRaw<NB, PR0>& r, const u4& wp, const QuadBufs& wb, unsigned cl) {
    const bool keep = quad_keep(wp);
    const unsigned va = keep ? wb.voffA : QUAD_OOB_VOFF;
    const unsigned vtl = keep ? wb.voffT : QUAD_OOB_VOFF;
    quad_for<QuadRaw<NB, PR0>::NPR + (NB & 1)>([&](auto pc) { quad_issue_piece<NB, PR0, decltype(pc)::value>(r, wb, cl, va, vtl); });
}

// w * (raw value of column block J); columns >= K (last block / block pair only) are zeroed by a select
template <int NB, int PR0, bool FULLK, int J>
__device__ __forceinline__ double quad_weighted(const QuadRaw<NB, PR0>& r, double wv, int K, int e) {
    if constexpr ((NB & 1) && J == NB - 1) {
        const double x = wv * __builtin_bit_cast(double, r.tail);
        if (FULLK) return x;
        return (16 * (NB - 1) + e < K) ? x : 0.0;
    } else {
        const double x = wv * __builtin_bit_cast(d2, r.pr[(J >> 1) - PR0])[J & 1];
        constexpr int last_pair = (NB & 1) ? -1 : NB / 2 - 1;
        if (FULLK || (J >> 1) != last_pair) return x;
        return (32 * (J >> 1) + 2 * e + (J & 1) < K) ? x : 0.0;
    }
}

#ifndef FSNAP_QUAD_SYNC
#define FSNAP_QUAD_SYNC 1
#endif
constexpr bool QUAD_SYNC = FSNAP_QUAD_SYNC != 0;
constexpr int QUAD_LDS_DOUBLES = 20480;   // all 160 KiB of the CU: the per-row pairs of the workgroup's rows
constexpr int QUAD_PACK_PAD = 12;         // chunk slots the unrolled loop may look past the last chunk (<= ncl + 9)

// One step: the MFMAs of the chunk held in V, with the preparation of the next chunk (raw set RN, pair PW) in
// between: the refill of the raw set consumed a step ago (rows of chunk cl_fill, gated by PKEEP) goes out one load
// per MFMA at the start of the step; V[j] takes the next chunk as soon as the wave's last tile row <= j is issued.
template <int NB, int W, bool FULLK, bool PACK, int NV>
__device__ __forceinline__ void quad_step(double (&V)[NB], d4 (&vt)[NV], QuadRaw<NB, (QuadPlanOf<NB, W>::P.jmin >> 1)>& RF,
                                          const QuadRaw<NB, (QuadPlanOf<NB, W>::P.jmin >> 1)>& RN, const QuadBufs& wb,
                                          const double* lpk, unsigned cl_fill, int K, int e, double (&cacc)[NB], const u4& PW,
                                          const u4& PKEEP, u4& PLOAD) {
    using PL = QuadPlanOf<NB, W>;
    constexpr int PR0 = PL::P.jmin >> 1;
    constexpr int NPIECE = QuadRaw<NB, PR0>::NPR + (NB & 1);
    const d2 wpn = __builtin_bit_cast(d2, PW);
    const double wv = wpn[0], wbv = wpn[1];
    const bool keep = quad_keep(PKEEP);
    const unsigned va = keep ? wb.voffA : QUAD_OOB_VOFF;
    const unsigned vtl = (NB & 1) ? (keep ? wb.voffT : QUAD_OOB_VOFF) : 0u;
    PLOAD = quad_pair<PACK>(wb, lpk, cl_fill + 2);      // before the row loads: vmcnt retires in order
    __builtin_amdgcn_sched_barrier(0);
    quad_for<PL::P.n>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        constexpr int TP = PL::P.tp[I], TQ = PL::P.tq[I], LO = PL::P.rf_lo[I], HI = PL::P.rf_hi[I];
        quad_mfma<I, NV>(V[TP], V[TQ], vt);
        if constexpr (I < NPIECE) {
            quad_issue_piece<NB, PR0, I>(RF, wb, cl_fill, va, vtl);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (HI > LO) {
            quad_for<HI - LO>([&](auto jc) {
                constexpr int J = LO + decltype(jc)::value;
                constexpr int OWN = PL::P.cown[J];
                V[J] = quad_weighted<NB, PR0, FULLK, J>(RN, wv, K, e);
                if constexpr (OWN != 0) cacc[J] = __builtin_fma(V[J], wbv, cacc[J]);
            });
            __builtin_amdgcn_sched_barrier(0);
        }
    });
}

// Flow control of a cluster (kernel 1QC): the workgroups of a cluster sweep the SAME rows, each for its share of the tile
// triangle; HBM should see a row once and the XCD's L2 (4 MiB for up to 16 clusters) serve the other reads, so no member
// may run more than `lead` trips (of six 4-row chunks) ahead of the slowest.  Wave 3 of every member (never the wave that owns tile row 0, the longest) publishes its trip
// count (+ the launch's tag: the words are never reset) and waits -- BOUNDED: this is a hint for the cache, nothing depends
// on it; a member that gives up simply streams on -- before the trip's barrier releases its workgroup.
struct QuadFlow {
    int* word;              // this cluster's words: one per member
    int members, me, tag;
    int lead;               // trips a member may run ahead of the slowest
    int mode;               // 0 = no flow control, 1 = publish + look at the end of a trip, 2 = publish + look at the START of the
                            // trip, judge at its end (the three device-scope round trips run beside the trip's MFMAs)
};
constexpr int QUAD_FLOW_SPINS = 4096;        // x s_sleep 2 (~130 cycles) ~ 0.2 ms per trip at most

struct QuadSeen {
    int p[3];
};

// lane 0 of wave 0 of a member: publish "trip `trip` done", note where the peers stand (relaxed device-scope loads: L2)
__device__ __forceinline__ QuadSeen quad_flow_look(const QuadFlow& f, int trip, int lane) {
    QuadSeen s;
    s.p[0] = s.p[1] = s.p[2] = 0x7FFFFFFF;
    if (lane == 0) {
        __hip_atomic_store(f.word + f.me, (int)((unsigned)f.tag + (unsigned)trip), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int peer = (f.me + 1 + k) & 3;
            if (peer < f.members && peer != f.me) s.p[k] = __hip_atomic_load(f.word + peer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    return s;
}

// ... and wait (bounded) for every peer that stood more than `lead` trips behind `trip`
__device__ __forceinline__ void quad_flow_wait(const QuadFlow& f, const QuadSeen& s, int trip, int lane) {
    if (lane == 0) {
        // serial-number arithmetic on 32-bit words: tags grow by 2^20 per launch and wrap after 4096 launches; a word of an
        // EARLIER launch compares below `need` (and a wrong verdict would only end or prolong a bounded wait)
        const unsigned need = (unsigned)f.tag + (unsigned)trip - (unsigned)f.lead;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int peer = (f.me + 1 + k) & 3;
            if (peer >= f.members || peer == f.me) continue;
            int v = s.p[k], n = 0;
            while ((int)((unsigned)v - need) < 0 && ++n < QUAD_FLOW_SPINS) {
                __builtin_amdgcn_s_sleep(2);
                v = __hip_atomic_load(f.word + peer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

template <int NB, int W, bool FULLK, bool PACK, bool FLOW = false>
__device__ __forceinline__ void quad_wave(const double* __restrict__ A, int64_t lda, const double* __restrict__ wpack, int K,
                                          int64_t row0, int64_t nrow, unsigned ncl, const double* lpk, double* __restrict__ pw,
                                          double* __restrict__ cw, int lane, const QuadFlow* flow = nullptr) {
    using PL = QuadPlanOf<NB, W>;
    constexpr int JMIN = PL::P.jmin, NTW = PL::P.n;
    constexpr int PR0 = JMIN >> 1;
    constexpr int NV = NTW > 32 ? NTW - 32 : 1;
    using Raw = QuadRaw<NB, PR0>;
    const int e = lane & 15, kr = lane >> 4;
    QuadBufs wb;
    wb.A = quad_rsrc(A + row0 * lda, (unsigned)(nrow * lda * 8 + (nrow ? 16 : 0)));
    wb.wp = quad_rsrc(PACK ? nullptr : wpack + 2 * row0, PACK ? 0u : (unsigned)(nrow * 16));
    wb.voffP = (unsigned)(kr * 16);
    wb.voffA = (unsigned)((kr * lda + 2 * e) * 8) + 256u * PR0;
    wb.voffT = (unsigned)((kr * lda + 16 * (NB - 1) + e) * 8);
    wb.chunk_bytes = (unsigned)(lda * 32);
    if (PACK) lpk += 2 * kr;

    d4 vt[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) vt[u] = d4{0.0, 0.0, 0.0, 0.0};
    double cacc[NB], V[NB];
#pragma unroll
    for (int p = 0; p < NB; ++p) {
        cacc[p] = 0.0;
        V[p] = 0.0;
    }
    auto pair_of = [&](unsigned cl) { return quad_pair<PACK>(wb, lpk, cl); };
    if (ncl > 0) {
        // raw sets in flight: three (kernel 1A's pipeline) up to 16 column blocks; two from 17 on, where the wave that owns tile
        // row 0 holds 36 registers per set and up to 11 accumulator tiles in VGPRs -- a chunk is 2 500+ cycles of matrix pipe
        // there, two of them cover the memory round trip
        constexpr int DEPTH = NB >= 17 ? 2 : 3;
        Raw r0, r1, r2;
        u4 pk0 = pair_of(0), pk1 = pair_of(1), pk2 = pair_of(2);
        quad_issue_rows<NB, PR0>(r0, pk0, wb, 0);
        quad_issue_rows<NB, PR0>(r1, pk1, wb, 1);
        if constexpr (DEPTH == 3) quad_issue_rows<NB, PR0>(r2, pk2, wb, 2);
        {   // chunk 0 -> V
            const d2 wp0 = __builtin_bit_cast(d2, pk0);
            const double wv = wp0[0], wbv = wp0[1];
            quad_for<NB - JMIN>([&](auto jc) {
                constexpr int J = JMIN + decltype(jc)::value;
                constexpr int OWN = PL::P.cown[J];
                V[J] = quad_weighted<NB, PR0, FULLK, J>(r0, wv, K, e);
                if constexpr (OWN != 0) cacc[J] = __builtin_fma(V[J], wbv, cacc[J]);
            });
        }
        // ring of pairs: slot (c mod 6) holds the pair of chunk c; a step uses c + 1 (weights), c + DEPTH (row mask of
        // the refill) and loads c + DEPTH + 2 (kernel 1A's loop, fsnap_syrk.hip)
        u4 pk3 = pair_of(3), pk4 = {0u, 0u, 0u, 0u}, pk5 = {0u, 0u, 0u, 0u};
        if constexpr (DEPTH == 3) pk4 = pair_of(4);
        (void)pk0;
        unsigned cl = 0;
        // keep the four waves on the same rows (the barrier at the end of a trip): their tile counts differ by up to two
        // (NB = 11: 17 / 17 / 17 / 15), and a wave that runs ahead by more than its share of the L2 (4 MiB for the 32 workgroups
        // of an XCD) makes the other three fetch the rows again -- 1 772 880 x 168: 1.5 x the algorithmic bytes past the L2
        // without it, 1.0 x with it.  A bare s_barrier: no wait for the loads in flight; every wave runs the same number of trips.
        if constexpr (DEPTH == 3) {
            for (; cl + 3 < ncl; cl += 6) {
                QuadSeen seen{};
                if constexpr (FLOW && (W & 3) == 3)
                    if (flow->mode == 2) seen = quad_flow_look(*flow, (int)(cl / 6), lane);
                quad_step<NB, W, FULLK, PACK>(V, vt, r0, r1, wb, lpk, cl + 3, K, e, cacc, pk1, pk3, pk5);
                quad_step<NB, W, FULLK, PACK>(V, vt, r1, r2, wb, lpk, cl + 4, K, e, cacc, pk2, pk4, pk0);
                quad_step<NB, W, FULLK, PACK>(V, vt, r2, r0, wb, lpk, cl + 5, K, e, cacc, pk3, pk5, pk1);
                quad_step<NB, W, FULLK, PACK>(V, vt, r0, r1, wb, lpk, cl + 6, K, e, cacc, pk4, pk0, pk2);
                quad_step<NB, W, FULLK, PACK>(V, vt, r1, r2, wb, lpk, cl + 7, K, e, cacc, pk5, pk1, pk3);
                quad_step<NB, W, FULLK, PACK>(V, vt, r2, r0, wb, lpk, cl + 8, K, e, cacc, pk0, pk2, pk4);
                if constexpr (FLOW && (W & 3) == 3) {
                    if (flow->mode == 1) seen = quad_flow_look(*flow, (int)(cl / 6) + 1, lane);
                    if (flow->mode) quad_flow_wait(*flow, seen, (int)(cl / 6) + (flow->mode == 1 ? 1 : 0), lane);
                }
                if (QUAD_SYNC) __builtin_amdgcn_s_barrier();
            }
            if (cl < ncl) {
                quad_step<NB, W, FULLK, PACK>(V, vt, r0, r1, wb, lpk, cl + 3, K, e, cacc, pk1, pk3, pk5);
                quad_step<NB, W, FULLK, PACK>(V, vt, r1, r2, wb, lpk, cl + 4, K, e, cacc, pk2, pk4, pk0);
                quad_step<NB, W, FULLK, PACK>(V, vt, r2, r0, wb, lpk, cl + 5, K, e, cacc, pk3, pk5, pk1);
            }
        } else {
            for (; cl + 3 < ncl; cl += 6) {
                QuadSeen seen{};
                if constexpr (FLOW && (W & 3) == 3)
                    if (flow->mode == 2) seen = quad_flow_look(*flow, (int)(cl / 6), lane);
                quad_step<NB, W, FULLK, PACK>(V, vt, r0, r1, wb, lpk, cl + 2, K, e, cacc, pk1, pk2, pk4);
                quad_step<NB, W, FULLK, PACK>(V, vt, r1, r0, wb, lpk, cl + 3, K, e, cacc, pk2, pk3, pk5);
                quad_step<NB, W, FULLK, PACK>(V, vt, r0, r1, wb, lpk, cl + 4, K, e, cacc, pk3, pk4, pk0);
                quad_step<NB, W, FULLK, PACK>(V, vt, r1, r0, wb, lpk, cl + 5, K, e, cacc, pk4, pk5, pk1);
                quad_step<NB, W, FULLK, PACK>(V, vt, r0, r1, wb, lpk, cl + 6, K, e, cacc, pk5, pk0, pk2);
                quad_step<NB, W, FULLK, PACK>(V, vt, r1, r0, wb, lpk, cl + 7, K, e, cacc, pk0, pk1, pk3);
                if constexpr (FLOW && (W & 3) == 3) {
                    if (flow->mode == 1) seen = quad_flow_look(*flow, (int)(cl / 6) + 1, lane);
                    if (flow->mode) quad_flow_wait(*flow, seen, (int)(cl / 6) + (flow->mode == 1 ? 1 : 0), lane);
                }
                if (QUAD_SYNC) __builtin_amdgcn_s_barrier();
            }
            if (cl < ncl) {
                quad_step<NB, W, FULLK, PACK>(V, vt, r0, r1, wb, lpk, cl + 2, K, e, cacc, pk1, pk2, pk4);
                quad_step<NB, W, FULLK, PACK>(V, vt, r1, r0, wb, lpk, cl + 3, K, e, cacc, pk2, pk3, pk5);
                quad_step<NB, W, FULLK, PACK>(V, vt, r0, r1, wb, lpk, cl + 4, K, e, cacc, pk3, pk4, pk0);
            }
        }
    }
    // the last MFMAs (16 passes) must have left the pipe before their accumulators are read
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(vt[0]));
#pragma unroll
    for (int u = 1; u < NV; ++u) asm volatile("" : "+v"(vt[u]));
    // every tile straight to its place in the workgroup's partial triangle
    quad_for<NTW>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        constexpr int T = quad_tri_index(PL::P.tp[I], PL::P.tq[I], NB);
        const d4 x = quad_read<I, NV>(vt);
#pragma unroll
        for (int i = 0; i < 4; ++i) pw[(T * 4 + i) * 64 + lane] = x[i];
    });
    quad_for<NB>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        constexpr int OWN = PL::P.cown[J];
        double sm = 0.0;
        if constexpr (OWN != 0) {
            sm = cacc[J];
            sm += __shfl_xor(sm, 16, 64);
            sm += __shfl_xor(sm, 32, 64);
        }
        if (kr == 0) cw[J * 16 + e] = sm;
    });
}

}  // namespace

template <int NB, bool FULLK, bool PACK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void
fsnap_syrk_quad(const double* __restrict__ A, int64_t lda, const double* __restrict__ wpack, int64_t m, int K, int64_t chunks_per_wg,
                double* __restrict__ part, double* __restrict__ cpart, const double* __restrict__ bvec,
                const double* __restrict__ wvec, const unsigned char* __restrict__ mask, double* __restrict__ spart) {
    constexpr int NTILE = NB * (NB + 1) / 2;
    __shared__ __attribute__((aligned(16))) double lds[PACK ? QUAD_LDS_DOUBLES : 2];
    const int lane = threadIdx.x & 63;
    const int rw = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t nchunks = (m + 3) >> 2;
    int64_t c0 = (int64_t)blockIdx.x * chunks_per_wg;
    int64_t c1 = c0 + chunks_per_wg;
    if (c1 > nchunks) c1 = nchunks;
    if (c0 > c1) c0 = c1;
    const int64_t row0 = c0 << 2;
    int64_t row1 = c1 << 2;
    if (row1 > m) row1 = m;
    const int64_t nrow = row1 > row0 ? row1 - row0 : 0;
    const unsigned ncl = (unsigned)(c1 - c0);

    // prologue: (w_eff, w_eff b) of the workgroup's rows -> LDS, thread t takes rows t, t + 256, ... (bounds-checked
    // loads: rows past the range read zeros and become (0, 0) pairs, what the loop's look-ahead expects); the
    // b-only scalars of a wave's share leave as that wave's partial
    if constexpr (PACK) {
        const unsigned wg_rows = (unsigned)chunks_per_wg * 4u;
        const unsigned region_rows = wg_rows + QUAD_PACK_PAD * 4u;
        const __amdgpu_buffer_rsrc_t rb = quad_rsrc(bvec + row0, (unsigned)(nrow * 8));
        const __amdgpu_buffer_rsrc_t rw_ = quad_rsrc(wvec + row0, (unsigned)(nrow * 8));
        const __amdgpu_buffer_rsrc_t rm = quad_rsrc(mask + row0, (unsigned)nrow);
        constexpr int PB = 8;
        double bb = 0.0, sb = 0.0, cnt = 0.0;
        for (unsigned r0 = 0; r0 < region_rows; r0 += 256u * PB) {
            u2 bv[PB], wv[PB];
            unsigned char mk[PB];
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const unsigned row = r0 + 256u * u + threadIdx.x;
                bv[u] = __builtin_amdgcn_raw_buffer_load_b64(rb, row * 8u, 0, 0);
                wv[u] = __builtin_amdgcn_raw_buffer_load_b64(rw_, row * 8u, 0, 0);
                mk[u] = __builtin_amdgcn_raw_buffer_load_b8(rm, row, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const unsigned row = r0 + 256u * u + threadIdx.x;
                const bool keep = (mk[u] != 0);
                const double wvv = keep ? __builtin_bit_cast(double, wv[u]) : 0.0;
                const double wbv = keep ? wvv * __builtin_bit_cast(double, bv[u]) : 0.0;
                if (row < region_rows) {
                    d2 o;
                    o[0] = wvv;
                    o[1] = wbv;
                    *reinterpret_cast<d2*>(lds + (size_t)row * 2) = o;
                }
                bb = __builtin_fma(wbv, wbv, bb);
                sb += wbv;
                cnt += keep ? 1.0 : 0.0;
            }
        }
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) {       // fixed butterfly: deterministic
            bb += __shfl_xor(bb, sh, 64);
            sb += __shfl_xor(sb, sh, 64);
            cnt += __shfl_xor(cnt, sh, 64);
        }
        if (lane == 0) {
            double* so = spart + ((int64_t)blockIdx.x * 4 + rw) * 4;
            so[0] = bb;
            so[1] = sb;
            so[2] = cnt;
            so[3] = 0.0;
        }
        __syncthreads();
    }

    // the compiler must count a[0:255] as used (register allocation granule of the kernel descriptor)
    asm volatile("" : : : "a0", "a255");
    quad_zero_all(std::make_integer_sequence<int, 256>{});
    double* pw = part + (int64_t)blockIdx.x * (int64_t)(NTILE * 256);
    double* cw = cpart + ((int64_t)blockIdx.x * 4 + rw) * (int64_t)(NB * 16);
    switch (rw) {
        case 0: quad_wave<NB, 0, FULLK, PACK>(A, lda, wpack, K, row0, nrow, ncl, lds, pw, cw, lane); break;
        case 1: quad_wave<NB, 1, FULLK, PACK>(A, lda, wpack, K, row0, nrow, ncl, lds, pw, cw, lane); break;
        case 2: quad_wave<NB, 2, FULLK, PACK>(A, lda, wpack, K, row0, nrow, ncl, lds, pw, cw, lane); break;
        default: quad_wave<NB, 3, FULLK, PACK>(A, lda, wpack, K, row0, nrow, ncl, lds, pw, cw, lane); break;
    }
}

// Kernel 1QC: kernel 1Q's plan on a cluster of C = quad_cluster(NB) workgroups (NB = 19 ... 32 column blocks: 289 ... 512
// columns).  Grid: 8 C clusters-per-XCD slots -- block b runs on XCD b % 8 (observed placement, used for speed only): member c
// of cluster (j, x) is block x + 8 (c + C j), so the members of a cluster share an L2.  Every member forms the per-row pairs
// of the cluster's rows in ITS LDS (17 bytes per row against 8 K: nothing), member 0 alone reports the b-only scalars.
// Partials in kernel 1Q's layout with "workgroup" read as "cluster": part[cluster][NT][4][64] (every tile written by the one
// wave of the cluster that owns it), cpart[cluster * 4 C + wave][NB][16], spart[cluster * 4 C + wave][4]; kernel 2b reduces
// them with cs_per_block = 4 C.  Always the select form of the last column block (FULLK = false) and fused packing.
template <int NB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void
fsnap_syrk_quadc(const double* __restrict__ A, int64_t lda, int64_t m, int K, int64_t chunks_per_cluster, int nclusters,
                 double* __restrict__ part, double* __restrict__ cpart, const double* __restrict__ bvec,
                 const double* __restrict__ wvec, const unsigned char* __restrict__ mask, double* __restrict__ spart,
                 int* __restrict__ flow_words, int flow_tag) {
    constexpr int NTILE = NB * (NB + 1) / 2;
    constexpr int C = quad_cluster(NB), NW = 4 * C;
    __shared__ __attribute__((aligned(16))) double lds[QUAD_LDS_DOUBLES];
    const int lane = threadIdx.x & 63;
    const int rw = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int xcd = (int)(blockIdx.x & 7), slot = (int)(blockIdx.x >> 3);
    const int member = slot % C, cluster = (slot / C) * 8 + xcd;
    if (cluster >= nclusters) return;
    const int64_t nchunks = (m + 3) >> 2;
    int64_t c0 = (int64_t)cluster * chunks_per_cluster;
    int64_t c1 = c0 + chunks_per_cluster;
    if (c1 > nchunks) c1 = nchunks;
    if (c0 > c1) c0 = c1;
    const int64_t row0 = c0 << 2;
    int64_t row1 = c1 << 2;
    if (row1 > m) row1 = m;
    const int64_t nrow = row1 > row0 ? row1 - row0 : 0;
    const unsigned ncl = (unsigned)(c1 - c0);
    const int W = 4 * member + rw;

    // prologue (kernel 1Q's): (w_eff, w_eff b) of the cluster's rows -> this member's LDS; the b-only scalars from member 0
    {
        const unsigned wg_rows = (unsigned)chunks_per_cluster * 4u;
        const unsigned region_rows = wg_rows + QUAD_PACK_PAD * 4u;
        const __amdgpu_buffer_rsrc_t rb = quad_rsrc(bvec + row0, (unsigned)(nrow * 8));
        const __amdgpu_buffer_rsrc_t rw_ = quad_rsrc(wvec + row0, (unsigned)(nrow * 8));
        const __amdgpu_buffer_rsrc_t rm = quad_rsrc(mask + row0, (unsigned)nrow);
        constexpr int PB = 8;
        double bb = 0.0, sb = 0.0, cnt = 0.0;
        for (unsigned r0 = 0; r0 < region_rows; r0 += 256u * PB) {
            u2 bv[PB], wv[PB];
            unsigned char mk[PB];
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const unsigned row = r0 + 256u * u + threadIdx.x;
                bv[u] = __builtin_amdgcn_raw_buffer_load_b64(rb, row * 8u, 0, 0);
                wv[u] = __builtin_amdgcn_raw_buffer_load_b64(rw_, row * 8u, 0, 0);
                mk[u] = __builtin_amdgcn_raw_buffer_load_b8(rm, row, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const unsigned row = r0 + 256u * u + threadIdx.x;
                const bool keep = (mk[u] != 0);
                const double wvv = keep ? __builtin_bit_cast(double, wv[u]) : 0.0;
                const double wbv = keep ? wvv * __builtin_bit_cast(double, bv[u]) : 0.0;
                if (row < region_rows) {
                    d2 o;
                    o[0] = wvv;
                    o[1] = wbv;
                    *reinterpret_cast<d2*>(lds + (size_t)row * 2) = o;
                }
                bb = __builtin_fma(wbv, wbv, bb);
                sb += wbv;
                cnt += keep ? 1.0 : 0.0;
            }
        }
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) {       // fixed butterfly: deterministic
            bb += __shfl_xor(bb, sh, 64);
            sb += __shfl_xor(sb, sh, 64);
            cnt += __shfl_xor(cnt, sh, 64);
        }
        if (lane == 0) {
            double* so = spart + ((int64_t)cluster * NW + W) * 4;
            const bool rep = member == 0;
            so[0] = rep ? bb : 0.0;
            so[1] = rep ? sb : 0.0;
            so[2] = rep ? cnt : 0.0;
            so[3] = 0.0;
        }
        __syncthreads();
    }

    asm volatile("" : : : "a0", "a255");
    quad_zero_all(std::make_integer_sequence<int, 256>{});
    double* pw = part + (int64_t)cluster * (int64_t)(NTILE * 256);
    double* cw = cpart + ((int64_t)cluster * NW + W) * (int64_t)(NB * 16);
    QuadFlow flow;
    flow.word = flow_words + (int64_t)cluster * 4;
    flow.members = C;
    flow.me = member;
    flow.tag = flow_tag & ~0xFF;
    flow.mode = flow_tag & 3;
    flow.lead = (flow_tag >> 2) & 63;
#define FSNAP_QC_CASE(WV) \
    case WV: quad_wave<NB, WV, false, true, true>(A, lda, nullptr, K, row0, nrow, ncl, lds, pw, cw, lane, &flow); break;
    switch (W) {
        FSNAP_QC_CASE(0) FSNAP_QC_CASE(1) FSNAP_QC_CASE(2) FSNAP_QC_CASE(3) FSNAP_QC_CASE(4) FSNAP_QC_CASE(5) FSNAP_QC_CASE(6)
        default:
            if constexpr (C == 2) {
                quad_wave<NB, 7, false, true, true>(A, lda, nullptr, K, row0, nrow, ncl, lds, pw, cw, lane, &flow);
            } else {
                switch (W) {
                    FSNAP_QC_CASE(7) FSNAP_QC_CASE(8) FSNAP_QC_CASE(9) FSNAP_QC_CASE(10) FSNAP_QC_CASE(11) FSNAP_QC_CASE(12)
                    FSNAP_QC_CASE(13) FSNAP_QC_CASE(14)
                    default: quad_wave<NB, (C == 4 ? 15 : 7), false, true, true>(A, lda, nullptr, K, row0, nrow, ncl, lds, pw, cw, lane, &flow); break;
                }
            }
            break;
    }
#undef FSNAP_QC_CASE
}

// The instantiations are compiled in three parts (fitsnap_amd/build.py: -DFSNAP_QUAD_PART=0 | 1 | 2, three objects in parallel:
// one translation unit with all 23 widths took three minutes, the longest of the build); without the macro (tools that
// include this file) everything is in one unit.
#ifndef FSNAP_QUAD_PART
#define FSNAP_QUAD_PART_ALL 1
#define FSNAP_QUAD_PART 0
#endif

namespace fsnap {

hipError_t launch_syrk_quad_part1(const SyrkArgs& a, hipStream_t st);     // NB = 19 ... 25
hipError_t launch_syrk_quad_part2(const SyrkArgs& a, hipStream_t st);     // NB = 26 ... 32

// chunks per workgroup up to which the workgroup's per-row pairs fit the LDS next to the look-ahead pad
static int64_t quad_max_cpg() { return QUAD_LDS_DOUBLES / 8 - QUAD_PACK_PAD; }

#if FSNAP_QUAD_PART == 0
int64_t syrk_quad_max_cpg() { return quad_max_cpg(); }

template <int NB>
static hipError_t launch_syrk_quad_nb(const SyrkArgs& a, hipStream_t st) {
    dim3 grid((unsigned)a.nblocks), block(256);
    if (a.fused_pack ? (!a.b || !a.w || !a.mask || !a.spart || a.chunks_per_wave > quad_max_cpg()) : !a.wpack)
        return hipErrorInvalidValue;
#define FSNAP_LAUNCH(FK, PK)                                                                                                   \
    hipLaunchKernelGGL((fsnap_syrk_quad<NB, FK, PK>), grid, block, 0, st, a.A, a.lda, a.wpack, a.m, a.K, a.chunks_per_wave, a.part, \
                       a.cpart, a.b, a.w, a.mask, a.spart)
    if (a.K == 16 * NB) {
        if (a.fused_pack) FSNAP_LAUNCH(true, true);
        else FSNAP_LAUNCH(true, false);
    } else {
        if (a.fused_pack) FSNAP_LAUNCH(false, true);
        else FSNAP_LAUNCH(false, false);
    }
#undef FSNAP_LAUNCH
    return hipGetLastError();
}
#endif

// kernel 1QC (NB = 19 ... 32): a.nblocks CLUSTERS of quad_cluster(NB) workgroups, a.chunks_per_wave = chunks per cluster,
// a.flow_words (4 ints per cluster, zero-initialised once, never reset) and a.flow_tag (+ 2^20 per launch)
template <int NB>
static hipError_t launch_syrk_quadc_nb(const SyrkArgs& a, hipStream_t st) {
    constexpr int C = quad_cluster(NB);
    if (!a.b || !a.w || !a.mask || !a.spart || !a.flow_words || a.chunks_per_wave > quad_max_cpg()) return hipErrorInvalidValue;
    const unsigned per_xcd = (unsigned)((a.nblocks + 7) / 8) * C;
    hipLaunchKernelGGL((fsnap_syrk_quadc<NB>), dim3(8 * per_xcd), dim3(256), 0, st, a.A, a.lda, a.m, a.K, a.chunks_per_wave,
                       (int)a.nblocks, a.part, a.cpart, a.b, a.w, a.mask, a.spart, a.flow_words, a.flow_tag);
    return hipGetLastError();
}

#if FSNAP_QUAD_PART == 1 || defined(FSNAP_QUAD_PART_ALL)
hipError_t launch_syrk_quad_part1(const SyrkArgs& a, hipStream_t st) {
    switch ((a.K + 15) / 16) {
        case 19: return launch_syrk_quadc_nb<19>(a, st);
        case 20: return launch_syrk_quadc_nb<20>(a, st);
        case 21: return launch_syrk_quadc_nb<21>(a, st);
        case 22: return launch_syrk_quadc_nb<22>(a, st);
        case 23: return launch_syrk_quadc_nb<23>(a, st);
        case 24: return launch_syrk_quadc_nb<24>(a, st);
        case 25: return launch_syrk_quadc_nb<25>(a, st);
        default: return hipErrorInvalidValue;
    }
}
#endif

#if FSNAP_QUAD_PART == 2 || defined(FSNAP_QUAD_PART_ALL)
hipError_t launch_syrk_quad_part2(const SyrkArgs& a, hipStream_t st) {
    switch ((a.K + 15) / 16) {
        case 26: return launch_syrk_quadc_nb<26>(a, st);
        case 27: return launch_syrk_quadc_nb<27>(a, st);
        case 28: return launch_syrk_quadc_nb<28>(a, st);
        case 29: return launch_syrk_quadc_nb<29>(a, st);
        case 30: return launch_syrk_quadc_nb<30>(a, st);
        case 31: return launch_syrk_quadc_nb<31>(a, st);
        case 32: return launch_syrk_quadc_nb<32>(a, st);
        default: return hipErrorInvalidValue;
    }
}
#endif

#if FSNAP_QUAD_PART == 0
int syrk_quad_cluster(int K) { return quad_cluster((K + 15) / 16); }

// kernel 1Q: a.nblocks workgroups, a.chunks_per_wave = 4-row chunks per WORKGROUP (its four waves sweep the same rows);
// a.fused_pack: the kernel forms the per-row pairs itself (b, w, mask, spart), otherwise it reads a.wpack
hipError_t launch_syrk_quad(const SyrkArgs& a, hipStream_t st) {
    const int nb = (a.K + 15) / 16;
    switch (nb) {
        case 10: return launch_syrk_quad_nb<10>(a, st);
        case 11: return launch_syrk_quad_nb<11>(a, st);
        case 12: return launch_syrk_quad_nb<12>(a, st);
        case 13: return launch_syrk_quad_nb<13>(a, st);
        case 14: return launch_syrk_quad_nb<14>(a, st);
        case 15: return launch_syrk_quad_nb<15>(a, st);
        case 16: return launch_syrk_quad_nb<16>(a, st);
        case 17: return launch_syrk_quad_nb<17>(a, st);
        case 18: return launch_syrk_quad_nb<18>(a, st);
        default: return nb >= 19 && nb <= 25 ? launch_syrk_quad_part1(a, st) : nb >= 26 && nb <= 32 ? launch_syrk_quad_part2(a, st) : hipErrorInvalidValue;
    }
}
#endif

}  // namespace fsnap
