// fsnap_syrk.hip — hand-written gfx950 (MI355X / CDNA4) kernels of the FitSNAP linear-fit hot path: the fused
// mask x weight x normal-equation (SYRK) kernels and the reduction of their partials.  The HBM-bound row kernels
// (weighting, GEMV, assembly, error statistics) are in fsnap_rows.hip, the device Cholesky solves in fsnap_chol.hip.  No CUDA compatibility layer, no dual paths: this file only
// builds for gfx950 (wave64, v_mfma_f64_16x16x4_f64).
//
// What the kernels replace in the reference (file:line into FitSNAP/FitSNAP):
//   * fitsnap3lib/solvers/svd.py:35-46, ridge.py:28-39, ard.py:18-20
//       training mask -> aw = w[:,None]*A[training], bw = w*b[training]
//   * fitsnap3lib/solvers/svd.py:50-51, ridge.py:41-43, ard.py:22-24,
//     fitsnap3lib/lib/ridge_solver/regressor.py:11-12,
//     examples/library/transpose_trick/example.py:234-240
//       G = aw.T @ aw ,  c = aw.T @ bw      (the "transpose trick")
//   * fitsnap3lib/solvers/solver.py:377   preds = a @ fit
//
// Data layout in HBM: A is row-major fp64, m rows x K columns, leading dimension
// lda (doubles, lda >= K); b, w are fp64[m]; mask is uint8[m] (1 = training row; the
// C-ABI layer substitutes an all-ones buffer when the caller passes no mask).  The allocation that holds A is padded by >= 256 B so that
// the vector loads of the last row may over-read (they are select-zeroed).
//
// Kernels (details at each definition):
//   1A  fsnap_syrk_acc         80 < K <= 144: one wave per SIMD owns the whole tile triangle (32 tiles in the
//                              accumulation registers, named in inline assembly), streams its own rows, no LDS
//   1P  fsnap_syrk_wave_p      K <= 80: the triangle (<= 15 tiles) in compiler-allocated registers, five waves per
//                              SIMD, rows masked by the loads
//   1T  fsnap_syrk_tiled       general K: 64-column superblock pairs x row splits
//   2b  fsnap_reduce_partials2 / fsnap_reduce_tiled*   deterministic fixed-order reduction of the
//                              per-workgroup partial triangles -> packed [G | c | scalars];
//                              un-permutes the even/odd column interleave, mirrors the triangle
//   (1Q / 1QC, 144 < K <= 512: fsnap_syrk_quad.hip.  Kernels 1, 1L, 1T2 and reduction kernel 2 of rounds 1-4 lost
//   their A/B runs and are gone; their records are in profiles/ and HISTORY.md.)
//
// Common operand trick of all SYRK kernels: v_mfma_f64_16x16x4_f64 takes
// A[i = lane&15][k = lane>>4] and B[k = lane>>4][j = lane&15]; with k = row inside a 4-row
// chunk and i/j = column inside a 16-column block, the SAME register (w * a[row][col]) is
// the A operand of tile (p, .) and the B operand of tile (., p): no transposes, A is read
// from HBM once.  16-byte loads give a lane two ADJACENT columns; they go to two different
// column blocks (even / odd columns of a 32-column group) and the column permutation is
// undone for free by the reduction kernel's scatter.  c = (wA)^T (wb), b^T W^2 b, sum(wb)
// and the training-row count ride along on the VALU.  No floating-point atomics anywhere:
// results are run-to-run bit-identical for a given (m, K, grid).

#include <utility>

#include "fsnap_device_common.h"
#include "fsnap_kernels.h"

namespace {

__host__ __device__ constexpr int tri_index(int p, int q, int NB) {
    // index of upper-triangle tile (p <= q) in row-major packed order
    return p * NB - (p * (p - 1)) / 2 + (q - p);
}

// column of A held by element e (0..15) of column block bq.
// Blocks 2j, 2j+1 (pair j) interleave the even / odd columns of [32j, 32j+32);
// if NB is odd the last block is a plain 16-column block.
__host__ __device__ inline int col_of(int bq, int e, int NB) {
    if ((NB & 1) && bq == NB - 1) return 16 * bq + e;
    return 32 * (bq >> 1) + 2 * e + (bq & 1);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    // dword3 0x00020000: raw buffer, 32-bit data format (gfx9-family encoding)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ double xlane_sum_rows(double x) {
    // sum over the four 16-lane row groups (lanes l, l^16, l^32, l^48)
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    return x;
}

}  // namespace

#ifdef FSNAP_TRACE
// tools/syrk_trace.hip only: per-workgroup {start, end} (100 MHz wall clock), HW_ID, XCC_ID, and
// per-wave shader-clock cycles spent in {MFMA phase, park + load issue, barrier} of the stage loop
__device__ unsigned long long fsnap_trace_buf[4096 * 8];
__device__ unsigned long long fsnap_trace_wave[4096 * 16 * 4];
#define FSNAP_TRACE_CLK(v) const unsigned long long v = __builtin_readcyclecounter()
#endif

// ---------------------------------------------------------------------------------
// Kernel 1A: fused mask x weight x SYRK with the WHOLE tile triangle resident in ONE wave
// (80 < K <= 128).  One wave per SIMD (4-wave workgroups, one per CU): the wave streams its own
// rows straight from HBM into MFMA-fragment registers (no LDS, no barrier in the loop, every
// row fetched once) and keeps all NB(NB+1)/2 <= 36 accumulator tiles: tiles 0..31 in the 256
// accumulation registers a[0:255], tiles 32..35 in VGPRs.  The compiler cannot allocate 288
// accumulator registers across both files (it shuffles every tile through v_accvgpr moves),
// so the MFMAs name their AGPR tiles explicitly in inline assembly; everything else (loads,
// weighting, masks, c) is ordinary compiler-scheduled code placed BETWEEN the MFMA rows:
// a wave issues in order, so VALU / VMEM instructions run in the shadow of the 64-cycle
// MFMAs only if they sit between them.
// Operand block p of the NEXT chunk overwrites V[p] right after row p of the CURRENT chunk
// (tiles (p, p..NB-1)) has been issued -- block p is not read again in this chunk -- so one
// operand set suffices and the loads run three chunks ahead.
// tools/mfma_f64_peak.hip ("stream step 4"): this instruction mix sustains ~66 TF/s on
// random data (one or two waves per SIMD), the LDS-shared kernel 1L ~50-54.
// Partial layout = kernel 1: part[workgroup][NT][4][64] (4 row-waves folded through LDS) |
// cpart[rowwave][NB][16] | spart[rowwave][4].
// ---------------------------------------------------------------------------------
namespace {

template <int T, int NV>
__device__ __forceinline__ void acc_mfma(double a, double b, d4 (&vt)[NV]) {
    if constexpr (T < 32) {
        asm volatile("v_mfma_f64_16x16x4_f64 a[%2:%3], %0, %1, a[%2:%3]" : : "v"(a), "v"(b), "n"(8 * T), "n"(8 * T + 7));
    } else {
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(vt[T - 32]) : "v"(a), "v"(b));
    }
}

template <int R>
__device__ __forceinline__ void acc_zero_reg() {
    asm volatile("v_accvgpr_write_b32 a[%0], 0" : : "n"(R));
}
template <int... R>
__device__ __forceinline__ void acc_zero_all(std::integer_sequence<int, R...>) {
    (acc_zero_reg<R>(), ...);
}

template <int T, int NV>
__device__ __forceinline__ d4 acc_read(const d4 (&vt)[NV]) {
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

template <int TB, int N, int NV, int... U>
__device__ __forceinline__ void acc_read_range(d4 (&tmp)[N], const d4 (&vt)[NV], std::integer_sequence<int, U...>) {
    ((tmp[U] = acc_read<TB + U, NV>(vt)), ...);
}

// row p of the triangle: tiles (p, p..NB-1)
template <int NB, int P, int NV, int... Q>
__device__ __forceinline__ void acc_row(const double (&V)[NB], d4 (&vt)[NV], std::integer_sequence<int, Q...>) {
    (acc_mfma<tri_index(P, P + Q, NB), NV>(V[P], V[P + Q], vt), ...);
}

// Raw loads of kernel 1A.  Per row the kernel needs three numbers that do not depend on A: keep (training row?),
// w and w*b.  They are packed once per (b, w, mask) by fsnap_pack_weights_k into wpack[row] = (w_eff, wb_eff) with
// w_eff = keep ? w : 0, wb_eff = keep ? w*b : 0 (16 bytes, ONE load per chunk instead of three), together with the
// scalars b^T W^2 b, sum(w b), n_train, which therefore leave this kernel altogether.
// The row mask is applied by the LOADS: a lane whose row has w_eff == 0 (test row, zero-weight row, or a row past
// the wave's range, whose packed entry reads back as zero) uses an out-of-range buffer offset for its A values, so
// the hardware bounds check returns zeros -- a = 0 and w_eff = 0 make every product of that row vanish without a
// select instruction, and garbage (NaN / Inf) in such a row is never even fetched.  Every VALU / VMEM instruction of
// this kernel costs matrix-pipe time (fp64 MFMAs and VALU instructions serialise on the SIMD): per 4-row chunk
// 8 multiplies, 8 FMAs (c), 1 compare, 1-2 offset selects and 5 loads remain.
template <int NB>
struct RawM {
    u4 pr[NB / 2 > 0 ? NB / 2 : 1];
    u2 tail;
    u4 wp;      // (w_eff, wb_eff) of the lane's row
};

constexpr unsigned FSNAP_OOB_VOFF = 0xFFFFF000u;   // > any wave's buffer size (plan_geometry: < 0xFFF00010), no 32-bit wrap

struct WaveBufsP {
    __amdgpu_buffer_rsrc_t A, wp;
    unsigned voffA;        // per-lane byte offset inside a chunk: (kr*lda + 2e)*8
    unsigned voffT;        // tail block: (kr*lda + 16*(NB-1) + e)*8
    unsigned voffP;        // kr * 16 (packed weights)
    unsigned chunk_bytes;  // 4*lda*8
};

__device__ __forceinline__ u4 load_pack(const WaveBufsP& wb, unsigned cl) {
    return __builtin_amdgcn_raw_buffer_load_b128(wb.wp, wb.voffP, cl * 64u, 0);
}
__device__ __forceinline__ bool pack_keep(const u4& wp) {
    return ((wp[0] | (wp[1] & 0x7FFFFFFFu)) != 0u);      // w_eff != +-0 (integer test: no fp64 VALU)
}

template <int NB, bool NT>
__device__ __forceinline__ void issue_rows(RawM<NB>& r, const WaveBufsP& wb, unsigned cl) {
    const unsigned soff = cl * wb.chunk_bytes;
    constexpr int AUX = NT ? 2 : 0;
    const bool keep = pack_keep(r.wp);
    const unsigned va = keep ? wb.voffA : FSNAP_OOB_VOFF;
#pragma unroll
    for (int j = 0; j < NB / 2; ++j) r.pr[j] = __builtin_amdgcn_raw_buffer_load_b128(wb.A, va + 256u * j, soff, AUX);
    if (NB & 1) {
        const unsigned vtl = keep ? wb.voffT : FSNAP_OOB_VOFF;
        r.tail = __builtin_amdgcn_raw_buffer_load_b64(wb.A, vtl, soff, AUX);
    }
}

// the same with the packed pair that gates the rows handed in (kernel 1P keeps the pairs in a ring of their own)
template <int NB, bool NT>
__device__ __forceinline__ void issue_rows_w(RawM<NB>& r, const u4& wp, const WaveBufsP& wb, unsigned cl) {
    const unsigned soff = cl * wb.chunk_bytes;
    constexpr int AUX = NT ? 2 : 0;
    const bool keep = pack_keep(wp);
    const unsigned va = keep ? wb.voffA : FSNAP_OOB_VOFF;
#pragma unroll
    for (int j = 0; j < NB / 2; ++j) r.pr[j] = __builtin_amdgcn_raw_buffer_load_b128(wb.A, va + 256u * j, soff, AUX);
    if (NB & 1) {
        const unsigned vtl = keep ? wb.voffT : FSNAP_OOB_VOFF;
        r.tail = __builtin_amdgcn_raw_buffer_load_b64(wb.A, vtl, soff, AUX);
    }
}

// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>) -- ring slots stay
// register names
template <class F, int... I>
__device__ __forceinline__ void wave_p_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void wave_p_for(F&& f) {
    wave_p_for_impl(f, std::make_integer_sequence<int, N>{});
}

// piece P of the refill of raw set r (one load instruction per MFMA slot: a block of back-to-back VMEM
// instructions holds the in-order wave at the address path while the matrix pipe drains)
template <int NB, bool NT, int P>
__device__ __forceinline__ void issue_piece(RawM<NB>& r, const WaveBufsP& wb, unsigned cl, unsigned va, unsigned vtl) {
    constexpr int AUX = NT ? 2 : 0;
    constexpr int NPR = NB / 2;
    const unsigned soff = cl * wb.chunk_bytes;
    if (P < NPR) r.pr[P] = __builtin_amdgcn_raw_buffer_load_b128(wb.A, va + 256u * P, soff, AUX);
    if ((NB & 1) && P == NPR) r.tail = __builtin_amdgcn_raw_buffer_load_b64(wb.A, vtl, soff, AUX);
    static_assert(NPR + (NB & 1) <= NB, "one load piece per MFMA slot");
}

// w * (raw value of block j); columns >= K (only possible in the last block / block pair when K is not a multiple
// of 16) are zeroed by a select
template <int NB, bool FULLK>
__device__ __forceinline__ double weighted_block(const RawM<NB>& r, int j, double wv, int K, int e) {
    if ((NB & 1) && j == NB - 1) {
        const double x = wv * __builtin_bit_cast(double, r.tail);
        if (FULLK) return x;
        return (16 * (NB - 1) + e < K) ? x : 0.0;
    }
    const double x = wv * __builtin_bit_cast(d2, r.pr[j >> 1])[j & 1];
    // K > 16 * (NB - 1): only the last even/odd block pair (NB even) can hold columns >= K
    constexpr int last_pair = (NB & 1) ? -1 : NB / 2 - 1;
    if (FULLK || (j >> 1) != last_pair) return x;
    return (32 * (j >> 1) + 2 * e + (j & 1) < K) ? x : 0.0;
}

// The packed pair of chunk cl.  PACK = false: from the wpack array in HBM (fsnap_pack_weights_k, or the per-row pairs
// of a row-space pass).  PACK = true: from the wave's own LDS region, filled by the kernel's prologue (pack_rows_to_lds)
// -- lpk = region + 2 * kr doubles; lanes of one row group read the same 16 bytes (broadcast).
template <bool PACK>
__device__ __forceinline__ u4 load_pack_x(const WaveBufsP& wb, const double* lpk, unsigned cl) {
    if constexpr (PACK) return __builtin_bit_cast(u4, *reinterpret_cast<const d2*>(lpk + (size_t)cl * 8));
    else return load_pack(wb, cl);
}

// Prologue of kernel 1A with PACK = true: the wave forms (w_eff, w_eff b) of ITS rows in its own LDS region -- what
// fsnap_pack_weights_k wrote to HBM in a launch of its own (9.8 us + a stream boundary in front of every fit whose
// weights changed) -- and the three statistics that do not involve A.  region_rows = 4 x (chunks per wave + the
// overshoot of the unrolled loop); rows past the wave's range read zeros through the bounds-checked descriptors and
// become (0, 0) pairs, which is what the loop's look-ahead expects there.  PB x 3 loads in flight per lane.
__device__ __forceinline__ void pack_rows_to_lds(const double* __restrict__ b, const double* __restrict__ w,
                                                 const unsigned char* __restrict__ mask, int64_t row0, int64_t nrow,
                                                 unsigned wave_rows, unsigned region_rows, double* lpw, int lane, double* sout) {
    constexpr int PB = 16;     // 16 x 64 rows per batch: the 980 rows a wave owns at 10^6 x 128 are ONE round trip to HBM
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(b + row0, (unsigned)(nrow * 8));
    const __amdgpu_buffer_rsrc_t rw_ = make_rsrc(w + row0, (unsigned)(nrow * 8));
    const __amdgpu_buffer_rsrc_t rm = make_rsrc(mask + row0, (unsigned)nrow);
    double bb = 0.0, sb = 0.0, cnt = 0.0;
    for (unsigned r0 = 0; r0 < wave_rows; r0 += 64u * PB) {
        u2 bv[PB], wv[PB];
        unsigned char mk[PB];
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const unsigned row = r0 + 64u * u + (unsigned)lane;
            bv[u] = __builtin_amdgcn_raw_buffer_load_b64(rb, row * 8u, 0, 0);
            wv[u] = __builtin_amdgcn_raw_buffer_load_b64(rw_, row * 8u, 0, 0);
            mk[u] = __builtin_amdgcn_raw_buffer_load_b8(rm, row, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const unsigned row = r0 + 64u * u + (unsigned)lane;
            const bool keep = (mk[u] != 0);
            const double wvv = keep ? __builtin_bit_cast(double, wv[u]) : 0.0;
            const double wbv = keep ? wvv * __builtin_bit_cast(double, bv[u]) : 0.0;
            if (row < wave_rows) {
                d2 o;
                o[0] = wvv;
                o[1] = wbv;
                *reinterpret_cast<d2*>(lpw + (size_t)row * 2) = o;
            }
            bb = __builtin_fma(wbv, wbv, bb);
            sb += wbv;
            cnt += keep ? 1.0 : 0.0;
        }
    }
    // the look-ahead pad behind the wave's rows: zeros, no loads
    for (unsigned row = wave_rows + (unsigned)lane; row < region_rows; row += 64u) {
        const d2 z = {0.0, 0.0};
        *reinterpret_cast<d2*>(lpw + (size_t)row * 2) = z;
    }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) {       // fixed butterfly: deterministic
        bb += __shfl_xor(bb, sh, 64);
        sb += __shfl_xor(sb, sh, 64);
        cnt += __shfl_xor(cnt, sh, 64);
    }
    if (lane == 0) {
        sout[0] = bb;
        sout[1] = sb;
        sout[2] = cnt;
        sout[3] = 0.0;
    }
}

// Slot P of a step (between row P and row P + 1 of the MFMAs).  The pieces are independent of each other (a wave
// issues in order: a dependent chain here would hold back row P + 1):
//   V[P] <- w * raw block P        cacc[P-1] += V[P-1] * wbv        one load of the refill
template <int NB, bool FULLK, bool NT, int P>
__device__ __forceinline__ void acc_slot(double (&V)[NB], const RawM<NB>& RN, double wv, double wbv, double& wbp, int K,
                                         int e, double (&cacc)[NB], RawM<NB>& RF, const WaveBufsP& wb, unsigned cl_fill,
                                         unsigned va, unsigned vtl) {
#if !defined(FSNAP_ACC_ABL) || !(FSNAP_ACC_ABL & 1)
    issue_piece<NB, NT, P>(RF, wb, cl_fill, va, vtl);
#endif
    if (P == 0) cacc[NB - 1] = __builtin_fma(V[NB - 1], wbp, cacc[NB - 1]);   // previous chunk's last block
    else cacc[P - 1] = __builtin_fma(V[P - 1], wbv, cacc[P - 1]);
    V[P] = weighted_block<NB, FULLK>(RN, P, wv, K, e);
    if (P == NB - 1) wbp = wbv;
    __builtin_amdgcn_sched_barrier(0);   // keep the slot between row P and row P + 1
}

// One chunk (step c): the MFMA rows of chunk c held in V, interleaved with the preparation of chunk c + 1 (raw set
// RN, packed weights PW) block by block.  RF (consumed one step ago) gets the rows of chunk c + 3, gated by the
// packed weights PKEEP of that chunk, which were loaded TWO steps ago: the packed pairs stream from HBM like the
// rows, and a wave that needs one only a step (~1 us) after issuing its load stalls on it (the packs live in a ring of
// six registers of their own, PLOAD is the slot for chunk c + 5).  The packed load goes out BEFORE the row loads:
// vmcnt retires in order, so a later wait for it does not drain the row loads issued here.  (PACK: the pairs come
// from LDS with the same ring; the distance is then far more than the ~100 cycles an LDS read takes.)
template <int NB, bool FULLK, bool NT, bool PACK, int NV, int... P>
__device__ __forceinline__ void acc_step(double (&V)[NB], d4 (&vt)[NV], RawM<NB>& RF, const RawM<NB>& RN, const WaveBufsP& wb,
                                         const double* lpk, unsigned cl_fill, int K, int e, double (&cacc)[NB], double& wbp,
                                         const u4& PW, const u4& PKEEP, u4& PLOAD, std::integer_sequence<int, P...>) {
    const d2 wpn = __builtin_bit_cast(d2, PW);
    const double wv = wpn[0], wbv = wpn[1];
    const bool keep = pack_keep(PKEEP);
    const unsigned va = keep ? wb.voffA : FSNAP_OOB_VOFF;
    const unsigned vtl = (NB & 1) ? (keep ? wb.voffT : FSNAP_OOB_VOFF) : 0u;
#if !defined(FSNAP_ACC_ABL) || !(FSNAP_ACC_ABL & 1)   // tools/syrk_trace.hip diagnostics: 1 = no loads, 2 = no VALU work
    PLOAD = load_pack_x<PACK>(wb, lpk, cl_fill + 2);
#endif
    __builtin_amdgcn_sched_barrier(0);
#if defined(FSNAP_ACC_ABL) && (FSNAP_ACC_ABL & 2)
    (acc_row<NB, P>(V, vt, std::make_integer_sequence<int, NB - P>{}), ...);
    (void)wbv;
#else
    ((acc_row<NB, P>(V, vt, std::make_integer_sequence<int, NB - P>{}),
      acc_slot<NB, FULLK, NT, P>(V, RN, wv, wbv, wbp, K, e, cacc, RF, wb, cl_fill, va, vtl)),
     ...);
#endif
}

// LDS of kernel 1A: all 160 KiB of the CU (one workgroup per CU).  The epilogue folds the four row-waves' triangles
// through it in parts of at most 20 tiles (4 slots x 20 x 2 KiB); with PACK the same space first holds the per-row
// pairs of the workgroup's rows: 4 x (chunks per wave + FSNAP_ACC_PACK_PAD) x 64 bytes.
constexpr int FSNAP_ACC_LDS_DOUBLES = 20480;
constexpr int FSNAP_ACC_PACK_PAD = 12;     // chunk slots the unrolled loop may look past a wave's last chunk (<= ncl + 9)

}  // namespace

template <int NB, bool FULLK, bool NT, bool PACK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void
fsnap_syrk_acc(const double* __restrict__ A, int64_t lda, const double* __restrict__ wpack, int64_t m, int K,
               int64_t chunks_per_wave, double* __restrict__ part, double* __restrict__ cpart,
               const double* __restrict__ bvec, const double* __restrict__ wvec, const unsigned char* __restrict__ mask,
               double* __restrict__ spart) {

    constexpr int NTILE = NB * (NB + 1) / 2;
    constexpr int NV = NTILE > 32 ? NTILE - 32 : 1;      // tiles beyond the 32 that fill a[0:255] live in VGPRs
    constexpr int NPART = (NTILE * 1024 + FSNAP_ACC_LDS_DOUBLES - 1) / FSNAP_ACC_LDS_DOUBLES;
    constexpr int PER = (NTILE + NPART - 1) / NPART;     // tiles per part of the epilogue fold (NB = 8: 18, NB = 9: 15)
    static_assert(4 * PER * 256 <= FSNAP_ACC_LDS_DOUBLES, "fold part does not fit the LDS");
    __shared__ __attribute__((aligned(16))) double lds[FSNAP_ACC_LDS_DOUBLES];
#ifdef FSNAP_TRACE
    const unsigned long long trace_t0 = wall_clock64();
    const unsigned long long trace_c0 = __builtin_readcyclecounter();
#endif
    const int lane = threadIdx.x & 63, e = lane & 15, kr = lane >> 4;
    const int rw = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t rowwave = (int64_t)blockIdx.x * 4 + rw;
    const int64_t nchunks = (m + 3) >> 2;
    int64_t c0 = rowwave * chunks_per_wave;
    int64_t c1 = c0 + chunks_per_wave;
    if (c1 > nchunks) c1 = nchunks;
    if (c0 > c1) c0 = c1;
    const int64_t row0 = c0 << 2;
    int64_t row1 = c1 << 2;
    if (row1 > m) row1 = m;
    const int64_t nrow = row1 > row0 ? row1 - row0 : 0;
    WaveBufsP wb;
    wb.A = make_rsrc(A + row0 * lda, (unsigned)(nrow * lda * 8 + (nrow ? 16 : 0)));
    wb.wp = make_rsrc(PACK ? nullptr : wpack + 2 * row0, PACK ? 0u : (unsigned)(nrow * 16));
    wb.voffA = (unsigned)((kr * lda + 2 * e) * 8);
    wb.voffT = (unsigned)((kr * lda + 16 * (NB - 1) + e) * 8);
    wb.voffP = (unsigned)(kr * 16);
    wb.chunk_bytes = (unsigned)(lda * 32);
    const unsigned ncl = (unsigned)(c1 - c0);

    const double* lpk = nullptr;
    if constexpr (PACK) {
        const unsigned wave_rows = (unsigned)chunks_per_wave * 4u;
        const unsigned region_rows = wave_rows + FSNAP_ACC_PACK_PAD * 4u;
        double* lpw = lds + (size_t)rw * region_rows * 2;
        pack_rows_to_lds(bvec, wvec, mask, row0, nrow, wave_rows, region_rows, lpw, lane, spart + rowwave * 4);
        lpk = lpw + 2 * kr;
    }

    // the compiler must count a[0:255] as used (register allocation granule of the kernel descriptor): the
    // clobber makes its resource analysis see the highest accumulation register
    asm volatile("" : : : "a0", "a255");
    acc_zero_all(std::make_integer_sequence<int, 256>{});
    d4 vt[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) vt[u] = d4{0.0, 0.0, 0.0, 0.0};
    double cacc[NB], V[NB];
#pragma unroll
    for (int p = 0; p < NB; ++p) cacc[p] = 0.0;
    // FSNAP_ACC_DEPTH raw sets in flight (3)
    RawM<NB> r0, r1, r2;
    constexpr auto rows = std::make_integer_sequence<int, NB>{};
    if (ncl > 0) {
        r0.wp = load_pack_x<PACK>(wb, lpk, 0);
        r1.wp = load_pack_x<PACK>(wb, lpk, 1);
        r2.wp = load_pack_x<PACK>(wb, lpk, 2);
        issue_rows<NB, NT>(r0, wb, 0);
        issue_rows<NB, NT>(r1, wb, 1);
        issue_rows<NB, NT>(r2, wb, 2);
        {   // chunk 0 -> V
            const d2 wp0 = __builtin_bit_cast(d2, r0.wp);
            const double wv = wp0[0], wbv = wp0[1];
#pragma unroll
            for (int p = 0; p < NB; ++p) {
                V[p] = weighted_block<NB, FULLK>(r0, p, wv, K, e);
                cacc[p] = __builtin_fma(V[p], wbv, cacc[p]);
            }
        }
        // ring of packed weights: slot (c mod 6) holds the pair of chunk c; steps use c + 1 (weights), c + 3 (row
        // mask of the refill) and load c + 5
        u4 pk0 = r0.wp, pk1 = r1.wp, pk2 = r2.wp, pk3 = load_pack_x<PACK>(wb, lpk, 3), pk4 = load_pack_x<PACK>(wb, lpk, 4),
           pk5 = {0u, 0u, 0u, 0u};
        (void)pk0;
        // step cl: MFMAs of chunk cl (in V), V <- chunk cl+1, rows of chunk cl+3 into the raw set freed one step ago,
        // packed weights of chunk cl+5.  Chunk slots past the wave's range read zeros (bounds-checked descriptors).
        double wbp = 0.0;   // chunk 0 is fully accounted for by the prologue
        // The ring of pairs has period six, the raw sets period three: the loop runs six steps per trip while more than three
        // chunks remain, and a last half trip of three steps for 1 ... 3 leftover chunks (a wave of a 125 000-row shard owns 31
        // chunks: 33 steps instead of 36; the steps past the range multiply zeros).
        unsigned cl = 0;
        for (; cl + 3 < ncl; cl += 6) {
            acc_step<NB, FULLK, NT, PACK>(V, vt, r0, r1, wb, lpk, cl + 3, K, e, cacc, wbp, pk1, pk3, pk5, rows);
            acc_step<NB, FULLK, NT, PACK>(V, vt, r1, r2, wb, lpk, cl + 4, K, e, cacc, wbp, pk2, pk4, pk0, rows);
            acc_step<NB, FULLK, NT, PACK>(V, vt, r2, r0, wb, lpk, cl + 5, K, e, cacc, wbp, pk3, pk5, pk1, rows);
            acc_step<NB, FULLK, NT, PACK>(V, vt, r0, r1, wb, lpk, cl + 6, K, e, cacc, wbp, pk4, pk0, pk2, rows);
            acc_step<NB, FULLK, NT, PACK>(V, vt, r1, r2, wb, lpk, cl + 7, K, e, cacc, wbp, pk5, pk1, pk3, rows);
            acc_step<NB, FULLK, NT, PACK>(V, vt, r2, r0, wb, lpk, cl + 8, K, e, cacc, wbp, pk0, pk2, pk4, rows);
        }
        if (cl < ncl) {
            acc_step<NB, FULLK, NT, PACK>(V, vt, r0, r1, wb, lpk, cl + 3, K, e, cacc, wbp, pk1, pk3, pk5, rows);
            acc_step<NB, FULLK, NT, PACK>(V, vt, r1, r2, wb, lpk, cl + 4, K, e, cacc, wbp, pk2, pk4, pk0, rows);
            acc_step<NB, FULLK, NT, PACK>(V, vt, r2, r0, wb, lpk, cl + 5, K, e, cacc, wbp, pk3, pk5, pk1, rows);
        }
        cacc[NB - 1] = __builtin_fma(V[NB - 1], wbp, cacc[NB - 1]);   // last prepared chunk (zeros past the range)
    }
    // the last MFMAs (16 passes) must have left the pipe before their accumulators are read
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(vt[0]));
#pragma unroll
    for (int u = 1; u < NV; ++u) asm volatile("" : "+v"(vt[u]));
#ifdef FSNAP_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 4096) {   // loop only (the epilogue is timed by the kernel duration)
        const int64_t wg = blockIdx.x;
        fsnap_trace_buf[wg * 8 + 0] = trace_t0;
        fsnap_trace_buf[wg * 8 + 1] = wall_clock64();
        fsnap_trace_buf[wg * 8 + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        fsnap_trace_buf[wg * 8 + 3] = __builtin_amdgcn_s_getreg((3 << 11) | 20);
        fsnap_trace_buf[wg * 8 + 4] = __builtin_readcyclecounter() - trace_c0;
    }
#endif

    // epilogue: fold the four row-waves through LDS in NPART parts of the triangle.  Every wave parks its tiles of
    // the part (4 slots x PER tiles x 2 KiB = 144 KiB at K = 128), then wave r sums a quarter of the tiles over the
    // four slots in a fixed order and stores them: one partial triangle per workgroup, all four waves busy.
    if constexpr (PACK) __syncthreads();       // the other waves may still be reading their pairs from this space
    double* pw = part + (int64_t)blockIdx.x * (int64_t)(NTILE * 256);
    auto fold_part = [&](auto part_tag) {
        constexpr int H = decltype(part_tag)::value;
        constexpr int TB = H * PER;
        constexpr int NT_H = (TB + PER <= NTILE) ? PER : (NTILE - TB);
        constexpr int Q = (NT_H + 3) / 4;
        double* slot = lds + (size_t)rw * PER * 256;
        {
            d4 tmp[NT_H];
            acc_read_range<TB>(tmp, vt, std::make_integer_sequence<int, NT_H>{});
#pragma unroll
            for (int u = 0; u < NT_H; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) slot[(u * 4 + i) * 64 + lane] = tmp[u][i];
        }
        __syncthreads();
        for (int u = rw * Q; u < (rw + 1) * Q && u < NT_H; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int o = (u * 4 + i) * 64 + lane;
                const double s01 = lds[o] + lds[PER * 256 + o];
                pw[((TB + u) * 4 + i) * 64 + lane] = (s01 + lds[2 * PER * 256 + o]) + lds[3 * PER * 256 + o];
            }
        }
        if (H + 1 < NPART) __syncthreads();
    };
    wave_p_for<NPART>(fold_part);

    double* cw = cpart + rowwave * (int64_t)(NB * 16);
#pragma unroll
    for (int p = 0; p < NB; ++p) {
        double sm = xlane_sum_rows(cacc[p]);
        if (kr == 0) cw[p * 16 + e] = sm;
    }
#ifdef FSNAP_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 4096) fsnap_trace_buf[blockIdx.x * 8 + 5] = wall_clock64();   // after the epilogue
#endif
}

// ---------------------------------------------------------------------------------
// Kernel 1P: kernel 1 (one wave = whole triangle in compiler-allocated registers, K <= 80: at most 15 tiles) on the
// packed weights of kernel 1A: one 16-byte (w_eff, w_eff b) pair per row instead of mask, b and w, the row mask
// applied by the loads (out-of-range offset for rows with w_eff == 0), the b-only scalars from the packing kernel.
// These shapes are HBM-bound or co-bound (K = 64: the MFMAs of a 2 KiB chunk take as long as its bytes at ~8 TB/s),
// and every VALU instruction still costs matrix-pipe time: ~14 instead of ~35 per chunk at K = 64.  Several waves
// per SIMD (compiler-scheduled code, no hand placement): the waves cover each other's load latency.
// Partial layout = kernel 1 / 1A: part[workgroup][NT][4][64] | cpart[rowwave][NB][16].
// ---------------------------------------------------------------------------------
// PACK (round 4): like kernel 1A the wave forms the (w_eff, w_eff b) pairs of its rows in LDS in a prologue
// (pack_rows_to_lds) instead of reading what fsnap_pack_weights_k wrote to HBM -- no packing launch in front of a fit of the
// Ta width (15 213 x 31: 35.6 -> 30 us per fit).  The LDS is dynamic: max(fold buffer, 4 x (chunks per wave + pad) x 64 B of
// pairs); the pairs go first, the fold reuses the space behind a barrier.  Contiguous chunk ranges only (no `interleave`).
constexpr int FSNAP_WAVE_P_PACK_PAD = 16;      // chunk slots the loop may look past a wave's last chunk (<= ncl + 2 W - 3, W <= 8)

template <int NB, bool FULLK, bool NT, bool PACK>
__global__ __launch_bounds__(256, 2) void fsnap_syrk_wave_p(const double* __restrict__ A, int64_t lda,
                                                            const double* __restrict__ wpack, int64_t m, int K,
                                                            int64_t chunks_per_wave,
                                                            double* __restrict__ part, double* __restrict__ cpart,
                                                            const double* __restrict__ bvec, const double* __restrict__ wvec,
                                                            const unsigned char* __restrict__ mask,
                                                            double* __restrict__ spart) {
    constexpr int NTILE = NB * (NB + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) double lds[];      // >= 2 * NTILE * 256 doubles (launcher)
    const int lane = threadIdx.x & 63, e = lane & 15, kr = lane >> 4;
    const int rw = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t rowwave = (int64_t)blockIdx.x * 4 + rw;
    const int64_t nchunks = (m + 3) >> 2;
    // a row-wave takes a contiguous range of chunks_per_wave chunks (one advancing front over the whole matrix -- chunk
    // rowwave, rowwave + NW, ... -- streamed at the same rate in round 3 and is gone)
    int64_t c0 = rowwave * chunks_per_wave, ncl64;
    const int64_t stride = 1;
    {
        int64_t c1 = c0 + chunks_per_wave;
        if (c1 > nchunks) c1 = nchunks;
        if (c0 > c1) c0 = c1;
        ncl64 = c1 - c0;
    }
    const int64_t row0 = c0 << 2;
    // the descriptors end where the wave's last chunk ends: chunk slots past the wave's range read zeros
    int64_t row1 = (c0 + ncl64) << 2;
    if (row1 > m) row1 = m;
    const int64_t nrow = (ncl64 > 0 && row1 > row0) ? row1 - row0 : 0;
    WaveBufsP wb;
    wb.A = make_rsrc(A + row0 * lda, (unsigned)(nrow * lda * 8 + (nrow ? 16 : 0)));
    wb.wp = make_rsrc(PACK ? nullptr : wpack + 2 * row0, PACK ? 0u : (unsigned)(nrow * 16));
    wb.voffA = (unsigned)((kr * lda + 2 * e) * 8);
    wb.voffT = (unsigned)((kr * lda + 16 * (NB - 1) + e) * 8);
    wb.voffP = (unsigned)(kr * 16);
    wb.chunk_bytes = (unsigned)(stride * lda * 32);
    const unsigned pack_bytes = (unsigned)(stride * 64);
    const unsigned ncl = (unsigned)ncl64;
    const double* lpk = nullptr;
    if constexpr (PACK) {
        const unsigned wave_rows = (unsigned)chunks_per_wave * 4u;
        const unsigned region_rows = wave_rows + FSNAP_WAVE_P_PACK_PAD * 4u;
        double* lpw = lds + (size_t)rw * region_rows * 2;
        pack_rows_to_lds(bvec, wvec, mask, row0, nrow, wave_rows, region_rows, lpw, lane, spart + rowwave * 4);
        lpk = lpw + 2 * kr;
    }
    auto load_pack_s = [&](unsigned cl) -> u4 {
        if constexpr (PACK) return __builtin_bit_cast(u4, *reinterpret_cast<const d2*>(lpk + (size_t)cl * 8));
        else return __builtin_amdgcn_raw_buffer_load_b128(wb.wp, wb.voffP, cl * pack_bytes, 0);
    };

    d4 acc[NTILE];
#pragma unroll
    for (int t = 0; t < NTILE; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
    double cacc[NB], V[NB];
#pragma unroll
    for (int p = 0; p < NB; ++p) {
        cacc[p] = 0.0;
        V[p] = 0.0;
    }
    double wbcur = 0.0;
    // Software pipeline over rings of registers with compile-time slots.  Step c: the packed pair of chunk c + 2 D - 1, the
    // rows of chunk c + D (gated by ITS packed pair, requested D - 1 steps ago), the MFMAs of chunk c (operands V prepared
    // one step ago), V <- chunk c + 1.  Rows are consumed D - 1 steps after their request.  With NB <= 2 a step is three
    // MFMAs (~200 cycles): the first version of this loop (three row sets, the packed pair one step ahead of the rows it
    // gates) waited for a memory round trip per step -- the ISA showed the row loads sunk behind the wait for their pair
    // and consumed at the top of the next iteration.  Chunk slots past the wave's range read zeros (bounds-checked
    // descriptors).
    constexpr int D = NB <= 2 ? 4 : 3, W = 2 * D;
    RawM<NB> R[D];
    u4 P[W];
    if (ncl > 0) {
        wave_p_for<W - 1>([&](auto x) { P[x] = load_pack_s((unsigned)x); });
        wave_p_for<D>([&](auto x) { issue_rows_w<NB, NT>(R[x], P[x], wb, (unsigned)x); });
        {
            const d2 wp0 = __builtin_bit_cast(d2, P[0]);
#pragma unroll
            for (int p = 0; p < NB; ++p) V[p] = weighted_block<NB, FULLK>(R[0], p, wp0[0], K, e);
            wbcur = wp0[1];
        }
        for (unsigned cl = 0; cl < ncl; cl += W) {
            wave_p_for<W>([&](auto u_) {
                constexpr int u = decltype(u_)::value;                  // chunk c = cl + u: c % W == u
                const d2 wpn = __builtin_bit_cast(d2, P[(u + 1) % W]);  // weights of chunk c + 1
                // the pair of chunk c - 1 + W goes into the slot of chunk c - 1: the slot of chunk c is still alive in
                // this step (wbcur is its upper half -- loading over it makes the allocator keep two copies of the ring
                // and rotate them with moves at the loop end, behind a wait for every load in flight)
                P[(u + W - 1) % W] = load_pack_s(cl + u + W - 1);
                issue_rows_w<NB, NT>(R[u % D], P[(u + D) % W], wb, cl + u + D);
                // the requests stay at the head of their step: left to itself the scheduler sinks them towards their
                // consumers' step and hoists the waits (the whole point is the distance between the two)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < NB; ++p) {
#pragma unroll
                    for (int q = p; q < NB; ++q) {
                        acc[tri_index(p, q, NB)] =
                            __builtin_amdgcn_mfma_f64_16x16x4f64(V[p], V[q], acc[tri_index(p, q, NB)], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int p = 0; p < NB; ++p) cacc[p] = __builtin_fma(V[p], wbcur, cacc[p]);
#pragma unroll
                for (int p = 0; p < NB; ++p) V[p] = weighted_block<NB, FULLK>(R[(u + 1) % D], p, wpn[0], K, e);
                wbcur = wpn[1];
                __builtin_amdgcn_sched_barrier(0);
            });
        }
    }

    // fold the four row-waves through LDS ({2,3} -> {0,1}, then 1 -> 0): one partial triangle per workgroup
    if constexpr (PACK) __syncthreads();       // the other waves may still be reading their pairs from this space
    {
        double* slot_hi = lds + (size_t)((rw & 1) * NTILE) * 256;
        if (rw >= 2) {
#pragma unroll
            for (int u = 0; u < NTILE; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) slot_hi[(u * 4 + i) * 64 + lane] = acc[u][i];
        }
        __syncthreads();
        if (rw < 2) {
#pragma unroll
            for (int u = 0; u < NTILE; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[u][i] += slot_hi[(u * 4 + i) * 64 + lane];
        }
        __syncthreads();
        if (rw == 1) {
#pragma unroll
            for (int u = 0; u < NTILE; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) lds[(u * 4 + i) * 64 + lane] = acc[u][i];
        }
        __syncthreads();
        if (rw == 0) {
            double* pw = part + (int64_t)blockIdx.x * (int64_t)(NTILE * 256);
#pragma unroll
            for (int u = 0; u < NTILE; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) pw[(u * 4 + i) * 64 + lane] = acc[u][i] + lds[(u * 4 + i) * 64 + lane];
        }
    }
    double* cw = cpart + rowwave * (int64_t)(NB * 16);
#pragma unroll
    for (int p = 0; p < NB; ++p) {
        double sm = xlane_sum_rows(cacc[p]);
        if (kr == 0) cw[p * 16 + e] = sm;
    }
}

// ---------------------------------------------------------------------------------
// Kernel 2b: the same reduction with every load of a thread in flight at once (default; option reduce = 1 keeps
// kernel 2 for A/B).  Kernel 2 gives a thread 8-byte loads and, for the 256 partials of a one-workgroup-per-CU launch,
// runs them as FOUR dependent round trips (its 8-deep loop needs >= 512 partials): 12.3 us for 19 MB that sit in
// L2 / Infinity Cache.  Here a workgroup of 256 threads owns 32 consecutive elements: thread (slice = tid >> 4,
// sub = tid & 15) takes elements 2 sub, 2 sub + 1 (one 16-byte load per partial) of partials slice, slice + 16, ...,
// sixteen loads in flight per batch (all of them when there are 256 partials); fixed summation order: per thread four
// interleaved accumulators, then the 16 slices in order through LDS.  Regions (triangle | c | scalars) start on
// workgroup boundaries.  upper_mirror: the host mirror receives every triangle element ONCE, at its upper-triangle
// position [min(r, c)][max(r, c)] (diagonal tiles in full) -- half the PCIe writes; the host solve reads the mirror as
// an upper triangle then (fsnap_solve_diag_upper).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fsnap_reduce_partials2(const double* __restrict__ part,
                                                              const double* __restrict__ cpart,
                                                              const double* __restrict__ spart, int nblocks,
                                                              int cs_per_block, int ns, int NB, int K,
                                                              double* __restrict__ out, double* __restrict__ mirror,
                                                              int accumulate, int upper_mirror) {
    __shared__ __attribute__((aligned(16))) double red[16][32];
    const int NTILE = NB * (NB + 1) / 2;
    const int nG = NTILE * 256, nC = NB * 16;
    const int gG = nG / 32, gC = (nC + 31) / 32;
    const int tid = threadIdx.x, sub = tid & 15, slice = tid >> 4;
    const int b = blockIdx.x;
    int region, e0, nel, np;
    const double* src;
    int64_t stride;
    if (b < gG) {
        region = 0; e0 = b * 32; nel = nG; np = nblocks; src = part; stride = nG;
    } else if (b < gG + gC) {
        region = 1; e0 = (b - gG) * 32; nel = nC; np = nblocks * cs_per_block; src = cpart; stride = nC;
    } else {
        region = 2; e0 = 0; nel = 4; np = ns >= 0 ? ns : nblocks * cs_per_block; src = spart; stride = 4;
    }
    const int el = e0 + 2 * sub;
    d2 a0 = {0.0, 0.0}, a1 = a0, a2 = a0, a3 = a0;
    if (el < nel) {
        const double* s0 = src + el;
        int p = slice;
        for (; p + 240 < np; p += 256) {
            d2 x[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) x[j] = *reinterpret_cast<const d2*>(s0 + (int64_t)(p + 16 * j) * stride);
#pragma unroll
            for (int j = 0; j < 16; j += 4) {
                a0 += x[j];
                a1 += x[j + 1];
                a2 += x[j + 2];
                a3 += x[j + 3];
            }
        }
        for (; p < np; p += 16) a0 += *reinterpret_cast<const d2*>(s0 + (int64_t)p * stride);
    }
    const d2 sv = (a0 + a1) + (a2 + a3);
    *reinterpret_cast<d2*>(&red[slice][2 * sub]) = sv;
    __syncthreads();
    if (tid >= 32) return;
    const int idx = e0 + tid;
    if (idx >= nel) return;
    double tot = red[0][tid];
#pragma unroll
    for (int k = 1; k < 16; ++k) tot += red[k][tid];
    if (region == 0) {
        const int t = idx >> 8, rem = idx & 255, i = rem >> 6, ln = rem & 63;
        int p = 0;
        while (p + 1 < NB && tri_index(p + 1, p + 1, NB) <= t) ++p;
        const int q = p + (t - tri_index(p, p, NB));
        const int ep = (ln >> 4) + 4 * i, eq = ln & 15;
        const int r = col_of(p, ep, NB), c = col_of(q, eq, NB);
        if (r < K && c < K) {
            const double val = accumulate ? out[(int64_t)r * K + c] + tot : tot;
            out[(int64_t)r * K + c] = val;
            if (p != q) out[(int64_t)c * K + r] = val;
            if (mirror) {
                if (upper_mirror) {
                    const int lo = r < c ? r : c, hi = r < c ? c : r;
                    if (p != q) mirror[(int64_t)lo * K + hi] = val;
                    else mirror[(int64_t)r * K + c] = val;
                } else {
                    mirror[(int64_t)r * K + c] = val;
                    if (p != q) mirror[(int64_t)c * K + r] = val;
                }
                if (r == c) mirror[(int64_t)K * K + K + 3 + r] = val;
            }
        }
    } else if (region == 1) {
        const int cidx = col_of(idx >> 4, idx & 15, NB);
        if (cidx < K) {
            const double val = accumulate ? out[(int64_t)K * K + cidx] + tot : tot;
            out[(int64_t)K * K + cidx] = val;
            if (mirror) mirror[(int64_t)K * K + cidx] = val;
        }
    } else if (idx < 3) {
        const double val = accumulate ? out[(int64_t)K * K + K + idx] + tot : tot;
        out[(int64_t)K * K + K + idx] = val;
        if (mirror) mirror[(int64_t)K * K + K + idx] = val;
    }
}

// ---------------------------------------------------------------------------------
// Kernel 1T: general-K fused mask x weight x SYRK (K > 128: ACE / quadratic SNAP widths).
// The column space is cut into superblocks of 64 columns (4 MFMA blocks, even/odd
// interleaved in pairs like kernel 1).  A workgroup owns ONE superblock pair (I <= J) over
// ONE row split; each of its 4 waves streams a quarter of the split's rows and keeps the
// 4 x 4 tiles of G[I-block, J-block] (16 tiles, 128 accumulator registers; 10 tiles on
// the diagonal I == J) in registers, so two workgroups share a CU.  Per 4-row chunk a
// lane loads 4 + 4 doubles and feeds 16 MFMAs.  Pairs are the fast grid index: the
// workgroups that run concurrently read the SAME rows (different column superblocks),
// so every row is fetched from HBM once per split and re-read from L2 / Infinity Cache.
// (A 2 x 2 arrangement -- the four waves of a workgroup on the four superblock pairs of a 128-column block pair
// over the same rows, every slice requested by two waves -- was measured 15-25 % slower: idle waves on diagonal and
// edge blocks, and no gain from the shared requests: the kernel is not bound by L2 traffic.)
// c rides on the diagonal pairs; the b-only scalars come from fsnap_pack_weights_k.
// Partials: partT[split*npairs + pair][16][4][64] | cpartT[(split*NSB + I)*4 + wave][4][16]
// ---------------------------------------------------------------------------------
namespace {

// Raw loads of the tiled kernel: like kernel 1A it reads ONE packed pair (w_eff, w_eff b) per row
// (fsnap_pack_weights_k) instead of mask, b and w, and applies the row mask through the loads (rows with
// w_eff == 0 get an out-of-range buffer offset: zeros come back, the row is never fetched).
struct RawT {
    u4 pi[2], pj[2];
    u4 wp;
};

struct WaveBufsT {
    __amdgpu_buffer_rsrc_t A, wp;
    unsigned voffP;        // kr * 16 (packed weights)
    unsigned chunk_bytes;  // 4*lda*8
};

__device__ __forceinline__ u4 load_pack_t(const WaveBufsT& wb, unsigned cl) {
    return __builtin_amdgcn_raw_buffer_load_b128(wb.wp, wb.voffP, cl * 64u, 0);
}
// w_eff only (off-diagonal items never use w_eff b): the unused half of a 16-byte load would be a dead register pair
// that the allocator reuses at once -- and every such reuse waits for the load (write-after-write, vmcnt(0))
__device__ __forceinline__ u2 load_pack_w(const WaveBufsT& wb, unsigned cl) {
    return __builtin_amdgcn_raw_buffer_load_b64(wb.wp, wb.voffP, cl * 64u, 0);
}
__device__ __forceinline__ bool pack_keep_w(const u2& wp) { return ((wp[0] | (wp[1] & 0x7FFFFFFFu)) != 0u); }
// kernel 1T: diagonal pairs need (w_eff, w_eff b), off-diagonal pairs w_eff only
template <bool DIAG>
__device__ __forceinline__ void load_pack_d(u4& dst, const WaveBufsT& wb, unsigned cl) {
    if (DIAG) {
        dst = load_pack_t(wb, cl);
    } else {
        const u2 t = load_pack_w(wb, cl);
        dst[0] = t[0];
        dst[1] = t[1];
    }
}

template <bool DIAG, int EDGE, bool NT>
__device__ __forceinline__ void issue_rows_t(RawT& r, const WaveBufsT& wb, unsigned voffI, unsigned voffJ, unsigned cl) {
    const unsigned soff = cl * wb.chunk_bytes;
    constexpr int AUX = NT ? 2 : 0;
    const bool keep = pack_keep(r.wp);
    const unsigned vi = keep ? voffI : FSNAP_OOB_VOFF;
    r.pi[0] = __builtin_amdgcn_raw_buffer_load_b128(wb.A, vi, soff, AUX);
    if (!(DIAG && EDGE == 2)) r.pi[1] = __builtin_amdgcn_raw_buffer_load_b128(wb.A, vi + 256u, soff, AUX);
    if (!DIAG) {
        const unsigned vj = keep ? voffJ : FSNAP_OOB_VOFF;
        r.pj[0] = __builtin_amdgcn_raw_buffer_load_b128(wb.A, vj, soff, AUX);
        if (EDGE != 2) r.pj[1] = __builtin_amdgcn_raw_buffer_load_b128(wb.A, vj + 256u, soff, AUX);
    }
}

// the same with the packed pair that gates the rows handed in (ring form of the pipeline)
template <bool DIAG, int EDGE, bool NT>
__device__ __forceinline__ void issue_rows_tw(RawT& r, const u4& wp, const WaveBufsT& wb, unsigned voffI, unsigned voffJ,
                                              unsigned cl) {
    const unsigned soff = cl * wb.chunk_bytes;
    constexpr int AUX = NT ? 2 : 0;
    const bool keep = pack_keep(wp);
    const unsigned vi = keep ? voffI : FSNAP_OOB_VOFF;
    r.pi[0] = __builtin_amdgcn_raw_buffer_load_b128(wb.A, vi, soff, AUX);
    if (!(DIAG && EDGE == 2)) r.pi[1] = __builtin_amdgcn_raw_buffer_load_b128(wb.A, vi + 256u, soff, AUX);
    if (!DIAG) {
        const unsigned vj = keep ? voffJ : FSNAP_OOB_VOFF;
        r.pj[0] = __builtin_amdgcn_raw_buffer_load_b128(wb.A, vj, soff, AUX);
        if (EDGE != 2) r.pj[1] = __builtin_amdgcn_raw_buffer_load_b128(wb.A, vj + 256u, soff, AUX);
    }
}

// MFMA operands of one chunk.  Diagonal pair (I == J): the same registers w a serve both sides.  Off-diagonal
// pair: the weight goes on ONE side, (w^2 a_I) x a_J -- the J side is used as loaded (no VALU instruction at all;
// fp64 MFMAs and VALU instructions serialise on the SIMD, the first version of this kernel spent ~40 VALU
// instructions per 16 MFMAs on per-value weighting and masking and stopped at 57-68 % of the matrix peak).
// Columns >= K exist only in the last superblock (EDGE): they are zeroed by selects there and nowhere else.
template <bool DIAG, int EDGE>
__device__ __forceinline__ void prep_t(double (&vI)[4], double (&vJ)[4], const RawT& r, double wv, int colI0, int colJ0,
                                       int K, int e) {
    // EDGE: 0 = every column of both superblocks is < K; 4 = the J superblock (= the I superblock on a diagonal
    // pair) is the last one and columns >= K are zeroed by selects; 2 = as 4, and its second 32-column group holds
    // no column < K at all: blocks 2, 3 are neither loaded nor multiplied
    const double f = DIAG ? wv : wv * wv;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (!(DIAG && EDGE == 2 && j == 1)) {
            const d2 x = __builtin_bit_cast(d2, r.pi[j]);
            vI[2 * j] = f * x[0];
            vI[2 * j + 1] = f * x[1];
            if (DIAG && EDGE) {
                if (!(colI0 + 32 * j + 2 * e < K)) vI[2 * j] = 0.0;
                if (!(colI0 + 32 * j + 2 * e + 1 < K)) vI[2 * j + 1] = 0.0;
            }
        }
        if (!DIAG && !(EDGE == 2 && j == 1)) {
            const d2 y = __builtin_bit_cast(d2, r.pj[j]);
            vJ[2 * j] = y[0];
            vJ[2 * j + 1] = y[1];
            if (EDGE) {
                if (!(colJ0 + 32 * j + 2 * e < K)) vJ[2 * j] = 0.0;
                if (!(colJ0 + 32 * j + 2 * e + 1 < K)) vJ[2 * j + 1] = 0.0;
            }
        }
    }
}

template <bool DIAG, int EDGE, bool NT, bool RING = false>
__device__ __forceinline__ void syrk_tiled_body(const double* __restrict__ A, int64_t lda,
                                                const double* __restrict__ wpack, int64_t m, int K, int I, int J,
                                                int64_t c0, int64_t c1, int wv_in_wg, double* lds,
                                                double* __restrict__ pw, double* __restrict__ cw) {
    constexpr int NTW = 16;
    const int lane = threadIdx.x & 63, e = lane & 15, kr = lane >> 4;
    d4 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
    double cacc[4] = {0.0, 0.0, 0.0, 0.0};

    const int64_t row0 = c0 << 2;
    int64_t row1 = c1 << 2;
    if (row1 > m) row1 = m;
    const int64_t nrow = row1 > row0 ? row1 - row0 : 0;
    WaveBufsT wb;
    wb.A = make_rsrc(A + row0 * lda, (unsigned)(nrow * lda * 8 + (nrow ? 16 : 0)));
    wb.wp = make_rsrc(wpack + 2 * row0, (unsigned)(nrow * 16));
    wb.voffP = (unsigned)(kr * 16);
    wb.chunk_bytes = (unsigned)(lda * 32);
    const unsigned voffI = (unsigned)((kr * lda + 64 * I + 2 * e) * 8);
    const unsigned voffJ = (unsigned)((kr * lda + 64 * J + 2 * e) * 8);
    const unsigned ncl = (unsigned)(c1 > c0 ? c1 - c0 : 0);

    RawT r0, r1, r2;
    double vI[4] = {0.0, 0.0, 0.0, 0.0}, vJ[4] = {0.0, 0.0, 0.0, 0.0};
    double wbcur = 0.0;
    constexpr int QMAX = (EDGE == 2) ? 2 : 4;              // 16-column blocks of the J side that hold columns < K
    constexpr int PMAX = (DIAG && EDGE == 2) ? 2 : 4;

    if constexpr (RING) {
        // Kernel 1P's pipeline (rings with compile-time slots: three row sets, six packed pairs, the pair of chunk c + 5
        // loaded into the slot of chunk c - 1).  The three-set form below keeps a pair in its row set and reloads that
        // slot while the old pair is still needed (its weight at the end of the step, its w b as wbcur in the next one):
        // the allocator then holds two copies of the pairs and rotates them with moves at the loop end -- behind a wait
        // for nearly every load in flight (diagonal items: s_waitcnt vmcnt(1) once per three chunks).
        constexpr int D = 3, W = 6;
        RawT R[D];
        u4 P[W];
        if (ncl > 0) {
            wave_p_for<W - 1>([&](auto x) { load_pack_d<DIAG>(P[x], wb, (unsigned)x); });
            wave_p_for<D>([&](auto x) { issue_rows_tw<DIAG, EDGE, NT>(R[x], P[x], wb, voffI, voffJ, (unsigned)x); });
            {
                const d2 wp0 = __builtin_bit_cast(d2, P[0]);
                prep_t<DIAG, EDGE>(vI, vJ, R[0], wp0[0], 64 * I, 64 * J, K, e);
                wbcur = wp0[1];
            }
            for (unsigned cl = 0; cl < ncl; cl += W) {
                wave_p_for<W>([&](auto u_) {
                    constexpr int u = decltype(u_)::value;
                    const d2 wpn = __builtin_bit_cast(d2, P[(u + 1) % W]);
                    load_pack_d<DIAG>(P[(u + W - 1) % W], wb, cl + u + W - 1);
                    issue_rows_tw<DIAG, EDGE, NT>(R[u % D], P[(u + D) % W], wb, voffI, voffJ, cl + u + D);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int p = 0; p < PMAX; ++p) {
#pragma unroll
                        for (int q = (DIAG ? p : 0); q < QMAX; ++q)
                            acc[p * 4 + q] = __builtin_amdgcn_mfma_f64_16x16x4f64(vI[p], DIAG ? vI[q] : vJ[q], acc[p * 4 + q], 0, 0, 0);
                    }
                    if (DIAG) {
#pragma unroll
                        for (int p = 0; p < PMAX; ++p) cacc[p] = __builtin_fma(vI[p], wbcur, cacc[p]);
                    }
                    prep_t<DIAG, EDGE>(vI, vJ, R[(u + 1) % D], wpn[0], 64 * I, 64 * J, K, e);
                    wbcur = wpn[1];
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
        }
    } else {
    // step c: MFMAs of chunk c (operands prepared one step ago), operands of chunk c + 1 from raw set RN, rows of
    // chunk c + 3 into the raw set RF consumed one step ago (its packed weights arrived during the last step),
    // packed weights of chunk c + 4 into RN.  The packed load goes out before the row loads (in-order vmcnt).
#define FSNAP_STEP_T(RF, RN, CLF)                                                               \
    {                                                                                           \
        const d2 wpn = __builtin_bit_cast(d2, RN.wp);                                           \
        load_pack_d<DIAG>(RN.wp, wb, (CLF) + 1);                                                \
        issue_rows_t<DIAG, EDGE, NT>(RF, wb, voffI, voffJ, (CLF));                                    \
        _Pragma("unroll") for (int p = 0; p < PMAX; ++p) {                                      \
            _Pragma("unroll") for (int q = (DIAG ? p : 0); q < QMAX; ++q) {                     \
                acc[p * 4 + q] = __builtin_amdgcn_mfma_f64_16x16x4f64(vI[p], DIAG ? vI[q] : vJ[q], acc[p * 4 + q], 0, 0, 0); \
            }                                                                                   \
        }                                                                                       \
        if (DIAG) {                                                                             \
            _Pragma("unroll") for (int p = 0; p < PMAX; ++p) cacc[p] = __builtin_fma(vI[p], wbcur, cacc[p]); \
        }                                                                                       \
        prep_t<DIAG, EDGE>(vI, vJ, RN, wpn[0], 64 * I, 64 * J, K, e);                           \
        wbcur = wpn[1];                                                                         \
    }
    if (ncl > 0) {
        load_pack_d<DIAG>(r0.wp, wb, 0);
        load_pack_d<DIAG>(r1.wp, wb, 1);
        load_pack_d<DIAG>(r2.wp, wb, 2);
        issue_rows_t<DIAG, EDGE, NT>(r0, wb, voffI, voffJ, 0);
        issue_rows_t<DIAG, EDGE, NT>(r1, wb, voffI, voffJ, 1);
        issue_rows_t<DIAG, EDGE, NT>(r2, wb, voffI, voffJ, 2);
        {
            const d2 wp0 = __builtin_bit_cast(d2, r0.wp);
            prep_t<DIAG, EDGE>(vI, vJ, r0, wp0[0], 64 * I, 64 * J, K, e);
            wbcur = wp0[1];
        }
        load_pack_d<DIAG>(r0.wp, wb, 3);
        for (unsigned cl = 0; cl < ncl; cl += 3) {
            FSNAP_STEP_T(r0, r1, cl + 3)
            FSNAP_STEP_T(r1, r2, cl + 4)
            FSNAP_STEP_T(r2, r0, cl + 5)
        }
    }
#undef FSNAP_STEP_T
    }

    // fold the 4 waves through LDS ({2,3} -> {0,1}, 1 -> 0), then one partial per workgroup
    {
        const int rw = wv_in_wg;
        double* slot_hi = lds + (size_t)((rw & 1) * NTW) * 256;
        if (rw >= 2) {
#pragma unroll
            for (int u = 0; u < NTW; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) slot_hi[(u * 4 + i) * 64 + lane] = acc[u][i];
        }
        __syncthreads();
        if (rw < 2) {
#pragma unroll
            for (int u = 0; u < NTW; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[u][i] += slot_hi[(u * 4 + i) * 64 + lane];
        }
        __syncthreads();
        if (rw == 1) {
#pragma unroll
            for (int u = 0; u < NTW; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) lds[(u * 4 + i) * 64 + lane] = acc[u][i];
        }
        __syncthreads();
        if (rw == 0) {
#pragma unroll
            for (int u = 0; u < NTW; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) pw[(u * 4 + i) * 64 + lane] = acc[u][i] + lds[(u * 4 + i) * 64 + lane];
        }
    }
    if (DIAG) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            double s = xlane_sum_rows(cacc[p]);
            if (kr == 0) cw[p * 16 + e] = s;
        }
    }
}

}  // namespace

template <bool NT>
__global__ __launch_bounds__(256, 2) void fsnap_syrk_tiled(const double* __restrict__ A, int64_t lda,
                                                           const double* __restrict__ wpack, int64_t m, int K,
                                                           int NSB, int npairs, int64_t chunks_per_split,
                                                           int nitems,
                                                           double* __restrict__ part, double* __restrict__ cpart) {
    __shared__ double lds[2 * 16 * 256];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // Work item = (split, pair), pair fastest.  Workgroups are dealt round-robin to the 8 XCDs
    // (blockIdx % 8); every XCD gets a CONTIGUOUS range of items, so the
    // workgroups that share an L2 sweep the same rows (different column pairs) together and a
    // row slab is fetched from HBM once per XCD instead of once per pair.
    unsigned item = blockIdx.x;
    int I, J, split;
    {
        const unsigned per = ((unsigned)nitems + 7u) >> 3;
        const unsigned slot = blockIdx.x >> 3;
        item = (blockIdx.x & 7u) * per + slot;
        if (slot >= per || item >= (unsigned)nitems) return;
    }
    const int id = (int)(item % (unsigned)npairs);
    split = (int)(item / (unsigned)npairs);
    // Items of a split: first the off-diagonal pairs (row-major strict upper triangle, 16 tiles each), then the
    // diagonal pairs (10 tiles): workgroups that run side by side then cost the same and sweep the split's rows in
    // step (a short item in their midst finishes early and its successor starts over at the first row, out of phase
    // with the rows its neighbours keep in L2).
    const int noff = npairs - NSB;
    if (id < noff) {
        I = 0;
        int rem = id;
        while (rem >= NSB - 1 - I) {
            rem -= NSB - 1 - I;
            ++I;
        }
        J = I + 1 + rem;
    } else {
        I = J = id - noff;
    }
    const int pair = I * NSB - (I * (I - 1)) / 2 + (J - I);      // slot in the packed upper triangle (partials)
    const int64_t nchunks = (m + 3) >> 2;
    const int64_t cpw = (chunks_per_split + 3) >> 2;
    int64_t s0 = (int64_t)split * chunks_per_split;
    int64_t s1 = s0 + chunks_per_split;
    if (s1 > nchunks) s1 = nchunks;
    int64_t c0 = s0 + (int64_t)wv * cpw;
    int64_t c1 = c0 + cpw;
    if (c1 > s1) c1 = s1;
    if (c0 > s1) c0 = s1;
    double* pw = part + ((int64_t)split * npairs + pair) * (16 * 256);
    double* cw = cpart + (((int64_t)split * NSB + I) * 4 + wv) * 64;
    // columns >= K live in the last superblock only: 4 = all of its 16-column blocks hold columns < K,
    // 2 = only the first 32-column group does (blocks 2, 3 are empty: their tiles are skipped), 0 = no edge
    const int tail = K & 63;
    const int edge = (J == NSB - 1 && tail != 0) ? (tail <= 32 ? 2 : 4) : 0;
    // (both kinds of items run the ring form of the load pipeline, see syrk_tiled_body)
    if (I == J) {
        if (edge == 2) syrk_tiled_body<true, 2, NT, true>(A, lda, wpack, m, K, I, J, c0, c1, wv, lds, pw, cw);
        else if (edge == 4) syrk_tiled_body<true, 4, NT, true>(A, lda, wpack, m, K, I, J, c0, c1, wv, lds, pw, cw);
        else syrk_tiled_body<true, 0, NT, true>(A, lda, wpack, m, K, I, J, c0, c1, wv, lds, pw, cw);
    } else {
        if (edge == 2) syrk_tiled_body<false, 2, NT, true>(A, lda, wpack, m, K, I, J, c0, c1, wv, lds, pw, cw);
        else if (edge == 4) syrk_tiled_body<false, 4, NT, true>(A, lda, wpack, m, K, I, J, c0, c1, wv, lds, pw, cw);
        else syrk_tiled_body<false, 0, NT, true>(A, lda, wpack, m, K, I, J, c0, c1, wv, lds, pw, cw);
    }
}

// G element idx of the tiled partial layout (pair, tile, register, lane) -> its place(s) in the packed buffer
__device__ __forceinline__ void reduce_tiled_store_g(int64_t idx, double tot, int NSB, int K, double* __restrict__ out,
                                                     int accumulate) {
    const int pair = (int)(idx >> 12), rem = (int)(idx & 4095);
    const int t = rem >> 8, i = (rem >> 6) & 3, ln = rem & 63;
    int I = 0, pr = pair;
    while (pr >= NSB - I) {
        pr -= NSB - I;
        ++I;
    }
    const int J = I + pr;
    const int p = t >> 2, q = t & 3;
    if (I == J && q < p) return;  // unused slots of a diagonal pair
    const int ep = (ln >> 4) + 4 * i, eq = ln & 15;
    const int r = 64 * I + 32 * (p >> 1) + 2 * ep + (p & 1);
    const int c = 64 * J + 32 * (q >> 1) + 2 * eq + (q & 1);
    if (r < K && c < K) {
        const double val = accumulate ? out[(int64_t)r * K + c] + tot : tot;
        out[(int64_t)r * K + c] = val;
        if (!(I == J && p == q)) out[(int64_t)c * K + r] = val;
    }
}


// Few row splits (<= 16, the usual case for wide matrices): one thread per G element sums its partials in order --
// the same order and therefore the same bits as the sliced kernel below, which for <= 16 partials keeps 15 of its 16
// slices idle (K = 1595, 3 splits: 55 us for 52 MB).  c and the scalars stay with the sliced kernel (idx0 = nG).
__global__ __launch_bounds__(256) void fsnap_reduce_tiled_small(const double* __restrict__ part, int nsplit, int NSB,
                                                                int npairs, int K, double* __restrict__ out,
                                                                int accumulate) {
    const int64_t nG = (int64_t)npairs * 4096;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nG) return;
    double tot = 0.0;
    for (int p = 0; p < nsplit; ++p) tot += part[(int64_t)p * nG + idx];
    reduce_tiled_store_g(idx, tot, NSB, K, out, accumulate);
}

// Reduction of the tiled partials into the packed buffer (same output as kernel 2).
// Element space: npairs*16*256 G elements (nsplit partials each), NSB*64 c elements and
// 4 scalars (nsplit*4 partials each).  1024 threads = 64 elements x 16 partial slices.
__global__ __launch_bounds__(1024) void fsnap_reduce_tiled(const double* __restrict__ part,
                                                           const double* __restrict__ cpart,
                                                           const double* __restrict__ spart, int ns, int nsplit, int NSB,
                                                           int npairs, int K, double* __restrict__ out, int accumulate,
                                                           int64_t idx0) {
    __shared__ double red[1024];
    const int64_t nG = (int64_t)npairs * 4096;
    const int nC = NSB * 64, nS = 4;
    const int tid = threadIdx.x, g = tid >> 6, l = tid & 63;
    const int64_t idx = idx0 + (int64_t)blockIdx.x * 64 + l;      // idx0: first element of this launch
    const double* src = nullptr;
    int64_t stride = 0;
    int np = 0;
    if (idx < nG) {
        src = part + idx;
        stride = nG;
        np = nsplit;
    } else if (idx < nG + nC) {
        // c element j of superblock I: partial index ((split*NSB + I)*4 + wave)
        const int j = (int)(idx - nG);
        src = cpart + (int64_t)(j >> 6) * 256 + (j & 63);
        stride = 0;  // handled below (two-level layout)
        np = nsplit * 4;
    } else if (idx < nG + nC + nS) {
        src = spart + (idx - nG - nC);      // per-workgroup partials of fsnap_pack_weights_k
        stride = 4;
        np = ns;
    }
    double s = 0.0;
    if (src) {
        if (idx >= nG && idx < nG + nC) {
            // partial q = split*4 + wave lives at cpart[((split*NSB + I)*4 + wave)*64 + e]
            for (int q = g; q < np; q += 16) {
                const int split = q >> 2, wave = q & 3;
                s += src[((int64_t)split * NSB * 4 + wave) * 64];
            }
        } else {
            int p = g;
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            for (; p + 48 < np; p += 64) {
                const double x0 = src[(int64_t)p * stride], x1 = src[(int64_t)(p + 16) * stride];
                const double x2 = src[(int64_t)(p + 32) * stride], x3 = src[(int64_t)(p + 48) * stride];
                a0 += x0; a1 += x1; a2 += x2; a3 += x3;
            }
            for (; p < np; p += 16) a0 += src[(int64_t)p * stride];
            s = (a0 + a1) + (a2 + a3);
        }
    }
    red[tid] = s;
    __syncthreads();
    if (g == 0 && src) {
        double tot = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) tot += red[k * 64 + l];
        if (idx < nG) {
            reduce_tiled_store_g(idx, tot, NSB, K, out, accumulate);
        } else if (idx < nG + nC) {
            const int j = (int)(idx - nG);
            const int Ib = j >> 6, bq = (j >> 4) & 3, e = j & 15;
            const int cidx = 64 * Ib + 32 * (bq >> 1) + 2 * e + (bq & 1);
            if (cidx < K) out[(int64_t)K * K + cidx] = accumulate ? out[(int64_t)K * K + cidx] + tot : tot;
        } else {
            const int j = (int)(idx - nG - nC);
            if (j < 3) out[(int64_t)K * K + K + j] = accumulate ? out[(int64_t)K * K + K + j] + tot : tot;
        }
    }
}

// ---------------------------------------------------------------------------------
// host-side launchers (C++ linkage, used by fsnap_capi.cpp)
// ---------------------------------------------------------------------------------
namespace fsnap {

int syrk_num_blocks(int K) { return (K + 15) / 16; }

// waves per SIMD the register budget of the (NB, SPLIT) instantiation admits
int syrk_waves_per_simd(int K, int split) {
    const int NB = syrk_num_blocks(K);
    const int ntw = (NB * (NB + 1) / 2 + split - 1) / split;
    const int regs = 8 * ntw + 12 * NB + 40;  // accumulators + 3 raw chunks + weighted chunk + misc
    int wps = 512 / regs;
    if (wps < 1) wps = 1;
    if (wps > 8) wps = 8;
    return wps;
}

template <int NB>
static hipError_t launch_syrk_acc_nb(const SyrkArgs& a, hipStream_t st) {
    dim3 grid((unsigned)a.nblocks), block(256);
    const bool fullk = (a.K == 16 * NB);
    if (a.fused_pack ? (!a.b || !a.w || !a.mask || !a.spart) : !a.wpack) return hipErrorInvalidValue;
#define FSNAP_LAUNCH(FK, NTL, PK)                                                                                   \
    hipLaunchKernelGGL((fsnap_syrk_acc<NB, FK, NTL, PK>), grid, block, 0, st, a.A, a.lda, a.wpack, a.m, a.K,        \
                       a.chunks_per_wave, a.part, a.cpart, a.b, a.w, a.mask, a.spart)
#define FSNAP_LAUNCH_PK(FK, NTL)                 \
    do {                                         \
        if (a.fused_pack) FSNAP_LAUNCH(FK, NTL, true); \
        else FSNAP_LAUNCH(FK, NTL, false);       \
    } while (0)
    if (fullk) {
        if (a.nontemporal) FSNAP_LAUNCH_PK(true, true);
        else FSNAP_LAUNCH_PK(true, false);
    } else {
        if (a.nontemporal) FSNAP_LAUNCH_PK(false, true);
        else FSNAP_LAUNCH_PK(false, false);
    }
#undef FSNAP_LAUNCH_PK
#undef FSNAP_LAUNCH
    return hipGetLastError();
}

template <int NB>
static hipError_t launch_syrk_wave_p_nb(const SyrkArgs& a, hipStream_t st) {
    dim3 grid((unsigned)a.nblocks), block(256);
    const bool fullk = (a.K == 16 * NB);
    if (a.fused_pack ? (!a.b || !a.w || !a.mask || !a.spart) : !a.wpack) return hipErrorInvalidValue;
    constexpr size_t fold_bytes = (size_t)2 * (NB * (NB + 1) / 2) * 256 * sizeof(double);
    const size_t pair_bytes = a.fused_pack ? (size_t)4 * (size_t)(a.chunks_per_wave + FSNAP_WAVE_P_PACK_PAD) * 64 : 0;
    const size_t lds = pair_bytes > fold_bytes ? pair_bytes : fold_bytes;
    if (lds > 64 * 1024) return hipErrorInvalidValue;           // (the planner keeps the pairs within the default dynamic limit)
#define FSNAP_LAUNCH(FK, NTL, PK)                                                                                      \
    hipLaunchKernelGGL((fsnap_syrk_wave_p<NB, FK, NTL, PK>), grid, block, lds, st, a.A, a.lda, a.wpack, a.m, a.K,       \
                       a.chunks_per_wave, a.part, a.cpart, a.b, a.w, a.mask, a.spart)
#define FSNAP_LAUNCH_PK(FK, NTL)                       \
    do {                                               \
        if (a.fused_pack) FSNAP_LAUNCH(FK, NTL, true); \
        else FSNAP_LAUNCH(FK, NTL, false);             \
    } while (0)
    if (fullk) {
        if (a.nontemporal) FSNAP_LAUNCH_PK(true, true);
        else FSNAP_LAUNCH_PK(true, false);
    } else {
        if (a.nontemporal) FSNAP_LAUNCH_PK(false, true);
        else FSNAP_LAUNCH_PK(false, false);
    }
#undef FSNAP_LAUNCH_PK
#undef FSNAP_LAUNCH
    return hipGetLastError();
}

// chunks per wave up to which kernel 1P packs its rows' pairs itself: the pairs of a workgroup must fit next to the other
// resident workgroups' LDS (wg_per_cu of them per CU) and within the 64 KiB a launch may ask for without an attribute
int64_t syrk_wave_p_max_fused_cpw(int K, int wg_per_cu) {
    const int NB = syrk_num_blocks(K);
    const int64_t fold_bytes = (int64_t)2 * (NB * (NB + 1) / 2) * 256 * 8;
    int64_t budget = (int64_t)160 * 1024 / (wg_per_cu > 0 ? wg_per_cu : 1);
    if (budget > 64 * 1024) budget = 64 * 1024;
    if (budget < fold_bytes) budget = fold_bytes;               // the fold buffer is there anyway: pairs up to its size cost nothing
    return budget / 256 - FSNAP_WAVE_P_PACK_PAD;
}

// kernel 1P (K <= 80): a.nblocks workgroups of 4 row-waves, a.chunks_per_wave chunks per row-wave
hipError_t launch_syrk_wave_p(const SyrkArgs& a, hipStream_t st) {
    switch (syrk_num_blocks(a.K)) {
        case 1: return launch_syrk_wave_p_nb<1>(a, st);
        case 2: return launch_syrk_wave_p_nb<2>(a, st);
        case 3: return launch_syrk_wave_p_nb<3>(a, st);
        case 4: return launch_syrk_wave_p_nb<4>(a, st);
        case 5: return launch_syrk_wave_p_nb<5>(a, st);
        default: return hipErrorInvalidValue;
    }
}

// LDS budget of the fused packing: chunks per wave that fit next to the look-ahead pad
int64_t syrk_acc_max_fused_cpw() { return FSNAP_ACC_LDS_DOUBLES / 32 - FSNAP_ACC_PACK_PAD; }

// kernel 1A: a.nblocks workgroups of 4 row-waves, a.chunks_per_wave chunks per row-wave
hipError_t launch_syrk_acc(const SyrkArgs& a, hipStream_t st) {
    switch (syrk_num_blocks(a.K)) {
        case 6: return launch_syrk_acc_nb<6>(a, st);
        case 7: return launch_syrk_acc_nb<7>(a, st);
        case 8: return launch_syrk_acc_nb<8>(a, st);
        case 9: return launch_syrk_acc_nb<9>(a, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_reduce(const double* part, const double* cpart, const double* spart, int nblocks,
                         int cs_per_block, int ns, int K, double* out, double* mirror, bool accumulate, hipStream_t st,
                         bool upper_mirror) {
    const int NB = syrk_num_blocks(K);
    const int ngroups = NB * (NB + 1) / 2 * 8 + (NB * 16 + 31) / 32 + 1;
    hipLaunchKernelGGL(fsnap_reduce_partials2, dim3((unsigned)ngroups), dim3(256), 0, st, part, cpart, spart, nblocks,
                       cs_per_block, ns, NB, K, out, mirror, accumulate ? 1 : 0, upper_mirror ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_syrk_tiled(const TiledArgs& a, hipStream_t st) {
    const int nitems = (int)((int64_t)a.npairs * a.nsplit);
    const int64_t per_xcd = (nitems + 7) / 8;
    dim3 grid((unsigned)(8 * per_xcd)), block(256);
    if (a.nontemporal)
        hipLaunchKernelGGL((fsnap_syrk_tiled<true>), grid, block, 0, st, a.A, a.lda, a.wpack, a.m, a.K, a.NSB,
                           a.npairs, a.chunks_per_split, nitems, a.part, a.cpart);
    else
        hipLaunchKernelGGL((fsnap_syrk_tiled<false>), grid, block, 0, st, a.A, a.lda, a.wpack, a.m, a.K, a.NSB,
                           a.npairs, a.chunks_per_split, nitems, a.part, a.cpart);
    return hipGetLastError();
}

hipError_t launch_reduce_tiled(const TiledArgs& a, double* out, bool accumulate, hipStream_t st) {
    const int64_t nG = (int64_t)a.npairs * 4096, nrest = a.NSB * 64 + 4;
    if (a.nsplit <= 16) {
        hipLaunchKernelGGL(fsnap_reduce_tiled_small, dim3((unsigned)((nG + 255) / 256)), dim3(256), 0, st, a.part, a.nsplit, a.NSB,
                           a.npairs, a.K, out, accumulate ? 1 : 0);
        hipLaunchKernelGGL(fsnap_reduce_tiled, dim3((unsigned)((nrest + 63) / 64)), dim3(1024), 0, st, a.part, a.cpart, a.spart, a.ns,
                           a.nsplit, a.NSB, a.npairs, a.K, out, accumulate ? 1 : 0, nG);
        return hipGetLastError();
    }
    const int64_t nelem = nG + nrest;
    dim3 grid((unsigned)((nelem + 63) / 64)), block(1024);
    hipLaunchKernelGGL(fsnap_reduce_tiled, grid, block, 0, st, a.part, a.cpart, a.spart, a.ns, a.nsplit, a.NSB, a.npairs, a.K,
                       out, accumulate ? 1 : 0, (int64_t)0);
    return hipGetLastError();
}

}  // namespace fsnap
