// fsnap_kernels.hip — hand-written gfx950 (MI355X / CDNA4) kernels for the FitSNAP
// linear-fit hot path.  No CUDA compatibility layer, no dual paths: this file only
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
//   1   fsnap_syrk_wave        fused mask x weight x [A|b]^T [A|b], one wave = whole block
//                              triangle in registers, no LDS (K <= 80; SPLIT = 2 variant for A/B)
//   1L  fsnap_syrk_lds_static  same statistics for 80 < K <= 128 (the BASELINE shape): rows
//                              read ONCE per workgroup, weighted once, shared through LDS in
//                              MFMA-fragment order; per-wave specialised bodies (default).
//       fsnap_syrk_lds         generic tile-table variant of 1L (A/B)
//   1T  fsnap_syrk_tiled       general K > 128: 64-column superblock pairs x row splits
//   2   fsnap_reduce_partials / fsnap_reduce_tiled   deterministic fixed-order reduction of the
//                              per-workgroup partial triangles -> packed [G | c | scalars];
//                              un-permutes the even/odd column interleave, mirrors the triangle
//   3   fsnap_weight_rows_k    stand-alone wavefront row weighting (HBM-bound)
//   4   fsnap_gemv_rows_k      preds = A @ beta (+ weighted SSE), HBM-bound
//   5   fsnap_assemble_k       post-LAMMPS assembly (_collect_lammps transform) into HBM rows
//   6   fsnap_chol_solve_k     K x K Cholesky solve on one workgroup (optional; host is faster)
//
// Common operand trick of kernels 1 / 1L / 1T: v_mfma_f64_16x16x4_f64 takes
// A[i = lane&15][k = lane>>4] and B[k = lane>>4][j = lane&15]; with k = row inside a 4-row
// chunk and i/j = column inside a 16-column block, the SAME register (w * a[row][col]) is
// the A operand of tile (p, .) and the B operand of tile (., p): no transposes, A is read
// from HBM once.  16-byte loads give a lane two ADJACENT columns; they go to two different
// column blocks (even / odd columns of a 32-column group) and the column permutation is
// undone for free by the reduction kernel's scatter.  c = (wA)^T (wb), b^T W^2 b, sum(wb)
// and the training-row count ride along on the VALU.  No floating-point atomics anywhere:
// results are run-to-run bit-identical for a given (m, K, grid).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

#include "fsnap_kernels.h"

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

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

template <int NB>
struct ChunkRegs {
    double v[NB];   // w * a[row][col_of(bq, lane&15)]  (0 for masked / out-of-range)
    double wb;      // w * b[row]
    double cnt;     // 1.0 if the row is a training row (only lanes with (lane&15)==0 use it)
};

// Raw loads of one 4-row chunk.  All global reads of the SYRK kernel go through buffer
// descriptors (V#) whose num_records ends at the end of the wave's row range: rows past
// the range (or past m) read back as zero in hardware, so the tail needs no branches
// and no address clamps, and the per-chunk address arithmetic is one scalar add
// (soffset) on loop-invariant per-lane voffsets.  Buffer-load intrinsics also keep the
// hand-written software pipeline intact: with plain pointer loads InstCombine folds
// phi(load, load) into load(phi) and moves every prefetch to the top of the next
// iteration, i.e. un-pipelines the loop.
template <int NB>
struct ChunkRaw {
    u4 pr[NB / 2 > 0 ? NB / 2 : 1];
    u2 tail;
    u2 bv, wv;
    unsigned char mk;
};

struct WaveBufs {
    __amdgpu_buffer_rsrc_t A, b, w, mask;
    unsigned voffA;   // per-lane byte offset inside a chunk: (kr*lda + 2e)*8
    unsigned voffT;   // tail block: (kr*lda + 16*(NB-1) + e)*8
    unsigned voffR;   // per-lane row offset kr*8 (b, w); mask uses kr
    unsigned chunk_bytes;  // 4*lda*8
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    // dword3 0x00020000: raw buffer, 32-bit data format (gfx9-family encoding)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

template <int NB, bool NT>
__device__ __forceinline__ void issue_chunk(ChunkRaw<NB>& r, const WaveBufs& wb, unsigned cl, int kr) {
    // cl = chunk index local to the wave (wave-uniform)
    const unsigned soff = cl * wb.chunk_bytes;
    constexpr int AUX = NT ? 2 : 0;  // 2 = nt (streamed once)
#pragma unroll
    for (int j = 0; j < NB / 2; ++j) r.pr[j] = __builtin_amdgcn_raw_buffer_load_b128(wb.A, wb.voffA + 256u * j, soff, AUX);
    if (NB & 1) r.tail = __builtin_amdgcn_raw_buffer_load_b64(wb.A, wb.voffT, soff, AUX);
    r.bv = __builtin_amdgcn_raw_buffer_load_b64(wb.b, wb.voffR, cl * 32u, 0);
    r.wv = __builtin_amdgcn_raw_buffer_load_b64(wb.w, wb.voffR, cl * 32u, 0);
    r.mk = __builtin_amdgcn_raw_buffer_load_b8(wb.mask, (unsigned)kr, cl * 4u, 0);
}

template <int NB, bool FULLK>
__device__ __forceinline__ void finish_chunk(ChunkRegs<NB>& c, const ChunkRaw<NB>& r, int K, int e) {
    const bool keep = (r.mk != 0);
    const double wv = __builtin_bit_cast(double, r.wv);
#pragma unroll
    for (int j = 0; j < NB / 2; ++j) {
        const d2 x = __builtin_bit_cast(d2, r.pr[j]);
        double x0 = wv * x[0];
        double x1 = wv * x[1];
        bool k0 = keep, k1 = keep;
        if (!FULLK) {
            k0 = k0 && (32 * j + 2 * e < K);
            k1 = k1 && (32 * j + 2 * e + 1 < K);
        }
        c.v[2 * j] = k0 ? x0 : 0.0;
        c.v[2 * j + 1] = k1 ? x1 : 0.0;
    }
    if (NB & 1) {
        double x = wv * __builtin_bit_cast(double, r.tail);
        bool kt = keep;
        if (!FULLK) kt = kt && (16 * (NB - 1) + e < K);
        c.v[NB - 1] = kt ? x : 0.0;
    }
    c.wb = keep ? wv * __builtin_bit_cast(double, r.bv) : 0.0;
    c.cnt = keep ? 1.0 : 0.0;
}

// One chunk of matrix work for sub-wave SUB of SPLIT: the tiles t of the packed upper
// triangle with t % SPLIT == SUB (local accumulator index t / SPLIT).
template <int NB, int SPLIT, int SUB>
__device__ __forceinline__ void mfma_chunk(d4 (&acc)[(NB * (NB + 1) / 2 + SPLIT - 1) / SPLIT],
                                           const ChunkRegs<NB>& c) {
#pragma unroll
    for (int p = 0; p < NB; ++p) {
#pragma unroll
        for (int q = p; q < NB; ++q) {
            constexpr int dummy = 0;
            (void)dummy;
            const int t = tri_index(p, q, NB);
            if (t % SPLIT == SUB) {
                acc[t / SPLIT] = __builtin_amdgcn_mfma_f64_16x16x4f64(c.v[p], c.v[q], acc[t / SPLIT], 0, 0, 0);
            }
        }
    }
}

template <int NB>
__device__ __forceinline__ void valu_c_chunk(double (&cacc)[NB], const ChunkRegs<NB>& c) {
#pragma unroll
    for (int p = 0; p < NB; ++p) cacc[p] = __builtin_fma(c.v[p], c.wb, cacc[p]);
}

template <int NB>
__device__ __forceinline__ void valu_s_chunk(double& bb, double& sbw, double& cnt, const ChunkRegs<NB>& c) {
    bb = __builtin_fma(c.wb, c.wb, bb);
    sbw += c.wb;
    cnt += c.cnt;
}

__device__ __forceinline__ double xlane_sum_rows(double x) {
    // sum over the four 16-lane row groups (lanes l, l^16, l^32, l^48)
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    return x;
}

// Whole life of one wave: stream its chunk range, keep its share of the triangle in
// registers, write per-row-wave partials.  SUB == 0 additionally carries c / scalars.
template <int NB, int SPLIT, int SUB, int DEPTH, bool FULLK, bool NT>
__device__ __forceinline__ void syrk_wave_body(const double* __restrict__ A, int64_t lda,
                                               const double* __restrict__ b, const double* __restrict__ w,
                                               const unsigned char* __restrict__ mask, int64_t m, int K,
                                               int64_t c0, int64_t c1, int64_t rowwave, double* lds,
                                               double* __restrict__ part, double* __restrict__ cpart,
                                               double* __restrict__ spart) {
    constexpr int NTILE = NB * (NB + 1) / 2;
    constexpr int NTW = (NTILE + SPLIT - 1) / SPLIT;
    const int lane = threadIdx.x & 63;
    const int e = lane & 15;
    const int kr = lane >> 4;

    d4 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
    double cacc[NB];
#pragma unroll
    for (int p = 0; p < NB; ++p) cacc[p] = 0.0;
    double bb = 0.0, sbw = 0.0, cnt = 0.0;

    ChunkRaw<NB> r0, r1, r2;
    ChunkRegs<NB> cr;

    // buffer descriptors covering exactly this wave's rows [row0, row1)
    const int64_t row0 = c0 << 2;
    int64_t row1 = c1 << 2;
    if (row1 > m) row1 = m;
    const int64_t nrow = row1 > row0 ? row1 - row0 : 0;
    WaveBufs wb;
    // +16 B: the 16-byte column over-read of the last row (select-zeroed) stays in range
    wb.A = make_rsrc(A + row0 * lda, (unsigned)(nrow * lda * 8 + (nrow ? 16 : 0)));
    wb.b = make_rsrc(b + row0, (unsigned)(nrow * 8));
    wb.w = make_rsrc(w + row0, (unsigned)(nrow * 8));
    wb.mask = make_rsrc(mask + row0, (unsigned)nrow);
    wb.voffA = (unsigned)((kr * lda + 2 * e) * 8);
    wb.voffT = (unsigned)((kr * lda + 16 * (NB - 1) + e) * 8);
    wb.voffR = (unsigned)(kr * 8);
    wb.chunk_bytes = (unsigned)(lda * 32);
    const unsigned ncl = (unsigned)(c1 > c0 ? c1 - c0 : 0);

    // Software pipeline, DEPTH chunks in flight per wave.  The loop body is branch-free
    // (one scheduling region: DEPTH x {weight, prefetch, MFMAs}); chunk slots past the
    // wave's range read zeros through the bounds-checked descriptors, so a wave wastes
    // at most DEPTH-1 chunks of matrix-pipe time.
#define FSNAP_STAGE(R, OFF)                                                    \
    finish_chunk<NB, FULLK>(cr, R, K, e);                                       \
    issue_chunk<NB, NT>(R, wb, cl + (OFF) + DEPTH, kr);                         \
    mfma_chunk<NB, SPLIT, SUB>(acc, cr);                                        \
    if (SUB == 0) valu_c_chunk<NB>(cacc, cr);                                   \
    if (SUB == SPLIT - 1) valu_s_chunk<NB>(bb, sbw, cnt, cr);
    if (ncl > 0) {
        issue_chunk<NB, NT>(r0, wb, 0, kr);
        issue_chunk<NB, NT>(r1, wb, 1, kr);
        if (DEPTH == 3) issue_chunk<NB, NT>(r2, wb, 2, kr);
        for (unsigned cl = 0; cl < ncl; cl += DEPTH) {
            FSNAP_STAGE(r0, 0)
            FSNAP_STAGE(r1, 1)
            if (DEPTH == 3) {
                FSNAP_STAGE(r2, 2)
            }
        }
    }
#undef FSNAP_STAGE

    // epilogue 1: fold the four row-waves of the workgroup through LDS (two rounds:
    // {2,3} -> {0,1}, then 1 -> 0) so that only ONE partial triangle per workgroup goes
    // to HBM.  Slot layout [slot][u][i][lane] doubles: conflict-free ds_write/read_b64.
    // All waves of the workgroup execute the same three barriers (wave-uniform paths).
    {
        const int rw = (int)(rowwave & 3);
        double* slot_hi = lds + (size_t)(((rw & 1) * SPLIT + SUB) * NTW) * 256;  // rounds use 2*SPLIT / SPLIT slots
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
        double* slot_lo = lds + (size_t)(SUB * NTW) * 256;
        if (rw == 1) {
#pragma unroll
            for (int u = 0; u < NTW; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) slot_lo[(u * 4 + i) * 64 + lane] = acc[u][i];
        }
        __syncthreads();
        if (rw == 0) {
            // epilogue 2: per-workgroup partial, coalesced 512-B stores
            double* pw = part + (rowwave >> 2) * (int64_t)(NTILE * 256);
#pragma unroll
            for (int u = 0; u < NTW; ++u) {
                const int t = u * SPLIT + SUB;
                if (t < NTILE) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        pw[(t * 4 + i) * 64 + lane] = acc[u][i] + slot_lo[(u * 4 + i) * 64 + lane];
                }
            }
        }
    }
    if (SUB == 0) {
        double* cw = cpart + rowwave * (int64_t)(NB * 16);
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            double s = xlane_sum_rows(cacc[p]);
            if (kr == 0) cw[p * 16 + e] = s;
        }
    }
    if (SUB == SPLIT - 1) {
        // every lane of a 16-lane row group carries the same bb/sbw/cnt
        double sb = xlane_sum_rows(bb), ss = xlane_sum_rows(sbw), sc = xlane_sum_rows(cnt);
        if (lane == 0) {
            double* sw = spart + rowwave * 4;
            sw[0] = sb;
            sw[1] = ss;
            sw[2] = sc;
            sw[3] = 0.0;
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------
// Kernel 1: fused mask x weight x SYRK.
// Workgroup = 4 row-waves x SPLIT sub-waves (256*SPLIT threads): row-wave g =
// blockIdx.x*4 + (wave & 3) streams chunks [g*cpw, (g+1)*cpw) of 4 rows; its SPLIT
// sub-waves (wave >> 2) read the same rows (second read is an L1/L2 hit) and own
// disjoint halves of the block triangle, so that at K = 128 each wave holds 18 tiles
// (144 accumulator registers) and two waves share each SIMD: one wave's loads and
// weighting overlap the other's MFMAs.
// Partial layout (doubles): part[workgroup][NT][4][64] (row-waves folded through LDS) |
// cpart[rowwave][NB][16] | spart[rowwave][4]
// ---------------------------------------------------------------------------------
template <int NB, int SPLIT, int DEPTH, bool FULLK, bool NT>
__global__ __launch_bounds__(256 * SPLIT, ((NB * (NB + 1) / 2 + SPLIT - 1) / SPLIT > 20) ? 1 : 2) void
fsnap_syrk_wave(const double* __restrict__ A, int64_t lda, const double* __restrict__ b,
                const double* __restrict__ w, const unsigned char* __restrict__ mask, int64_t m, int K,
                int64_t chunks_per_wave, double* __restrict__ part, double* __restrict__ cpart,
                double* __restrict__ spart) {
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t rowwave = (int64_t)blockIdx.x * 4 + (wv & 3);
    const int sub = wv >> 2;
    __shared__ double lds[2 * SPLIT * ((NB * (NB + 1) / 2 + SPLIT - 1) / SPLIT) * 256];
    const int64_t nchunks = (m + 3) >> 2;
    int64_t c0 = rowwave * chunks_per_wave;
    int64_t c1 = c0 + chunks_per_wave;
    if (c1 > nchunks) c1 = nchunks;
    if (SPLIT == 1 || sub == 0) {
        syrk_wave_body<NB, SPLIT, 0, DEPTH, FULLK, NT>(A, lda, b, w, mask, m, K, c0, c1, rowwave, lds, part, cpart, spart);
    } else {
        syrk_wave_body<NB, SPLIT, (SPLIT > 1 ? 1 : 0), DEPTH, FULLK, NT>(A, lda, b, w, mask, m, K, c0, c1, rowwave, lds,
                                                                 part, cpart, spart);
    }
}

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

template <int T>
__device__ __forceinline__ void acc_mfma(double a, double b, d4 (&vt)[4]) {
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

template <int T>
__device__ __forceinline__ d4 acc_read(const d4 (&vt)[4]) {
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

template <int TB, int N, int... U>
__device__ __forceinline__ void acc_read_range(d4 (&tmp)[N], const d4 (&vt)[4], std::integer_sequence<int, U...>) {
    ((tmp[U] = acc_read<TB + U>(vt)), ...);
}

// row p of the triangle: tiles (p, p..NB-1)
template <int NB, int P, int... Q>
__device__ __forceinline__ void acc_row(const double (&V)[NB], d4 (&vt)[4], std::integer_sequence<int, Q...>) {
    (acc_mfma<tri_index(P, P + Q, NB)>(V[P], V[P + Q], vt), ...);
}

// Raw loads of kernel 1A.  The row mask is applied by the LOADS: a lane whose row is a test row (or lies past the
// wave's range) uses an out-of-range buffer offset, so the hardware bounds check returns zeros for its A, b and w
// values -- w = 0, b = 0 and a = 0 make every product of that row vanish without a single select instruction
// (and garbage such as NaN / Inf in a masked row is never even fetched).  Every VALU instruction of this kernel
// costs matrix-pipe time (fp64 MFMAs and VALU instructions serialise on the SIMD), hence: two offset selects per
// chunk instead of two selects per value.
template <int NB>
struct RawM {
    u4 pr[NB / 2 > 0 ? NB / 2 : 1];
    u2 tail;
    u2 bv, wv;
};

constexpr unsigned FSNAP_OOB_VOFF = 0xFFFFF000u;   // > any wave's buffer size (plan_geometry: < 0xFFF00010), no 32-bit wrap

template <int NB, bool NT>
__device__ __forceinline__ void issue_masked(RawM<NB>& r, const WaveBufs& wb, unsigned cl, unsigned mk) {
    const unsigned soff = cl * wb.chunk_bytes;
    constexpr int AUX = NT ? 2 : 0;
    const bool keep = (mk != 0);
    const unsigned va = keep ? wb.voffA : FSNAP_OOB_VOFF;
    const unsigned vr = keep ? wb.voffR : FSNAP_OOB_VOFF;
#pragma unroll
    for (int j = 0; j < NB / 2; ++j) r.pr[j] = __builtin_amdgcn_raw_buffer_load_b128(wb.A, va + 256u * j, soff, AUX);
    if (NB & 1) {
        const unsigned vtl = keep ? wb.voffT : FSNAP_OOB_VOFF;
        r.tail = __builtin_amdgcn_raw_buffer_load_b64(wb.A, vtl, soff, AUX);
    }
    r.bv = __builtin_amdgcn_raw_buffer_load_b64(wb.b, vr, cl * 32u, 0);
    r.wv = __builtin_amdgcn_raw_buffer_load_b64(wb.w, vr, cl * 32u, 0);
}

// piece P of the refill of raw set r (one load instruction per MFMA slot: a block of back-to-back VMEM
// instructions holds the in-order wave at the address path while the matrix pipe drains)
template <int NB, bool NT, int P>
__device__ __forceinline__ void issue_piece(RawM<NB>& r, const WaveBufs& wb, unsigned cl, unsigned va, unsigned vtl,
                                            unsigned vr) {
    constexpr int AUX = NT ? 2 : 0;
    constexpr int NPR = NB / 2;
    const unsigned soff = cl * wb.chunk_bytes;
    if (P < NPR) r.pr[P] = __builtin_amdgcn_raw_buffer_load_b128(wb.A, va + 256u * P, soff, AUX);
    if ((NB & 1) && P == NPR) r.tail = __builtin_amdgcn_raw_buffer_load_b64(wb.A, vtl, soff, AUX);
    constexpr int PB = NPR + (NB & 1);
    if (P == PB) r.bv = __builtin_amdgcn_raw_buffer_load_b64(wb.b, vr, cl * 32u, 0);
    if (P == PB + 1) r.wv = __builtin_amdgcn_raw_buffer_load_b64(wb.w, vr, cl * 32u, 0);
    static_assert(NPR + (NB & 1) + 2 <= NB, "one load piece per MFMA slot");
}

// mask byte of the lane's row in chunk cl, normalised to 0 / 1 (rows past the range read 0)
__device__ __forceinline__ unsigned load_mask(const WaveBufs& wb, unsigned cl, int kr) {
    return (unsigned)__builtin_amdgcn_raw_buffer_load_b8(wb.mask, (unsigned)kr, cl * 4u, 0);
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

// Slot P of a step (between row P and row P + 1 of the MFMAs).  The pieces are independent of each other (a wave
// issues in order: a dependent chain here would hold back row P + 1):
//   V[P] <- w * raw block P        cacc[P-1] += V[P-1] * wbv        slots 1, 2: bb, sum_bw
template <int NB, bool FULLK, bool NT, int P>
__device__ __forceinline__ void acc_slot(double (&V)[NB], const RawM<NB>& RN, double wv, double wbv, double& wbp, int K,
                                         int e, double (&cacc)[NB], double& bb, double& sbw, RawM<NB>& RF,
                                         const WaveBufs& wb, unsigned cl_fill, unsigned va, unsigned vtl, unsigned vr) {
#if !defined(FSNAP_ACC_ABL) || !(FSNAP_ACC_ABL & 1)
    issue_piece<NB, NT, P>(RF, wb, cl_fill, va, vtl, vr);
#endif
    if (P == 0) cacc[NB - 1] = __builtin_fma(V[NB - 1], wbp, cacc[NB - 1]);   // previous chunk's last block
    else cacc[P - 1] = __builtin_fma(V[P - 1], wbv, cacc[P - 1]);
    V[P] = weighted_block<NB, FULLK>(RN, P, wv, K, e);
    if (P == 1) bb = __builtin_fma(wbv, wbv, bb);
    if (P == 2) sbw += wbv;
    if (P == NB - 1) wbp = wbv;
    __builtin_amdgcn_sched_barrier(0);   // keep the slot between row P and row P + 1
}

// One chunk: the MFMA rows of the chunk held in V, interleaved with the preparation of the next chunk (raw
// registers RN) block by block.  First the raw set RF (consumed one step ago) is refilled three chunks ahead,
// using the mask byte fetched during the previous step, and the mask of the following chunk is requested.
template <int NB, bool FULLK, bool NT, int... P>
__device__ __forceinline__ void acc_step(double (&V)[NB], d4 (&vt)[4], RawM<NB>& RF, const RawM<NB>& RN,
                                         const WaveBufs& wb, unsigned cl_fill, unsigned& mk, unsigned& cnt, int K, int e,
                                         int kr, double (&cacc)[NB], double& bb, double& sbw, double& wbp,
                                         std::integer_sequence<int, P...>) {
    const bool keep = (mk != 0);                    // mask bytes may be any non-zero value for "training row"
    cnt += keep ? 1u : 0u;
    const unsigned va = keep ? wb.voffA : FSNAP_OOB_VOFF;
    const unsigned vr = keep ? wb.voffR : FSNAP_OOB_VOFF;
    const unsigned vtl = (NB & 1) ? (keep ? wb.voffT : FSNAP_OOB_VOFF) : 0u;
#if !defined(FSNAP_ACC_ABL) || !(FSNAP_ACC_ABL & 1)   // tools/syrk_trace.hip diagnostics: 1 = no loads, 2 = no VALU work
    // the mask request goes out BEFORE this step's row loads: vmcnt retires in order, so the next step can wait for
    // its mask byte without also draining the row loads issued here (they stay two steps ahead of their use)
    mk = load_mask(wb, cl_fill + 1, kr);
#endif
    const double wv = __builtin_bit_cast(double, RN.wv);
    const double wbv = wv * __builtin_bit_cast(double, RN.bv);
    __builtin_amdgcn_sched_barrier(0);
#if defined(FSNAP_ACC_ABL) && (FSNAP_ACC_ABL & 2)
    (acc_row<NB, P>(V, vt, std::make_integer_sequence<int, NB - P>{}), ...);
    (void)wbv;
#else
    ((acc_row<NB, P>(V, vt, std::make_integer_sequence<int, NB - P>{}),
      acc_slot<NB, FULLK, NT, P>(V, RN, wv, wbv, wbp, K, e, cacc, bb, sbw, RF, wb, cl_fill, va, vtl, vr)),
     ...);
#endif
}

}  // namespace

template <int NB, bool FULLK, bool NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void
fsnap_syrk_acc(const double* __restrict__ A, int64_t lda, const double* __restrict__ b, const double* __restrict__ w,
               const unsigned char* __restrict__ mask, int64_t m, int K, int64_t chunks_per_wave,
               double* __restrict__ part, double* __restrict__ cpart, double* __restrict__ spart) {
    constexpr int NTILE = NB * (NB + 1) / 2;
    constexpr int HALF = (NTILE + 1) / 2;
    __shared__ double lds[4 * HALF * 256];
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
    WaveBufs wb;
    wb.A = make_rsrc(A + row0 * lda, (unsigned)(nrow * lda * 8 + (nrow ? 16 : 0)));
    wb.b = make_rsrc(b + row0, (unsigned)(nrow * 8));
    wb.w = make_rsrc(w + row0, (unsigned)(nrow * 8));
    wb.mask = make_rsrc(mask + row0, (unsigned)nrow);
    wb.voffA = (unsigned)((kr * lda + 2 * e) * 8);
    wb.voffT = (unsigned)((kr * lda + 16 * (NB - 1) + e) * 8);
    wb.voffR = (unsigned)(kr * 8);
    wb.chunk_bytes = (unsigned)(lda * 32);
    const unsigned ncl = (unsigned)(c1 - c0);

    // the compiler must count a[0:255] as used (register allocation granule of the kernel descriptor): the
    // clobber makes its resource analysis see the highest accumulation register
    asm volatile("" : : : "a0", "a255");
    acc_zero_all(std::make_integer_sequence<int, 256>{});
    d4 vt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) vt[u] = d4{0.0, 0.0, 0.0, 0.0};
    double cacc[NB], V[NB];
#pragma unroll
    for (int p = 0; p < NB; ++p) cacc[p] = 0.0;
    double bb = 0.0, sbw = 0.0;
    unsigned cnt = 0;

    // FSNAP_ACC_DEPTH raw sets in flight (3 by default; 4 = one more chunk of prefetch, tools/syrk_trace.hip A/B)
#ifndef FSNAP_ACC_DEPTH
#define FSNAP_ACC_DEPTH 3
#endif
    RawM<NB> r0, r1, r2;
#if FSNAP_ACC_DEPTH == 4
    RawM<NB> r3;
#endif
    constexpr auto rows = std::make_integer_sequence<int, NB>{};
    if (ncl > 0) {
        const unsigned m0 = load_mask(wb, 0, kr) ? 1u : 0u, m1 = load_mask(wb, 1, kr) ? 1u : 0u;
        const unsigned m2 = load_mask(wb, 2, kr) ? 1u : 0u;
#if FSNAP_ACC_DEPTH == 4
        const unsigned m3 = load_mask(wb, 3, kr) ? 1u : 0u;
        unsigned mk = load_mask(wb, 4, kr);
        cnt = m0 + m1 + m2 + m3;
#else
        unsigned mk = load_mask(wb, 3, kr);
        cnt = m0 + m1 + m2;
#endif
        issue_masked<NB, NT>(r0, wb, 0, m0);
        issue_masked<NB, NT>(r1, wb, 1, m1);
        issue_masked<NB, NT>(r2, wb, 2, m2);
#if FSNAP_ACC_DEPTH == 4
        issue_masked<NB, NT>(r3, wb, 3, m3);
#endif
        {   // chunk 0 -> V
            const double wv = __builtin_bit_cast(double, r0.wv);
            const double wbv = wv * __builtin_bit_cast(double, r0.bv);
            bb = __builtin_fma(wbv, wbv, bb);
            sbw += wbv;
#pragma unroll
            for (int p = 0; p < NB; ++p) {
                V[p] = weighted_block<NB, FULLK>(r0, p, wv, K, e);
                cacc[p] = __builtin_fma(V[p], wbv, cacc[p]);
            }
        }
        // step cl: MFMAs of chunk cl (in V), V <- chunk cl+1, refill of the raw set freed one step ago with chunk
        // cl+DEPTH.  Chunk slots past the wave's range read zeros through the bounds-checked descriptors.
        double wbp = 0.0;   // chunk 0 is fully accounted for by the prologue
#if FSNAP_ACC_DEPTH == 4
        for (unsigned cl = 0; cl < ncl; cl += 4) {
            acc_step<NB, FULLK, NT>(V, vt, r0, r1, wb, cl + 4, mk, cnt, K, e, kr, cacc, bb, sbw, wbp, rows);
            acc_step<NB, FULLK, NT>(V, vt, r1, r2, wb, cl + 5, mk, cnt, K, e, kr, cacc, bb, sbw, wbp, rows);
            acc_step<NB, FULLK, NT>(V, vt, r2, r3, wb, cl + 6, mk, cnt, K, e, kr, cacc, bb, sbw, wbp, rows);
            acc_step<NB, FULLK, NT>(V, vt, r3, r0, wb, cl + 7, mk, cnt, K, e, kr, cacc, bb, sbw, wbp, rows);
        }
#else
        for (unsigned cl = 0; cl < ncl; cl += 3) {
            acc_step<NB, FULLK, NT>(V, vt, r0, r1, wb, cl + 3, mk, cnt, K, e, kr, cacc, bb, sbw, wbp, rows);
            acc_step<NB, FULLK, NT>(V, vt, r1, r2, wb, cl + 4, mk, cnt, K, e, kr, cacc, bb, sbw, wbp, rows);
            acc_step<NB, FULLK, NT>(V, vt, r2, r0, wb, cl + 5, mk, cnt, K, e, kr, cacc, bb, sbw, wbp, rows);
        }
#endif
        cacc[NB - 1] = __builtin_fma(V[NB - 1], wbp, cacc[NB - 1]);   // last prepared chunk (zeros past the range)
    }
    // the last MFMAs (16 passes) must have left the pipe before their accumulators are read
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(vt[0]), "+v"(vt[1]), "+v"(vt[2]), "+v"(vt[3]));
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

    // epilogue: fold the four row-waves through LDS in two halves of the triangle.  Every wave parks its tiles of
    // the half (4 slots x HALF tiles x 2 KiB = 144 KiB at K = 128), then wave r sums a quarter of the tiles over the
    // four slots in a fixed order and stores them: one partial triangle per workgroup, all four waves busy.
    double* pw = part + (int64_t)blockIdx.x * (int64_t)(NTILE * 256);
    auto fold_half = [&](auto half_tag) {
        constexpr int H = decltype(half_tag)::value;
        constexpr int TB = H * HALF;
        constexpr int NT_H = (TB + HALF <= NTILE) ? HALF : (NTILE - TB);
        constexpr int Q = (NT_H + 3) / 4;
        double* slot = lds + (size_t)rw * HALF * 256;
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
                const double s01 = lds[o] + lds[HALF * 256 + o];
                pw[((TB + u) * 4 + i) * 64 + lane] = (s01 + lds[2 * HALF * 256 + o]) + lds[3 * HALF * 256 + o];
            }
        }
        if (H == 0) __syncthreads();
    };
    fold_half(std::integral_constant<int, 0>{});
    fold_half(std::integral_constant<int, 1>{});

    double* cw = cpart + rowwave * (int64_t)(NB * 16);
#pragma unroll
    for (int p = 0; p < NB; ++p) {
        double sm = xlane_sum_rows(cacc[p]);
        if (kr == 0) cw[p * 16 + e] = sm;
    }
    double sb = xlane_sum_rows(bb), ss = xlane_sum_rows(sbw), sc = xlane_sum_rows((double)cnt);
    if (lane == 0) {
        double* sw = spart + rowwave * 4;
        sw[0] = sb;
        sw[1] = ss;
        sw[2] = sc;
        sw[3] = 0.0;
    }
#ifdef FSNAP_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 4096) fsnap_trace_buf[blockIdx.x * 8 + 5] = wall_clock64();   // after the epilogue
#endif
}

// ---------------------------------------------------------------------------------
// Kernel 2: deterministic reduction of the partials into the packed statistics buffer
//   out = [G (K*K row-major) | c (K) | bTb, sum_bw, n_train].
// One workgroup (1024 threads) = 16 consecutive partial elements x 64 slices of the
// partial range: thread (g = tid>>4, l = tid&15) sums partials g, g+64, g+128, ... in a
// fixed order with 8 independent loads in flight, then a fixed-order 64-way LDS
// combine.  Element space: [0, NT*256) triangle elements (one partial per workgroup of
// kernel 1), then NB*16 c elements and 4 scalars (one partial per row-wave).
// The scatter undoes the even/odd column interleave and mirrors the upper triangle.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void fsnap_reduce_partials(const double* __restrict__ part,
                                                              const double* __restrict__ cpart,
                                                              const double* __restrict__ spart, int nblocks,
                                                              int cs_per_block, int NB, int K,
                                                              double* __restrict__ out, double* __restrict__ mirror,
                                                              int accumulate) {
    // One workgroup (1024 threads) = 16 consecutive elements (one 128-B line per partial)
    // x 64 slices of the partial range; ~585 workgroups at K = 128, every load in flight.
    __shared__ double red[1024];
    __shared__ double red2[64];
    const int NTILE = NB * (NB + 1) / 2;
    const int nG = NTILE * 256, nC = NB * 16, nS = 4;
    const int tid = threadIdx.x, g = tid >> 4, l = tid & 15;
    const int idx = blockIdx.x * 16 + l;
    const double* src = nullptr;
    int64_t stride = 0;
    int np = nblocks * cs_per_block;  // c / scalar partials per workgroup of kernel 1 / 1L
    if (idx < nG) {
        src = part + idx;
        stride = nG;
        np = nblocks;
    } else if (idx < nG + nC) {
        src = cpart + (idx - nG);
        stride = nC;
    } else if (idx < nG + nC + nS) {
        src = spart + (idx - nG - nC);
        stride = nS;
    }
    double s = 0.0;
    if (src) {
        int p = g;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0, a5 = 0.0, a6 = 0.0, a7 = 0.0;
        for (; p + 448 < np; p += 512) {
            const double x0 = src[(int64_t)p * stride], x1 = src[(int64_t)(p + 64) * stride];
            const double x2 = src[(int64_t)(p + 128) * stride], x3 = src[(int64_t)(p + 192) * stride];
            const double x4 = src[(int64_t)(p + 256) * stride], x5 = src[(int64_t)(p + 320) * stride];
            const double x6 = src[(int64_t)(p + 384) * stride], x7 = src[(int64_t)(p + 448) * stride];
            a0 += x0; a1 += x1; a2 += x2; a3 += x3; a4 += x4; a5 += x5; a6 += x6; a7 += x7;
        }
        for (; p < np; p += 64) a0 += src[(int64_t)p * stride];
        s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    }
    red[tid] = s;
    __syncthreads();
    // fixed-order 64-way combine: 4 threads per element sum 16 slices each, then 4 -> 1
    if (tid < 64) {
        const int el = tid & 15, q = tid >> 4;
        double t4 = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t4 += red[(q * 16 + k) * 16 + el];
        red2[tid] = t4;
    }
    __syncthreads();
    if (tid < 16 && src) {
        const double tot = (red2[tid] + red2[16 + tid]) + (red2[32 + tid] + red2[48 + tid]);
        if (idx < nG) {
            int t = idx >> 8, rem = idx & 255, i = rem >> 6, ln = rem & 63;
            // invert tri_index: find p with tri_index(p,p) <= t
            int p = 0;
            while (p + 1 < NB && tri_index(p + 1, p + 1, NB) <= t) ++p;
            int q = p + (t - tri_index(p, p, NB));
            int ep = (ln >> 4) + 4 * i, eq = ln & 15;
            int r = col_of(p, ep, NB), c = col_of(q, eq, NB);
            if (r < K && c < K) {
                // accumulate: out += statistics of these rows (streaming per-batch accumulation, the reference's
                // `c += cm; d += dm`, transpose_trick/example.py:236-237); (r, c) and (c, r) hold equal values
                const double val = accumulate ? out[(int64_t)r * K + c] + tot : tot;
                out[(int64_t)r * K + c] = val;
                if (p != q) out[(int64_t)c * K + r] = val;
                if (mirror) {        // page-locked host copy written by the same kernel (no separate D2H copy)
                    mirror[(int64_t)r * K + c] = val;
                    if (p != q) mirror[(int64_t)c * K + r] = val;
                    // compact copy of the diagonal behind the packed statistics: the host's scaling pass needs it
                    // first, and 128 entries with a 1 KiB stride are 128 cold cache lines
                    if (r == c) mirror[(int64_t)K * K + K + 3 + r] = val;
                }
            }
        } else if (idx < nG + nC) {
            int j = idx - nG;
            int cidx = col_of(j >> 4, j & 15, NB);
            if (cidx < K) {
                const double val = accumulate ? out[(int64_t)K * K + cidx] + tot : tot;
                out[(int64_t)K * K + cidx] = val;
                if (mirror) mirror[(int64_t)K * K + cidx] = val;
            }
        } else {
            int j = idx - nG - nC;
            if (j < 3) {
                const double val = accumulate ? out[(int64_t)K * K + K + j] + tot : tot;
                out[(int64_t)K * K + K + j] = val;
                if (mirror) mirror[(int64_t)K * K + K + j] = val;
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// Kernel 1L: fused mask x weight x SYRK for 80 < K <= 128 with the weighted rows SHARED
// through LDS (the production kernel for the BASELINE 10^6 x 128 shape).
// A workgroup of NW waves owns a contiguous row range and the WHOLE block triangle; the
// NT = NB(NB+1)/2 tiles are dealt to the waves in contiguous runs (at NB = 8, NW = 8:
// waves 0-3 own 5 tiles, waves 4-7 own 4, so every SIMD carries 9).  Rows move in stages
// of NW chunks (NW x 4 rows): wave v loads chunk v of the next stage from HBM (16-byte
// buffer loads, each row read exactly ONCE per launch), applies mask and weight once,
// accumulates c / scalars for that chunk on the VALU and writes the weighted values to
// LDS in MFMA-fragment order ([chunk][block][lane], conflict-free ds_write/read_b64).
// After one barrier per stage every wave reads, per chunk, only the operand blocks of
// its own tiles and issues its MFMAs.  Versus kernel 1 with SPLIT = 2 this halves the
// L2/HBM request traffic (measured: the duplicate `nt` reads of the two sub-waves MISS in
// L2 and the chip fetched 2x the algorithmic bytes), and cuts the fp64 VALU work 8x.
// LDS: 2 stages x NW chunks x NB blocks x 512 B (64 KiB at NB = 8, NW = 8), so two
// workgroups share a CU.  Partials: part[wg][NT][4][64] | cpart[wg][NB][16] | spart[wg][4].
// ---------------------------------------------------------------------------------
namespace {

// NTM = number of tiles THIS wave owns (compile-time: the kernel dispatches on the wave's
// class so that the stage loop is one branch-free scheduling region).
template <int NB, int NW, int NTM, bool FULLK, bool NT>
__device__ __forceinline__ void syrk_lds_body(double* lds, const WaveBufs& wb, int K, unsigned nstage, int wv, int t0,
                                              double* __restrict__ pw, double* __restrict__ cw,
                                              double* __restrict__ sw) {
    constexpr int NTILE = NB * (NB + 1) / 2;
    constexpr int STAGE_DOUBLES = NW * NB * 64;
    const int lane = threadIdx.x & 63, e = lane & 15, kr = lane >> 4;

    int tp[NTM], tq[NTM];
#pragma unroll
    for (int u = 0; u < NTM; ++u) {
        int t = t0 + u;
        if (t >= NTILE) t = NTILE - 1;
        int p = 0;
        while (p + 1 < NB && tri_index(p + 1, p + 1, NB) <= t) ++p;
        tp[u] = p * 64;                                   // LDS offsets (doubles) of the operand blocks
        tq[u] = (p + (t - tri_index(p, p, NB))) * 64;
    }
    d4 acc[NTM];
#pragma unroll
    for (int u = 0; u < NTM; ++u) acc[u] = d4{0.0, 0.0, 0.0, 0.0};
    double cacc[NB];
#pragma unroll
    for (int p = 0; p < NB; ++p) cacc[p] = 0.0;
    double bb = 0.0, sbw = 0.0, cnt = 0.0;

    ChunkRaw<NB> raw;
    ChunkRegs<NB> cr;
    // weight the raw chunk in `raw`, accumulate c / scalars, park it in LDS stage `buf`
    auto park = [&](int buf) {
        finish_chunk<NB, FULLK>(cr, raw, K, e);
        valu_c_chunk<NB>(cacc, cr);
        valu_s_chunk<NB>(bb, sbw, cnt, cr);
        double* dst = lds + buf * STAGE_DOUBLES + (wv * NB) * 64 + lane;
#pragma unroll
        for (int bq = 0; bq < NB; ++bq) dst[bq * 64] = cr.v[bq];
    };

    if (nstage > 0) {
        issue_chunk<NB, NT>(raw, wb, (unsigned)wv, kr);
        park(0);
        issue_chunk<NB, NT>(raw, wb, (unsigned)(NW + wv), kr);
        __syncthreads();
        for (unsigned s = 0; s < nstage; ++s) {
            const double* src = lds + (s & 1) * STAGE_DOUBLES + lane;
#pragma unroll
            for (int c = 0; c < NW; ++c) {
#pragma unroll
                for (int u = 0; u < NTM; ++u) {
                    const double va = src[c * NB * 64 + tp[u]];
                    const double vb = src[c * NB * 64 + tq[u]];
                    acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(va, vb, acc[u], 0, 0, 0);
                }
            }
            // next stage: weight + park what was prefetched, then prefetch the stage after
            park((s + 1) & 1);
            issue_chunk<NB, NT>(raw, wb, (s + 2) * NW + wv, kr);
            __syncthreads();
        }
    }

    // epilogue: tiles are disjoint across waves -> straight to the per-workgroup partial
#pragma unroll
    for (int u = 0; u < NTM; ++u) {
        const int t = t0 + u;
        if (t < NTILE) {
#pragma unroll
            for (int i = 0; i < 4; ++i) pw[(t * 4 + i) * 64 + lane] = acc[u][i];
        }
    }
    // c / scalars: fold the NW per-wave partials through LDS in a fixed order -> one per workgroup
    // (the stage buffers are free: every wave is past the last barrier of the stage loop)
    constexpr int CS = NB * 16 + 4;
    double* fold = lds + wv * CS;
#pragma unroll
    for (int p = 0; p < NB; ++p) {
        double sm = xlane_sum_rows(cacc[p]);
        if (kr == 0) fold[p * 16 + e] = sm;
    }
    double sb = xlane_sum_rows(bb), ss = xlane_sum_rows(sbw), sc = xlane_sum_rows(cnt);
    if (lane == 0) {
        fold[NB * 16 + 0] = sb;
        fold[NB * 16 + 1] = ss;
        fold[NB * 16 + 2] = sc;
        fold[NB * 16 + 3] = 0.0;
    }
    __syncthreads();
    if (wv == 0) {
        for (int j = lane; j < CS; j += 64) {
            double tot = 0.0;
#pragma unroll
            for (int k = 0; k < NW; ++k) tot += lds[k * CS + j];
            if (j < NB * 16) cw[j] = tot;
            else sw[j - NB * 16] = tot;
        }
    }
}

}  // namespace

template <int NB, int NW, bool FULLK, bool NT>
__global__ __launch_bounds__(64 * NW, 4) void fsnap_syrk_lds(const double* __restrict__ A, int64_t lda,
                                                             const double* __restrict__ b,
                                                             const double* __restrict__ w,
                                                             const unsigned char* __restrict__ mask, int64_t m, int K,
                                                             int64_t chunks_per_wg, double* __restrict__ part,
                                                             double* __restrict__ cpart, double* __restrict__ spart) {
    constexpr int NTILE = NB * (NB + 1) / 2;
    constexpr int NTW = (NTILE + NW - 1) / NW;        // tiles of a "big" wave
    constexpr int BIG = NTILE - (NTW - 1) * NW;       // number of waves owning NTW tiles (the rest own NTW - 1)
    __shared__ double lds[2 * NW * NB * 64];

    const int lane = threadIdx.x & 63, e = lane & 15, kr = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t wg = blockIdx.x;

    // row range of this workgroup and the descriptors that bound it
    const int64_t nchunks = (m + 3) >> 2;
    int64_t c0 = wg * chunks_per_wg;
    int64_t c1 = c0 + chunks_per_wg;
    if (c1 > nchunks) c1 = nchunks;
    if (c0 > c1) c0 = c1;
    const int64_t row0 = c0 << 2;
    int64_t row1 = c1 << 2;
    if (row1 > m) row1 = m;
    const int64_t nrow = row1 > row0 ? row1 - row0 : 0;
    WaveBufs wb;
    wb.A = make_rsrc(A + row0 * lda, (unsigned)(nrow * lda * 8 + (nrow ? 16 : 0)));
    wb.b = make_rsrc(b + row0, (unsigned)(nrow * 8));
    wb.w = make_rsrc(w + row0, (unsigned)(nrow * 8));
    wb.mask = make_rsrc(mask + row0, (unsigned)nrow);
    wb.voffA = (unsigned)((kr * lda + 2 * e) * 8);
    wb.voffT = (unsigned)((kr * lda + 16 * (NB - 1) + e) * 8);
    wb.voffR = (unsigned)(kr * 8);
    wb.chunk_bytes = (unsigned)(lda * 32);
    const unsigned ncl = (unsigned)(c1 - c0);
    const unsigned nstage = (ncl + NW - 1) / NW;

    double* pw = part + wg * (int64_t)(NTILE * 256);
    double* cw = cpart + wg * (int64_t)(NB * 16);
    double* sw = spart + wg * 4;
    if (wv < BIG) {
        syrk_lds_body<NB, NW, NTW, FULLK, NT>(lds, wb, K, nstage, wv, wv * NTW, pw, cw, sw);
    } else {
        syrk_lds_body<NB, NW, (NTW > 1 ? NTW - 1 : 1), FULLK, NT>(lds, wb, K, nstage, wv,
                                                                  BIG * NTW + (wv - BIG) * (NTW - 1), pw, cw, sw);
    }
}

// ---------------------------------------------------------------------------------
// Kernel 1L, statically specialised per wave (the default): same algorithm as
// fsnap_syrk_lds, but every wave runs a body instantiated for ITS tile list, so the
// operand offsets are instruction immediates (no per-MFMA address VALU) and an A operand
// shared by consecutive tiles of a row is read from LDS once (common-subexpression of the
// identical ds_read).  Measured in-kernel (tools/mfma_f64_peak.hip): the matrix pipe issues
// one fp64 MFMA per 64 cycles; two ds_read_b64 + wait per MFMA cost ~15 %, one fp64 VALU op
// per MFMA ~7 % — hence fewer LDS reads and fewer VALU ops per MFMA.
// ---------------------------------------------------------------------------------

namespace {

template <int NB>
__host__ __device__ constexpr int tile_p_of(int t) {
    int p = 0;
    while (p + 1 < NB && tri_index(p + 1, p + 1, NB) <= t) ++p;
    return p;
}
template <int NB>
__host__ __device__ constexpr int tile_q_of(int t) {
    return tile_p_of<NB>(t) + (t - tri_index(tile_p_of<NB>(t), tile_p_of<NB>(t), NB));
}

template <int NB, int NW>
struct LdsPlan {
    static constexpr int NTILE = NB * (NB + 1) / 2;
    static constexpr int NTW = (NTILE + NW - 1) / NW;
    static constexpr int BIG = NTILE - (NTW - 1) * NW;
    static constexpr int ntm(int wv) { return wv < BIG ? NTW : NTW - 1; }
    static constexpr int t0(int wv) { return wv < BIG ? wv * NTW : BIG * NTW + (wv - BIG) * (NTW - 1); }
};

template <int P, int Q>
__device__ __forceinline__ void tile_mfma(d4& acc, const double* src) {
    const double va = src[P * 64];
    const double vb = src[Q * 64];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va, vb, acc, 0, 0, 0);
}

template <int NB, int T0, int NTM, int... U>
__device__ __forceinline__ void chunk_tiles(d4 (&acc)[NTM], const double* src, std::integer_sequence<int, U...>) {
    (tile_mfma<tile_p_of<NB>(T0 + U), tile_q_of<NB>(T0 + U)>(acc[U], src), ...);
}

// split form for operand prefetch: LDS reads of one chunk, then (later) its MFMAs
template <int NB, int T0, int NTM, int... U>
__device__ __forceinline__ void chunk_load(double (&va)[NTM], double (&vb)[NTM], const double* src,
                                           std::integer_sequence<int, U...>) {
    ((va[U] = src[tile_p_of<NB>(T0 + U) * 64], vb[U] = src[tile_q_of<NB>(T0 + U) * 64]), ...);
}
template <int NTM, int... U>
__device__ __forceinline__ void chunk_mfma(d4 (&acc)[NTM], const double (&va)[NTM], const double (&vb)[NTM],
                                           std::integer_sequence<int, U...>) {
    ((acc[U] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[U], vb[U], acc[U], 0, 0, 0)), ...);
}

template <int NB, int NW, int WV, bool FULLK, bool NT, int ABL>
__device__ __forceinline__ void syrk_lds_static_body(double* lds, const WaveBufs& wb, int K, unsigned nstage,
                                                     double* __restrict__ pw, double* __restrict__ cw,
                                                     double* __restrict__ sw) {
    using Plan = LdsPlan<NB, NW>;
    constexpr int NTILE = Plan::NTILE;
    constexpr int NTM = Plan::ntm(WV) > 0 ? Plan::ntm(WV) : 1;
    constexpr bool HAS_TILES = Plan::ntm(WV) > 0;
    constexpr int T0 = Plan::t0(WV);
    constexpr int STAGE_DOUBLES = NW * NB * 64;
    const int lane = threadIdx.x & 63, e = lane & 15, kr = lane >> 4;

    d4 acc[NTM];
#pragma unroll
    for (int u = 0; u < NTM; ++u) acc[u] = d4{0.0, 0.0, 0.0, 0.0};
    double cacc[NB];
#pragma unroll
    for (int p = 0; p < NB; ++p) cacc[p] = 0.0;
    double bb = 0.0, sbw = 0.0, cnt = 0.0;

    ChunkRaw<NB> raw;
    ChunkRegs<NB> cr;
    // ABL != 0: diagnostic ablations (wrong results, timing only; option "ablate"):
    //   1 = no HBM loads in the stage loop, 2 = no weighting / c VALU work in park,
    //   3 = no barriers in the stage loop, 4 = MFMA operands not re-read from LDS
    auto park = [&](int buf) {
        if (ABL == 2) {
#pragma unroll
            for (int j = 0; j < NB / 2; ++j) {
                const d2 x = __builtin_bit_cast(d2, raw.pr[j]);
                cr.v[2 * j] = x[0];
                cr.v[2 * j + 1] = x[1];
            }
        } else {
            finish_chunk<NB, FULLK>(cr, raw, K, e);
            valu_c_chunk<NB>(cacc, cr);
            valu_s_chunk<NB>(bb, sbw, cnt, cr);
        }
        double* dst = lds + buf * STAGE_DOUBLES + (WV * NB) * 64 + lane;
#pragma unroll
        for (int bq = 0; bq < NB; ++bq) dst[bq * 64] = cr.v[bq];
    };

    // co-resident workgroups differ in blockIdx / (number of CUs); gridDim / 2 (or / 3) separates the layers
    const unsigned prio_phase = (ABL == 8) ? (unsigned)(blockIdx.x >= (gridDim.x + 1) / 2) : 0u;
    if (nstage > 0) {
        issue_chunk<NB, NT>(raw, wb, (unsigned)WV, kr);
        park(0);
        issue_chunk<NB, NT>(raw, wb, (unsigned)(NW + WV), kr);
        __syncthreads();
#if defined(FSNAP_TRACE) && FSNAP_TRACE >= 2
        unsigned long long tr_mfma = 0, tr_park = 0, tr_bar = 0;
#endif
        for (unsigned s = 0; s < nstage; ++s) {
#if defined(FSNAP_TRACE) && FSNAP_TRACE >= 2
            FSNAP_TRACE_CLK(tr0);
#endif
            const double* src = lds + ((ABL == 4) ? 0 : (s & 1)) * STAGE_DOUBLES + lane;
            // ABL == 6: the younger wave of each SIMD (WV >= NW/2; the SIMD serves its older wave
            // first) parks the next stage's chunk BEFORE its MFMA phase, i.e. while the older wave
            // owns the matrix pipe, instead of after it on the stage's critical path
            constexpr bool EARLY = (ABL == 6) && (WV >= NW / 2);
            if (EARLY) {
                park((s + 1) & 1);
                issue_chunk<NB, NT>(raw, wb, (s + 2) * NW + WV, kr);
            }
            // ABL 7 / 8: wave priorities.  The SIMD arbiter serves the highest s_setprio level first and
            // only then the oldest wave; a wave in its MFMA phase outranks waves that are parking /
            // issuing loads, so their VALU / LDS / VMEM instructions fill the 60 idle issue cycles
            // between two MFMAs instead of delaying one.  ABL 8 additionally alternates, stage by
            // stage, which of the co-resident workgroups wins ties (fair progress, no tail in which
            // the younger workgroup runs alone).
            if (ABL == 7) __builtin_amdgcn_s_setprio(2);
            if (ABL == 8) {
                if ((s ^ prio_phase) & 1u) __builtin_amdgcn_s_setprio(3);
                else __builtin_amdgcn_s_setprio(2);
            }
            if (ABL == 9 || ABL == 10) {
                // Interleaved park: the weighting / mask / c work of this wave's next chunk is cut into NB
                // per-block pieces that are issued BETWEEN the MFMA groups of the stage (a wave issues in
                // order: VALU / LDS-write instructions placed after the last MFMA of a stage run with the
                // matrix pipe idle, placed between MFMAs they run in the shadow of the 64-cycle MFMA).
                const bool keep = (raw.mk != 0);
                const double wv = __builtin_bit_cast(double, raw.wv);
                const double wbv = keep ? wv * __builtin_bit_cast(double, raw.bv) : 0.0;
                double* dst = lds + ((s + 1) & 1) * STAGE_DOUBLES + (WV * NB) * 64 + lane;
                // ABL == 10: additionally the LDS operands of chunk c + 1 are requested before the MFMAs of chunk c
                constexpr auto seq = std::make_integer_sequence<int, NTM>{};
                double oa[2][NTM], ob[2][NTM];
                if (ABL == 10 && HAS_TILES) chunk_load<NB, T0, NTM>(oa[0], ob[0], src, seq);
#pragma unroll
                for (int c = 0; c < NW; ++c) {
                    if (ABL == 10) {
                        if (HAS_TILES) {
                            if (c + 1 < NW) chunk_load<NB, T0, NTM>(oa[(c + 1) & 1], ob[(c + 1) & 1], src + (c + 1) * NB * 64, seq);
                            chunk_mfma<NTM>(acc, oa[c & 1], ob[c & 1], seq);
                        }
                    } else if (HAS_TILES) {
                        chunk_tiles<NB, T0, NTM>(acc, src + c * NB * 64, seq);
                    }
#pragma unroll
                    for (int j = c * NB / NW; j < (c + 1) * NB / NW; ++j) {
                        double x;
                        bool kj = keep;
                        if ((NB & 1) && j == NB - 1) {
                            x = __builtin_bit_cast(double, raw.tail);
                            if (!FULLK) kj = kj && (16 * (NB - 1) + e < K);
                        } else {
                            x = __builtin_bit_cast(d2, raw.pr[j >> 1])[j & 1];
                            if (!FULLK) kj = kj && (32 * (j >> 1) + 2 * e + (j & 1) < K);
                        }
                        const double vj = kj ? wv * x : 0.0;
                        cacc[j] = __builtin_fma(vj, wbv, cacc[j]);
                        dst[j * 64] = vj;
                    }
                    if (c == NW - 1) {
                        bb = __builtin_fma(wbv, wbv, bb);
                        sbw += wbv;
                        cnt += keep ? 1.0 : 0.0;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                issue_chunk<NB, NT>(raw, wb, (s + 2) * NW + WV, kr);
                __syncthreads();
                continue;
            }
            if (HAS_TILES) {
                if (ABL == 5) {
                    // operand prefetch: the LDS reads of chunk c + 1 are issued BEFORE the MFMAs of chunk c
                    // (double-buffered operand registers; sched_barriers pin the order), so that the matrix
                    // pipe never waits on lgkmcnt inside a stage
                    constexpr auto seq = std::make_integer_sequence<int, NTM>{};
                    double oa[2][NTM], ob[2][NTM];
                    chunk_load<NB, T0, NTM>(oa[0], ob[0], src, seq);
#pragma unroll
                    for (int c = 0; c < NW; ++c) {
                        if (c + 1 < NW) chunk_load<NB, T0, NTM>(oa[(c + 1) & 1], ob[(c + 1) & 1], src + (c + 1) * NB * 64, seq);
                        __builtin_amdgcn_sched_barrier(0);
                        chunk_mfma<NTM>(acc, oa[c & 1], ob[c & 1], seq);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < NW; ++c)
                        chunk_tiles<NB, T0, NTM>(acc, src + ((ABL == 4) ? 0 : c) * NB * 64, std::make_integer_sequence<int, NTM>{});
                }
            }
#if defined(FSNAP_TRACE) && FSNAP_TRACE >= 2
            FSNAP_TRACE_CLK(tr1);
#endif
            if (ABL == 7 || ABL == 8) __builtin_amdgcn_s_setprio(0);
            if (!EARLY) {
                park((s + 1) & 1);
                if (ABL != 1) issue_chunk<NB, NT>(raw, wb, (s + 2) * NW + WV, kr);
            }
#if defined(FSNAP_TRACE) && FSNAP_TRACE >= 2
            FSNAP_TRACE_CLK(tr2);
#endif
            if (ABL != 3) __syncthreads();
#if defined(FSNAP_TRACE) && FSNAP_TRACE >= 2
            FSNAP_TRACE_CLK(tr3);
            tr_mfma += tr1 - tr0;
            tr_park += tr2 - tr1;
            tr_bar += tr3 - tr2;
#endif
        }
#if defined(FSNAP_TRACE) && FSNAP_TRACE >= 2
        if (lane == 0 && blockIdx.x < 4096) {
            unsigned long long* o = fsnap_trace_wave + ((size_t)blockIdx.x * 16 + WV) * 4;
            o[0] = tr_mfma;
            o[1] = tr_park;
            o[2] = tr_bar;
            o[3] = nstage;
        }
#endif
    }

    if (HAS_TILES) {
#pragma unroll
        for (int u = 0; u < NTM; ++u) {
            const int t = T0 + u;
#pragma unroll
            for (int i = 0; i < 4; ++i) pw[(t * 4 + i) * 64 + lane] = acc[u][i];
        }
    }
    constexpr int CS = NB * 16 + 4;
    double* fold = lds + WV * CS;
#pragma unroll
    for (int p = 0; p < NB; ++p) {
        double sm = xlane_sum_rows(cacc[p]);
        if (kr == 0) fold[p * 16 + e] = sm;
    }
    double sb = xlane_sum_rows(bb), ss = xlane_sum_rows(sbw), sc = xlane_sum_rows(cnt);
    if (lane == 0) {
        fold[NB * 16 + 0] = sb;
        fold[NB * 16 + 1] = ss;
        fold[NB * 16 + 2] = sc;
        fold[NB * 16 + 3] = 0.0;
    }
    __syncthreads();
    if (WV == 0) {
        for (int j = lane; j < CS; j += 64) {
            double tot = 0.0;
#pragma unroll
            for (int k = 0; k < NW; ++k) tot += lds[k * CS + j];
            if (j < NB * 16) cw[j] = tot;
            else sw[j - NB * 16] = tot;
        }
    }
}

template <int NB, int NW, bool FULLK, bool NT, int ABL, int... W>
__device__ __forceinline__ void syrk_lds_dispatch(int wv, double* lds, const WaveBufs& wb, int K, unsigned nstage,
                                                  double* pw, double* cw, double* sw, std::integer_sequence<int, W...>) {
    // every wave of the workgroup takes exactly one branch; all bodies execute the same barriers
    ((wv == W ? (syrk_lds_static_body<NB, NW, W, FULLK, NT, ABL>(lds, wb, K, nstage, pw, cw, sw), 0) : 0), ...);
}

}  // namespace

template <int NB, int NW, bool FULLK, bool NT, int ABL = 0>
__global__ __launch_bounds__(64 * NW, (NW == 2 ? 2 : NW == 4 ? 3 : 4)) void fsnap_syrk_lds_static(const double* __restrict__ A, int64_t lda,
                                                                    const double* __restrict__ b,
                                                                    const double* __restrict__ w,
                                                                    const unsigned char* __restrict__ mask, int64_t m,
                                                                    int K, int64_t chunks_per_wg,
                                                                    double* __restrict__ part,
                                                                    double* __restrict__ cpart,
                                                                    double* __restrict__ spart) {
    constexpr int NTILE = NB * (NB + 1) / 2;
    __shared__ double lds[2 * NW * NB * 64];
#ifdef FSNAP_TRACE
    const unsigned long long trace_t0 = wall_clock64();
    const unsigned long long trace_c0 = __builtin_readcyclecounter();
#endif
    const int lane = threadIdx.x & 63, e = lane & 15, kr = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t wg = blockIdx.x;
    const int64_t nchunks = (m + 3) >> 2;
    int64_t c0 = wg * chunks_per_wg;
    int64_t c1 = c0 + chunks_per_wg;
    if (c1 > nchunks) c1 = nchunks;
    if (c0 > c1) c0 = c1;
    const int64_t row0 = c0 << 2;
    int64_t row1 = c1 << 2;
    if (row1 > m) row1 = m;
    const int64_t nrow = row1 > row0 ? row1 - row0 : 0;
    WaveBufs wb;
    wb.A = make_rsrc(A + row0 * lda, (unsigned)(nrow * lda * 8 + (nrow ? 16 : 0)));
    wb.b = make_rsrc(b + row0, (unsigned)(nrow * 8));
    wb.w = make_rsrc(w + row0, (unsigned)(nrow * 8));
    wb.mask = make_rsrc(mask + row0, (unsigned)nrow);
    wb.voffA = (unsigned)((kr * lda + 2 * e) * 8);
    wb.voffT = (unsigned)((kr * lda + 16 * (NB - 1) + e) * 8);
    wb.voffR = (unsigned)(kr * 8);
    wb.chunk_bytes = (unsigned)(lda * 32);
    const unsigned ncl = (unsigned)(c1 - c0);
    const unsigned nstage = (ncl + NW - 1) / NW;
    double* pw = part + wg * (int64_t)(NTILE * 256);
    double* cw = cpart + wg * (int64_t)(NB * 16);
    double* sw = spart + wg * 4;
    syrk_lds_dispatch<NB, NW, FULLK, NT, ABL>(wv, lds, wb, K, nstage, pw, cw, sw, std::make_integer_sequence<int, NW>{});
#ifdef FSNAP_TRACE
    if (threadIdx.x == 0 && wg < 4096) {
        fsnap_trace_buf[wg * 8 + 0] = trace_t0;
        fsnap_trace_buf[wg * 8 + 1] = wall_clock64();
        fsnap_trace_buf[wg * 8 + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        fsnap_trace_buf[wg * 8 + 3] = __builtin_amdgcn_s_getreg((3 << 11) | 20);    // HW_REG_XCC_ID[3:0]
        fsnap_trace_buf[wg * 8 + 4] = __builtin_readcyclecounter() - trace_c0;      // shader-clock cycles of the workgroup's life
    }
#endif
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
// c and the scalars ride on the diagonal pairs / pair 0.
// Partials: partT[split*npairs + pair][16][4][64] | cpartT[(split*NSB + I)*4 + wave][4][16]
//           | spartT[split*4 + wave][4]
// ---------------------------------------------------------------------------------
namespace {

template <bool NT>
struct RawT {
    u4 pi[2], pj[2];
    u2 bv, wv;
    unsigned char mk;
};

template <bool DIAG, bool NT>
__device__ __forceinline__ void issue_chunk_t(RawT<NT>& r, const WaveBufs& wb, unsigned voffI, unsigned voffJ,
                                              unsigned cl, int kr) {
    const unsigned soff = cl * wb.chunk_bytes;
    constexpr int AUX = NT ? 2 : 0;
    r.pi[0] = __builtin_amdgcn_raw_buffer_load_b128(wb.A, voffI, soff, AUX);
    r.pi[1] = __builtin_amdgcn_raw_buffer_load_b128(wb.A, voffI + 256u, soff, AUX);
    if (!DIAG) {
        r.pj[0] = __builtin_amdgcn_raw_buffer_load_b128(wb.A, voffJ, soff, AUX);
        r.pj[1] = __builtin_amdgcn_raw_buffer_load_b128(wb.A, voffJ + 256u, soff, AUX);
    }
    r.bv = __builtin_amdgcn_raw_buffer_load_b64(wb.b, wb.voffR, cl * 32u, 0);
    r.wv = __builtin_amdgcn_raw_buffer_load_b64(wb.w, wb.voffR, cl * 32u, 0);
    r.mk = __builtin_amdgcn_raw_buffer_load_b8(wb.mask, (unsigned)kr, cl * 4u, 0);
}

__device__ __forceinline__ void weight4(double (&v)[4], const u4 (&p)[2], double wv, bool keep, int col0, int K, int e) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const d2 x = __builtin_bit_cast(d2, p[j]);
        const bool k0 = keep && (col0 + 32 * j + 2 * e < K);
        const bool k1 = keep && (col0 + 32 * j + 2 * e + 1 < K);
        v[2 * j] = k0 ? wv * x[0] : 0.0;
        v[2 * j + 1] = k1 ? wv * x[1] : 0.0;
    }
}

template <bool DIAG, bool NT>
__device__ __forceinline__ void syrk_tiled_body(const double* __restrict__ A, int64_t lda,
                                                const double* __restrict__ b, const double* __restrict__ w,
                                                const unsigned char* __restrict__ mask, int64_t m, int K, int I, int J,
                                                int64_t c0, int64_t c1, int wv_in_wg, bool do_c, bool do_s,
                                                double* lds, double* __restrict__ pw, double* __restrict__ cw,
                                                double* __restrict__ sw) {
    constexpr int NTW = 16;
    const int lane = threadIdx.x & 63, e = lane & 15, kr = lane >> 4;
    d4 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
    double cacc[4] = {0.0, 0.0, 0.0, 0.0};
    double bb = 0.0, sbw = 0.0, cnt = 0.0;

    const int64_t row0 = c0 << 2;
    int64_t row1 = c1 << 2;
    if (row1 > m) row1 = m;
    const int64_t nrow = row1 > row0 ? row1 - row0 : 0;
    WaveBufs wb;
    wb.A = make_rsrc(A + row0 * lda, (unsigned)(nrow * lda * 8 + (nrow ? 16 : 0)));
    wb.b = make_rsrc(b + row0, (unsigned)(nrow * 8));
    wb.w = make_rsrc(w + row0, (unsigned)(nrow * 8));
    wb.mask = make_rsrc(mask + row0, (unsigned)nrow);
    wb.voffA = 0;
    wb.voffT = 0;
    wb.voffR = (unsigned)(kr * 8);
    wb.chunk_bytes = (unsigned)(lda * 32);
    const unsigned voffI = (unsigned)((kr * lda + 64 * I + 2 * e) * 8);
    const unsigned voffJ = (unsigned)((kr * lda + 64 * J + 2 * e) * 8);
    const unsigned ncl = (unsigned)(c1 > c0 ? c1 - c0 : 0);

    RawT<NT> r0, r1, r2;
    double vI[4], vJ[4];

#define FSNAP_STAGE_T(R, OFF)                                                                   \
    {                                                                                           \
        const bool keep = (R.mk != 0);                                                          \
        const double wgt = __builtin_bit_cast(double, R.wv);                                    \
        weight4(vI, R.pi, wgt, keep, 64 * I, K, e);                                             \
        if (!DIAG) weight4(vJ, R.pj, wgt, keep, 64 * J, K, e);                                  \
        const double wbv = keep ? wgt * __builtin_bit_cast(double, R.bv) : 0.0;                 \
        const double one = keep ? 1.0 : 0.0;                                                    \
        issue_chunk_t<DIAG, NT>(R, wb, voffI, voffJ, cl + (OFF) + 3, kr);                       \
        _Pragma("unroll") for (int p = 0; p < 4; ++p) {                                         \
            _Pragma("unroll") for (int q = (DIAG ? p : 0); q < 4; ++q) {                        \
                acc[p * 4 + q] = __builtin_amdgcn_mfma_f64_16x16x4f64(vI[p], DIAG ? vI[q] : vJ[q], acc[p * 4 + q], 0, 0, 0); \
            }                                                                                   \
        }                                                                                       \
        if (DIAG) {                                                                             \
            if (do_c) {                                                                         \
                _Pragma("unroll") for (int p = 0; p < 4; ++p) cacc[p] = __builtin_fma(vI[p], wbv, cacc[p]); \
            }                                                                                   \
            if (do_s) {                                                                         \
                bb = __builtin_fma(wbv, wbv, bb);                                               \
                sbw += wbv;                                                                     \
                cnt += one;                                                                     \
            }                                                                                   \
        }                                                                                       \
    }
    if (ncl > 0) {
        issue_chunk_t<DIAG, NT>(r0, wb, voffI, voffJ, 0, kr);
        issue_chunk_t<DIAG, NT>(r1, wb, voffI, voffJ, 1, kr);
        issue_chunk_t<DIAG, NT>(r2, wb, voffI, voffJ, 2, kr);
        for (unsigned cl = 0; cl < ncl; cl += 3) {
            FSNAP_STAGE_T(r0, 0)
            FSNAP_STAGE_T(r1, 1)
            FSNAP_STAGE_T(r2, 2)
        }
    }
#undef FSNAP_STAGE_T

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
    if (DIAG && do_c) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            double s = xlane_sum_rows(cacc[p]);
            if (kr == 0) cw[p * 16 + e] = s;
        }
    }
    if (DIAG && do_s) {
        double sb = xlane_sum_rows(bb), ss = xlane_sum_rows(sbw), sc = xlane_sum_rows(cnt);
        if (lane == 0) {
            sw[0] = sb;
            sw[1] = ss;
            sw[2] = sc;
            sw[3] = 0.0;
        }
    }
}

}  // namespace

template <bool NT>
__global__ __launch_bounds__(256, 2) void fsnap_syrk_tiled(const double* __restrict__ A, int64_t lda,
                                                           const double* __restrict__ b,
                                                           const double* __restrict__ w,
                                                           const unsigned char* __restrict__ mask, int64_t m, int K,
                                                           int NSB, int npairs, int64_t chunks_per_split,
                                                           int nitems, int xcd_map,
                                                           double* __restrict__ part, double* __restrict__ cpart,
                                                           double* __restrict__ spart) {
    __shared__ double lds[2 * 16 * 256];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // Work item = (split, pair), pair fastest.  Workgroups are dealt round-robin to the 8 XCDs
    // (blockIdx % 8); with xcd_map every XCD gets a CONTIGUOUS range of items, so the
    // workgroups that share an L2 sweep the same rows (different column pairs) together and a
    // row slab is fetched from HBM once per XCD instead of once per pair.
    unsigned item = blockIdx.x;
    if (xcd_map) {
        const unsigned per = ((unsigned)nitems + 7u) >> 3;
        const unsigned slot = blockIdx.x >> 3;
        item = (blockIdx.x & 7u) * per + slot;
        if (slot >= per || item >= (unsigned)nitems) return;
    }
    const int pair = (int)(item % (unsigned)npairs);
    const int split = (int)(item / (unsigned)npairs);
    // decode pair -> (I, J), I <= J, row-major packed triangle over NSB superblocks
    int I = 0, rem = pair;
    while (rem >= NSB - I) {
        rem -= NSB - I;
        ++I;
    }
    const int J = I + rem;
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
    double* sw = spart + ((int64_t)split * 4 + wv) * 4;
    if (I == J) {
        syrk_tiled_body<true, NT>(A, lda, b, w, mask, m, K, I, J, c0, c1, wv, true, pair == 0, lds, pw, cw, sw);
    } else {
        syrk_tiled_body<false, NT>(A, lda, b, w, mask, m, K, I, J, c0, c1, wv, false, false, lds, pw, cw, sw);
    }
}

// Reduction of the tiled partials into the packed buffer (same output as kernel 2).
// Element space: npairs*16*256 G elements (nsplit partials each), NSB*64 c elements and
// 4 scalars (nsplit*4 partials each).  1024 threads = 64 elements x 16 partial slices.
__global__ __launch_bounds__(1024) void fsnap_reduce_tiled(const double* __restrict__ part,
                                                           const double* __restrict__ cpart,
                                                           const double* __restrict__ spart, int nsplit, int NSB,
                                                           int npairs, int K, double* __restrict__ out, int accumulate) {
    __shared__ double red[1024];
    const int64_t nG = (int64_t)npairs * 4096;
    const int nC = NSB * 64, nS = 4;
    const int tid = threadIdx.x, g = tid >> 6, l = tid & 63;
    const int64_t idx = (int64_t)blockIdx.x * 64 + l;
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
        src = spart + (idx - nG - nC);
        stride = 4;
        np = nsplit * 4;
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
// Kernel 3: stand-alone wavefront row weighting (svd.py:46 / ridge.py:39).
//   aw[i,:] = w[i]*A[i,:], bw[i] = w[i]*b[i] for every row; masked rows are written
//   as zeros (row compaction is the host shim's business, see fsnap_weight_rows()).
// One wave per row-slab, 16-byte vector accesses, grid-stride.  HBM-bound:
// 16K + 24 bytes per row.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fsnap_weight_rows_k(const double* __restrict__ A, int64_t lda,
                                                           const double* __restrict__ b,
                                                           const double* __restrict__ w,
                                                           const unsigned char* __restrict__ mask, int64_t m,
                                                           int K, double* __restrict__ aw, int64_t ldaw,
                                                           double* __restrict__ bw) {
    // One wave handles 4 consecutive rows per iteration (4 independent 16-byte loads per lane in
    // flight before the first store: memory-level parallelism for the HBM stream).
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * 4;
    const bool vec2 = ((K & 1) == 0) && ((lda & 1) == 0) && ((ldaw & 1) == 0);
    for (int64_t row0 = wave * 4; row0 < m; row0 += nwave * 4) {
        double wv[4];
        bool keep[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = row0 + r;
            const bool in = row < m;
            keep[r] = in && (mask[in ? row : 0] != 0);
            wv[r] = in ? w[row] : 0.0;
        }
        if (vec2) {
            for (int c = 2 * lane; c < K; c += 128) {
                d2u x[4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (row0 + r < m) x[r] = __builtin_nontemporal_load(reinterpret_cast<const d2u*>(A + (row0 + r) * lda + c));
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (row0 + r < m) {
                        d2u y;
                        y[0] = keep[r] ? wv[r] * x[r][0] : 0.0;
                        y[1] = keep[r] ? wv[r] * x[r][1] : 0.0;
                        __builtin_nontemporal_store(y, reinterpret_cast<d2u*>(aw + (row0 + r) * ldaw + c));
                    }
                }
            }
        } else {
            for (int c = lane; c < K; c += 64) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (row0 + r < m) aw[(row0 + r) * ldaw + c] = keep[r] ? wv[r] * A[(row0 + r) * lda + c] : 0.0;
            }
        }
        if (lane < 4 && row0 + lane < m) {
            const int64_t row = row0 + lane;
            const bool kp = mask[row] != 0;
            bw[row] = kp ? w[row] * b[row] : 0.0;
        }
    }
}

// ---------------------------------------------------------------------------------
// Kernel 4: preds = A @ beta (solver.py:377) and, optionally, per-workgroup partial
// sums of the weighted squared residual sum_i mask_i (w_i (b_i - preds_i))^2
// (the SSE that sklearn's ARD loop recomputes each iteration, _bayes.py `rmse_`).
// 16 lanes per row (4 rows per wave pass), beta staged once in LDS.  HBM-bound.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fsnap_gemv_rows_k(const double* __restrict__ A, int64_t lda,
                                                         const double* __restrict__ beta, int64_t m, int K,
                                                         double* __restrict__ preds,
                                                         const double* __restrict__ b,
                                                         const double* __restrict__ w,
                                                         const unsigned char* __restrict__ mask,
                                                         double* __restrict__ sse_part,
                                                         double* __restrict__ uout) {
    // uout (optional): u_i = mask_i * w_i^2 * (b_i - a_i . beta), the row weights of the
    // refinement right-hand side  s = (wA)^T (wb - wA beta) = A^T u   (kernel 7)
    extern __shared__ __attribute__((aligned(16))) double sbeta[];
    for (int c = threadIdx.x; c < K; c += 256) sbeta[c] = beta[c];
    __syncthreads();
    const int lane = threadIdx.x & 63, e = lane & 15, kr = lane >> 4;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * 4;
    double sse = 0.0;
    for (int64_t r0 = wave * 4; r0 < m; r0 += nwave * 4) {
        const int64_t row = r0 + kr;
        double s = 0.0;
        if (row < m) {
            const double* src = A + row * lda;
            if (((K | lda) & 1) == 0) {   // 16-byte loads: two adjacent columns per lane, two accumulators
                double s1 = 0.0;
                for (int c = 2 * e; c < K; c += 32) {
                    const d2u x = __builtin_nontemporal_load(reinterpret_cast<const d2u*>(src + c));
                    s = __builtin_fma(x[0], sbeta[c], s);
                    s1 = __builtin_fma(x[1], sbeta[c + 1], s1);
                }
                s += s1;
            } else {
                for (int c = e; c < K; c += 16) s = __builtin_fma(src[c], sbeta[c], s);
            }
        }
        // reduce over the 16 lanes of the row group
        s += __shfl_xor(s, 8, 64);
        s += __shfl_xor(s, 4, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 1, 64);
        if (row < m && e == 0) {
            if (preds) preds[row] = s;
            if (sse_part || uout) {
                const bool keep = mask ? (mask[row] != 0) : true;
                const double wr = w[row];
                const double rr = keep ? wr * (b[row] - s) : 0.0;
                if (sse_part) sse = __builtin_fma(rr, rr, sse);
                if (uout) uout[row] = keep ? wr * rr : 0.0;
            }
        }
    }
    if (sse_part) {
        __shared__ double wsum[4];
        sse += __shfl_xor(sse, 16, 64);
        sse += __shfl_xor(sse, 32, 64);
        if (lane == 0) wsum[threadIdx.x >> 6] = sse;
        __syncthreads();
        if (threadIdx.x == 0) sse_part[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    }
}

// ---------------------------------------------------------------------------------
// Kernel 5: post-LAMMPS assembly — the `_collect_lammps` transform
// (fitsnap3lib/calculators/lammps_snap.py:391-556, lammps_pace.py:369-509) for a batch of
// configurations: raw `compute snap|pace` rows -> rows of A, b, w, written straight into the
// resident HBM arrays.  One wave per output row, lanes stride the K output columns.
//   raw      : row-major raw rows, leading dimension raw_ld = ncoeff*ntypes + 1; the last
//              column (icolref) is the reference-potential contribution
//   per output row r (SoA plan): src_row[r] raw row, kind[r], d[r], truth[r], weight[r],
//              frac[r] (index of the per-type atom fractions of its configuration, or -1)
//   kind 0 energy       : A = x / d              b = (truth - ref) / d   w = weight   (d = N)
//   kind 1 force        : A = x                  b = truth - ref         w = weight
//   kind 2 virial       : A = (1.6021765e6 x)/d  b = truth - ref         w = weight   (d = volume)
//   kind 3 per-atom-energy rows after the first (bikflag): A = x / d, b = 0, w = 0
//   column k -> type t = k / (ncoeff + off), j = k % (ncoeff + off); with off = 1
//   (bzeroflag = 0) column j = 0 is the per-type offset column: atom fraction of type t on
//   energy rows, 0 elsewhere; every column is multiplied by blank2J[k].
// The arithmetic order is the reference's (divide, not multiply by a reciprocal), so
// rows are bit-identical to the numpy path.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fsnap_assemble_k(const double* __restrict__ raw, int64_t raw_ld,
                                                        int64_t nrows, const int64_t* __restrict__ src_row,
                                                        const int* __restrict__ kind, const int* __restrict__ frac,
                                                        const double* __restrict__ dval,
                                                        const double* __restrict__ truth,
                                                        const double* __restrict__ weight,
                                                        const double* __restrict__ fractions,
                                                        const double* __restrict__ blank2J, int ntypes, int ncoeff,
                                                        int off, double* __restrict__ A, int64_t lda,
                                                        double* __restrict__ b, double* __restrict__ w) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * 4;
    const int stride = ncoeff + off;
    const int K = ntypes * stride;
    const int icolref = ntypes * ncoeff;
    for (int64_t r = wave; r < nrows; r += nwave) {
        const double* src = raw + src_row[r] * raw_ld;
        const int kd = kind[r];
        const double d = dval[r];
        const int fr = frac[r];
        double* dst = A + r * lda;
        for (int k = lane; k < K; k += 64) {
            const int t = k / stride, j = k - t * stride;
            double v;
            if (off && j == 0) {
                v = (kd == 0 && fr >= 0) ? fractions[(int64_t)fr * ntypes + t] : 0.0;
            } else {
                const double x = src[t * ncoeff + (j - off)];
                v = (kd == 1) ? x : (kd == 2) ? (1.6021765e6 * x) / d : x / d;
            }
            dst[k] = v * blank2J[k];
        }
        if (lane == 0) {
            const double ref = src[icolref];
            double bv, wv = weight[r];
            if (kd == 0) bv = (truth[r] - ref) / d;
            else if (kd == 3) {
                bv = 0.0;
                wv = 0.0;
            } else bv = truth[r] - ref;
            b[r] = bv;
            w[r] = wv;
        }
    }
}

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
// Kernel 7: s = A^T u  (transposed streaming GEMV, HBM-bound) — with u from kernel 4 this is
// the right-hand side of one step of iterative refinement of the least-squares solution
// ("corrected semi-normal equations": G delta = (wA)^T (wb - wA beta), beta += delta), which
// takes the error of the normal-equation solve from ~kappa^2 eps back to ~kappa eps — what
// keeps the GPU path within 1e-6 of the reference's lstsq (svd.py:54) on ill-conditioned A.
// Workgroup = row range; wave v takes rows v, v+4, ...; lane l owns columns 2l, 2l+1 (+128 j).
// Per-workgroup partial vectors are written to spart2[wg][K] and summed in fixed order.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fsnap_gemvT_rows_k(const double* __restrict__ A, int64_t lda,
                                                          const double* __restrict__ u, int64_t m, int K,
                                                          int64_t rows_per_wg, double* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) double sacc[];   // 4 waves x Kpad
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int Kpad = (K + 1) & ~1;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
    int64_t r1 = r0 + rows_per_wg;
    if (r1 > m) r1 = m;
    const bool vec2 = ((K | lda) & 1) == 0;
    for (int c0 = 0; c0 < K; c0 += 128) {
        const int c = c0 + 2 * lane;
        double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
        if (c < K) {
            int64_t row = r0 + wv;
            for (; row + 4 < r1; row += 8) {   // two rows in flight per wave
                const double u0 = u[row], u1 = u[row + 4];
                double x0, x1, y0, y1;
                if (vec2) {
                    const d2u x = *reinterpret_cast<const d2u*>(A + row * lda + c);
                    const d2u y = *reinterpret_cast<const d2u*>(A + (row + 4) * lda + c);
                    x0 = x[0]; x1 = x[1]; y0 = y[0]; y1 = y[1];
                } else {
                    x0 = A[row * lda + c]; x1 = (c + 1 < K) ? A[row * lda + c + 1] : 0.0;
                    y0 = A[(row + 4) * lda + c]; y1 = (c + 1 < K) ? A[(row + 4) * lda + c + 1] : 0.0;
                }
                a0 = __builtin_fma(x0, u0, a0);
                a1 = __builtin_fma(x1, u0, a1);
                b0 = __builtin_fma(y0, u1, b0);
                b1 = __builtin_fma(y1, u1, b1);
            }
            for (; row < r1; row += 4) {
                const double u0 = u[row];
                const double x0 = A[row * lda + c];
                const double x1 = (c + 1 < K) ? A[row * lda + c + 1] : 0.0;
                a0 = __builtin_fma(x0, u0, a0);
                a1 = __builtin_fma(x1, u0, a1);
            }
            sacc[wv * Kpad + c] = a0 + b0;
            if (c + 1 < K) sacc[wv * Kpad + c + 1] = a1 + b1;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < K; c += 256)
        partial[(int64_t)blockIdx.x * K + c] = (sacc[c] + sacc[Kpad + c]) + (sacc[2 * Kpad + c] + sacc[3 * Kpad + c]);
}

__global__ __launch_bounds__(256) void fsnap_colsum_partials_k(const double* __restrict__ partial, int nparts, int K,
                                                               double* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= K) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int p = 0;
    for (; p + 3 < nparts; p += 4) {
        s0 += partial[(int64_t)p * K + c];
        s1 += partial[(int64_t)(p + 1) * K + c];
        s2 += partial[(int64_t)(p + 2) * K + c];
        s3 += partial[(int64_t)(p + 3) * K + c];
    }
    for (; p < nparts; ++p) s0 += partial[(int64_t)p * K + c];
    out[c] = (s0 + s1) + (s2 + s3);
}

// ---------------------------------------------------------------------------------
// host-side launchers (C++ linkage, used by fsnap_capi.cpp)
// ---------------------------------------------------------------------------------
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
// Kernel 9: grouped error statistics of Solver.error_analysis (solver.py:108-133, 391-429).
// Every row carries a category id (group x train/test x row type, built by the host shim); per category the
// reference needs  n, count_nonzero(w), mean|r|, sum r^2, sum (t - mean t)^2  and the same for w r, w t
// (r = truth - prediction).  The centred sums need the category means first, hence two passes:
//   pass 0:  [n, n_w, sum t, sum w t]                       (4 values per category)
//   pass 1:  [sum|r|, sum r^2, sum (t - mean)^2, sum|w r|, sum (w r)^2, sum (w t - wmean)^2]   (6 values)
// A workgroup accumulates its rows into an LDS table (ds_add_f64) and writes one partial table; the host sums the
// partial tables in a fixed order.  HBM-bound: 8 (t) + 8 (w) + 8 (pred) + 4 (cat) bytes per row and pass.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fsnap_error_stats_k(const double* __restrict__ truth,
                                                          const double* __restrict__ pred,
                                                          const double* __restrict__ wgt, const int* __restrict__ cat,
                                                          int64_t m, int ncat, int pass,
                                                          const double* __restrict__ means /* [ncat][2] */,
                                                          double* __restrict__ partial /* [grid][ncat][nv] */) {
    extern __shared__ double tab[];
    const int nv = pass == 0 ? 4 : 6;
    for (int i = threadIdx.x; i < ncat * nv; i += 256) tab[i] = 0.0;
    __syncthreads();
    for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < m; row += (int64_t)gridDim.x * 256) {
        const int c = cat[row];
        if (c < 0 || c >= ncat) continue;
        const double t = truth[row], w = wgt[row];
        double* e = tab + (size_t)c * nv;
        if (pass == 0) {
            atomicAdd(e + 0, 1.0);
            atomicAdd(e + 1, w != 0.0 ? 1.0 : 0.0);
            atomicAdd(e + 2, t);
            atomicAdd(e + 3, w * t);
        } else {
            const double r = t - pred[row], wr = w * r;
            const double dt = t - means[2 * c], dwt = w * t - means[2 * c + 1];
            atomicAdd(e + 0, fabs(r));
            atomicAdd(e + 1, r * r);
            atomicAdd(e + 2, dt * dt);
            atomicAdd(e + 3, fabs(wr));
            atomicAdd(e + 4, wr * wr);
            atomicAdd(e + 5, dwt * dwt);
        }
    }
    __syncthreads();
    double* out = partial + (size_t)blockIdx.x * ncat * nv;
    for (int i = threadIdx.x; i < ncat * nv; i += 256) out[i] = tab[i];
}

// Kernel 10: small device -> page-locked host copy done by a kernel (the copy engine's start-up latency, ~12 us on
// these boxes, is several times the transfer time of the 132 KB statistics)
__global__ __launch_bounds__(256) void fsnap_copy_to_host_k(const double* __restrict__ src, double* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

namespace fsnap {

int syrk_num_blocks(int K) { return (K + 15) / 16; }

template <int NB, int SPLIT>
static hipError_t launch_syrk_nb(const SyrkArgs& a, hipStream_t st) {
    constexpr int DEPTH = (SPLIT == 2 && NB >= 7) ? 2 : 3;
    dim3 grid((unsigned)a.nblocks), block(256 * SPLIT);
    const bool fullk = (a.K == 16 * NB);
#define FSNAP_LAUNCH(FK, NTL)                                                                              \
    hipLaunchKernelGGL((fsnap_syrk_wave<NB, SPLIT, DEPTH, FK, NTL>), grid, block, 0, st, a.A, a.lda, a.b, a.w, a.mask, \
                       a.m, a.K, a.chunks_per_wave, a.part, a.cpart, a.spart)
    if (fullk) {
        if (a.nontemporal) FSNAP_LAUNCH(true, true);
        else FSNAP_LAUNCH(true, false);
    } else {
        if (a.nontemporal) FSNAP_LAUNCH(false, true);
        else FSNAP_LAUNCH(false, false);
    }
#undef FSNAP_LAUNCH
    return hipGetLastError();
}

int syrk_default_split(int K) { return syrk_num_blocks(K) >= 6 ? 2 : 1; }

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

template <int NB, int NW>
static hipError_t launch_syrk_lds_nb(const SyrkArgs& a, hipStream_t st) {
    dim3 grid((unsigned)a.nblocks), block(64 * NW);
    const bool fullk = (a.K == 16 * NB);
#define FSNAP_LAUNCH(FK, NTL)                                                                                  \
    hipLaunchKernelGGL((fsnap_syrk_lds<NB, NW, FK, NTL>), grid, block, 0, st, a.A, a.lda, a.b, a.w, a.mask, a.m, \
                       a.K, a.chunks_per_wave, a.part, a.cpart, a.spart)
    if (fullk) {
        if (a.nontemporal) FSNAP_LAUNCH(true, true);
        else FSNAP_LAUNCH(true, false);
    } else {
        if (a.nontemporal) FSNAP_LAUNCH(false, true);
        else FSNAP_LAUNCH(false, false);
    }
#undef FSNAP_LAUNCH
    return hipGetLastError();
}

template <int NB, int NW>
static hipError_t launch_syrk_lds_static_nb(const SyrkArgs& a, hipStream_t st) {
    dim3 grid((unsigned)a.nblocks), block(64 * NW);
    const bool fullk = (a.K == 16 * NB);
#define FSNAP_ABL(N)                                                                                               \
    hipLaunchKernelGGL((fsnap_syrk_lds_static<8, 8, true, true, N>), grid, block, 0, st, a.A, a.lda, a.b, a.w, a.mask, \
                       a.m, a.K, a.chunks_per_wave, a.part, a.cpart, a.spart)
    if (NB == 8 && fullk && a.ablate == 5) {        // operand-prefetch variant (correct results; A/B)
        hipLaunchKernelGGL((fsnap_syrk_lds_static<8, NW, true, true, 5>), grid, block, 0, st, a.A, a.lda, a.b, a.w, a.mask,
                           a.m, a.K, a.chunks_per_wave, a.part, a.cpart, a.spart);
        return hipGetLastError();
    }
#define FSNAP_VARIANT(N)                                                                                           \
    if (NB == 8 && fullk && a.ablate == N) {                                                                       \
        hipLaunchKernelGGL((fsnap_syrk_lds_static<8, NW, true, true, N>), grid, block, 0, st, a.A, a.lda, a.b, a.w, \
                           a.mask, a.m, a.K, a.chunks_per_wave, a.part, a.cpart, a.spart);                         \
        return hipGetLastError();                                                                                  \
    }
    FSNAP_VARIANT(6)   // early park (correct results; A/B)
    FSNAP_VARIANT(7)   // wave priorities by phase
    FSNAP_VARIANT(8)   // wave priorities by phase + alternating tie-break between co-resident workgroups
    FSNAP_VARIANT(9)   // park work interleaved with the MFMA groups
    FSNAP_VARIANT(10)  // ... plus LDS operand prefetch one chunk ahead
#undef FSNAP_VARIANT
    if (NB == 8 && NW == 8 && fullk && a.ablate) {   // timing-only diagnostic variants (option "ablate")
        switch (a.ablate) {
            case 1: FSNAP_ABL(1); break;
            case 2: FSNAP_ABL(2); break;
            case 3: FSNAP_ABL(3); break;
            default: FSNAP_ABL(4); break;
        }
        return hipGetLastError();
    }
#undef FSNAP_ABL
    if (fullk)
        hipLaunchKernelGGL((fsnap_syrk_lds_static<NB, NW, true, true>), grid, block, 0, st, a.A, a.lda, a.b, a.w, a.mask,
                           a.m, a.K, a.chunks_per_wave, a.part, a.cpart, a.spart);
    else
        hipLaunchKernelGGL((fsnap_syrk_lds_static<NB, NW, false, true>), grid, block, 0, st, a.A, a.lda, a.b, a.w, a.mask,
                           a.m, a.K, a.chunks_per_wave, a.part, a.cpart, a.spart);
    return hipGetLastError();
}

// a.split = 8: statically specialised kernel 1L (default); a.split = -8: the generic
// (run-time tile table) variant kept for A/B; a.chunks_per_wave = chunks per workgroup
hipError_t launch_syrk_lds(const SyrkArgs& a, hipStream_t st) {
    const int nb = syrk_num_blocks(a.K);
    if (a.split == 16) {   // one 16-wave workgroup per CU (A/B variant, NB = 8 only)
        if (nb != 8) return hipErrorInvalidValue;
        return launch_syrk_lds_static_nb<8, 16>(a, st);
    }
    if (a.split == 2) {    // 2-wave workgroups, 18 tiles per wave (two waves per SIMD)
        switch (nb) {
            case 6: return launch_syrk_lds_static_nb<6, 2>(a, st);
            case 7: return launch_syrk_lds_static_nb<7, 2>(a, st);
            case 8: return launch_syrk_lds_static_nb<8, 2>(a, st);
            default: return hipErrorInvalidValue;
        }
    }
    if (a.split == 4) {    // four 4-wave workgroups per CU (one wave per SIMD each), 9 tiles per wave
        switch (nb) {
            case 6: return launch_syrk_lds_static_nb<6, 4>(a, st);
            case 7: return launch_syrk_lds_static_nb<7, 4>(a, st);
            case 8: return launch_syrk_lds_static_nb<8, 4>(a, st);
            default: return hipErrorInvalidValue;
        }
    }
    if (a.split == 8) {
        switch (nb) {
            case 6: return launch_syrk_lds_static_nb<6, 8>(a, st);
            case 7: return launch_syrk_lds_static_nb<7, 8>(a, st);
            case 8: return launch_syrk_lds_static_nb<8, 8>(a, st);
            default: return hipErrorInvalidValue;
        }
    }
    switch (nb) {
        case 6: return launch_syrk_lds_nb<6, 8>(a, st);
        case 7: return launch_syrk_lds_nb<7, 8>(a, st);
        case 8: return launch_syrk_lds_nb<8, 8>(a, st);
        default: return hipErrorInvalidValue;
    }
}

template <int NB>
static hipError_t launch_syrk_acc_nb(const SyrkArgs& a, hipStream_t st) {
    dim3 grid((unsigned)a.nblocks), block(256);
    const bool fullk = (a.K == 16 * NB);
#define FSNAP_LAUNCH(FK, NTL)                                                                                       \
    hipLaunchKernelGGL((fsnap_syrk_acc<NB, FK, NTL>), grid, block, 0, st, a.A, a.lda, a.b, a.w, a.mask, a.m, a.K, \
                       a.chunks_per_wave, a.part, a.cpart, a.spart)
    if (fullk) {
        if (a.nontemporal) FSNAP_LAUNCH(true, true);
        else FSNAP_LAUNCH(true, false);
    } else {
        if (a.nontemporal) FSNAP_LAUNCH(false, true);
        else FSNAP_LAUNCH(false, false);
    }
#undef FSNAP_LAUNCH
    return hipGetLastError();
}

// kernel 1A: a.nblocks workgroups of 4 row-waves, a.chunks_per_wave chunks per row-wave
hipError_t launch_syrk_acc(const SyrkArgs& a, hipStream_t st) {
    switch (syrk_num_blocks(a.K)) {
        case 6: return launch_syrk_acc_nb<6>(a, st);
        case 7: return launch_syrk_acc_nb<7>(a, st);
        case 8: return launch_syrk_acc_nb<8>(a, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_syrk(const SyrkArgs& a, hipStream_t st) {
    const int nb = syrk_num_blocks(a.K);
    if (a.split == 2) {
        switch (nb) {
            case 6: return launch_syrk_nb<6, 2>(a, st);
            case 7: return launch_syrk_nb<7, 2>(a, st);
            case 8: return launch_syrk_nb<8, 2>(a, st);
            default: return hipErrorInvalidValue;
        }
    }
    if (a.split != 1) return hipErrorInvalidValue;
    switch (nb) {
        case 1: return launch_syrk_nb<1, 1>(a, st);
        case 2: return launch_syrk_nb<2, 1>(a, st);
        case 3: return launch_syrk_nb<3, 1>(a, st);
        case 4: return launch_syrk_nb<4, 1>(a, st);
        case 5: return launch_syrk_nb<5, 1>(a, st);
        case 6: return launch_syrk_nb<6, 1>(a, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_reduce(const double* part, const double* cpart, const double* spart, int nblocks,
                         int cs_per_block, int K, double* out, double* mirror, bool accumulate, hipStream_t st) {
    const int NB = syrk_num_blocks(K);
    const int nelem = NB * (NB + 1) / 2 * 256 + NB * 16 + 4;
    dim3 grid((unsigned)((nelem + 15) / 16)), block(1024);
    hipLaunchKernelGGL(fsnap_reduce_partials, grid, block, 0, st, part, cpart, spart, nblocks, cs_per_block, NB, K, out, mirror,
                       accumulate ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_syrk_tiled(const TiledArgs& a, hipStream_t st) {
    const int nitems = (int)((int64_t)a.npairs * a.nsplit);
    dim3 grid((unsigned)(a.xcd_map ? 8 * ((nitems + 7) / 8) : nitems)), block(256);
    if (a.nontemporal)
        hipLaunchKernelGGL((fsnap_syrk_tiled<true>), grid, block, 0, st, a.A, a.lda, a.b, a.w, a.mask, a.m, a.K, a.NSB,
                           a.npairs, a.chunks_per_split, nitems, (int)a.xcd_map, a.part, a.cpart, a.spart);
    else
        hipLaunchKernelGGL((fsnap_syrk_tiled<false>), grid, block, 0, st, a.A, a.lda, a.b, a.w, a.mask, a.m, a.K, a.NSB,
                           a.npairs, a.chunks_per_split, nitems, (int)a.xcd_map, a.part, a.cpart, a.spart);
    return hipGetLastError();
}

hipError_t launch_reduce_tiled(const TiledArgs& a, double* out, bool accumulate, hipStream_t st) {
    const int64_t nelem = (int64_t)a.npairs * 4096 + a.NSB * 64 + 4;
    dim3 grid((unsigned)((nelem + 63) / 64)), block(1024);
    hipLaunchKernelGGL(fsnap_reduce_tiled, grid, block, 0, st, a.part, a.cpart, a.spart, a.nsplit, a.NSB, a.npairs, a.K,
                       out, accumulate ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_weight_rows(const double* A, int64_t lda, const double* b, const double* w,
                              const unsigned char* mask, int64_t m, int K, double* aw, int64_t ldaw,
                              double* bw, hipStream_t st) {
    int64_t nb = (m + 15) / 16;
    if (nb > 256 * 8) nb = 256 * 8;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(fsnap_weight_rows_k, dim3((unsigned)nb), dim3(256), 0, st, A, lda, b, w, mask, m, K, aw,
                       ldaw, bw);
    return hipGetLastError();
}

hipError_t launch_assemble(const double* raw, int64_t raw_ld, int64_t nrows, const int64_t* src_row, const int* kind,
                           const int* frac, const double* dval, const double* truth, const double* weight,
                           const double* fractions, const double* blank2J, int ntypes, int ncoeff, int off, double* A,
                           int64_t lda, double* b, double* w, hipStream_t st) {
    int64_t nb = (nrows + 3) / 4;
    if (nb > 256 * 8) nb = 256 * 8;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(fsnap_assemble_k, dim3((unsigned)nb), dim3(256), 0, st, raw, raw_ld, nrows, src_row, kind, frac,
                       dval, truth, weight, fractions, blank2J, ntypes, ncoeff, off, A, lda, b, w);
    return hipGetLastError();
}

int gemvT_num_blocks(int64_t m) {
    int64_t nb = (m + 63) / 64;
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    return (int)nb;
}

hipError_t launch_gemvT_rows(const double* A, int64_t lda, const double* u, int64_t m, int K, double* partial,
                             double* out, hipStream_t st) {
    const int nb = gemvT_num_blocks(m);
    const int64_t rpw = (m + nb - 1) / nb;
    const size_t lds = (size_t)4 * ((K + 1) & ~1) * sizeof(double);
    if (lds > 160 * 1024 - 256) return hipErrorInvalidValue;
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)fsnap_gemvT_rows_k, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024 - 256);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(fsnap_gemvT_rows_k, dim3((unsigned)nb), dim3(256), lds, st, A, lda, u, m, K, rpw, partial);
    hipLaunchKernelGGL(fsnap_colsum_partials_k, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, st, partial, nb, K, out);
    return hipGetLastError();
}

int error_stats_num_blocks(int64_t m) {
    int64_t nb = (m + 256 * 16 - 1) / (256 * 16);
    if (nb > 512) nb = 512;
    if (nb < 1) nb = 1;
    return (int)nb;
}

hipError_t launch_error_stats(const double* truth, const double* pred, const double* wgt, const int* cat, int64_t m, int ncat,
                              int pass, const double* means, double* partial, hipStream_t st) {
    const int nv = pass == 0 ? 4 : 6;
    const size_t lds = (size_t)ncat * nv * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)fsnap_error_stats_k, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024 - 64);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(fsnap_error_stats_k, dim3((unsigned)error_stats_num_blocks(m)), dim3(256), lds, st, truth, pred, wgt, cat,
                       m, ncat, pass, means, partial);
    return hipGetLastError();
}

hipError_t launch_copy_to_host(const double* src, double* dst_pinned, int64_t n, hipStream_t st) {
    int64_t nb = (n + 255) / 256;
    if (nb > 256) nb = 256;
    hipLaunchKernelGGL(fsnap_copy_to_host_k, dim3((unsigned)nb), dim3(256), 0, st, src, dst_pinned, n);
    return hipGetLastError();
}

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

int gemv_num_blocks(int64_t m) {
    int64_t nb = (m + 15) / 16;
    if (nb > 256 * 8) nb = 256 * 8;
    if (nb < 1) nb = 1;
    return (int)nb;
}

hipError_t launch_gemv_rows(const double* A, int64_t lda, const double* beta, int64_t m, int K, double* preds,
                            const double* b, const double* w, const unsigned char* mask, double* sse_part,
                            double* uout, hipStream_t st) {
    const int nb = gemv_num_blocks(m);
    hipLaunchKernelGGL(fsnap_gemv_rows_k, dim3((unsigned)nb), dim3(256), (size_t)K * sizeof(double), st, A, lda,
                       beta, m, K, preds, b, w, mask, sse_part, uout);
    return hipGetLastError();
}

}  // namespace fsnap
