// fsnap_fused.hip — post-LAMMPS assembly fused into the accumulation of the normal equations (gfx950 only):
// the per-configuration loop of examples/library/transpose_trick/example.py:230-237
//     a, b, w = process_single(configuration);  aw = w[:, None] * a;  c += aw.T @ aw;  d += aw.T @ (w * b)
// with the rows of `a` formed in registers from the raw `compute snap|pace` arrays of a batch and fed straight to the
// matrix pipe: A is never written to HBM (and never read back).
//   15  fsnap_assemble_bw_k     b, w and a 16-byte row record per output row (what of a row is not A: 32 B per row)
//   16  fsnap_assemble_syrk_k   kernel 1T's work decomposition (64-column superblock pairs x row splits, 16 tiles per
//                               wave, four waves over a split's rows, fold through LDS, one partial per workgroup)
//                               with kernel 5's arithmetic as the loader of the MFMA operands
// Both halves keep the arithmetic ORDER of the two-step path: a value of A is formed exactly as fsnap_assemble_k forms
// it (divide, not multiply by a reciprocal; blank2J last), weighted exactly as fsnap_syrk_tiled weights it (w a on both
// sides of a diagonal pair, (w^2 a_I) x a_J on an off-diagonal one), chunks of four rows reach a tile's accumulator in
// the same order from the same wave, and partials have kernel 1T's layout and go through ITS reduction kernels.  The
// statistics are therefore bit-identical to fsnap_assemble + fsnap_normal_eq_accumulate with the tiled kernel (the
// default for K > 128, option tiled = 1 below) -- tests/test_gpu_assembly.py holds that to the last bit.
// Cost model: the rows are HBM traffic in neither direction; raw is read once per superblock pair that needs it
// (L2-resident: a batch is a few MB).  A batch is a few thousand rows, so the launch is latency-bound; what the fusion
// buys is the memory profile of the reference's loop (no m x K array anywhere) without a second pass over the rows.
#include "fsnap_device_common.h"
#include "fsnap_kernels.h"

namespace {

struct __attribute__((aligned(16))) RowRec {
    double d;      // atoms (energy rows) or cell volume (virial rows)
    int srow;      // row of raw
    int kf;        // kind | (frac + 1) << 2
};

__device__ __forceinline__ double xlane_sum_rows4(double x) {
    // sum over the four 16-lane row groups (lanes l, l^16, l^32, l^48)
    x += __shfl_xor(x, 16, 64);
    x += __shfl_xor(x, 32, 64);
    return x;
}

// column k of A as the assembly sees it: 0 = descriptor column (raw column rc), 1 = per-type offset column of type rc
// (bzeroflag = 0), 2 = k >= K (edge of the last superblock: zero)
struct ColDesc {
    int code;          // rc | cls << 30
    double blank;
    __device__ __forceinline__ int cls() const { return (int)((unsigned)code >> 30); }
    __device__ __forceinline__ int rc() const { return code & 0x3FFFFFFF; }
};

__device__ __forceinline__ ColDesc describe_column(int k, int K, int ncoeff, int off, const double* __restrict__ blank2J) {
    ColDesc c;
    if (k >= K) {
        c.code = 2 << 30;
        c.blank = 0.0;
        return c;
    }
    const int stride = ncoeff + off;
    const int t = k / stride, j = k - t * stride;
    if (off && j == 0) c.code = t | (1 << 30);
    else c.code = t * ncoeff + (j - off);
    c.blank = blank2J[k];
    return c;
}

struct RowMeta {
    double w, wb;   // (w_eff, w_eff b) of the lane's row of the chunk (w_eff b on diagonal pairs only); zeros past the batch
    RowRec rec;
    __device__ __forceinline__ bool keep() const {
        // rows with w_eff == +-0 contribute exact zeros (kernel 1T never fetches them)
        return (__double_as_longlong(w) & 0x7FFFFFFFFFFFFFFFll) != 0ll;
    }
};

template <bool DIAG>
struct RawVals {
    double xi[4], xj[4];
};

template <bool DIAG>
__device__ __forceinline__ void fused_body(const double* __restrict__ raw, int64_t raw_ld,
                                           const RowRec* __restrict__ recs, const double* __restrict__ dval,
                                           const double* __restrict__ wpack, int64_t nrows, const double* __restrict__ fractions,
                                           const double* __restrict__ blank2J, int ntypes, int ncoeff, int off, int K,
                                           int I, int J, int64_t c0, int64_t c1, int wv_in_wg, double* lds,
                                           double* __restrict__ pw, double* __restrict__ cw) {
    constexpr int NTW = 16;
    const int lane = threadIdx.x & 63, e = lane & 15, kr = lane >> 4;
    d4 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
    double cacc[4] = {0.0, 0.0, 0.0, 0.0};

    // the lane's four columns per side: block p of superblock S holds column 64 S + 32 (p >> 1) + 2 e + (p & 1)
    // (kernel 1T's even / odd interleave, undone by the reduction kernel)
    ColDesc ci[4], cj[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        ci[p] = describe_column(64 * I + 32 * (p >> 1) + 2 * e + (p & 1), K, ncoeff, off, blank2J);
        if (!DIAG) cj[p] = describe_column(64 * J + 32 * (p >> 1) + 2 * e + (p & 1), K, ncoeff, off, blank2J);
    }

    // the loads of the loop are unconditional (clamped addresses, results selected afterwards): with a branch around a
    // load the compiler no longer knows how many are in flight and waits for ALL of them before the next use -- the
    // prefetched chunk included
    auto load_meta = [&](int64_t c) -> RowMeta {
        RowMeta mt;
        const int64_t r = (c << 2) + kr;
        const bool in = c < c1 && r < nrows;
        const int64_t rr = r < nrows ? r : nrows - 1;
        if (DIAG) {
            const d2u wp = *reinterpret_cast<const d2u*>(wpack + 2 * rr);
            mt.w = in ? wp[0] : 0.0;
            mt.wb = in ? wp[1] : 0.0;
        } else {
            const double wv = wpack[2 * rr];
            mt.w = in ? wv : 0.0;
            mt.wb = 0.0;
        }
        mt.rec = recs[rr];
        return mt;
    };
    // branch-free: every lane loads from an address that exists (row 0 of raw for rows that are not kept, fractions[0]
    // where a row has no offset entry) and what does not belong is selected away afterwards -- with a branch per value
    // the loads of a chunk end up in separate basic blocks and go out one round trip after the other
    auto fetch = [&](const ColDesc& cd, const RowMeta& mt) -> double {
        const int cls = cd.cls(), rc = cd.rc();
        const int kd = mt.rec.kf & 3, fr = (mt.rec.kf >> 2) - 1;
        const bool fcol = off != 0 && cls == 1;
        const bool fvalid = kd == 0 && fr >= 0;
        const double* src = fcol ? fractions + (int64_t)(fvalid ? fr : 0) * ntypes + rc
                                 : raw + (int64_t)mt.rec.srow * raw_ld + rc;
        const double x = *src;
        const bool ok = mt.keep() && cls != 2 && (!fcol || fvalid);
        return ok ? x : 0.0;
    };
    auto load_raw = [&](const RowMeta& mt) -> RawVals<DIAG> {
        RawVals<DIAG> x;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            x.xi[p] = fetch(ci[p], mt);
            x.xj[p] = DIAG ? 0.0 : fetch(cj[p], mt);
        }
        return x;
    };
    // fsnap_assemble_k's value of A[r][k] from the raw value (same operations, same order).  Force rows (kind 1) take
    // the raw value as it is; the others divide -- a dozen fp64 instructions per value that serialise with the MFMAs on
    // the SIMD, so the dividing form runs only for chunks that hold such a row (3 N of a configuration's 3 N + 7 rows
    // are force rows): the branch is wave-uniform.
    auto assembled = [&](double x, const ColDesc& cd, const RowMeta& mt, bool divides) -> double {
        double v = x;
        if (divides) {
            const int kd = mt.rec.kf & 3;
            const double q = ((kd == 2) ? 1.6021765e6 * x : x) / mt.rec.d;       // one division, no branch per lane
            v = (kd != 1 && cd.cls() == 0) ? q : x;
        }
        return v * cd.blank;
    };

    double vI[4] = {0.0, 0.0, 0.0, 0.0}, vJ[4] = {0.0, 0.0, 0.0, 0.0};
    // one chunk: MFMA operands from its raw values (kernel 1T's prep_t: the weight on ONE side of an off-diagonal
    // pair), then its 16 (10) MFMAs
    auto chunk = [&](const RawVals<DIAG>& x, const RowMeta& mt) {
        const double wv = mt.w;
        const double f = DIAG ? wv : wv * wv;
        const bool keep = mt.keep();
        const bool divides = __builtin_amdgcn_ballot_w64(keep && (mt.rec.kf & 3) != 1) != 0ull;
        if (divides) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const double a = assembled(x.xi[p], ci[p], mt, true);
                vI[p] = (ci[p].cls() == 2 || !keep) ? 0.0 : f * a;
                if (!DIAG) vJ[p] = (cj[p].cls() == 2 || !keep) ? 0.0 : assembled(x.xj[p], cj[p], mt, true);
            }
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const double a = assembled(x.xi[p], ci[p], mt, false);
                vI[p] = (ci[p].cls() == 2 || !keep) ? 0.0 : f * a;
                if (!DIAG) vJ[p] = (cj[p].cls() == 2 || !keep) ? 0.0 : assembled(x.xj[p], cj[p], mt, false);
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
            for (int q = (DIAG ? p : 0); q < 4; ++q)
                acc[p * 4 + q] = __builtin_amdgcn_mfma_f64_16x16x4f64(vI[p], DIAG ? vI[q] : vJ[q], acc[p * 4 + q], 0, 0, 0);
        }
        if (DIAG) {
            const double wb = mt.wb;
#pragma unroll
            for (int p = 0; p < 4; ++p) cacc[p] = __builtin_fma(vI[p], wb, cacc[p]);
        }
    };
    // Software pipeline over three named register sets (no copies between them: a copy of a set that is still being
    // loaded would wait for it).  Step c: raw values of chunk c + 1 (its row records arrived during the last step), row
    // records of chunk c + 2, then the arithmetic of chunk c.  Chunks past the wave's range are zeros, as in kernel 1T.
    if (c0 < c1) {
        RowMeta m0 = load_meta(c0), m1 = load_meta(c0 + 1), m2;
        RawVals<DIAG> x0 = load_raw(m0), x1, x2;
        for (int64_t c = c0; c < c1; c += 3) {
            // (records before raw values: the counter of outstanding loads retires in order, so the next step's wait for
            // the records leaves the eight raw loads behind them in flight)
            m2 = load_meta(c + 2);
            x1 = load_raw(m1);
            chunk(x0, m0);
            m0 = load_meta(c + 3);
            x2 = load_raw(m2);
            chunk(x1, m1);
            m1 = load_meta(c + 4);
            x0 = load_raw(m0);
            chunk(x2, m2);
        }
    }

    // fold the 4 waves through LDS ({2,3} -> {0,1}, 1 -> 0), then one partial per workgroup: kernel 1T's order
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
            const double s = xlane_sum_rows4(cacc[p]);
            if (kr == 0) cw[p * 16 + e] = s;
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------
// Kernel 15: what a row carries besides A -- b and w exactly as fsnap_assemble_k forms them (lammps_snap.py:391-556:
// energy rows (truth - ref) / N, force / virial rows truth - ref, the extra per-atom-energy rows b = w = 0), and the
// 16-byte record kernel 16 reads per row.  The (w_eff, w_eff b) pairs and the b-only statistics then come from the
// packing kernel of the two-step path (fsnap_pack_weights_k), so they carry the same bits.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fsnap_assemble_bw_k(const double* __restrict__ raw, int64_t raw_ld, int64_t nrows,
                                                           const int64_t* __restrict__ src_row,
                                                           const int* __restrict__ kind, const int* __restrict__ frac,
                                                           const double* __restrict__ dval,
                                                           const double* __restrict__ truth,
                                                           const double* __restrict__ weight, int icolref,
                                                           double* __restrict__ b, double* __restrict__ w,
                                                           RowRec* __restrict__ recs) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrows) return;
    const int64_t sr = src_row[r];
    const int kd = kind[r];
    const double d = dval[r];
    const double ref = raw[sr * raw_ld + icolref];
    double bv, wv = weight[r];
    if (kd == 0) bv = (truth[r] - ref) / d;
    else if (kd == 3) {
        bv = 0.0;
        wv = 0.0;
    } else bv = truth[r] - ref;
    b[r] = bv;
    w[r] = wv;
    RowRec rec;
    rec.d = d;
    rec.srow = (int)sr;
    rec.kf = kd | ((frac[r] + 1) << 2);
    recs[r] = rec;
}

// ---------------------------------------------------------------------------------
// Kernel 16: G, c partials of a batch straight from the raw arrays.  Grid and partial layout = fsnap_syrk_tiled
// (item = (split, pair), off-diagonal pairs of a split first; contiguous item ranges per XCD).
//   partT[split*npairs + pair][16][4][64] | cpartT[(split*NSB + I)*4 + wave][4][16]
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void fsnap_assemble_syrk_k(const double* __restrict__ raw, int64_t raw_ld,
                                                                const void* __restrict__ recs_,
                                                                const double* __restrict__ dval,
                                                                const double* __restrict__ wpack, int64_t nrows,
                                                                const double* __restrict__ fractions,
                                                                const double* __restrict__ blank2J, int ntypes,
                                                                int ncoeff, int off, int K, int NSB, int npairs,
                                                                int64_t chunks_per_split, int nitems, int xcd_map,
                                                                double* __restrict__ part, double* __restrict__ cpart) {
    __shared__ double lds[2 * 16 * 256];
    const RowRec* recs = reinterpret_cast<const RowRec*>(recs_);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned item = blockIdx.x;
    if (xcd_map) {
        const unsigned per = ((unsigned)nitems + 7u) >> 3;
        const unsigned slot = blockIdx.x >> 3;
        item = (blockIdx.x & 7u) * per + slot;
        if (slot >= per || item >= (unsigned)nitems) return;
    }
    const int id = (int)(item % (unsigned)npairs);
    const int split = (int)(item / (unsigned)npairs);
    const int noff = npairs - NSB;
    int I, J;
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
    const int pair = I * NSB - (I * (I - 1)) / 2 + (J - I);
    const int64_t nchunks = (nrows + 3) >> 2;
    const int64_t cpw = (chunks_per_split + 3) >> 2;
    const int64_t s0 = (int64_t)split * chunks_per_split;
    int64_t s1 = s0 + chunks_per_split;
    if (s1 > nchunks) s1 = nchunks;
    int64_t c0 = s0 + (int64_t)wv * cpw;
    int64_t c1 = c0 + cpw;
    if (c1 > s1) c1 = s1;
    if (c0 > s1) c0 = s1;
    double* pw = part + ((int64_t)split * npairs + pair) * (16 * 256);
    double* cw = cpart + (((int64_t)split * NSB + I) * 4 + wv) * 64;
    if (I == J)
        fused_body<true>(raw, raw_ld, recs, dval, wpack, nrows, fractions, blank2J, ntypes, ncoeff, off, K, I, J, c0, c1, wv, lds, pw, cw);
    else
        fused_body<false>(raw, raw_ld, recs, dval, wpack, nrows, fractions, blank2J, ntypes, ncoeff, off, K, I, J, c0, c1, wv, lds, pw, cw);
}

namespace fsnap {

size_t assemble_row_record_bytes() { return sizeof(RowRec); }

hipError_t launch_assemble_bw(const double* raw, int64_t raw_ld, int64_t nrows, const int64_t* src_row, const int* kind,
                              const int* frac, const double* d, const double* truth, const double* weight, int icolref,
                              double* b, double* w, void* recs, hipStream_t st) {
    hipLaunchKernelGGL(fsnap_assemble_bw_k, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, st, raw, raw_ld, nrows, src_row,
                       kind, frac, d, truth, weight, icolref, b, w, reinterpret_cast<RowRec*>(recs));
    return hipGetLastError();
}

hipError_t launch_assemble_syrk(const double* raw, int64_t raw_ld, const void* recs, const double* dval, const double* fractions,
                                const double* blank2J, int ntypes, int ncoeff, int off, const TiledArgs& a, hipStream_t st) {
    const int nitems = (int)((int64_t)a.npairs * a.nsplit);
    dim3 grid((unsigned)(8 * ((nitems + 7) / 8))), block(256);
    hipLaunchKernelGGL(fsnap_assemble_syrk_k, grid, block, 0, st, raw, raw_ld, recs, dval, a.wpack, a.m, fractions, blank2J, ntypes,
                       ncoeff, off, a.K, a.NSB, a.npairs, a.chunks_per_split, nitems, 1, a.part, a.cpart);
    return hipGetLastError();
}

}  // namespace fsnap
