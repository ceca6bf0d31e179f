// fsnap_rows.hip — HBM-bound row-streaming kernels of the FitSNAP linear-fit path (gfx950 only):
//   3   fsnap_weight_rows_k    stand-alone wavefront row weighting            (svd.py:46 / ridge.py:39)
//   4   fsnap_gemv_rows_k      preds = A @ beta (+ weighted SSE, refinement u) (solver.py:377)
//   5   fsnap_assemble_k       post-LAMMPS assembly (_collect_lammps)         (lammps_snap.py:391-556)
//   7   fsnap_gemvT_rows_k     s = A^T u (right-hand side of a refinement step)
//   9   fsnap_error_stats_k    grouped error statistics of error_analysis     (solver.py:108-133)
//   11  fsnap_pack_weights_k   (w_eff, w_eff b) per row + the b-only statistics, once per (b, w, mask)
//   12  fsnap_mirror_copy_k    packed statistics + diag(G) into the page-locked host mirror (multi-GPU path)
//   14  fsnap_expand_weights_k one weight per TRAINING row -> one weight per row (the reference's explicit-array quirk)
// Every kernel here moves each byte once; the roofline is HBM bandwidth.
#include "fsnap_device_common.h"
#include "fsnap_kernels.h"

// ---------------------------------------------------------------------------------
// Kernel 3: stand-alone wavefront row weighting (svd.py:46 / ridge.py:39).
//   aw[i,:] = w[i]*A[i,:], bw[i] = w[i]*b[i] for every row; masked rows are written
//   as zeros (row compaction is the host shim's business, see fsnap_weight_rows()).
// One wave per 4-row group (grid-stride only beyond 2^34 rows), 16-byte vector accesses.  HBM-bound:
// 16K + 24 bytes per row.
// ---------------------------------------------------------------------------------
template <int lanes_log2>
__global__ __launch_bounds__(256) void fsnap_weight_rows_k(const double* __restrict__ A, int64_t lda,
                                                           const double* __restrict__ b,
                                                           const double* __restrict__ w,
                                                           const unsigned char* __restrict__ mask, int64_t m,
                                                           int K, double* __restrict__ aw,
                                                           int64_t ldaw, double* __restrict__ bw) {
    // L = 2^lanes_log2 lanes share a row (two adjacent columns per lane and pass over the row: L = 64 from K = 65 on,
    // 16 at the Ta width K = 31 -- with 64 lanes on a 31-column row three quarters of every access are idle lanes
    // and the kernel streamed 2.5 TB/s), a wave covers 64 / L rows per pass and four passes per iteration: four
    // independent 16-byte loads per lane in flight before the first store.
    // 16-byte accesses for every K and leading dimension: the vector types are 8-byte aligned (rows of an odd width
    // start on odd multiples of 8 bytes), the upper half of a row's last pair is the neighbour's first element when K
    // is odd -- read (the allocation is readable 16 bytes past its end) but neither used nor overwritten.
    const int lane = threadIdx.x & 63;
    constexpr int L = 1 << lanes_log2, G = 64 >> lanes_log2;
    const int sub = lane >> lanes_log2, cl = lane & (L - 1);      // L = 64: sub == 0, every row test below is wave-uniform
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * 4;
    constexpr int64_t per_iter = 4 * (int64_t)G;
    for (int64_t row0 = wave * per_iter; row0 < m; row0 += nwave * per_iter) {
        double wv[4];
        bool keep[4], in[4];
        int64_t rows[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rows[r] = row0 + (int64_t)r * G + sub;
            in[r] = rows[r] < m;
            keep[r] = in[r] && (mask[in[r] ? rows[r] : 0] != 0);
            wv[r] = in[r] ? w[rows[r]] : 0.0;
        }
        for (int c = 2 * cl; c < K; c += 2 * L) {
            // the four loads go out unconditionally (rows past the end re-read the last row): a lane-dependent branch
            // around a load puts it into a basic block of its own, and the loads of an iteration leave one by one
            d2u x[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                x[r] = __builtin_nontemporal_load(reinterpret_cast<const d2u*>(A + (in[r] ? rows[r] : m - 1) * lda + c));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (in[r]) {
                    d2u y;
                    y[0] = keep[r] ? wv[r] * x[r][0] : 0.0;
                    y[1] = keep[r] ? wv[r] * x[r][1] : 0.0;
                    double* dst = aw + rows[r] * ldaw + c;
                    if (c + 1 < K) __builtin_nontemporal_store(y, reinterpret_cast<d2u*>(dst));
                    else __builtin_nontemporal_store(y[0], dst);
                }
            }
        }
        if (cl == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (in[r]) bw[rows[r]] = keep[r] ? wv[r] * b[rows[r]] : 0.0;
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
                                                         double* __restrict__ uout, int uplain) {
    // uout (optional): u_i = mask_i * w_i^2 * (b_i - a_i . beta), the row weights of the
    // refinement right-hand side  s = (wA)^T (wb - wA beta) = A^T u   (kernel 7);
    // uplain: u_i = mask_i * w_i * (b_i - a_i . beta) instead, the weighted residual itself (its product with the
    // orthonormal factor Q of the row-space solve is the refinement right-hand side there)
    extern __shared__ __attribute__((aligned(16))) double sbeta[];
    for (int c = threadIdx.x; c < K; c += 256) sbeta[c] = beta[c];
    __syncthreads();
    const int lane = threadIdx.x & 63, e = lane & 15, kr = lane >> 4;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * 4;
    double sse = 0.0;
    // two 4-row groups per iteration (rows r0 + kr and r0 + 4 + kr): on narrow rows one group is a single load per lane
    // followed by the dependent lane reduction -- K = 31 ran at 4.1 TB/s with one
    for (int64_t r0 = wave * 8; r0 < m; r0 += nwave * 8) {
        const int64_t rowA = r0 + kr, rowB = r0 + 4 + kr;
        const bool inA = rowA < m, inB = rowB < m;
        double sA = 0.0, sB = 0.0, tA = 0.0, tB = 0.0;
        {
            // 16-byte loads (8-byte aligned vector type: any K, any lda): two adjacent columns per lane, two
            // accumulators; the upper half of an odd row's last pair belongs to the next row and is selected away
            const double* srcA = A + (inA ? rowA : 0) * lda;
            const double* srcB = A + (inB ? rowB : 0) * lda;
            for (int c = 2 * e; c < K; c += 32) {
                const d2u x = __builtin_nontemporal_load(reinterpret_cast<const d2u*>(srcA + c));
                const d2u y = __builtin_nontemporal_load(reinterpret_cast<const d2u*>(srcB + c));
                const bool two = c + 1 < K;
                const double be0 = sbeta[c], be1 = two ? sbeta[c + 1] : 0.0;
                sA = __builtin_fma(x[0], be0, sA);
                tA = __builtin_fma(two ? x[1] : 0.0, be1, tA);
                sB = __builtin_fma(y[0], be0, sB);
                tB = __builtin_fma(two ? y[1] : 0.0, be1, tB);
            }
            sA += tA;
            sB += tB;
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int64_t row = half ? rowB : rowA;
            double s = half ? sB : sA;
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
                    if (uout) uout[row] = keep ? (uplain ? rr : wr * rr) : 0.0;
                }
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
// Kernel 7: s = A^T u  (transposed streaming GEMV, HBM-bound) — with u from kernel 4 this is
// the right-hand side of one step of iterative refinement of the least-squares solution
// ("corrected semi-normal equations": G delta = (wA)^T (wb - wA beta), beta += delta), which
// takes the error of the normal-equation solve from ~kappa^2 eps back to ~kappa eps — what
// keeps the GPU path within 1e-6 of the reference's lstsq (svd.py:54) on ill-conditioned A.
// Workgroup = row range, cut into 4 * 64 / L interleaved row streams; a lane owns two adjacent columns per pass.
// Per-workgroup partial vectors are written to spart2[wg][K] and summed in fixed order.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fsnap_gemvT_rows_k(const double* __restrict__ A, int64_t lda,
                                                          const double* __restrict__ u, int64_t m, int K,
                                                          int lanes_log2, int64_t rows_per_wg,
                                                          double* __restrict__ partial) {
    // L = 2^lanes_log2 lanes cover a row (two adjacent columns per lane and pass), so a wave runs 64 / L row streams
    // side by side (L = 64: one; the Ta width K = 31: four -- with one stream three quarters of the lanes of every load
    // were idle: 2.9 TB/s); stream i of the workgroup's 4 * 64 / L takes rows i, i + NS, ..., two rows in flight each.
    extern __shared__ __attribute__((aligned(16))) double sacc[];   // NS streams x Kpad
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int L = 1 << lanes_log2, G = 64 >> lanes_log2, NS = 4 * G;
    const int sub = lane >> lanes_log2, cl = lane & (L - 1);
    const int sidx = wv * G + sub;
    const int Kpad = (K + 1) & ~1;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
    int64_t r1 = r0 + rows_per_wg;
    if (r1 > m) r1 = m;
    for (int c0 = 0; c0 < K; c0 += 2 * L) {
        const int c = c0 + 2 * cl;
        double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
        if (c < K) {
            // 16-byte loads for any K and lda (8-byte aligned vector type); the upper half of an odd row's last
            // pair is the next row's first element: selected away
            const bool two = c + 1 < K;
            int64_t row = r0 + sidx;
            for (; row + NS < r1; row += 2 * NS) {   // two rows in flight per stream
                const double u0 = u[row], u1 = u[row + NS];
                const d2u x = *reinterpret_cast<const d2u*>(A + row * lda + c);
                const d2u y = *reinterpret_cast<const d2u*>(A + (row + NS) * lda + c);
                const double x0 = x[0], x1 = two ? x[1] : 0.0, y0 = y[0], y1 = two ? y[1] : 0.0;
                a0 = __builtin_fma(x0, u0, a0);
                a1 = __builtin_fma(x1, u0, a1);
                b0 = __builtin_fma(y0, u1, b0);
                b1 = __builtin_fma(y1, u1, b1);
            }
            for (; row < r1; row += NS) {
                const double u0 = u[row];
                const double x0 = A[row * lda + c];
                const double x1 = two ? A[row * lda + c + 1] : 0.0;
                a0 = __builtin_fma(x0, u0, a0);
                a1 = __builtin_fma(x1, u0, a1);
            }
            sacc[sidx * Kpad + c] = a0 + b0;
            if (two) sacc[sidx * Kpad + c + 1] = a1 + b1;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < K; c += 256) {
        double t = 0.0;
        for (int i = 0; i < NS; i += 4)          // fixed order: streams in fours, as the four waves used to be summed
            t += (sacc[i * Kpad + c] + sacc[(i + 1) * Kpad + c]) + (sacc[(i + 2) * Kpad + c] + sacc[(i + 3) * Kpad + c]);
        partial[(int64_t)blockIdx.x * K + c] = t;
    }
}

// ---------------------------------------------------------------------------------
// Kernel 4+7: one-pass refinement right-hand side  s = (wA)^T (wb - wA beta)  of the SVD solver's corrected semi-normal
// step (svd.py:54 is lstsq on the weighted rows; this keeps the normal-equation solve at its accuracy) -- kernels 4 and 7
// fused: a row is read ONCE and stays in registers between the two uses,
//     r_i = keep_i w_i (b_i - a_i . beta)      lane reduction over the 16 lanes of the row (the bits of kernel 4)
//     s  += a_i (w_i r_i)                      per-lane column accumulators, folded in a fixed order at the end
// plus the weighted SSE sum r_i^2.  16 lanes per row, a wave takes 8 rows per step (two groups of four); the latency of a
// step's loads is covered by the other waves of the SIMD (4 at K = 128, 8 at K = 31; prefetching the next step's rows into a
// second register set instead halves the waves per SIMD and measured slower at every shape, round 4).  Rows that do not take part (test
// rows, rows past m) are zeroed by selects: NaN / Inf in them reach nothing.  K <= 32 NJ (NJ <= 9: 72 VGPRs of row data
// per set); wider systems keep the two-kernel form.  HBM-bound: 8K + 17 bytes per row.
// Per-workgroup partial vectors partial[wg][K] (fold: kernel fsnap_colsum_partials_k), sse_part[wg].
// ---------------------------------------------------------------------------------
template <int NJ>
struct ResidualRows {
    d2u x[2][NJ];
    double bb[2], ww[2];
    bool keep[2];
};

template <int NJ>
__global__ __launch_bounds__(256) void fsnap_residual_rows_k(const double* __restrict__ A, int64_t lda,
                                                             const double* __restrict__ beta, int64_t m, int K,
                                                             const double* __restrict__ b, const double* __restrict__ w,
                                                             const unsigned char* __restrict__ mask,
                                                             double* __restrict__ partial, double* __restrict__ sse_part) {
    constexpr int KP = NJ * 32;
    __shared__ __attribute__((aligned(16))) double sbeta[KP];
    __shared__ double fold[4][KP];
    __shared__ double wsum[4];
    for (int c = threadIdx.x; c < KP; c += 256) sbeta[c] = c < K ? beta[c] : 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, e = lane & 15, kr = lane >> 4, wv = threadIdx.x >> 6;
    bool v1[NJ], v2[NJ];
    int coff[NJ];
    double be0[NJ], be1[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = 2 * e + 32 * j;
        v1[j] = c < K;
        v2[j] = c + 1 < K;
        coff[j] = v1[j] ? c : 0;          // lanes past the row's end re-read its first pair (selected away)
        be0[j] = sbeta[c];
        be1[j] = sbeta[c + 1];
    }
    double a0[NJ], a1[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) a0[j] = a1[j] = 0.0;
    double sse = 0.0;
    const int64_t wave = (int64_t)blockIdx.x * 4 + wv, step = (int64_t)gridDim.x * 4 * 8;

    auto fetch = [&](int64_t r0, ResidualRows<NJ>& R) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t row = r0 + 4 * h + kr;
            const bool in = row < m;
            const int64_t rr = in ? row : 0;
            R.keep[h] = in && (mask[rr] != 0);
            R.bb[h] = b[rr];
            R.ww[h] = w[rr];
            const double* src = A + rr * lda;
#pragma unroll
            for (int j = 0; j < NJ; ++j) R.x[h][j] = __builtin_nontemporal_load(reinterpret_cast<const d2u*>(src + coff[j]));
        }
    };
    auto process = [&](const ResidualRows<NJ>& R) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double x0[NJ], x1[NJ];
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                x0[j] = (R.keep[h] && v1[j]) ? R.x[h][j][0] : 0.0;
                x1[j] = (R.keep[h] && v2[j]) ? R.x[h][j][1] : 0.0;
                s0 = __builtin_fma(x0[j], be0[j], s0);
                s1 = __builtin_fma(x1[j], be1[j], s1);
            }
            double sd = s0 + s1;
            sd += __shfl_xor(sd, 8, 64);
            sd += __shfl_xor(sd, 4, 64);
            sd += __shfl_xor(sd, 2, 64);
            sd += __shfl_xor(sd, 1, 64);
            const double rr = R.keep[h] ? R.ww[h] * (R.bb[h] - sd) : 0.0;
            const double u = R.keep[h] ? R.ww[h] * rr : 0.0;
            if (e == 0) sse = __builtin_fma(rr, rr, sse);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                a0[j] = __builtin_fma(x0[j], u, a0[j]);
                a1[j] = __builtin_fma(x1[j], u, a1[j]);
            }
        }
    };

    int64_t r0 = wave * 8;
    {                   // one register set: the waves of a SIMD cover each other's load latency
        ResidualRows<NJ> R0;
        for (; r0 < m; r0 += step) {
            fetch(r0, R0);
            process(R0);
        }
    }
    // fold: the four row groups of a wave (lanes e, e + 16, e + 32, e + 48), then the four waves through LDS, fixed order
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        a0[j] += __shfl_xor(a0[j], 16, 64);
        a0[j] += __shfl_xor(a0[j], 32, 64);
        a1[j] += __shfl_xor(a1[j], 16, 64);
        a1[j] += __shfl_xor(a1[j], 32, 64);
        if (kr == 0) {
            fold[wv][2 * e + 32 * j] = a0[j];
            fold[wv][2 * e + 32 * j + 1] = a1[j];
        }
    }
    sse += __shfl_xor(sse, 16, 64);
    sse += __shfl_xor(sse, 32, 64);
    if (lane == 0) wsum[wv] = sse;
    __syncthreads();
    for (int c = threadIdx.x; c < K; c += 256)
        partial[(int64_t)blockIdx.x * K + c] = (fold[0][c] + fold[1][c]) + (fold[2][c] + fold[3][c]);
    if (threadIdx.x == 0 && sse_part) sse_part[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// out[c] = sum over the per-workgroup partial vectors, fixed order.  A workgroup owns 16 columns: thread (column
// c0 + (tid & 15), lane group g = tid >> 4) adds the partials g, g + 16, ... (four independent accumulators: the loads
// of a thread are dependent only through the sums), then the 16 groups are folded through LDS in a fixed tree.  (One
// thread per column walked 2048 partials one dependent load after the other: 143 us at K = 128 -- as long as the
// streaming pass it finishes.)
__global__ __launch_bounds__(256) void fsnap_colsum_partials_k(const double* __restrict__ partial, int nparts, int K,
                                                               double* __restrict__ out) {
    __shared__ double fold[16][17];
    const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (c < K) {
        int p = g;
        for (; p + 48 < nparts; p += 64) {
            s0 += partial[(int64_t)p * K + c];
            s1 += partial[(int64_t)(p + 16) * K + c];
            s2 += partial[(int64_t)(p + 32) * K + c];
            s3 += partial[(int64_t)(p + 48) * K + c];
        }
        for (; p < nparts; p += 16) s0 += partial[(int64_t)p * K + c];
    }
    fold[g][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int h = 8; h > 0; h >>= 1) {
        if (g < h) fold[g][cl] += fold[g + h][cl];
        __syncthreads();
    }
    if (g == 0 && c < K) out[c] = fold[0][cl];
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

// ---------------------------------------------------------------------------------
// Kernel 11: per-row weights of the SYRK kernel 1A, packed once per (b, w, mask):
//   wpack[row] = (w_eff, wb_eff),  w_eff = keep ? w : 0,  wb_eff = keep ? w * b : 0     (svd.py:44-46: w[training], w * b)
// and the statistics that do not involve A:  b^T W^2 b = sum wb_eff^2,  sum wb_eff,  n_train = sum keep
// (per-workgroup partials, summed in fixed order by the reduction kernel).  HBM-bound, 33 B per row.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fsnap_pack_weights_k(const double* __restrict__ b, const double* __restrict__ w,
                                                           const unsigned char* __restrict__ mask, int64_t m,
                                                           double* __restrict__ wpack, double* __restrict__ spart) {
    __shared__ double red[3][256];
    double bb = 0.0, sb = 0.0, cnt = 0.0;
    for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < m; row += (int64_t)gridDim.x * 256) {
        const bool keep = mask ? (mask[row] != 0) : true;
        const double wv = keep ? w[row] : 0.0;
        const double wb = keep ? wv * b[row] : 0.0;
        d2u o;
        o[0] = wv;
        o[1] = wb;
        *reinterpret_cast<d2u*>(wpack + 2 * row) = o;
        bb = __builtin_fma(wb, wb, bb);
        sb += wb;
        cnt += keep ? 1.0 : 0.0;
    }
    red[0][threadIdx.x] = bb;
    red[1][threadIdx.x] = sb;
    red[2][threadIdx.x] = cnt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {        // fixed-order tree: deterministic
        if ((int)threadIdx.x < s) {
            red[0][threadIdx.x] += red[0][threadIdx.x + s];
            red[1][threadIdx.x] += red[1][threadIdx.x + s];
            red[2][threadIdx.x] += red[2][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double* o = spart + (size_t)blockIdx.x * 4;
        o[0] = red[0][0];
        o[1] = red[1][0];
        o[2] = red[2][0];
        o[3] = 0.0;
    }
}

// ---------------------------------------------------------------------------------
// Kernel 14: expand the weights of the TRAINING rows to one weight per row.  The reference's explicit-array quirk
// (svd.py:46, ridge.py:39: ``w`` is multiplied into ``a[training]`` without being masked) hands the solver one weight
// per training row; rank[row] = number of training rows before it (exclusive prefix sum of the mask, resident next to
// the mask).  Replaces a scatter of m doubles on the host and the upload of the non-training entries.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fsnap_expand_weights_k(const double* __restrict__ wtrain,
                                                             const unsigned char* __restrict__ mask,
                                                             const int* __restrict__ rank, int64_t m,
                                                             double* __restrict__ w) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row < m) w[row] = mask[row] ? wtrain[rank[row]] : 0.0;
}

// ---------------------------------------------------------------------------------
// host-side launchers (C++ linkage, used by fsnap_capi.cpp)
// ---------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------
// Kernel 12: packed statistics [G | c | scalars] + a compact copy of diag(G) from HBM into page-locked, coherent host
// memory (the "mirror" of fsnap_solve_device).  The single-GPU reduction kernel writes its mirror itself; this one
// serves statistics that were changed afterwards -- the all-reduced buffer of the multi-GPU path -- and replaces a
// D2H copy (SDMA launch latency + a blocking stream wait) by a 3 us kernel and an event the host polls.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fsnap_mirror_copy_k(const double* __restrict__ src, int K, double* __restrict__ mirror) {
    // Row i of G from the pair that holds its diagonal to the end of the row (the host solve reads the mirror as an upper
    // triangle: fsnap_solve_diag_upper), two columns per thread when the row base is 16-byte aligned; the last grid row
    // carries c, the scalars and the compact diagonal.  Half the PCIe writes of a full copy.
    const int i = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (i < K) {
        const int64_t base = (int64_t)i * K;
        const int j = (i & ~1) + 2 * t;
        if (j >= K) return;
        if (((base + j) & 1) == 0 && j + 1 < K) {
            *reinterpret_cast<d2*>(mirror + base + j) = *reinterpret_cast<const d2*>(src + base + j);
        } else {
            mirror[base + j] = src[base + j];
            if (j + 1 < K) mirror[base + j + 1] = src[base + j + 1];
        }
    } else {
        const int64_t n = (int64_t)K * K;
        for (int u = t; u < 2 * K + 3; u += (int)gridDim.x * 256)
            mirror[n + u] = u < K + 3 ? src[n + u] : src[(int64_t)(u - K - 3) * K + (u - K - 3)];
    }
}

// ---------------------------------------------------------------------------------
// Kernels 17 / 18: packed statistics [G | c | scalars] <-> [upper triangle of G, row-major | c | scalars].  The multi-GPU
// fit all-reduces K (K + 1) / 2 + K + 3 doubles instead of K^2 + K + 3 (20.4 -> 10.2 MB at K = 1595: the collective is
// most of what a wide fit costs beyond one GPU) and mirrors the triangle afterwards -- which also makes the reduced G
// symmetric to the last bit (two positions of a full buffer fall into different chunks of the ring and may be summed
// over the ranks in different orders).  One workgroup row per matrix row; the tail (c, scalars) rides in the last rows.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fsnap_tri_pack_k(const double* __restrict__ packed, int K, double* __restrict__ tri) {
    const int64_t i = blockIdx.y;
    const int64_t T = (int64_t)K * (K + 1) / 2;
    if (i < K) {
        const int64_t off = i * K - i * (i - 1) / 2 - i;          // tri index of (i, j) = off + j
        for (int64_t j = i + (int64_t)blockIdx.x * 256 + threadIdx.x; j < K; j += (int64_t)gridDim.x * 256)
            tri[off + j] = packed[i * K + j];
    } else {
        for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < K + 3; t += (int64_t)gridDim.x * 256)
            tri[T + t] = packed[(int64_t)K * K + t];
    }
}

__global__ __launch_bounds__(256) void fsnap_tri_unpack_k(const double* __restrict__ tri, int K, double* __restrict__ packed) {
    const int64_t i = blockIdx.y;
    const int64_t T = (int64_t)K * (K + 1) / 2;
    if (i < K) {
        const int64_t off = i * K - i * (i - 1) / 2 - i;
        for (int64_t j = i + (int64_t)blockIdx.x * 256 + threadIdx.x; j < K; j += (int64_t)gridDim.x * 256) {
            const double v = tri[off + j];
            packed[i * K + j] = v;
            packed[j * K + i] = v;
        }
    } else {
        for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < K + 3; t += (int64_t)gridDim.x * 256)
            packed[(int64_t)K * K + t] = tri[T + t];
    }
}

namespace fsnap {

hipError_t launch_weight_rows(const double* A, int64_t lda, const double* b, const double* w,
                              const unsigned char* mask, int64_t m, int K, double* aw, int64_t ldaw,
                              double* bw, hipStream_t st) {
    // one group of rows per wave, no grid-stride loop (tools/weight_rows_variants.hip, 10^6 x 128: 6.35 TB/s against
    // 5.25 TB/s for 2048 looping workgroups and 5.5-5.8 TB/s for a plain 16-byte copy of the same bytes in a loop)
    int lanes_log2 = 6;                         // lanes per row: the power of two that covers K / 2 column pairs, <= 64
    while (lanes_log2 > 2 && (1 << (lanes_log2 - 1)) * 2 >= K) --lanes_log2;
    const int64_t rows_per_wg = 4 * 4 * (int64_t)(64 >> lanes_log2);
    int64_t nb = (m + rows_per_wg - 1) / rows_per_wg;
    if (nb > (1ll << 30)) nb = 1ll << 30;   // the kernel's loop covers the rest
    if (nb < 1) nb = 1;
#define FSNAP_LAUNCH(LG)                                                                                        \
    hipLaunchKernelGGL((fsnap_weight_rows_k<LG>), dim3((unsigned)nb), dim3(256), 0, st, A, lda, b, w, mask, m, K, aw, \
                       ldaw, bw)
    switch (lanes_log2) {
        case 2: FSNAP_LAUNCH(2); break;
        case 3: FSNAP_LAUNCH(3); break;
        case 4: FSNAP_LAUNCH(4); break;
        case 5: FSNAP_LAUNCH(5); break;
        default: FSNAP_LAUNCH(6); break;
    }
#undef FSNAP_LAUNCH
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
    int lanes_log2 = 6;                         // lanes per row: the power of two that covers K / 2 column pairs, <= 64
    while (lanes_log2 > 2 && (1 << (lanes_log2 - 1)) * 2 >= K) --lanes_log2;
    const size_t lds = (size_t)4 * (64 >> lanes_log2) * ((K + 1) & ~1) * sizeof(double);
    if (lds > 160 * 1024 - 256) return hipErrorInvalidValue;
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)fsnap_gemvT_rows_k, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024 - 256);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(fsnap_gemvT_rows_k, dim3((unsigned)nb), dim3(256), lds, st, A, lda, u, m, K, lanes_log2, rpw, partial);
    hipLaunchKernelGGL(fsnap_colsum_partials_k, dim3((unsigned)((K + 15) / 16)), dim3(256), 0, st, partial, nb, K, out);
    return hipGetLastError();
}

// out[c] = sum_p partial[p][c] for an nparts x ncols table, fixed order (kernel fsnap_colsum_partials_k)
hipError_t launch_colsum(const double* partial, int nparts, int ncols, double* out, hipStream_t st) {
    hipLaunchKernelGGL(fsnap_colsum_partials_k, dim3((unsigned)((ncols + 15) / 16)), dim3(256), 0, st, partial, nparts, ncols, out);
    return hipGetLastError();
}

int pack_weights_num_blocks(int64_t m) {
    int64_t nb = (m + 2047) / 2048;
    if (nb > 512) nb = 512;
    if (nb < 1) nb = 1;
    return (int)nb;
}

hipError_t launch_pack_weights(const double* b, const double* w, const unsigned char* mask, int64_t m, double* wpack,
                               double* spart, hipStream_t st) {
    hipLaunchKernelGGL(fsnap_pack_weights_k, dim3((unsigned)pack_weights_num_blocks(m)), dim3(256), 0, st, b, w, mask, m,
                       wpack, spart);
    return hipGetLastError();
}

hipError_t launch_expand_weights(const double* wtrain, const unsigned char* mask, const int* rank, int64_t m, double* w,
                                 hipStream_t st) {
    const int64_t nb = (m + 255) / 256;
    if (nb > 0x7FFFFFFF) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fsnap_expand_weights_k, dim3((unsigned)nb), dim3(256), 0, st, wtrain, mask, rank, m, w);
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

int gemv_num_blocks(int64_t m) {
    int64_t nb = (m + 31) / 32;
    if (nb > 256 * 8) nb = 256 * 8;   // measured at 10^6 x 128: 2048 workgroups 6.9 TB/s, 8192: 6.8, one pass per wave: 5.8
    if (nb < 1) nb = 1;
    return (int)nb;
}

hipError_t launch_gemv_rows(const double* A, int64_t lda, const double* beta, int64_t m, int K, double* preds,
                            const double* b, const double* w, const unsigned char* mask, double* sse_part,
                            double* uout, hipStream_t st, bool uplain) {
    const int nb = gemv_num_blocks(m);
    hipLaunchKernelGGL(fsnap_gemv_rows_k, dim3((unsigned)nb), dim3(256), (size_t)K * sizeof(double), st, A, lda,
                       beta, m, K, preds, b, w, mask, sse_part, uout, uplain ? 1 : 0);
    return hipGetLastError();
}

// workgroups of the one-pass residual kernel: one resident round of the chip -- 256 CUs x the workgroups per CU the
// register budget of the instantiation admits (waves per SIMD: 8 / 7 / 5 / 4 / 4 / 3 / 2 for NJ = 1 / 2 / 3 / 4 / 5 / 6 / 8).
// Measured (scripts/residual_probe.py): 10^6 x 128 0.198 ms per call with 1536 workgroups, 0.188 with 1024, 0.199 with 512;
// 4 10^6 x 31 0.238 / 0.312 / 0.492.
int residual_num_blocks(int64_t m, int K) {
    static const int forced = [] {
        const char* e = getenv("FSNAP_RESIDUAL_BLOCKS");       // tuning aid
        const int v = e ? atoi(e) : 0;
        return v > 0 ? v : 0;
    }();
    const int nj = (K + 31) / 32;
    const int per_cu = nj <= 1 ? 8 : nj == 2 ? 7 : nj == 3 ? 5 : nj <= 5 ? 4 : nj == 6 ? 3 : 2;      // (7 ... 9: 2)
    const int cap = forced ? forced : 256 * per_cu;
    int64_t nb = (m + 31) / 32;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return (int)nb;
}

// fused refinement right-hand side (K <= 288): partial[residual_num_blocks(m)][K], sse_part[residual_num_blocks(m)] (or
// nullptr), out[K]
hipError_t launch_residual_rows(const double* A, int64_t lda, const double* beta, int64_t m, int K, const double* b,
                                const double* w, const unsigned char* mask, double* partial, double* sse_part, double* out,
                                hipStream_t st) {
    const int nb = residual_num_blocks(m, K);
    const int nj = (K + 31) / 32;
#define FSNAP_LAUNCH(NJ) \
    hipLaunchKernelGGL((fsnap_residual_rows_k<NJ>), dim3((unsigned)nb), dim3(256), 0, st, A, lda, beta, m, K, b, w, mask, partial, sse_part)
    switch (nj) {
        case 1: FSNAP_LAUNCH(1); break;
        case 2: FSNAP_LAUNCH(2); break;
        case 3: FSNAP_LAUNCH(3); break;
        case 4: FSNAP_LAUNCH(4); break;
        case 5: FSNAP_LAUNCH(5); break;
        case 6: FSNAP_LAUNCH(6); break;
        case 7: case 8: FSNAP_LAUNCH(8); break;
        case 9: FSNAP_LAUNCH(9); break;          // 257 ... 288 columns: the widths kernel 1Q still takes
        default: return hipErrorInvalidValue;
    }
#undef FSNAP_LAUNCH
    hipLaunchKernelGGL(fsnap_colsum_partials_k, dim3((unsigned)((K + 15) / 16)), dim3(256), 0, st, partial, nb, K, out);
    return hipGetLastError();
}

// packed [G | c | scalars] -> tri [upper triangle | c | scalars] (K (K + 1) / 2 + K + 3 doubles) and back (mirrors the triangle)
hipError_t launch_tri_pack(const double* packed, int K, double* tri, hipStream_t st) {
    hipLaunchKernelGGL(fsnap_tri_pack_k, dim3((unsigned)((K + 1023) / 1024), (unsigned)(K + 1)), dim3(256), 0, st, packed, K, tri);
    return hipGetLastError();
}
hipError_t launch_tri_unpack(const double* tri, int K, double* packed, hipStream_t st) {
    hipLaunchKernelGGL(fsnap_tri_unpack_k, dim3((unsigned)((K + 1023) / 1024), (unsigned)(K + 1)), dim3(256), 0, st, tri, K, packed);
    return hipGetLastError();
}

hipError_t launch_mirror_copy(const double* src, int K, double* mirror, hipStream_t st) {
    hipLaunchKernelGGL(fsnap_mirror_copy_k, dim3((unsigned)((K / 2 + 256) / 256), (unsigned)(K + 1)), dim3(256), 0, st, src, K, mirror);
    return hipGetLastError();
}

}  // namespace fsnap
