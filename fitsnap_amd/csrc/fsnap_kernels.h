// fsnap_kernels.h — internal C++ interface between the gfx950 kernels
// (fsnap_syrk.hip, fsnap_rows.hip, fsnap_chol.hip) and the C-ABI layer (fsnap_capi.cpp).  Not part of the public
// boundary; the public boundary is include/fsnap_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fsnap {

struct SyrkArgs {
    const double* A;            // device, row-major m x K, leading dimension lda
    int64_t lda;
    const double* b;            // device, m
    const double* w;            // device, m
    const unsigned char* mask;  // device, m (1 = training row) or nullptr
    int64_t m;
    int K;
    int nblocks;                // workgroups (4 row-waves each)
    int split;                  // (1)
    int64_t chunks_per_wave;    // 4-row chunks per wave
    bool nontemporal;           // use nt loads for the A stream
    double* part;               // [nblocks][NT][4][64]
    double* cpart;              // [nblocks*4][NB][16]
    double* spart;              // [nblocks*4][4]
    const double* wpack = nullptr;  // kernel 1A: packed (w_eff, w_eff * b) per row (launch_pack_weights)
    int* flow_words = nullptr;      // kernel 1QC: flow-control words, 4 ints per cluster (zero once, never reset)
    int flow_tag = 0;               // kernel 1QC: + 2^20 per launch
    bool fused_pack = false;        // kernels 1A / 1P: the kernel packs (w_eff, w_eff b) of its rows into LDS itself (b, w, mask,
                                    // spart instead of wpack; needs chunks_per_wave <= syrk_acc_max_fused_cpw())
};

struct TiledArgs {
    const double* A;
    int64_t lda;
    const double* wpack;        // (w_eff, w_eff b) per row (fsnap_pack_weights_k)
    int64_t m;
    int K;
    int NSB;                    // 64-column superblocks
    int npairs;                 // NSB*(NSB+1)/2
    int nsplit;                 // row splits
    int64_t chunks_per_split;   // 4-row chunks per split (each split = 4 waves)
    bool nontemporal;
    double* part;               // [nsplit*npairs][16][4][64]
    double* cpart;              // [(nsplit*NSB)*4][4][16]
    const double* spart;        // [ns][4]: partial b-only scalars of fsnap_pack_weights_k
    int ns;
};

int syrk_num_blocks(int K);
int syrk_waves_per_simd(int K, int split);
hipError_t launch_syrk_wave_p(const SyrkArgs& a, hipStream_t st);   // K <= 80, packed weights (a.wpack)
hipError_t launch_syrk_acc(const SyrkArgs& a, hipStream_t st);
// kernel 1Q (144 < K <= 288, fsnap_syrk_quad.hip): a.chunks_per_wave = chunks per WORKGROUP; pairs from a.wpack or (a.fused_pack,
// chunks per workgroup <= syrk_quad_max_cpg()) formed by the kernel
hipError_t launch_syrk_quad(const SyrkArgs& a, hipStream_t st);
int64_t syrk_quad_max_cpg();
// workgroups per cluster for K columns: 1 = kernel 1Q (K <= 288), 2 / 4 = kernel 1QC (289 ... 512 columns; a.nblocks = clusters,
// a.chunks_per_wave = chunks per cluster, fused packing only)
int syrk_quad_cluster(int K);
int64_t syrk_acc_max_fused_cpw();
// kernel 1S (fsnap_syrk_short.hip; 80 < K <= 144, short systems): a.nblocks = CHUNKS of a.chunks_per_wave ROWS (a multiple of 4),
// two workgroups per chunk; part[chunk][NT][4][64] | cpart[chunk][NB][16] | spart[chunk][4] (one c / scalar partial per chunk);
// pairs from a.wpack or (a.fused_pack) formed by the kernel, any chunk length (phases of syrk_short_phase_rows() rows)
hipError_t launch_syrk_short(const SyrkArgs& a, hipStream_t st);
bool syrk_short_takes(int K);
int syrk_short_phase_rows(int K);
int64_t syrk_wave_p_max_fused_cpw(int K, int wg_per_cu);   // kernel 1P: same for its (smaller, shared) LDS budget
// mirror: optional page-locked HOST buffer that receives the same packed statistics (zero-copy D2H)
// accumulate: out += statistics instead of out = statistics
// ns: number of scalar partials in spart (< 0: nblocks * cs_per_block, like the c partials)
// upper_mirror: the mirror receives the triangle at its upper positions only
hipError_t launch_reduce(const double* part, const double* cpart, const double* spart, int nblocks,
                         int cs_per_block, int ns, int K, double* out, double* mirror, bool accumulate, hipStream_t st,
                         bool upper_mirror = false);
int pack_weights_num_blocks(int64_t m);
// wpack[m][2] = (w_eff, w_eff * b); spart[pack_weights_num_blocks(m)][4] = partial b^T W^2 b, sum(w b), n_train, 0
hipError_t launch_pack_weights(const double* b, const double* w, const unsigned char* mask, int64_t m, double* wpack,
                               double* spart, hipStream_t st);
hipError_t launch_syrk_tiled(const TiledArgs& a, hipStream_t st);
hipError_t launch_reduce_tiled(const TiledArgs& a, double* out, bool accumulate, hipStream_t st);
hipError_t launch_weight_rows(const double* A, int64_t lda, const double* b, const double* w,
                              const unsigned char* mask, int64_t m, int K, double* aw, int64_t ldaw, double* bw,
                              hipStream_t st);
hipError_t launch_assemble(const double* raw, int64_t raw_ld, int64_t nrows, const int64_t* src_row, const int* kind,
                           const int* frac, const double* dval, const double* truth, const double* weight,
                           const double* fractions, const double* blank2J, int ntypes, int ncoeff, int off, double* A,
                           int64_t lda, double* b, double* w, hipStream_t st);
// fused assembly + accumulation (fsnap_fused.hip): b, w, 16-byte row records; then kernel 1T's partials from the raw batch
size_t assemble_row_record_bytes();
hipError_t launch_assemble_bw(const double* raw, int64_t raw_ld, int64_t nrows, const int64_t* src_row, const int* kind,
                              const int* frac, const double* d, const double* truth, const double* weight, int icolref,
                              double* b, double* w, void* recs, hipStream_t st);
hipError_t launch_assemble_syrk(const double* raw, int64_t raw_ld, const void* recs, const double* dval, const double* fractions,
                                const double* blank2J, int ntypes, int ncoeff, int off, const TiledArgs& a, hipStream_t st);
hipError_t launch_mirror_copy(const double* src, int K, double* mirror, hipStream_t st);
// packed [G | c | scalars] <-> [upper triangle of G row-major | c | scalars]: the payload of the multi-GPU all-reduce for wide systems
hipError_t launch_tri_pack(const double* packed, int K, double* tri, hipStream_t st);
hipError_t launch_tri_unpack(const double* tri, int K, double* packed, hipStream_t st);
// blocked Cholesky solve for large K: work (chol_large_work_doubles(K) doubles), dsc, z (np = K rounded up to 64),
// beta (K), status (1 int), minpiv (np / 64) are device scratch / outputs
// cvec: device right-hand side (NULL = the c part of packed)
size_t chol_large_work_doubles(int n);
hipError_t launch_chol_large(const double* packed, const double* cvec, int n, double alpha, double* work, double* dsc, double* z,
                             double* beta, int* status, double* minpiv, double* host_out, bool clear_status, double* probe_out,
                             hipStream_t st);
// probe_out (may be NULL): receives the CHOL_PROBES x CHOL_PROBES matrix Z^T Z of the probe vectors the strip carries (condition
// estimate, see fsnap_chol_probe_gram_k); the probe at (row, p), p = 1 .. CHOL_PROBES, is chol_probe(row, p)
constexpr int CHOL_PROBES = 31;
double chol_probe(int row, int p);
// one more right-hand side (n doubles, device) for the factor the last launch_chol_large of the same n left in `work`
hipError_t launch_chol_resolve(const double* d_rhs, int n, double* work, const double* dsc, double* z, double* beta, int* status,
                               const double* minpiv, double* host_out, hipStream_t st);
// factor only (pass factor of the row-space solve, K >= 384): R = chol(D^-1 G D^-1 + shift I) D for the n x n Gram matrix G in
// device memory, written as the K16 x K16 padded factor + inverse blocks that launch_trsm_rows reads (trsm_factor_doubles(K16)
// doubles at Rout); status: bit 0 non-finite input, bit 1 failed pivot (retry with a larger shift).  work / dsc / minpiv as
// for launch_chol_large.
hipError_t launch_chol_factor(const double* G, int n, double shift, double* work, double* dsc, int* status, double* minpiv,
                              int K16, double* Rout, hipStream_t st);
// out[0 .. n) = row maxima of |G - I|, out[n .. 2 n) = row sums of the squared Jacobi-scaled entries (active columns; NaN row
// maximum = non-finite input): the steering numbers of a row-space pass, so that G itself can stay in HBM
hipError_t launch_gram_scan(const double* G, int n, double* out, hipStream_t st);
int gemv_num_blocks(int64_t m);
hipError_t launch_gemv_rows(const double* A, int64_t lda, const double* beta, int64_t m, int K, double* preds,
                            const double* b, const double* w, const unsigned char* mask, double* sse_part,
                            double* uout, hipStream_t st, bool uplain = false);   // uplain: u = w (b - a.beta), not w^2 (...)
int gemvT_num_blocks(int64_t m);
int residual_num_blocks(int64_t m, int K);
// one-pass s = (wA)^T (wb - wA beta) for K <= 288 (kernels 4 + 7 fused): partial[residual_num_blocks(m, K)][K] scratch,
// sse_part[residual_num_blocks(m, K)] per-workgroup partial SSE (nullptr: none), out[K]
hipError_t launch_residual_rows(const double* A, int64_t lda, const double* beta, int64_t m, int K, const double* b,
                                const double* w, const unsigned char* mask, double* partial, double* sse_part, double* out,
                                hipStream_t st);
hipError_t launch_colsum(const double* partial, int nparts, int ncols, double* out, hipStream_t st);
// w[row] = mask[row] ? wtrain[rank[row]] : 0   (rank = exclusive prefix sum of the mask)
hipError_t launch_expand_weights(const double* wtrain, const unsigned char* mask, const int* rank, int64_t m, double* w,
                                 hipStream_t st);
int error_stats_num_blocks(int64_t m);
// pass 0: partial[grid][ncat][4] = n, n_w, sum t, sum w t; pass 1: partial[grid][ncat][6] (see kernel 9)
hipError_t launch_error_stats(const double* truth, const double* pred, const double* wgt, const int* cat, int64_t m, int ncat,
                              int pass, const double* means, double* partial, hipStream_t st);
hipError_t launch_gemvT_rows(const double* A, int64_t lda, const double* u, int64_t m, int K, double* partial,
                             double* out, hipStream_t st);

// Row-space solve (fsnap_trsm.hip).  Q <- X R^-1 by blocked substitution over the columns, one wave per 64 rows:
// first pass X = diag(w_eff) A (src = A, leading dimension lds, per-row pairs wpack = (w_eff, w_eff b); rows with
// w_eff = 0 become zero rows), later passes X = Q in place (src = Q, wpack = nullptr).  R: device, K16 x K16 row-major
// upper triangular, K16 = K rounded up to 16, identity in the padding, FOLLOWED by the inverses of its K16 / 16 diagonal
// 16 x 16 blocks ([block][16][16], row-major, upper triangular: what kernel 13B multiplies by and refines against):
// trsm_factor_doubles(K16) doubles in all, the second part filled by trsm_invert_diagonal_blocks on the host.
inline size_t trsm_factor_doubles(int K16) { return (size_t)K16 * K16 + (size_t)K16 * 16; }
inline void trsm_invert_diagonal_blocks(double* Rpad, int K16) {
    double* inv = Rpad + (size_t)K16 * K16;
    for (int jb = 0; jb < K16 / 16; ++jb) {
        const double* T = Rpad + (size_t)(jb * 16) * K16 + jb * 16;      // T[i][j] = T[i * K16 + j]
        double* X = inv + (size_t)jb * 256;
        // X = T^-1 column by column: T x_c = e_c, back substitution (x_c has zeros below row c)
        for (int c = 0; c < 16; ++c) {
            for (int i = 15; i >= 0; --i) {
                double v = (i == c) ? 1.0 : 0.0;
                for (int k = i + 1; k <= c; ++k) v -= T[(size_t)i * K16 + k] * X[k * 16 + c];
                X[i * 16 + c] = (i <= c) ? v / T[(size_t)i * K16 + i] : 0.0;
            }
        }
    }
}
hipError_t launch_trsm_rows(const double* src, int64_t lds, const double* wpack, double* Q, int64_t ldq, int64_t m, int K,
                            const double* R, int K16, hipStream_t st);
// qpack[row] = (w_eff != 0 ? 1 : 0, w_eff b): per-row pairs that make the SYRK kernels compute Q^T Q and Q^T (w b)
hipError_t launch_qpack(const double* wpack, int64_t m, double* qpack, hipStream_t st);

}  // namespace fsnap
