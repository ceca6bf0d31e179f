// fsnap_capi.cpp — the C-ABI layer of libfsnap_hip.so (declared in include/fsnap_hip.h).
// Owns the device-resident copy of the reference's shared arrays a / b / w
// (fitsnap3lib/parallel_tools.py:352-389, calculator.py:287-289), the training mask
// derived from fitsnap_dict['Testing'] (svd.py:35-40), launch geometry, HIP-event
// timing, and error text.  All compute is in fsnap_syrk.hip / fsnap_rows.hip / fsnap_chol.hip; the host K x K solve in
// fsnap_solve.cpp.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <queue>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fsnap_hip.h"
#include "fsnap_kernels.h"

// fsnap_solve with a contiguous copy of diag(G) (fsnap_solve.cpp; not part of the public ABI)
extern "C" int fsnap_solve_diag_tagged(int kind, double param, int64_t K, const double* G, const double* c, const double* diag,
                                       double* beta, int* rank, double* rcond_est, int upper, const void* owner,
                                       unsigned long long generation);
extern "C" int fsnap_solve_diag_upper(int kind, double param, int64_t K, const double* G, const double* c, const double* diag,
                                      double* beta, int* rank, double* rcond_est);
extern "C" int fsnap_solve_diag(int kind, double param, int64_t K, const double* G, const double* c, const double* diag,
                                double* beta, int* rank, double* rcond_est);
extern "C" void fsnap_cond_note(double min_pivot, double lambda_min, int steps, int where);
extern "C" double fsnap_gen_eig_max(const double* M, const double* N, int k);

#include "fsnap_ctx.h"
#include "fsnap_condest.h"

namespace fsnap {
std::string& library_error() {
    static thread_local std::string text;
    return text;
}
}  // namespace fsnap

#define g_last_error (fsnap::library_error())

namespace {

// process-wide fill counter of the host mirrors: a (context, generation) tag is never reused, not even by a context that is
// created at the address of a destroyed one
unsigned long long next_mirror_generation() {
    static std::atomic<unsigned long long> counter{0};
    return ++counter;
}

// the one rule for "fsnap_solve_device factorises this K x K system on the GPU" (blocked kernels 8a-8s): launch_normal_eq and
// fsnap_mirror_packed skip the host mirror then, fsnap_solve_device_rhs takes the device path
inline bool device_factor(const fsnap_ctx* ctx, int64_t K) {
    if (ctx->opt_device_solve == 2) return false;
    return K >= fsnap::DEVICE_CHOL_MIN_K || (K > 128 && ctx->opt_device_solve == 1);
}

struct Geometry {
    int nblocks, split, threads;
    int64_t cpw;
    int NB;
    bool acc;       // kernel 1A: whole triangle in one wave (accumulation registers), one wave per SIMD
    bool packed;    // kernel 1P (K <= 80)
    bool fused_pack = false;   // kernel 1A packs (w_eff, w_eff b) of its rows into LDS itself: no fsnap_pack_weights_k launch
    bool quad = false;         // kernel 1Q: the triangle dealt to the four waves of a workgroup (144 < K <= 288); cpw = chunks per workgroup
    bool shortk = false;       // kernel 1S (80 < K <= 144, short systems): nblocks = chunks, cpw = ROWS per chunk, two workgroups per chunk
    int cluster = 1;           // kernel 1QC (288 < K <= 512): workgroups per cluster; nblocks = clusters, cpw = chunks per cluster
};

// rows below which the tiled kernel keeps 145 ... 288 columns: kernel 1Q writes one partial triangle per workgroup
// (2 KiB x 55 ... 171 tiles), which short systems do not amortise
constexpr int64_t QUAD_MIN_ROWS = 8192;
constexpr int64_t QUAD_MIN_CPG = 24;        // fewest 4-row chunks per workgroup before the grid shrinks (profiles/r05_quad_min_cpg_ab.txt)
// widest system the accumulator-resident kernel 1A takes (NB = 9 column blocks: the ACE width 142 of examples/Ta_PACE_RIDGE)
constexpr int64_t ACC_MAX_K = 144;
// longest system kernel 1S takes by default, in staging phases (128 rows, 112 at NB = 9) per workgroup with half the CUs' worth of
// chunks (two workgroups share a chunk): 43 008 ... 49 152 rows on 256 CUs.  Measured (profiles/r06_short_kernel.txt, kernel us,
// 1S | 1A): 13 035 x 142 18.3 | 32.1, 20 000 x 142 22.9 | 35.1, 28 672 x 142 25.2 | 36.7, 40 000 x 142 32.0 | 37.0 (x 128: 27.7 | 32.3,
// x 96: 23.3 | 22.8), 60 000 x 142 43.5 | 41.3, 125 000 x 128 62.4 | 55.2
constexpr int64_t SHORT_MAX_PHASES = 3;
constexpr int64_t SHORT_MIN_CHUNK_ROWS = 32;
constexpr int64_t ACC_MIN_CPW = 12;         // kernel 1A: fewest 4-row chunks per row-wave before its grid shrinks below one workgroup per CU
// kernel 1QC (289 ... 512 columns on clusters of workgroups) against the tiled kernel, round 5 (profiles/r05_quadc_ab.txt):
// 500 000 x 368 1.19 against 1.24 ms, 367 900 x 480 1.40 / 1.38 ms (with 1.43 instead of 6.14 GB of HBM reads and a 17 instead of
// 42 us reduction: the complete fit 1.61 / 1.62 ms), 200 000 x 320 0.41 / 0.37, 100 000 x 512 0.51 / 0.43: a cluster's partial
// triangle (up to 1 MiB) and its prologue want ~5 000 rows per cluster to amortise
constexpr int64_t QUADC_MIN_ROWS = 300000;

// chunks per workgroup of kernel 1Q (per CLUSTER of kernel 1QC, 288 < K <= 512) for this context's rows, 0 = neither kernel
// takes them
int64_t quad_chunks_per_wg(const fsnap_ctx* ctx, int64_t* nblocks_out) {
    if (ctx->opt_tiled) return 0;
    if (ctx->K <= ACC_MAX_K || ctx->K > 512) return 0;
    const int cluster = fsnap::syrk_quad_cluster((int)ctx->K);
    // kernel 1QC exists with fused packing only: pairs brought by a row-space pass, or rows beyond the LDS, stay on the tiled kernel
    if (cluster > 1 && (!ctx->opt_fused_pack || ctx->wpack_override)) return 0;
    const int64_t min_rows = ctx->opt_quad_min_rows >= 0 ? ctx->opt_quad_min_rows : (cluster > 1 ? QUADC_MIN_ROWS : QUAD_MIN_ROWS);
    if (ctx->m < min_rows || ctx->m < 4) return 0;
    const int64_t nchunks = (ctx->m + 3) / 4;
    int64_t nblocks = ctx->opt_nblocks > 0 ? ctx->opt_nblocks : (int64_t)ctx->num_cu;
    nblocks /= cluster;                                  // clusters
    const int64_t max_blocks = (nchunks + QUAD_MIN_CPG - 1) / QUAD_MIN_CPG;
    if (nblocks > max_blocks) nblocks = max_blocks;
    if (nblocks < 1) nblocks = 1;
    const int64_t cpg = (nchunks + nblocks - 1) / nblocks;
    const int64_t off_limit = (int64_t)0xFFF00000;
    if (cpg > off_limit / (ctx->lda * 32)) return 0;    // a workgroup's rows beyond 32-bit buffer offsets: tiled kernel
    if (cluster > 1 && cpg > fsnap::syrk_quad_max_cpg()) return 0;
    nblocks = (nchunks + cpg - 1) / cpg;
    if (nblocks_out) *nblocks_out = nblocks;
    return cpg;
}

inline bool use_tiled(const fsnap_ctx* ctx) {
    if (ctx->opt_tiled) return true;
    if (ctx->K <= ACC_MAX_K) return false;
    return quad_chunks_per_wg(ctx, nullptr) == 0;
}

int plan_geometry(fsnap_ctx* ctx, Geometry* g) {
    const int K = (int)ctx->K;
    const int64_t m = ctx->m;
    g->NB = fsnap::syrk_num_blocks(K);
    const int64_t off_limit = (int64_t)0xFFF00000;  // 32-bit buffer offsets, 1 MiB of slack for prefetch overshoot
    g->acc = false;
    g->packed = false;
    if (g->NB >= 10) {
        // kernel 1Q (use_tiled() sent everything else of this width to the tiled kernel)
        int64_t nblocks = 0;
        const int64_t cpg = quad_chunks_per_wg(ctx, &nblocks);
        if (cpg <= 0) return ctx->fail(FSNAP_E_ARG, "no accumulator-resident kernel for %d columns", K);
        g->nblocks = (int)nblocks;
        g->cpw = cpg;
        g->split = 1;
        g->threads = 256;
        g->quad = true;
        g->cluster = fsnap::syrk_quad_cluster(K);
        // fused packing: the workgroup's per-row pairs must fit the LDS; the row-space passes bring pairs of their own
        g->fused_pack = ctx->opt_fused_pack && !ctx->wpack_override && cpg <= fsnap::syrk_quad_max_cpg();
        return FSNAP_OK;
    }
    if (g->NB >= 6 && ctx->opt_short != 0 && (ctx->opt_short == 1 || m <= SHORT_MAX_PHASES * (int64_t)fsnap::syrk_short_phase_rows(K) * (ctx->num_cu / 2)) &&
        ctx->lda * 8 * (int64_t)136 + 16 <= (int64_t)0xFFFFF000) {
        // kernel 1S: chunks of rows staged through LDS, the triangle dealt over the 16 waves of the two workgroups of a chunk
        int64_t want = ctx->opt_nblocks > 0 ? ctx->opt_nblocks : std::max<int64_t>(1, (int64_t)ctx->num_cu / 2);
        int64_t rpc = ((m + want - 1) / want + 3) / 4 * 4;
        if (rpc < SHORT_MIN_CHUNK_ROWS) rpc = SHORT_MIN_CHUNK_ROWS;
        const int64_t nchunk = (m + rpc - 1) / rpc;
        if (rpc <= 0x7FFFFFF0 && nchunk <= 0x3FFFFFF) {
            g->nblocks = (int)nchunk;
            g->cpw = rpc;
            g->split = 1;
            g->threads = 512;
            g->shortk = true;
            g->fused_pack = ctx->opt_fused_pack && !ctx->wpack_override;
            return FSNAP_OK;
        }
    }
    if (g->NB >= 6) {
        // kernel 1A: one 4-wave workgroup per CU, every wave streams its own rows and owns the whole triangle
        const int64_t nchunks = (m + 3) / 4;
        int64_t nblocks = ctx->opt_nblocks > 0 ? ctx->opt_nblocks : (int64_t)ctx->num_cu;
        const int64_t max_blocks = (nchunks + 4 * ACC_MIN_CPW - 1) / (4 * ACC_MIN_CPW);   // >= ACC_MIN_CPW chunks per row-wave
        if (nblocks > max_blocks) nblocks = max_blocks;
        if (nblocks < 1) nblocks = 1;
        int64_t cpw = (nchunks + nblocks * 4 - 1) / (nblocks * 4);
        const int64_t max_cpw = off_limit / (ctx->lda * 32);
        if (max_cpw < 1) return ctx->fail(FSNAP_E_ARG, "leading dimension %lld too large", (long long)ctx->lda);
        if (cpw > max_cpw) cpw = max_cpw;
        nblocks = (nchunks + cpw * 4 - 1) / (cpw * 4);
        if (nblocks > 0x7FFFFFF) return ctx->fail(FSNAP_E_ARG, "too many workgroups");
        g->nblocks = (int)nblocks;
        g->cpw = cpw;
        g->split = 1;
        g->threads = 256;
        g->acc = true;
        // fused packing: the workgroup's per-row pairs must fit the LDS; the row-space passes bring pairs of their own
        g->fused_pack = ctx->opt_fused_pack && !ctx->wpack_override && cpw <= fsnap::syrk_acc_max_fused_cpw();
        return FSNAP_OK;
    }
    // K <= 80: kernel 1P
    g->packed = true;
    g->split = 1;
    g->threads = 256;
    const int64_t nchunks = (m + 3) / 4;
    // workgroups resident per CU: waves per SIMD the register budget admits
    int wg_per_cu = fsnap::syrk_waves_per_simd(K, 1);
    if (wg_per_cu < 1) wg_per_cu = 1;
    int64_t nblocks = ctx->opt_nblocks > 0 ? ctx->opt_nblocks : (int64_t)ctx->num_cu * wg_per_cu;
    // keep >= 8 chunks (32 rows) per row-wave so the pipeline prologue amortises
    const int64_t max_blocks = (nchunks + 31) / 32;
    if (nblocks > max_blocks) nblocks = max_blocks;
    if (nblocks < 1) nblocks = 1;
    int64_t cpw = (nchunks + nblocks * 4 - 1) / (nblocks * 4);
    if (cpw < 1) cpw = 1;
    // 32-bit buffer offsets: a row-wave's byte range must stay below 4 GiB
    const int64_t max_cpw = off_limit / (ctx->lda * 32);
    if (max_cpw < 1) return ctx->fail(FSNAP_E_ARG, "leading dimension %lld too large", (long long)ctx->lda);
    if (cpw > max_cpw) cpw = max_cpw;
    nblocks = (nchunks + cpw * 4 - 1) / (cpw * 4);
    if (nblocks < 1) nblocks = 1;
    if (nblocks > 0x7FFFFFF) return ctx->fail(FSNAP_E_ARG, "too many workgroups");
    g->nblocks = (int)nblocks;
    g->cpw = cpw;
    // kernel 1P packs the pairs of its rows itself when they fit its LDS budget (contiguous chunk ranges only)
    g->fused_pack = ctx->opt_fused_pack && !ctx->wpack_override && cpw <= fsnap::syrk_wave_p_max_fused_cpw(K, wg_per_cu);
    return FSNAP_OK;
}

// Host memory -> device memory on the context's stream through the page-locked staging slots, WITHOUT a stream
// synchronisation: the bytes are copied into a slot in 1 MiB pieces, each piece's DMA starts as soon as it is in the
// slot (the memcpy of piece i + 1 overlaps the DMA of piece i), and the caller's buffer is free on return.  A slot is
// reused only after the event of its previous DMA has completed.  Up to two independent transfers per slot rotation
// (weights + mask); larger requests than a slot holds grow the slot.
// A few MB of host data (one weight per row, a mask, the rank array) -> device memory, ordered before everything launched later
// on the context's stream; the caller's buffer is free when the function returns.  Two ways:
//   1 staged    pieces of 1 MB through a page-locked slot: the caller's thread copies piece i + 1 while the DMA engine drains
//               piece i; returns when the last piece is queued.  Host time 0.23 ms for 8 MB (memcpy 0.16 + 8 DMA calls).
//   0 pageable  ONE hipMemcpyAsync from the caller's buffer and a wait for it (the runtime may read the buffer after the call
//               returns, so the wait is what frees it).  0.15 ms for 8 MB, 0.26 for 14 MB on the round-5 boxes
//               (profiles/r05_class_overhead.txt) -- the runtime pins the pages in place and the copy runs at PCIe speed,
//               where the staged form is bound by one core's memcpy.
// Which one is faster depends on the box (a pageable copy of 1 GB ran at 10 GB/s on one and 37 GB/s on another): from 1 MB on,
// the first three uploads go the pageable way and the next three the staged way, each timed until the data is on the device
// (the first of either kind not counted); the better one is kept for the context.  FSNAP_H2D_SMALL=pageable|staged fixes the choice.
static int h2d_forced() {
    static const int v = [] {
        const char* e = getenv("FSNAP_H2D_SMALL");
        if (!e) return -1;
        if (!strcmp(e, "pageable")) return 0;
        if (!strcmp(e, "staged")) return 1;
        return -1;
    }();
    return v;
}

int staged_h2d(fsnap_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return FSNAP_OK;
    int method = 1, probing = 0;
    if (bytes >= ((size_t)1 << 20)) {
        if (h2d_forced() >= 0) method = h2d_forced();
        else if (ctx->h2d_method >= 0) method = ctx->h2d_method;
        else {
            probing = 1;
            method = ctx->h2d_probes < 3 ? 0 : 1;
        }
    }
    if ((method == 0 || probing) && !ctx->h2d_ev &&
        hipEventCreateWithFlags(&ctx->h2d_ev, hipEventDisableTiming) != hipSuccess) {
        ctx->h2d_ev = nullptr;
        method = 1;
        probing = 0;
    }
    std::chrono::steady_clock::time_point t0;
    if (probing) {
        int wrc;
        if ((wrc = fsnap::wait_stream(ctx, nullptr, "upload probe"))) return wrc;      // time the copy, not what is queued before it
        t0 = std::chrono::steady_clock::now();
    }
    if (method == 0) {
        FSNAP_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream), "hipMemcpy(H2D)");
        FSNAP_HIP(hipEventRecord(ctx->h2d_ev, ctx->stream), "hipEventRecord");
        int wrc;
        if ((wrc = fsnap::wait_stream(ctx, ctx->h2d_ev, "upload"))) return wrc;
    } else {
        const int slot = ctx->wstage_next;
        ctx->wstage_next ^= 1;
        if (ctx->wstage_ev[slot]) {
            FSNAP_HIP(hipEventSynchronize(ctx->wstage_ev[slot]), "hipEventSynchronize(staging)");   // normally long done
        } else {
            FSNAP_HIP(hipEventCreateWithFlags(&ctx->wstage_ev[slot], hipEventDisableTiming), "hipEventCreate");
        }
        if (ctx->wstage_bytes[slot] < bytes) {
            if (ctx->wstage[slot]) (void)hipHostFree(ctx->wstage[slot]);
            ctx->wstage[slot] = nullptr;
            ctx->wstage_bytes[slot] = 0;
            if (hipHostMalloc((void**)&ctx->wstage[slot], bytes, hipHostMallocDefault) != hipSuccess)
                return ctx->fail(FSNAP_E_NOMEM, "hipHostMalloc(%zu) for the weight staging failed", bytes);
            ctx->wstage_bytes[slot] = bytes;
        }
        const size_t piece = (size_t)1 << 20;
        for (size_t off = 0; off < bytes; off += piece) {
            const size_t n = bytes - off < piece ? bytes - off : piece;
            memcpy(ctx->wstage[slot] + off, (const char*)src + off, n);
            FSNAP_HIP(hipMemcpyAsync((char*)dst + off, ctx->wstage[slot] + off, n, hipMemcpyHostToDevice, ctx->stream), "hipMemcpy(staged H2D)");
        }
        FSNAP_HIP(hipEventRecord(ctx->wstage_ev[slot], ctx->stream), "hipEventRecord");
        if (probing) {
            int wrc;
            if ((wrc = fsnap::wait_stream(ctx, ctx->wstage_ev[slot], "upload probe"))) return wrc;
        }
    }
    if (probing) {
        // per byte, so that a mask (1 byte per row) and the weights (8) of the same fit can share the probes
        const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / (double)bytes;
        const bool first_of_its_kind = ctx->h2d_probes == 0 || ctx->h2d_probes == 3;      // pays the runtime's first-use costs: not counted
        if (!first_of_its_kind && t < ctx->h2d_best[method]) ctx->h2d_best[method] = t;
        if (++ctx->h2d_probes >= 6) ctx->h2d_method = ctx->h2d_best[0] <= ctx->h2d_best[1] ? 0 : 1;
    }
    return FSNAP_OK;
}

// Rows of a host matrix -> device memory through two page-locked slots: the host copies piece i + 1 into one slot (a few
// threads, each a contiguous part) while the DMA engine drains piece i from the other.  A pageable hipMemcpyAsync of the
// 1 GB matrix of the headline shape ran at 10 GB/s on one box and 37 GB/s on another (the runtime pins or stages the
// user's pages itself, single-threaded); rows with a leading dimension wider than the row are packed on the way.
// Returns with every byte handed to the stream (the caller's buffer is free); not synchronised.
// probe (optional): after two pieces the host-side fill rate (bytes copied into the slots per second of memcpy time) is
// compared with min_fill_rate; below it the function stops, *rows_done says how far it got, and the caller sends the rest
// another way.
int staged_rows_h2d(fsnap_ctx* ctx, void* dst, const void* src, size_t rows, size_t row_bytes, size_t src_pitch,
                    double min_fill_rate = 0.0, size_t* rows_done = nullptr, double* fill_rate = nullptr) {
    const size_t total = rows * row_bytes;
    if (rows_done) *rows_done = 0;
    if (total == 0) return FSNAP_OK;
    const size_t slot_bytes = (size_t)16 << 20;
    double fill_s = 0.0;
    size_t fill_bytes = 0;
    int pieces = 0;
    for (int i = 0; i < 2; ++i) {
        if (!ctx->rstage[i] && hipHostMalloc((void**)&ctx->rstage[i], slot_bytes, hipHostMallocDefault) != hipSuccess) {
            ctx->rstage[i] = nullptr;
            return FSNAP_E_NOMEM;               // the caller falls back on the plain copy
        }
        if (!ctx->rstage_ev[i]) FSNAP_HIP(hipEventCreateWithFlags(&ctx->rstage_ev[i], hipEventDisableTiming), "hipEventCreate");
    }
    static const int nthreads = [] {
        const char* e = getenv("FSNAP_UPLOAD_THREADS");
        int n = e ? atoi(e) : 4;
        const unsigned hc = std::thread::hardware_concurrency();
        if (hc > 0 && n > (int)hc) n = (int)hc;
        return n < 1 ? 1 : (n > 16 ? 16 : n);
    }();
    const bool dense = src_pitch == row_bytes;
    size_t rows_per_piece = slot_bytes / row_bytes;
    if (rows_per_piece < 1) return FSNAP_E_NOMEM;      // a row wider than a slot: the caller falls back on the plain copy
    // `rstage_busy[slot]`: a DMA out of this slot was enqueued and its event not waited for since -- kept in the CONTEXT: a
    // call that returned early (a failed copy further on, a failing plan copy of the assembly before its stream
    // synchronisation) must not let the next call fill a slot the DMA engine is still reading
    bool* used = ctx->rstage_busy;
    int slot = 0;
    for (size_t r0 = 0; r0 < rows; r0 += rows_per_piece, slot ^= 1) {
        const size_t nr = rows - r0 < rows_per_piece ? rows - r0 : rows_per_piece;
        if (used[slot]) {
            FSNAP_HIP(hipEventSynchronize(ctx->rstage_ev[slot]), "hipEventSynchronize(row staging)");
            used[slot] = false;
        }
        char* stage = ctx->rstage[slot];
        const char* from = (const char*)src + r0 * src_pitch;
        auto copy_part = [&](size_t a, size_t b) {           // rows [a, b) of this piece
            if (dense) {
                memcpy(stage + a * row_bytes, from + a * row_bytes, (b - a) * row_bytes);
            } else {
                for (size_t r = a; r < b; ++r) memcpy(stage + r * row_bytes, from + r * src_pitch, row_bytes);
            }
        };
        const int nt = (nr * row_bytes >= ((size_t)4 << 20)) ? nthreads : 1;
        const auto tf0 = std::chrono::steady_clock::now();
        if (nt <= 1) {
            copy_part(0, nr);
        } else {
            std::vector<std::thread> th;
            th.reserve(nt - 1);
            for (int t = 1; t < nt; ++t) th.emplace_back(copy_part, nr * t / nt, nr * (t + 1) / nt);
            copy_part(0, nr / nt);
            for (auto& x : th) x.join();
        }
        fill_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - tf0).count();
        fill_bytes += nr * row_bytes;
        FSNAP_HIP(hipMemcpyAsync((char*)dst + r0 * row_bytes, stage, nr * row_bytes, hipMemcpyHostToDevice, ctx->stream),
                  "hipMemcpy(staged rows)");
        FSNAP_HIP(hipEventRecord(ctx->rstage_ev[slot], ctx->stream), "hipEventRecord");
        used[slot] = true;
        if (rows_done) *rows_done = r0 + nr;
        if (fill_rate) *fill_rate = fill_s > 0.0 ? (double)fill_bytes / fill_s : 0.0;
        if (++pieces == 2 && min_fill_rate > 0.0 && fill_s > 0.0 && (double)fill_bytes / fill_s < min_fill_rate) return FSNAP_OK;
    }
    return FSNAP_OK;
}

int check_rows(fsnap_ctx* ctx) {
    if (!ctx->dA || !ctx->db || ctx->m <= 0) return ctx->fail(FSNAP_E_STATE, "no rows: call fsnap_upload_rows/fsnap_bind_rows first");
    return FSNAP_OK;
}

int check_weights(fsnap_ctx* ctx) {
    if (!ctx->dw) return ctx->fail(FSNAP_E_STATE, "no weights: call fsnap_set_weights/fsnap_bind_weights first");
    return FSNAP_OK;
}

int ensure_ones(fsnap_ctx* ctx) {
    const size_t need = (size_t)ctx->m;
    if (ctx->ones.p && ctx->ones.bytes >= need) return FSNAP_OK;
    if (!ctx->ones.ensure(need)) return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(mask) failed");
    FSNAP_HIP(hipMemsetAsync(ctx->ones.p, 1, need, ctx->stream), "hipMemsetAsync(mask)");
    return FSNAP_OK;
}

struct TiledGeometry {
    int NSB, npairs, nsplit;
    int64_t cps;  // chunks per split
};

// event triple {before SYRK, after SYRK, after reduction} of the fit being launched: a ring, so that the kernel times
// of many consecutive fits can be read AFTER a timed loop instead of synchronising inside it
// Option timing_every = N: only every N-th launch is bracketed (0: none).  An event record between two dependent kernels
// costs ~5.6 us of idle stream on this runtime (rocprof timeline of the C1 shape: pack 5.0 | 5.6 | kernel 4.0 | 5.8 |
// reduce 4.3 us), i.e. 11 us of a 0.34 ms headline step; *slot stays null on the launches that are not sampled.
int fit_events(fsnap_ctx* ctx, hipEvent_t** slot) {
    *slot = nullptr;
    ctx->cur_events = nullptr;
    ++ctx->nlaunch;
    const int64_t phase = ctx->timing_phase++;         // 0 right after the option was set: that launch is sampled
    if (ctx->opt_timing_every <= 0 || phase % ctx->opt_timing_every != 0) return FSNAP_OK;
    hipEvent_t* sl = ctx->ring[ctx->nfit % fsnap_ctx::RING];
    for (int i = 0; i < 4; ++i)
        if (!sl[i]) FSNAP_HIP(hipEventCreate(&sl[i]), "hipEventCreate");
    ctx->ring_comm[ctx->nfit % fsnap_ctx::RING] = false;
    ++ctx->nfit;
    *slot = sl;
    ctx->cur_events = sl;
    return FSNAP_OK;
}

// Row splits of the tiled kernel.  Work items (split, superblock pair) are NOT equal: an off-diagonal pair runs 16
// MFMAs per chunk, a diagonal pair 10, pairs with the (half-empty) last superblock 8 / 3, and two workgroups per CU
// are resident -- with one or two rounds of items the short ones leave their slots idle (367 900 x 480: 27 splits
// 1.84 ms, 57 splits 1.60 ms).  So the split count comes from a small model: greedy list scheduling of one XCD's
// item range on its slots (the hardware dispatches the next workgroup when a slot frees up) + the reduction of the
// partial triangles; it reproduces the measured kernel times within ~5 %.  The result is cached per shape.
double tiled_makespan_units(const std::vector<int>& tiles, int64_t n, int64_t chunks_per_wave, int64_t groups, int64_t slots,
                            double ovh) {
    const int64_t npairs = (int64_t)tiles.size();
    const int64_t nitems = npairs * n;
    const int64_t per = (nitems + groups - 1) / groups;
    std::priority_queue<double, std::vector<double>, std::greater<double>> free_at;
    for (int64_t i = 0; i < slots; ++i) free_at.push(0.0);
    double makespan = 0.0;
    for (int64_t it = 0; it < per && it < nitems; ++it) {
        const double t0 = free_at.top();
        free_at.pop();
        // ovh: prologue + 4-wave fold + partial store, in units of one chunk of one tile
        const double t1 = t0 + ovh + (double)chunks_per_wave * tiles[(size_t)(it % npairs)];
        free_at.push(t1);
        if (t1 > makespan) makespan = t1;
    }
    return makespan;
}

// geometry of the tiled kernel for m rows of K columns (leading dimension lda).  Pure function of the shape and the context's options.
int tiled_geometry(fsnap_ctx* ctx, int64_t m, int64_t K, int64_t lda, TiledGeometry* g) {
    if (K > 32768) return ctx->fail(FSNAP_E_ARG, "K = %lld too large", (long long)K);
    g->NSB = (int)((K + 63) / 64);
    g->npairs = g->NSB * (g->NSB + 1) / 2;
    const int64_t nchunks = (m + 3) / 4;
    const int64_t bytes = m * lda * 8;
    // keep a split's rows resident in the Infinity Cache (256 MiB) while all pairs sweep them
    const int64_t min_split_cache = (bytes + (96ll << 20) - 1) / (96ll << 20);
    // work items of one split and their cost in MFMA tiles per chunk
    const int tail = (int)(K & 63);
    const bool half = tail != 0 && tail <= 32;                   // last superblock: second 32-column group empty
    std::vector<int> tiles;
    for (int I = 0; I < g->NSB; ++I)                 // kernel 1T's item order: off-diagonal pairs, then the diagonal
        for (int J = I + 1; J < g->NSB; ++J) tiles.push_back((half && J == g->NSB - 1) ? 8 : 16);
    for (int I = 0; I < g->NSB; ++I) tiles.push_back((half && I == g->NSB - 1) ? 3 : 10);
    int64_t nsplit = ctx->opt_nsplit;
    if (nsplit <= 0) {
        const int64_t groups = 8;         // contiguous work-item ranges per XCD
        // two workgroups (8 waves) per CU share the matrix pipe, 66 ns per tile and chunk at ~80 % issue
        const int64_t slots = (int64_t)ctx->num_cu * 2 / groups;
        const double unit = 66.0e-9, ovh = 128.0;
        const int64_t part_bytes = (int64_t)g->npairs * 32768;                        // partial triangles of one split
        int64_t n_hi = std::max<int64_t>(1, std::min<int64_t>(1024, nchunks / 128));  // >= 32 chunks per wave
        n_hi = std::min(n_hi, std::max<int64_t>(1, (256ll << 20) / part_bytes));      // <= 256 MiB of partials
        const int64_t n_lo = std::min(n_hi, std::max<int64_t>(1, min_split_cache));
        double best = 1.0e300;
        for (int64_t n = n_lo; n <= n_hi; ++n) {
            const int64_t cps = (nchunks + n - 1) / n;
            const int64_t cpw = (cps + 3) / 4;
            // Long items drift apart: the pairs of a split start together and share every row segment through their
            // XCD's L2, but they cost 16 / 10 / 8 / 3 tiles per chunk, and the longer they run the less of a split's rows is
            // still in L2 when the slower ones get there (367 900 x 480: 7.4 GB fetched per launch for a 1.4 GB matrix, the
            // kernel at 5.2 TB/s of fabric traffic as much memory- as MFMA-bound).  Measured with the split count forced
            // (round 3): the kernel time grows by ~2e-4 per chunk a wave owns -- 367 900 x 480: 64 splits 1.44 ms, 128
            // 1.40, 192 1.37; 200 000 x 1 000: 23 splits 3.83 ms, 64 3.48 -- and by a quarter of that when the whole
            // matrix stays in the Infinity Cache (15 213 x 1 595).
            // (scaled by how much there is to share: a row segment is wanted by NSB pairs -- with 3 superblocks at most 3 x
            // the matrix can be fetched, and many splits only buy a long reduction: 1 772 880 x 142 took 744)
            const double share = std::min(1.0, (double)(g->NSB - 1) / 7.0);
            const double drift = 1.0 + share * (bytes > (200ll << 20) ? 2.0e-4 : 0.5e-4) * (double)cpw;
            const double cost = unit * drift * tiled_makespan_units(tiles, n, cpw, groups, slots, ovh) + (double)n * (double)part_bytes / 3.0e12;
            if (cost < best) {
                best = cost;
                nsplit = n;
            }
        }
    }
    // >= 8 chunks per wave (4 waves per split)
    const int64_t max_split = (nchunks + 31) / 32;
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    int64_t cps = (nchunks + nsplit - 1) / nsplit;
    cps = (cps + 3) / 4 * 4;
    // 32-bit buffer offsets per wave
    const int64_t max_cpw = (int64_t)0xFFF00000 / (lda * 32);
    if (max_cpw < 1) return ctx->fail(FSNAP_E_ARG, "leading dimension %lld too large", (long long)lda);
    if (cps / 4 > max_cpw) cps = max_cpw * 4;
    nsplit = (nchunks + cps - 1) / cps;
    if ((int64_t)g->npairs * nsplit > 0x7FFFFFFF) return ctx->fail(FSNAP_E_ARG, "too many workgroups");
    g->nsplit = (int)nsplit;
    g->cps = cps;
    return FSNAP_OK;
}

// the same for the resident rows, cached per shape
int plan_tiled(fsnap_ctx* ctx, TiledGeometry* g) {
    const int64_t m = ctx->m, K = ctx->K;
    if (ctx->tplan_valid && ctx->tplan_key[0] == m && ctx->tplan_key[1] == K && ctx->tplan_key[2] == ctx->lda &&
        ctx->tplan_key[3] == ctx->opt_nsplit) {
        g->NSB = ctx->tplan[0];
        g->npairs = ctx->tplan[1];
        g->nsplit = ctx->tplan[2];
        g->cps = ctx->tplan_cps;
        return FSNAP_OK;
    }
    int rc;
    if ((rc = tiled_geometry(ctx, m, K, ctx->lda, g))) return rc;
    ctx->tplan_key[0] = m;
    ctx->tplan_key[1] = K;
    ctx->tplan_key[2] = ctx->lda;
    ctx->tplan_key[3] = ctx->opt_nsplit;
    ctx->tplan[0] = g->NSB;
    ctx->tplan[1] = g->npairs;
    ctx->tplan[2] = g->nsplit;
    ctx->tplan_cps = g->cps;
    ctx->tplan_valid = true;
    return FSNAP_OK;
}

// (w_eff, w_eff b) per row for kernels 1A / 1T, packed once per (b, w, mask) -- every time if any of them lives in
// memory the caller owns (fsnap_bind_rows / fsnap_bind_weights: it may have changed without notice).  *npk = number
// of partial b-only scalars in ctx->wpack_spart.
int ensure_wpack(fsnap_ctx* ctx, int* npk) {
    *npk = fsnap::pack_weights_num_blocks(ctx->m);
    if (ctx->wpack_override) {                     // rows and per-row pairs of the row-space passes (normal_eq_launch_on)
        if (!ctx->wpack_spart.ensure((size_t)*npk * 4 * 8)) return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(packed weights) failed");
        return FSNAP_OK;
    }
    if (!ctx->wpack.ensure((size_t)ctx->m * 16 + 64) || !ctx->wpack_spart.ensure((size_t)*npk * 4 * 8))
        return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(packed weights) failed");
    const bool caller_owned = ctx->db != (const double*)ctx->ownb.p || ctx->dw != (const double*)ctx->ownw.p ||
                              (ctx->dmask && ctx->dmask != (const unsigned char*)ctx->ownmask.p);
    if (!ctx->wpack_valid || caller_owned || ctx->opt_repack) {
        FSNAP_HIP(fsnap::launch_pack_weights(ctx->db, ctx->dw, ctx->dmask, ctx->m, (double*)ctx->wpack.p,
                                             (double*)ctx->wpack_spart.p, ctx->stream),
                  "launch fsnap_pack_weights_k");
        ctx->wpack_valid = true;
    }
    return FSNAP_OK;
}

int launch_normal_eq_tiled(fsnap_ctx* ctx, double* d_packed, bool accumulate) {
    int rc;
    TiledGeometry g;
    if ((rc = plan_tiled(ctx, &g))) return rc;
    int npk = 0;
    if ((rc = ensure_wpack(ctx, &npk))) return rc;
    if (!ctx->part.ensure((size_t)g.nsplit * g.npairs * 4096 * sizeof(double)) ||
        !ctx->cpart.ensure((size_t)g.nsplit * g.NSB * 4 * 64 * sizeof(double)))
        return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(partials) failed");
    fsnap::TiledArgs a;
    a.A = ctx->dA;
    a.lda = ctx->lda;
    a.wpack = ctx->wpack_override ? ctx->wpack_override : (const double*)ctx->wpack.p;
    a.m = ctx->m;
    a.K = (int)ctx->K;
    a.NSB = g.NSB;
    a.npairs = g.npairs;
    a.nsplit = g.nsplit;
    a.chunks_per_split = g.cps;
    a.nontemporal = false;  // rows are re-read by the other column pairs: keep them cached
    a.part = (double*)ctx->part.p;
    a.cpart = (double*)ctx->cpart.p;
    a.spart = (const double*)ctx->wpack_spart.p;
    a.ns = npk;
    // every c slot (split, superblock, wave) is rewritten by the diagonal pair of its split on every launch (also by
    // waves whose row range is empty: they store zeros), so the buffer is cleared only when its geometry changes -- a
    // memset per fit was a 4 us kernel plus a stream boundary in front of a 27 us SYRK at the ACE shape 13 035 x 142
    const int64_t cpart_key = ((int64_t)g.nsplit << 32) | (int64_t)g.NSB;
    if (ctx->cpart_key != cpart_key || ctx->cpart_ptr != ctx->cpart.p) {
        FSNAP_HIP(hipMemsetAsync(a.cpart, 0, (size_t)g.nsplit * g.NSB * 4 * 64 * sizeof(double), ctx->stream), "hipMemsetAsync");
        ctx->cpart_key = cpart_key;
        ctx->cpart_ptr = ctx->cpart.p;
    }
    hipEvent_t* evs;
    if ((rc = fit_events(ctx, &evs))) return rc;
    if (evs) FSNAP_HIP(hipEventRecord(evs[0], ctx->stream), "hipEventRecord");
    FSNAP_HIP(fsnap::launch_syrk_tiled(a, ctx->stream), "launch fsnap_syrk_tiled");
    if (evs) FSNAP_HIP(hipEventRecord(evs[1], ctx->stream), "hipEventRecord");
    FSNAP_HIP(fsnap::launch_reduce_tiled(a, d_packed, accumulate, ctx->stream), "launch fsnap_reduce_tiled");
    if (evs) FSNAP_HIP(hipEventRecord(evs[2], ctx->stream), "hipEventRecord");
    if (evs) ctx->t_syrk = true;
    return FSNAP_OK;
}

int launch_normal_eq(fsnap_ctx* ctx, double* d_packed, bool want_mirror = false, bool accumulate = false) {
    int rc;
    ctx->mirror_of = nullptr;
    ctx->chol_factor_of = nullptr;              // the statistics are about to change: the factor on the device is of the old ones
    if ((rc = check_rows(ctx)) || (rc = check_weights(ctx))) return rc;
    if (use_tiled(ctx)) return launch_normal_eq_tiled(ctx, d_packed, accumulate);
    Geometry g;
    if ((rc = plan_geometry(ctx, &g))) return rc;
    const unsigned char* mask = ctx->dmask;
    if (!mask) {
        if ((rc = ensure_ones(ctx))) return rc;
        mask = (const unsigned char*)ctx->ones.p;
    }
    const int NT = g.NB * (g.NB + 1) / 2;
    const int cs_per_block = g.shortk ? 1 : 4 * g.cluster;   // c / scalar partials per workgroup (kernel 1QC: 4 per member; kernel 1S: one per chunk)
    if (!ctx->part.ensure((size_t)g.nblocks * NT * 256 * sizeof(double)) ||
        !ctx->cpart.ensure((size_t)g.nblocks * cs_per_block * g.NB * 16 * sizeof(double)) ||
        !ctx->spart.ensure((size_t)g.nblocks * cs_per_block * 4 * sizeof(double)))
        return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(partials) failed");
    fsnap::SyrkArgs a;
    a.A = ctx->dA;
    a.lda = ctx->lda;
    a.b = ctx->db;
    a.w = ctx->dw;
    a.mask = mask;
    a.m = ctx->m;
    a.K = (int)ctx->K;
    a.nblocks = g.nblocks;
    a.split = g.split;
    a.chunks_per_wave = g.cpw;
    a.nontemporal = true;
    a.part = (double*)ctx->part.p;
    a.cpart = (double*)ctx->cpart.p;
    a.spart = (double*)ctx->spart.p;
    int ns = -1;                      // scalar partials: from the SYRK kernel, or (kernel 1A) from the weight packing
    const double* spart_src = a.spart;
    if (g.fused_pack) {
        a.fused_pack = true;          // b, w, mask -> pairs in LDS + the b-only scalars per row-wave, inside the SYRK launch
    } else {
        int npk = 0;
        if ((rc = ensure_wpack(ctx, &npk))) return rc;
        a.wpack = ctx->wpack_override ? ctx->wpack_override : (const double*)ctx->wpack.p;
        spart_src = (const double*)ctx->wpack_spart.p;
        ns = npk;
    }
    hipEvent_t* evs;
    if ((rc = fit_events(ctx, &evs))) return rc;
    if (evs) FSNAP_HIP(hipEventRecord(evs[0], ctx->stream), "hipEventRecord");
    if (g.quad && g.cluster > 1) {
        const size_t need = (size_t)g.nblocks * 4 * sizeof(int);
        if (ctx->quad_flow.bytes < need) {
            if (!ctx->quad_flow.ensure(need)) return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(flow words) failed");
            FSNAP_HIP(hipMemsetAsync(ctx->quad_flow.p, 0, ctx->quad_flow.bytes, ctx->stream), "hipMemset(flow words)");
            ctx->quad_flow_tag = 0;
        }
        ctx->quad_flow_tag += 1u << 20;                                  // (unsigned: wraps after 4096 launches, see quad_flow_wait)
        a.flow_words = (int*)ctx->quad_flow.p;
        // low byte: flow-control mode (2 bits) and lead (6 bits).  Lead 63 = by cluster size, as measured (HBM reads / algorithmic):
        // clusters of 4 (367 900 x 480) lead 2: 1.01 x, lead 4: 1.5 x; clusters of 2 (500 000 x 368: twice as many clusters share an
        // XCD's L2) lead 0: 1.01 x, lead 1: 1.33 x, lead 2: 1.78 x -- at the same kernel time in every case
        const int flow = 2 | ((g.cluster == 2 ? 0 : 2) << 2);      // mode 2: publish at the start of a trip, judge at its end
        a.flow_tag = (int)(ctx->quad_flow_tag | (unsigned)flow);
    }
    if (g.shortk) FSNAP_HIP(fsnap::launch_syrk_short(a, ctx->stream), "launch fsnap_syrk_short");
    else if (g.quad) FSNAP_HIP(fsnap::launch_syrk_quad(a, ctx->stream), "launch fsnap_syrk_quad");
    else if (g.acc) FSNAP_HIP(fsnap::launch_syrk_acc(a, ctx->stream), "launch fsnap_syrk_acc");
    else FSNAP_HIP(fsnap::launch_syrk_wave_p(a, ctx->stream), "launch fsnap_syrk_wave_p");
    if (evs) FSNAP_HIP(hipEventRecord(evs[1], ctx->stream), "hipEventRecord");
    // K <= 128 and the context owns the output: the reduction also writes a page-locked host mirror, so the solve
    // needs no D2H copy (the copy's launch latency was 12 us of a 440 us step)
    double* mirror = nullptr;
    // (a system the GPU factorises -- fsnap_solve_device_rhs's rule -- needs no mirror: DEVICE_CHOL_MIN_K columns and more)
    if (want_mirror && !device_factor(ctx, ctx->K)) {
        const size_t need = ((size_t)FSNAP_PACKED_LEN(ctx->K) + (size_t)ctx->K) * 8;   // + compact diagonal
        if (ctx->mirror_bytes < need) {
            if (ctx->mirror) (void)hipHostFree(ctx->mirror);
            ctx->mirror = nullptr;
            ctx->mirror_bytes = 0;
            // coherent (fine-grained) host memory: the kernel's stores are visible to the host once the event has completed
            if (hipHostMalloc((void**)&ctx->mirror, need, hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess)
                ctx->mirror_bytes = need;
        }
        if (ctx->mirror && !ctx->mirror_ev && hipEventCreateWithFlags(&ctx->mirror_ev, hipEventDisableTiming) != hipSuccess)
            ctx->mirror_ev = nullptr;
        if (ctx->mirror && ctx->mirror_ev) mirror = ctx->mirror;
    }
    const bool upper_mirror = mirror != nullptr;      // the mirror's triangle is written once per element, at its upper position
    FSNAP_HIP(fsnap::launch_reduce(a.part, a.cpart, spart_src, g.nblocks, cs_per_block, ns, a.K, d_packed, mirror, accumulate,
                                   ctx->stream, upper_mirror),
              "launch fsnap_reduce_partials");
    if (evs) FSNAP_HIP(hipEventRecord(evs[2], ctx->stream), "hipEventRecord");
    if (mirror) {
        FSNAP_HIP(hipEventRecord(ctx->mirror_ev, ctx->stream), "hipEventRecord");
        ctx->mirror_of = d_packed;
        ctx->mirror_K = ctx->K;
        ctx->mirror_upper = upper_mirror;
        ctx->mirror_gen = next_mirror_generation();
    }
    if (evs) ctx->t_syrk = true;
    return FSNAP_OK;
}

}  // namespace

namespace fsnap {

int wpack_current(fsnap_ctx* ctx) {
    int rc, npk = 0;
    if ((rc = check_rows(ctx)) || (rc = check_weights(ctx))) return rc;
    return ensure_wpack(ctx, &npk);
}

int normal_eq_launch_on(fsnap_ctx* ctx, const double* Q, int64_t ldq, const double* qpack, double* d_packed) {
    const double* const save_A = ctx->dA;
    const int64_t save_lda = ctx->lda;
    ctx->dA = Q;
    ctx->lda = ldq;
    ctx->wpack_override = qpack;
    const int rc = launch_normal_eq(ctx, d_packed);
    ctx->wpack_override = nullptr;
    ctx->dA = save_A;
    ctx->lda = save_lda;
    return rc;
}

}  // namespace fsnap

extern "C" {

int fsnap_version(void) { return 200; }

int fsnap_device_count(int* count) {
    if (!count) return FSNAP_E_ARG;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        g_last_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e);
        *count = 0;
        return FSNAP_E_HIP;
    }
    *count = n;
    return FSNAP_OK;
}

const char* fsnap_last_error(const fsnap_ctx* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int fsnap_ctx_create(int device, fsnap_ctx** out) {
    if (!out) return FSNAP_E_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_last_error = std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return FSNAP_E_HIP;
    }
    if (device < 0 || device >= n) {
        g_last_error = "device index out of range";
        return FSNAP_E_ARG;
    }
    fsnap_ctx* ctx = new (std::nothrow) fsnap_ctx();
    if (!ctx) return FSNAP_E_NOMEM;
    ctx->device = device;
    if ((e = hipSetDevice(device)) != hipSuccess) {
        g_last_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
        delete ctx;
        return FSNAP_E_HIP;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            g_last_error = std::string("libfsnap_hip is built for gfx950 (MI355X) only; device is ") + prop.gcnArchName;
            delete ctx;
            return FSNAP_E_HIP;
        }
    }
    if ((e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess) {
        g_last_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
        delete ctx;
        return FSNAP_E_HIP;
    }
    ctx->stream = ctx->own_stream;
    for (auto& ev : ctx->ev) {
        if ((e = hipEventCreate(&ev)) != hipSuccess) {
            g_last_error = std::string("hipEventCreate: ") + hipGetErrorString(e);
            fsnap_ctx_destroy(ctx);
            return FSNAP_E_HIP;
        }
    }
    *out = ctx;
    return FSNAP_OK;
}

int fsnap_ctx_destroy(fsnap_ctx* ctx) {
    if (!ctx) return FSNAP_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream && !ctx->comm_broken) (void)hipStreamSynchronize(ctx->stream);
    (void)fsnap_comm_destroy(ctx);                    // a broken communicator is aborted: its stuck kernel leaves the stream
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    fsnap::rowspace_release(ctx);
    ctx->commbuf.release();
    DevBuf* bufs[] = {&ctx->ownA, &ctx->ownb, &ctx->ownw, &ctx->ownmask, &ctx->ones, &ctx->part, &ctx->cpart,
                      &ctx->spart, &ctx->packed, &ctx->beta, &ctx->preds, &ctx->sse, &ctx->aw, &ctx->bw,
                      &ctx->st_raw, &ctx->st_plan, &ctx->st_frac, &ctx->st_blank, &ctx->dsolve, &ctx->dchol, &ctx->dcat, &ctx->dstat, &ctx->wpack, &ctx->wpack_spart,
                      &ctx->du, &ctx->dspart, &ctx->dsvec};
    for (DevBuf* b : bufs) b->release();
    for (int i = 0; i < 2; ++i) {
        if (ctx->rstage[i]) (void)hipHostFree(ctx->rstage[i]);
        if (ctx->rstage_ev[i]) (void)hipEventDestroy(ctx->rstage_ev[i]);
        if (ctx->wstage[i]) (void)hipHostFree(ctx->wstage[i]);
        if (ctx->wstage_ev[i]) (void)hipEventDestroy(ctx->wstage_ev[i]);
    }
    ctx->wtrain.release();
    ctx->wrank.release();
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->mirror) (void)hipHostFree(ctx->mirror);
    if (ctx->mirror_ev) (void)hipEventDestroy(ctx->mirror_ev);
    if (ctx->h2d_ev) (void)hipEventDestroy(ctx->h2d_ev);
    if (ctx->chol_host) (void)hipHostFree(ctx->chol_host);
    if (ctx->chol_ev) (void)hipEventDestroy(ctx->chol_ev);
    for (auto& ev : ctx->ev)
        if (ev) (void)hipEventDestroy(ev);
    for (auto& sl : ctx->ring)
        for (auto& ev : sl)
            if (ev) (void)hipEventDestroy(ev);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return FSNAP_OK;
}

int fsnap_ctx_set_stream(fsnap_ctx* ctx, void* hip_stream) {
    if (!ctx) return FSNAP_E_ARG;
    FSNAP_HIP(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    ctx->stream = (hipStream_t)hip_stream;  // NULL = the HIP legacy default stream
    return FSNAP_OK;
}

int fsnap_ctx_use_own_stream(fsnap_ctx* ctx) {
    if (!ctx) return FSNAP_E_ARG;
    FSNAP_HIP(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    ctx->stream = ctx->own_stream;
    return FSNAP_OK;
}

int fsnap_set_option(fsnap_ctx* ctx, const char* key, int64_t value) {
    if (!ctx || !key) return FSNAP_E_ARG;
    if (!strcmp(key, "device_solve")) {
        if (value < 0 || value > 2) return ctx->fail(FSNAP_E_ARG, "device_solve must be 0 (auto), 1 (always) or 2 (never)");
        ctx->opt_device_solve = (int)value;
    } else if (!strcmp(key, "tiled")) {
        ctx->opt_tiled = value != 0;
    } else if (!strcmp(key, "short")) {
        if (value < -1 || value > 1) return ctx->fail(FSNAP_E_ARG, "short must be -1 (by the row count), 0 (never) or 1 (always)");
        ctx->opt_short = (int)value;
    } else if (!strcmp(key, "timing_every")) {
        if (value < 0 || value > (1 << 20)) return ctx->fail(FSNAP_E_ARG, "timing_every out of range");
        ctx->opt_timing_every = (int)value;
        ctx->timing_phase = 0;
    } else if (!strcmp(key, "repack")) {
        ctx->opt_repack = value != 0;
    } else if (!strcmp(key, "comm_timeout")) {
        if (value < 0 || value > 86400) return ctx->fail(FSNAP_E_ARG, "comm_timeout must be 0 (FSNAP_COMM_TIMEOUT) ... 86400 seconds");
        ctx->opt_comm_timeout = (int)value;
    } else if (!strcmp(key, "chol_reuse")) {
        ctx->opt_chol_reuse = value != 0;
        ctx->chol_factor_of = nullptr;
    } else if (!strcmp(key, "rowspace_reuse_stats")) {
        ctx->opt_rowspace_reuse = value != 0;
    } else if (!strcmp(key, "quad_min_rows")) {
        if (value < -1) return ctx->fail(FSNAP_E_ARG, "quad_min_rows must be >= -1");
        ctx->opt_quad_min_rows = value;
    } else if (!strcmp(key, "reduce_triangle")) {
        if (value < -1 || value > 1) return ctx->fail(FSNAP_E_ARG, "reduce_triangle must be -1 (K >= 256), 0 (never) or 1 (always)");
        ctx->opt_reduce_triangle = (int)value;
    } else if (!strcmp(key, "staged_upload")) {
        if (value < 0 || value > 2) return ctx->fail(FSNAP_E_ARG, "staged_upload must be 0 (pageable copy), 1 (probe) or 2 (double buffer)");
        ctx->opt_staged_upload = (int)value;
    } else if (!strcmp(key, "fused_residual")) {
        if (value < 0 || value > 1) return ctx->fail(FSNAP_E_ARG, "fused_residual must be 0 (two kernels) or 1 (one pass)");
        ctx->opt_fused_residual = (int)value;
    } else if (!strcmp(key, "fused_pack")) {
        ctx->opt_fused_pack = value != 0;
    } else if (!strcmp(key, "nsplit")) {
        if (value < 0 || value > (1 << 24)) return ctx->fail(FSNAP_E_ARG, "nsplit out of range");
        ctx->opt_nsplit = (int)value;
    } else if (!strcmp(key, "nblocks")) {
        if (value < 0 || value > (1 << 24)) return ctx->fail(FSNAP_E_ARG, "nblocks out of range");
        ctx->opt_nblocks = (int)value;
    } else {
        return ctx->fail(FSNAP_E_ARG, "unknown option '%s'", key);
    }
    return FSNAP_OK;
}

int fsnap_upload_rows(fsnap_ctx* ctx, const double* A, int64_t m, int64_t K, int64_t lda, const double* b) {
    if (!ctx) return FSNAP_E_ARG;
    if (!A || !b || m <= 0 || K <= 0 || lda < K) return ctx->fail(FSNAP_E_ARG, "fsnap_upload_rows: bad argument");
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    // rows are stored densely (lda_dev = K) + 256 B of zeroed tail padding for the
    // 16-byte over-read of the last row
    const size_t abytes = (size_t)m * K * sizeof(double);
    if (!ctx->ownA.ensure(abytes + 256) || !ctx->ownb.ensure((size_t)m * sizeof(double)))
        return ctx->fail(FSNAP_E_NOMEM, "hipMalloc of %zu bytes for A failed", abytes);
    FSNAP_HIP(hipEventRecord(ctx->ev[3], ctx->stream), "hipEventRecord");
    FSNAP_HIP(hipMemsetAsync((char*)ctx->ownA.p + abytes, 0, 256, ctx->stream), "hipMemsetAsync");
    // Large matrices: which way is faster depends on the box.  The runtime's pageable copy of a matrix this size pins the
    // caller's pages and lets the DMA engine read them in place: 1.03 GB in 27 ms (38 GB/s) on the boxes of round 4, 100 ms
    // on the driver's box of round 3.  The page-locked double buffer (staged_rows_h2d) costs host memcpy time instead: 104 ms
    // on the round-4 boxes, whose CPU quota holds four copying threads at ~10 GB/s together, and the first touch of its
    // freshly pinned slots alone cost ~60 ms there (0.6 GB/s over the first 32 MiB) -- a probe of either path on a small
    // piece says nothing about the large copy (a 64 MiB pageable probe ran at 7 GB/s where the 1 GB copy ran at 38).
    // So the default stays the pageable copy (option staged_upload = 0); 2 = double buffer (hosts with free cores and slow
    // pinning), 1 = double buffer that hands the rest to the pageable copy when its first two slots fill at < 20 GB/s.
    size_t done_rows = 0;
    int staged = FSNAP_E_STATE;
    const bool fits = (size_t)K * 8 <= ((size_t)16 << 20);
    if (ctx->opt_staged_upload == 2 && abytes >= ((size_t)8 << 20) && fits) {
        staged = staged_rows_h2d(ctx, ctx->ownA.p, A, (size_t)m, (size_t)K * 8, (size_t)lda * 8, 0.0, &done_rows);
        if (staged != FSNAP_OK && staged != FSNAP_E_NOMEM) return staged;
        if (staged != FSNAP_OK) done_rows = 0;
    } else if (ctx->opt_staged_upload == 1 && abytes >= ((size_t)256 << 20) && fits) {
        double rate = 0.0;
        staged = staged_rows_h2d(ctx, ctx->ownA.p, A, (size_t)m, (size_t)K * 8, (size_t)lda * 8, 20.0e9, &done_rows, &rate);
        if (staged != FSNAP_OK && staged != FSNAP_E_NOMEM) return staged;
        if (staged != FSNAP_OK) done_rows = 0;
        ctx->upload_probe_gbps = rate / 1e9;
    }
    const size_t rest = (size_t)m - done_rows;
    if (rest > 0) {
        staged = FSNAP_E_STATE;                 // (part of) the matrix goes through the runtime's pageable copy
        if (lda == K) {
            FSNAP_HIP(hipMemcpyAsync((char*)ctx->ownA.p + done_rows * (size_t)K * 8, A + done_rows * (size_t)lda, rest * (size_t)K * 8,
                                     hipMemcpyHostToDevice, ctx->stream),
                      "hipMemcpy(A)");
        } else {
            FSNAP_HIP(hipMemcpy2DAsync((char*)ctx->ownA.p + done_rows * (size_t)K * 8, (size_t)K * 8, A + done_rows * (size_t)lda,
                                       (size_t)lda * 8, (size_t)K * 8, rest, hipMemcpyHostToDevice, ctx->stream),
                      "hipMemcpy2D(A)");
        }
    }
    ctx->upload_staged = staged == FSNAP_OK;
    FSNAP_HIP(hipMemcpyAsync(ctx->ownb.p, b, (size_t)m * 8, hipMemcpyHostToDevice, ctx->stream), "hipMemcpy(b)");
    FSNAP_HIP(hipEventRecord(ctx->ev[4], ctx->stream), "hipEventRecord");
    FSNAP_HIP(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");  // host buffers may be reused now
    ctx->t_upload = true;
    ctx->dA = (const double*)ctx->ownA.p;
    ctx->db = (const double*)ctx->ownb.p;
    ctx->wpack_valid = false;
    if (m != ctx->m) {  // weights / mask of a previous matrix no longer apply
        ctx->dw = nullptr;
        ctx->dmask = nullptr;
        ctx->ntrain_resident = -1;
        ctx->ones.release();
    }
    ctx->m = m;
    ctx->K = K;
    ctx->lda = K;
    return FSNAP_OK;
}

int fsnap_drop_rows(fsnap_ctx* ctx) {
    if (!ctx) return FSNAP_E_ARG;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream && !ctx->comm_broken) (void)hipStreamSynchronize(ctx->stream);   // a launch may still read them
    ctx->dA = nullptr;
    ctx->db = nullptr;
    ctx->dw = nullptr;
    ctx->dmask = nullptr;
    ctx->m = 0;
    ctx->ntrain_resident = -1;
    ctx->wpack_valid = false;
    ctx->mirror_of = nullptr;
    ctx->ownA.release();
    ctx->ownb.release();
    ctx->ownw.release();
    ctx->ownmask.release();
    ctx->ones.release();
    return FSNAP_OK;
}

int fsnap_bind_rows(fsnap_ctx* ctx, const double* dA, int64_t m, int64_t K, int64_t lda, const double* db) {
    if (!ctx) return FSNAP_E_ARG;
    if (!dA || !db || m <= 0 || K <= 0 || lda < K) return ctx->fail(FSNAP_E_ARG, "fsnap_bind_rows: bad argument");
    if (m != ctx->m) {
        ctx->dw = nullptr;
        ctx->dmask = nullptr;
        ctx->ntrain_resident = -1;
        ctx->ones.release();
    }
    ctx->dA = dA;
    ctx->db = db;
    ctx->wpack_valid = false;
    ctx->m = m;
    ctx->K = K;
    ctx->lda = lda;
    return FSNAP_OK;
}

int fsnap_rows_alloc(fsnap_ctx* ctx, int64_t m, int64_t K) {
    if (!ctx) return FSNAP_E_ARG;
    if (m <= 0 || K <= 0) return ctx->fail(FSNAP_E_ARG, "fsnap_rows_alloc: bad argument");
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    const size_t abytes = (size_t)m * K * sizeof(double);
    if (!ctx->ownA.ensure(abytes + 256) || !ctx->ownb.ensure((size_t)m * 8) || !ctx->ownw.ensure((size_t)m * 8))
        return ctx->fail(FSNAP_E_NOMEM, "hipMalloc of %zu bytes for A failed", abytes);
    FSNAP_HIP(hipMemsetAsync(ctx->ownA.p, 0, abytes + 256, ctx->stream), "hipMemsetAsync(A)");
    FSNAP_HIP(hipMemsetAsync(ctx->ownb.p, 0, (size_t)m * 8, ctx->stream), "hipMemsetAsync(b)");
    FSNAP_HIP(hipMemsetAsync(ctx->ownw.p, 0, (size_t)m * 8, ctx->stream), "hipMemsetAsync(w)");
    FSNAP_HIP(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    ctx->dA = (const double*)ctx->ownA.p;
    ctx->db = (const double*)ctx->ownb.p;
    ctx->wpack_valid = false;
    ctx->dw = (const double*)ctx->ownw.p;
    ctx->dmask = nullptr;
    ctx->ntrain_resident = -1;
    ctx->ones.release();
    ctx->m = m;
    ctx->K = K;
    ctx->lda = K;
    return FSNAP_OK;
}

namespace {

struct AssemblyPlanDev {           // device views of a staged batch (fsnap_assemble / fsnap_assemble_accumulate)
    const double* raw;
    const int64_t* src_row;
    const int *kind, *frac;
    const double *d, *truth, *weight, *fractions, *blank2J;
};

// argument checks and H2D staging shared by the two assembly entry points; K = ntypes * (ncoeff + offcol)
int stage_assembly(fsnap_ctx* ctx, const char* who, const double* raw, int64_t raw_rows, int64_t raw_ld, int64_t nrows,
                   const int64_t* src_row, const int32_t* kind, const int32_t* frac, const double* d, const double* truth,
                   const double* weight, const double* fractions, int64_t nfrac, const double* blank2J, int32_t ntypes,
                   int32_t ncoeff, int32_t offcol, AssemblyPlanDev* out) {
    if (!raw || !src_row || !kind || !frac || !d || !truth || !weight || !blank2J || raw_rows <= 0 || nrows <= 0 ||
        ntypes <= 0 || ncoeff <= 0 || (offcol != 0 && offcol != 1) || raw_ld < (int64_t)ntypes * ncoeff + 1 ||
        (nfrac > 0 && !fractions) || raw_rows > 0x7FFFFFFF)
        return ctx->fail(FSNAP_E_ARG, "%s: bad argument", who);
    for (int64_t r = 0; r < nrows; ++r)
        if (src_row[r] < 0 || src_row[r] >= raw_rows || kind[r] < 0 || kind[r] > 3 || frac[r] >= nfrac || frac[r] < -1)
            return ctx->fail(FSNAP_E_ARG, "%s: plan entry %lld out of range", who, (long long)r);
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    const int64_t K = (int64_t)ntypes * (ncoeff + offcol);
    const size_t rawb = (size_t)raw_rows * raw_ld * 8;
    // plan layout on the device: src_row (8n) | d (8n) | truth (8n) | weight (8n) | kind (4n) | frac (4n)
    const size_t n = (size_t)nrows;
    const size_t planb = n * 40;
    const size_t fracb = (size_t)(nfrac > 0 ? nfrac : 1) * ntypes * 8;
    if (!ctx->st_raw.ensure(rawb) || !ctx->st_plan.ensure(planb) || !ctx->st_frac.ensure(fracb) ||
        !ctx->st_blank.ensure((size_t)K * 8))
        return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(assembly staging) failed");
    char* pl = (char*)ctx->st_plan.p;
    hipStream_t st = ctx->stream;
    // the raw LAMMPS block of the batch (tens of MB): through the page-locked double buffer like fsnap_upload_rows
    int staged = FSNAP_E_STATE;
    if (ctx->opt_staged_upload == 2 && rawb >= ((size_t)8 << 20) && st == ctx->stream) {
        staged = staged_rows_h2d(ctx, ctx->st_raw.p, raw, (size_t)raw_rows, (size_t)raw_ld * 8, (size_t)raw_ld * 8);
        if (staged != FSNAP_OK && staged != FSNAP_E_NOMEM) return staged;
    }
    if (staged != FSNAP_OK) FSNAP_HIP(hipMemcpyAsync(ctx->st_raw.p, raw, rawb, hipMemcpyHostToDevice, st), "hipMemcpy(raw)");
    FSNAP_HIP(hipMemcpyAsync(pl, src_row, n * 8, hipMemcpyHostToDevice, st), "hipMemcpy(plan)");
    FSNAP_HIP(hipMemcpyAsync(pl + n * 8, d, n * 8, hipMemcpyHostToDevice, st), "hipMemcpy(plan)");
    FSNAP_HIP(hipMemcpyAsync(pl + n * 16, truth, n * 8, hipMemcpyHostToDevice, st), "hipMemcpy(plan)");
    FSNAP_HIP(hipMemcpyAsync(pl + n * 24, weight, n * 8, hipMemcpyHostToDevice, st), "hipMemcpy(plan)");
    FSNAP_HIP(hipMemcpyAsync(pl + n * 32, kind, n * 4, hipMemcpyHostToDevice, st), "hipMemcpy(plan)");
    FSNAP_HIP(hipMemcpyAsync(pl + n * 36, frac, n * 4, hipMemcpyHostToDevice, st), "hipMemcpy(plan)");
    if (nfrac > 0)
        FSNAP_HIP(hipMemcpyAsync(ctx->st_frac.p, fractions, (size_t)nfrac * ntypes * 8, hipMemcpyHostToDevice, st),
                  "hipMemcpy(fractions)");
    FSNAP_HIP(hipMemcpyAsync(ctx->st_blank.p, blank2J, (size_t)K * 8, hipMemcpyHostToDevice, st), "hipMemcpy(blank2J)");
    out->raw = (const double*)ctx->st_raw.p;
    out->src_row = (const int64_t*)pl;
    out->d = (const double*)(pl + n * 8);
    out->truth = (const double*)(pl + n * 16);
    out->weight = (const double*)(pl + n * 24);
    out->kind = (const int*)(pl + n * 32);
    out->frac = (const int*)(pl + n * 36);
    out->fractions = (const double*)ctx->st_frac.p;
    out->blank2J = (const double*)ctx->st_blank.p;
    return FSNAP_OK;
}

}  // namespace

int fsnap_assemble(fsnap_ctx* ctx, const double* raw, int64_t raw_rows, int64_t raw_ld, int64_t nrows, int64_t row0,
                   const int64_t* src_row, const int32_t* kind, const int32_t* frac, const double* d,
                   const double* truth, const double* weight, const double* fractions, int64_t nfrac,
                   const double* blank2J, int32_t ntypes, int32_t ncoeff, int32_t offcol) {
    if (!ctx) return FSNAP_E_ARG;
    int rc;
    if ((rc = check_rows(ctx))) return rc;
    if (ctx->dA != (const double*)ctx->ownA.p || ctx->dw != (const double*)ctx->ownw.p)
        return ctx->fail(FSNAP_E_STATE, "fsnap_assemble needs rows allocated by fsnap_rows_alloc");
    if (nrows <= 0 || row0 < 0 || row0 + nrows > ctx->m || ntypes <= 0 || ncoeff <= 0 || (offcol != 0 && offcol != 1) ||
        (int64_t)ntypes * (ncoeff + offcol) != ctx->K)
        return ctx->fail(FSNAP_E_ARG, "fsnap_assemble: bad argument");
    AssemblyPlanDev pd;
    if ((rc = stage_assembly(ctx, "fsnap_assemble", raw, raw_rows, raw_ld, nrows, src_row, kind, frac, d, truth, weight,
                             fractions, nfrac, blank2J, ntypes, ncoeff, offcol, &pd)))
        return rc;
    hipStream_t st = ctx->stream;
    ctx->wpack_valid = false;      // the kernel below writes b and w of these rows
    FSNAP_HIP(fsnap::launch_assemble(pd.raw, raw_ld, nrows, pd.src_row, pd.kind, pd.frac, pd.d, pd.truth, pd.weight,
                                     pd.fractions, pd.blank2J, ntypes, ncoeff, offcol,
                                     (double*)ctx->ownA.p + row0 * ctx->lda, ctx->lda, (double*)ctx->ownb.p + row0,
                                     (double*)ctx->ownw.p + row0, st),
              "launch fsnap_assemble_k");
    FSNAP_HIP(hipStreamSynchronize(st), "hipStreamSynchronize");   // host staging may be reused by the caller
    return FSNAP_OK;
}

int fsnap_assemble_accumulate(fsnap_ctx* ctx, const double* raw, int64_t raw_rows, int64_t raw_ld, int64_t nrows,
                              const int64_t* src_row, const int32_t* kind, const int32_t* frac, const double* d,
                              const double* truth, const double* weight, const double* fractions, int64_t nfrac,
                              const double* blank2J, int32_t ntypes, int32_t ncoeff, int32_t offcol, double* d_packed) {
    if (!ctx) return FSNAP_E_ARG;
    if (!d_packed) return ctx->fail(FSNAP_E_ARG, "fsnap_assemble_accumulate: d_packed is NULL");
    ctx->chol_factor_of = nullptr;              // the statistics in d_packed change: neither a factor on the device ...
    if (ctx->mirror_of == d_packed) ctx->mirror_of = nullptr;      // ... nor the host mirror is of them any more
    int rc;
    AssemblyPlanDev pd;
    if ((rc = stage_assembly(ctx, "fsnap_assemble_accumulate", raw, raw_rows, raw_ld, nrows, src_row, kind, frac, d, truth,
                             weight, fractions, nfrac, blank2J, ntypes, ncoeff, offcol, &pd)))
        return rc;
    const int64_t K = (int64_t)ntypes * (ncoeff + offcol);
    // geometry of the tiled kernel for a batch of this shape as fsnap_rows_alloc would hold it (lda = K): the fused
    // launch sums in the order of fsnap_assemble + fsnap_normal_eq_accumulate on the tiled kernel
    TiledGeometry g;
    if ((rc = tiled_geometry(ctx, nrows, K, K, &g))) return rc;
    const int npk = fsnap::pack_weights_num_blocks(nrows);
    const size_t recb = fsnap::assemble_row_record_bytes();
    // per-row scratch: b | w | (w_eff, w_eff b) pairs | row records -- 48 bytes per row, nothing of A
    if (!ctx->fz_rows.ensure((size_t)nrows * (16 + recb + 16) + 64) || !ctx->fz_spart.ensure((size_t)npk * 4 * 8) ||
        !ctx->fz_part.ensure((size_t)g.nsplit * g.npairs * 4096 * sizeof(double)) ||
        !ctx->fz_cpart.ensure((size_t)g.nsplit * g.NSB * 4 * 64 * sizeof(double)))
        return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(fused assembly scratch) failed");
    hipStream_t st = ctx->stream;
    double* db = (double*)ctx->fz_rows.p;
    double* dw = db + nrows;
    double* wpack = dw + nrows;
    char* recs = (char*)(wpack + 2 * nrows);
    FSNAP_HIP(fsnap::launch_assemble_bw(pd.raw, raw_ld, nrows, pd.src_row, pd.kind, pd.frac, pd.d, pd.truth, pd.weight,
                                        ntypes * ncoeff, db, dw, recs, st),
              "launch fsnap_assemble_bw_k");
    FSNAP_HIP(fsnap::launch_pack_weights(db, dw, nullptr, nrows, wpack, (double*)ctx->fz_spart.p, st),
              "launch fsnap_pack_weights_k");
    fsnap::TiledArgs a;
    a.A = nullptr;
    a.lda = K;
    a.wpack = wpack;
    a.m = nrows;
    a.K = (int)K;
    a.NSB = g.NSB;
    a.npairs = g.npairs;
    a.nsplit = g.nsplit;
    a.chunks_per_split = g.cps;
    a.nontemporal = false;
    a.part = (double*)ctx->fz_part.p;
    a.cpart = (double*)ctx->fz_cpart.p;
    a.spart = (const double*)ctx->fz_spart.p;
    a.ns = npk;
    // c slots of waves that own no rows are written (zeros) by the kernel, like kernel 1T's: no memset
    FSNAP_HIP(fsnap::launch_assemble_syrk(pd.raw, raw_ld, recs, pd.d, pd.fractions, pd.blank2J, ntypes, ncoeff, offcol, a, st),
              "launch fsnap_assemble_syrk_k");
    FSNAP_HIP(fsnap::launch_reduce_tiled(a, d_packed, true, st), "launch fsnap_reduce_tiled");
    FSNAP_HIP(hipStreamSynchronize(st), "hipStreamSynchronize");   // host staging may be reused by the caller
    return FSNAP_OK;
}

int fsnap_download_rows(fsnap_ctx* ctx, double* A, int64_t lda, double* b, double* w) {
    if (!ctx) return FSNAP_E_ARG;
    int rc;
    if ((rc = check_rows(ctx))) return rc;
    if (A && lda < ctx->K) return ctx->fail(FSNAP_E_ARG, "fsnap_download_rows: lda < K");
    if (w && !ctx->dw) return ctx->fail(FSNAP_E_STATE, "no weights on the device");
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    const size_t m = (size_t)ctx->m, K = (size_t)ctx->K;
    if (A)
        FSNAP_HIP(hipMemcpy2DAsync(A, (size_t)lda * 8, ctx->dA, (size_t)ctx->lda * 8, K * 8, m, hipMemcpyDeviceToHost,
                                   ctx->stream),
                  "hipMemcpy2D(A)");
    if (b) FSNAP_HIP(hipMemcpyAsync(b, ctx->db, m * 8, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpy(b)");
    if (w) FSNAP_HIP(hipMemcpyAsync(w, ctx->dw, m * 8, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpy(w)");
    FSNAP_HIP(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    return FSNAP_OK;
}

int fsnap_set_weights(fsnap_ctx* ctx, const double* w, const uint8_t* mask) {
    if (!ctx) return FSNAP_E_ARG;
    int rc;
    if ((rc = check_rows(ctx))) return rc;
    if (!w) return ctx->fail(FSNAP_E_ARG, "fsnap_set_weights: w is NULL");
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    const size_t m = (size_t)ctx->m;
    if (!ctx->ownw.ensure(m * 8)) return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(w) failed");
    if ((rc = staged_h2d(ctx, ctx->ownw.p, w, m * 8))) return rc;
    ctx->dw = (const double*)ctx->ownw.p;
    ctx->wpack_valid = false;
    if (mask) {
        if (!ctx->ownmask.ensure(m)) return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(mask) failed");
        if ((rc = staged_h2d(ctx, ctx->ownmask.p, mask, m))) return rc;
        ctx->dmask = (const unsigned char*)ctx->ownmask.p;
        ctx->ntrain_resident = -1;          // the mask buffer no longer matches the resident prefix sum
    } else {
        ctx->dmask = nullptr;
    }
    return FSNAP_OK;     // asynchronous: the copies are ordered before everything launched later on this context
}

int fsnap_set_weights_train(fsnap_ctx* ctx, const double* w_train, int64_t ntrain, const uint8_t* mask, const int32_t* rank) {
    if (!ctx) return FSNAP_E_ARG;
    int rc;
    if ((rc = check_rows(ctx))) return rc;
    if (!w_train || ntrain < 0 || ntrain > ctx->m || (mask == nullptr) != (rank == nullptr))
        return ctx->fail(FSNAP_E_ARG, "fsnap_set_weights_train: bad argument");
    if (!mask && ctx->ntrain_resident != ntrain)
        return ctx->fail(FSNAP_E_STATE, "fsnap_set_weights_train: no resident training mask with %lld training rows", (long long)ntrain);
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    const size_t m = (size_t)ctx->m;
    if (!ctx->ownw.ensure(m * 8) || !ctx->wtrain.ensure((size_t)(ntrain > 0 ? ntrain : 1) * 8))
        return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(w) failed");
    if (mask) {
        if (!ctx->ownmask.ensure(m) || !ctx->wrank.ensure(m * 4)) return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(mask) failed");
        if ((rc = staged_h2d(ctx, ctx->ownmask.p, mask, m))) return rc;
        if ((rc = staged_h2d(ctx, ctx->wrank.p, rank, m * 4))) return rc;
        ctx->ntrain_resident = ntrain;
    }
    if ((rc = staged_h2d(ctx, ctx->wtrain.p, w_train, (size_t)ntrain * 8))) return rc;
    FSNAP_HIP(fsnap::launch_expand_weights((const double*)ctx->wtrain.p, (const unsigned char*)ctx->ownmask.p,
                                           (const int*)ctx->wrank.p, ctx->m, (double*)ctx->ownw.p, ctx->stream),
              "launch fsnap_expand_weights_k");
    ctx->dw = (const double*)ctx->ownw.p;
    ctx->dmask = (const unsigned char*)ctx->ownmask.p;
    ctx->wpack_valid = false;
    return FSNAP_OK;
}

int fsnap_bind_weights(fsnap_ctx* ctx, const double* dw, const uint8_t* dmask) {
    if (!ctx) return FSNAP_E_ARG;
    int rc;
    if ((rc = check_rows(ctx))) return rc;
    if (!dw) return ctx->fail(FSNAP_E_ARG, "fsnap_bind_weights: dw is NULL");
    ctx->dw = dw;
    ctx->dmask = dmask;
    ctx->wpack_valid = false;
    ctx->ntrain_resident = -1;
    return FSNAP_OK;
}

int fsnap_normal_eq_async(fsnap_ctx* ctx, double* d_packed) {
    if (!ctx) return FSNAP_E_ARG;
    if (!d_packed) return ctx->fail(FSNAP_E_ARG, "fsnap_normal_eq_async: d_packed is NULL");
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    return launch_normal_eq(ctx, d_packed);
}

int fsnap_normal_eq(fsnap_ctx* ctx, double* G, double* c, double* scalars) {
    if (!ctx) return FSNAP_E_ARG;
    int rc;
    if ((rc = check_rows(ctx))) return rc;
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    const int64_t K = ctx->K;
    if (!ctx->packed.ensure((size_t)FSNAP_PACKED_LEN(K) * 8)) return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(packed) failed");
    double* dp = (double*)ctx->packed.p;
    if ((rc = launch_normal_eq(ctx, dp))) return rc;
    if (G) FSNAP_HIP(hipMemcpyAsync(G, dp, (size_t)K * K * 8, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpy(G)");
    if (c) FSNAP_HIP(hipMemcpyAsync(c, dp + K * K, (size_t)K * 8, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpy(c)");
    if (scalars)
        FSNAP_HIP(hipMemcpyAsync(scalars, dp + K * K + K, 3 * 8, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpy(scalars)");
    FSNAP_HIP(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    return FSNAP_OK;
}

int fsnap_normal_eq_accumulate(fsnap_ctx* ctx, double* d_packed) {
    if (!ctx) return FSNAP_E_ARG;
    if (!d_packed) return ctx->fail(FSNAP_E_ARG, "fsnap_normal_eq_accumulate: d_packed is NULL");
    int rc;
    if ((rc = check_rows(ctx))) return rc;
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    return launch_normal_eq(ctx, d_packed, false, true);
}

int fsnap_normal_eq_resident(fsnap_ctx* ctx, double** d_packed) {
    if (!ctx) return FSNAP_E_ARG;
    if (!d_packed) return ctx->fail(FSNAP_E_ARG, "fsnap_normal_eq_resident: d_packed is NULL");
    int rc;
    if ((rc = check_rows(ctx))) return rc;
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    if (!ctx->packed.ensure((size_t)FSNAP_PACKED_LEN(ctx->K) * 8)) return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(packed) failed");
    if ((rc = launch_normal_eq(ctx, (double*)ctx->packed.p, true))) return rc;
    *d_packed = (double*)ctx->packed.p;
    return FSNAP_OK;
}

int fsnap_mirror_packed(fsnap_ctx* ctx, const double* d_packed, int64_t K) {
    if (!ctx) return FSNAP_E_ARG;
    if (!d_packed || K <= 0) return ctx->fail(FSNAP_E_ARG, "fsnap_mirror_packed: bad argument");
    ctx->mirror_of = nullptr;
    if (device_factor(ctx, K)) return FSNAP_OK;     // factorised on the GPU: nothing to mirror
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    const size_t need = ((size_t)FSNAP_PACKED_LEN(K) + (size_t)K) * 8;
    if (ctx->mirror_bytes < need) {
        if (ctx->mirror) (void)hipHostFree(ctx->mirror);
        ctx->mirror = nullptr;
        ctx->mirror_bytes = 0;
        if (hipHostMalloc((void**)&ctx->mirror, need, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess)
            return FSNAP_OK;                                 // no mirror: fsnap_solve_device copies instead
        ctx->mirror_bytes = need;
    }
    if (!ctx->mirror_ev && hipEventCreateWithFlags(&ctx->mirror_ev, hipEventDisableTiming) != hipSuccess) {
        ctx->mirror_ev = nullptr;
        return FSNAP_OK;
    }
    FSNAP_HIP(fsnap::launch_mirror_copy(d_packed, (int)K, ctx->mirror, ctx->stream), "launch fsnap_mirror_copy_k");
    FSNAP_HIP(hipEventRecord(ctx->mirror_ev, ctx->stream), "hipEventRecord");
    ctx->mirror_of = d_packed;
    ctx->mirror_K = K;
    ctx->mirror_upper = true;                   // the copy kernel fills the upper triangle (from the pair of the diagonal on)
    ctx->mirror_gen = next_mirror_generation();
    return FSNAP_OK;
}

int fsnap_download_packed(fsnap_ctx* ctx, const double* d_packed, int64_t K, double* G, double* c, double* scalars) {
    if (!ctx) return FSNAP_E_ARG;
    if (!d_packed || K <= 0) return ctx->fail(FSNAP_E_ARG, "fsnap_download_packed: bad argument");
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    if (G) FSNAP_HIP(hipMemcpyAsync(G, d_packed, (size_t)K * K * 8, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpy(G)");
    if (c) FSNAP_HIP(hipMemcpyAsync(c, d_packed + K * K, (size_t)K * 8, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpy(c)");
    if (scalars)
        FSNAP_HIP(hipMemcpyAsync(scalars, d_packed + K * K + K, 3 * 8, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpy(s)");
    FSNAP_HIP(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    return FSNAP_OK;
}

int fsnap_weight_rows_device(fsnap_ctx* ctx, double* d_aw, int64_t ldaw, double* d_bw) {
    if (!ctx) return FSNAP_E_ARG;
    int rc;
    if ((rc = check_rows(ctx)) || (rc = check_weights(ctx))) return rc;
    if (!d_aw || !d_bw || ldaw < ctx->K) return ctx->fail(FSNAP_E_ARG, "fsnap_weight_rows: bad argument");
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    const unsigned char* mask = ctx->dmask;
    if (!mask) {
        if ((rc = ensure_ones(ctx))) return rc;
        mask = (const unsigned char*)ctx->ones.p;
    }
    FSNAP_HIP(hipEventRecord(ctx->ev[5], ctx->stream), "hipEventRecord");
    FSNAP_HIP(fsnap::launch_weight_rows(ctx->dA, ctx->lda, ctx->db, ctx->dw, mask, ctx->m, (int)ctx->K, d_aw, ldaw,
                                        d_bw, ctx->stream),
              "launch fsnap_weight_rows_k");
    FSNAP_HIP(hipEventRecord(ctx->ev[6], ctx->stream), "hipEventRecord");
    ctx->t_weight = true;
    return FSNAP_OK;
}

int fsnap_weight_rows(fsnap_ctx* ctx, double* aw, int64_t ldaw, double* bw) {
    if (!ctx) return FSNAP_E_ARG;
    int rc;
    if ((rc = check_rows(ctx))) return rc;
    if (!aw || !bw || ldaw < ctx->K) return ctx->fail(FSNAP_E_ARG, "fsnap_weight_rows: bad argument");
    const size_t m = (size_t)ctx->m, K = (size_t)ctx->K;
    if (!ctx->aw.ensure(m * K * 8) || !ctx->bw.ensure(m * 8)) return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(aw) failed");
    if ((rc = fsnap_weight_rows_device(ctx, (double*)ctx->aw.p, (int64_t)K, (double*)ctx->bw.p))) return rc;
    FSNAP_HIP(hipMemcpy2DAsync(aw, (size_t)ldaw * 8, ctx->aw.p, K * 8, K * 8, m, hipMemcpyDeviceToHost, ctx->stream),
              "hipMemcpy2D(aw)");
    FSNAP_HIP(hipMemcpyAsync(bw, ctx->bw.p, m * 8, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpy(bw)");
    FSNAP_HIP(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    return FSNAP_OK;
}

int fsnap_predict(fsnap_ctx* ctx, const double* beta, double* preds, double* sse) {
    if (!ctx) return FSNAP_E_ARG;
    int rc;
    if ((rc = check_rows(ctx))) return rc;
    if (!beta) return ctx->fail(FSNAP_E_ARG, "fsnap_predict: beta is NULL");
    if (sse && (rc = check_weights(ctx))) return rc;
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    const size_t m = (size_t)ctx->m, K = (size_t)ctx->K;
    const int nb = fsnap::gemv_num_blocks(ctx->m);
    if (!ctx->beta.ensure(K * 8) || (preds && !ctx->preds.ensure(m * 8)) || (sse && !ctx->sse.ensure((size_t)nb * 8)))
        return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(preds) failed");
    const unsigned char* mask = ctx->dmask;
    if (sse && !mask) {
        if ((rc = ensure_ones(ctx))) return rc;
        mask = (const unsigned char*)ctx->ones.p;
    }
    FSNAP_HIP(hipMemcpyAsync(ctx->beta.p, beta, K * 8, hipMemcpyHostToDevice, ctx->stream), "hipMemcpy(beta)");
    FSNAP_HIP(hipEventRecord(ctx->ev[7], ctx->stream), "hipEventRecord");
    FSNAP_HIP(fsnap::launch_gemv_rows(ctx->dA, ctx->lda, (const double*)ctx->beta.p, ctx->m, (int)ctx->K,
                                      preds ? (double*)ctx->preds.p : nullptr, ctx->db, ctx->dw, mask,
                                      sse ? (double*)ctx->sse.p : nullptr, nullptr, ctx->stream),
              "launch fsnap_gemv_rows_k");
    FSNAP_HIP(hipEventRecord(ctx->ev[8], ctx->stream), "hipEventRecord");
    ctx->t_predict = true;
    if (preds) FSNAP_HIP(hipMemcpyAsync(preds, ctx->preds.p, m * 8, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpy(preds)");
    if (sse) {
        std::string tmp;
        tmp.resize((size_t)nb * 8);
        FSNAP_HIP(hipMemcpyAsync(&tmp[0], ctx->sse.p, (size_t)nb * 8, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpy(sse)");
        FSNAP_HIP(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
        const double* ps = (const double*)tmp.data();
        long double s = 0.0L;  // fixed-order host sum of the per-workgroup partials
        for (int i = 0; i < nb; ++i) s += ps[i];
        *sse = (double)s;
    } else {
        FSNAP_HIP(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    }
    return FSNAP_OK;
}

int fsnap_error_stats(fsnap_ctx* ctx, const double* beta, const int32_t* cat, int ncat, double* stats) {
    if (!ctx) return FSNAP_E_ARG;
    int rc;
    if ((rc = check_rows(ctx)) || (rc = check_weights(ctx))) return rc;
    if (!beta || !stats || ncat <= 0) return ctx->fail(FSNAP_E_ARG, "fsnap_error_stats: bad argument");
    if (!cat && (ctx->dcat_rows != ctx->m || !ctx->dcat.p))
        return ctx->fail(FSNAP_E_STATE, "fsnap_error_stats: no categories on the device for these rows");
    if (ncat > 3000) return ctx->fail(FSNAP_E_ARG, "fsnap_error_stats: more than 3000 categories");   // LDS table
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    const size_t m = (size_t)ctx->m, K = (size_t)ctx->K;
    const int nb = fsnap::error_stats_num_blocks(ctx->m);
    if (!ctx->beta.ensure(K * 8) || !ctx->preds.ensure(m * 8) || !ctx->dcat.ensure(m * 4) ||
        !ctx->dstat.ensure(((size_t)nb * ncat * 6 + (size_t)ncat * 2 + (size_t)ncat * 6) * 8))
        return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(error statistics) failed");
    double* d_partial = (double*)ctx->dstat.p;
    double* d_means = d_partial + (size_t)nb * ncat * 6;
    double* d_table = d_means + (size_t)ncat * 2;                 // per-category sums of a pass, folded over the workgroups
    FSNAP_HIP(hipMemcpyAsync(ctx->beta.p, beta, K * 8, hipMemcpyHostToDevice, ctx->stream), "hipMemcpy(beta)");
    if (cat) {
        FSNAP_HIP(hipMemcpyAsync(ctx->dcat.p, cat, m * 4, hipMemcpyHostToDevice, ctx->stream), "hipMemcpy(categories)");
        ctx->dcat_rows = ctx->m;
    }
    FSNAP_HIP(fsnap::launch_gemv_rows(ctx->dA, ctx->lda, (const double*)ctx->beta.p, ctx->m, (int)ctx->K, (double*)ctx->preds.p,
                                      ctx->db, ctx->dw, nullptr, nullptr, nullptr, ctx->stream),
              "launch fsnap_gemv_rows_k");
    std::vector<double> h((size_t)ncat * 6), means((size_t)ncat * 2);
    // pass 0: counts and sums -> category means.  The per-workgroup tables are folded on the device (fixed order); only
    // ncat x 4 doubles come back (the host used to add 245 tables of ncat x 4 entries itself: ~1 ms at 240 categories)
    FSNAP_HIP(fsnap::launch_error_stats(ctx->db, (const double*)ctx->preds.p, ctx->dw, (const int*)ctx->dcat.p, ctx->m, ncat, 0,
                                        nullptr, d_partial, ctx->stream),
              "launch fsnap_error_stats_k");
    FSNAP_HIP(fsnap::launch_colsum(d_partial, nb, ncat * 4, d_table, ctx->stream), "launch fsnap_colsum_partials_k");
    FSNAP_HIP(hipMemcpyAsync(h.data(), d_table, (size_t)ncat * 4 * 8, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpy(stats)");
    FSNAP_HIP(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    for (int c = 0; c < ncat; ++c) {
        const double* e = h.data() + (size_t)c * 4;
        double* o = stats + (size_t)c * 10;
        o[0] = e[0]; o[1] = e[1]; o[2] = e[2]; o[3] = e[3];
        means[2 * c] = e[0] > 0 ? e[2] / e[0] : 0.0;            // (truths / n).sum()
        means[2 * c + 1] = e[1] > 0 ? e[3] / e[1] : 0.0;        // (w * truths / n_w).sum()
    }
    FSNAP_HIP(hipMemcpyAsync(d_means, means.data(), (size_t)ncat * 2 * 8, hipMemcpyHostToDevice, ctx->stream), "hipMemcpy(means)");
    FSNAP_HIP(fsnap::launch_error_stats(ctx->db, (const double*)ctx->preds.p, ctx->dw, (const int*)ctx->dcat.p, ctx->m, ncat, 1,
                                        d_means, d_partial, ctx->stream),
              "launch fsnap_error_stats_k");
    FSNAP_HIP(fsnap::launch_colsum(d_partial, nb, ncat * 6, d_table, ctx->stream), "launch fsnap_colsum_partials_k");
    FSNAP_HIP(hipMemcpyAsync(h.data(), d_table, (size_t)ncat * 6 * 8, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpy(stats)");
    FSNAP_HIP(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    for (int c = 0; c < ncat; ++c) {
        double* o = stats + (size_t)c * 10;
        for (int k = 0; k < 6; ++k) o[4 + k] = h[(size_t)c * 6 + k];
    }
    return FSNAP_OK;
}

int fsnap_solve_device(fsnap_ctx* ctx, int kind, double param, int64_t K, const double* d_packed, double* beta,
                       int* rank, double* rcond_est) {
    return fsnap_solve_device_rhs(ctx, kind, param, K, d_packed, nullptr, beta, rank, rcond_est);
}

int fsnap_solve_device_rhs(fsnap_ctx* ctx, int kind, double param, int64_t K, const double* d_packed, const double* rhs,
                           double* beta, int* rank, double* rcond_est) {
    if (!ctx) return FSNAP_E_ARG;
    if (!d_packed || !beta || K <= 0 || kind < FSNAP_SOLVE_CHOL || kind > FSNAP_SOLVE_RIDGE_INV_PROBE)
        return ctx->fail(FSNAP_E_ARG, "fsnap_solve_device: bad argument");
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    const double alpha = (kind == FSNAP_SOLVE_RIDGE || kind == FSNAP_SOLVE_RIDGE_INV || kind == FSNAP_SOLVE_RIDGE_PROBE ||
                          kind == FSNAP_SOLVE_RIDGE_INV_PROBE) ? param : 0.0;
    // large systems: blocked Cholesky on the GPU (kernels 8a-8e); option device_solve = 2 disables it
    // (threshold and measurements: fsnap::DEVICE_CHOL_MIN_K; below it the host is faster: the panel kernels are latency-bound
    // launches)
    if (device_factor(ctx, K)) {
        const int n = (int)K, np = (n + 63) / 64 * 64, npanel = np / 64;
        const size_t head = (size_t)n + npanel + 1;            // [beta | min pivots | status]
        constexpr int NPR = fsnap::CHOL_PROBES;
        const size_t head_all = head + (size_t)NPR * NPR;      // ... | Z^T Z of the probes (condition estimate, LSTSQ kinds)
        if (!ctx->dchol.ensure(fsnap::chol_large_work_doubles(n) * 8) || !ctx->dsolve.ensure((head + 2 * (size_t)np) * 8))
            return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(device Cholesky) failed");
        double* dv = (double*)ctx->dsolve.p;
        double* d_beta = dv;
        double* d_minpiv = dv + n;
        int* d_status = (int*)(dv + n + npanel);
        double* d_dsc = dv + head;
        double* d_z = d_dsc + np;
        // results come back through page-locked, GPU-visible host memory written by the last launch of the chain (the
        // host polls an event: no D2H copy, whose launch latency was ~10 us); the plain staging buffer + copy remain
        // as the fallback when that allocation fails
        if (ctx->chol_host_bytes < head_all * 8) {
            if (ctx->chol_host) (void)hipHostFree(ctx->chol_host);
            ctx->chol_host = nullptr;
            ctx->chol_host_bytes = 0;
            if (hipHostMalloc((void**)&ctx->chol_host, head_all * 8, hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess)
                ctx->chol_host_bytes = head_all * 8;
        }
        if (ctx->chol_host && !ctx->chol_ev && hipEventCreateWithFlags(&ctx->chol_ev, hipEventDisableTiming) != hipSuccess)
            ctx->chol_ev = nullptr;
        double* host_out = (ctx->chol_host && ctx->chol_ev) ? ctx->chol_host : nullptr;
        if (!host_out && ctx->pinned_bytes < head * 8) {
            if (ctx->pinned) (void)hipHostFree(ctx->pinned);
            ctx->pinned = nullptr;
            ctx->pinned_bytes = 0;
            if (hipHostMalloc((void**)&ctx->pinned, head * 8, hipHostMallocDefault) != hipSuccess)
                return ctx->fail(FSNAP_E_NOMEM, "hipHostMalloc(%zu) failed", head * 8);
            ctx->pinned_bytes = head * 8;
        }
        const double* d_rhs = nullptr;
        if (rhs) {
            if (!ctx->dsvec.ensure((size_t)n * 8)) return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(rhs) failed");
            FSNAP_HIP(hipMemcpyAsync(ctx->dsvec.p, rhs, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream), "hipMemcpy(rhs)");
            d_rhs = (const double*)ctx->dsvec.p;
        }
        // a right-hand side of its own for statistics this context has just factorised (the refinement steps of a fit): the
        // factor, the scaling and the inverses of the diagonal blocks are still in the work buffers -- two sweeps instead of a
        // factorisation (kernel 8f + the backward sweep)
        const bool lstsq = kind == FSNAP_SOLVE_LSTSQ || kind == FSNAP_SOLVE_LSTSQ_PROBE;
        const bool reuse = rhs && host_out && ctx->opt_chol_reuse && ctx->chol_factor_of == d_packed && ctx->chol_factor_K == K &&
                           ctx->chol_factor_alpha == alpha;
        if (reuse) {
            FSNAP_HIP(fsnap::launch_chol_resolve(d_rhs, n, (double*)ctx->dchol.p, d_dsc, d_z, d_beta, d_status, d_minpiv, host_out,
                                                 ctx->stream),
                      "launch device Cholesky (sweeps)");
            FSNAP_HIP(hipEventRecord(ctx->chol_ev, ctx->stream), "hipEventRecord");
            int wrc;
            if ((wrc = fsnap::wait_stream(ctx, ctx->chol_ev, "device Cholesky sweeps"))) return wrc;
            int status;
            memcpy(&status, host_out + n + npanel, sizeof(int));
            bool fin = status == 0;
            for (int i = 0; i < n && fin; ++i) fin = (host_out[i] - host_out[i] == 0.0);
            if (fin) {
                double mp = 1.0e300;
                for (int p = 0; p < npanel; ++p) mp = host_out[n + p] < mp ? host_out[n + p] : mp;
                for (int i = 0; i < n; ++i) beta[i] = host_out[i];
                if (rank) *rank = n;
                if (rcond_est) *rcond_est = ctx->chol_factor_rcond;         // of the factor, as its own solve reported it
                fsnap_cond_note(ctx->chol_factor_piv, ctx->chol_factor_lam, 0, 1);
                return FSNAP_OK;
            }
            ctx->chol_factor_of = nullptr;          // (a non-finite right-hand side: the full path below reports it)
        }
        ctx->chol_factor_of = nullptr;
        // the status word lives behind beta and the pivots: its address depends on K.  The chain leaves it cleared; a
        // clearing launch is needed only when this word has not been through a chain yet
        const bool clear_status = ctx->chol_status_word != (const void*)d_status;
        ctx->chol_status_word = host_out ? (const void*)d_status : nullptr;
        FSNAP_HIP(fsnap::launch_chol_large(d_packed, d_rhs, n, alpha, (double*)ctx->dchol.p, d_dsc, d_z, d_beta, d_status,
                                           d_minpiv, host_out, clear_status, (lstsq && host_out) ? host_out + head : nullptr, ctx->stream),
                  "launch device Cholesky");
        const double* h;
        if (host_out) {
            FSNAP_HIP(hipEventRecord(ctx->chol_ev, ctx->stream), "hipEventRecord");
            int wrc;
            if ((wrc = fsnap::wait_stream(ctx, ctx->chol_ev, "device Cholesky"))) return wrc;
            h = host_out;
        } else {
            FSNAP_HIP(hipMemcpyAsync(ctx->pinned, dv, head * 8, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpy(beta)");
            int wrc;
            if ((wrc = fsnap::wait_stream(ctx, nullptr, "device Cholesky"))) return wrc;
            h = ctx->pinned;
        }
        int status;
        memcpy(&status, h + n + npanel, sizeof(int));
        double mp = 1.0e300;
        for (int p = 0; p < npanel; ++p) mp = h[n + p] < mp ? h[n + p] : mp;
        if (getenv("FSNAP_SOLVE_TIMING"))
            fprintf(stderr, "[fsnap_solve_device] blocked Cholesky on the GPU: K = %d, status %d, min pivot %.3e, beta[0] %.6e\n", n,
                    status, mp, h[0]);
        if (status == 0 && mp > 1.0e-3) {
            bool fin = true;
            for (int i = 0; i < n; ++i) fin = fin && (h[i] - h[i] == 0.0);
            if (fin) {
                for (int i = 0; i < n; ++i) beta[i] = h[i];
                // LSTSQ stands in for an SVD of the rows, which knows the conditioning; the smallest pivot bounds lambda_min of
                // the scaled matrix from above only.  The factorisation carried 31 probe vectors in its right-hand-side strip:
                // Z^T Z = B^T S^-1 B came back with the results, its largest generalised eigenvalue against B^T B is a lower
                // bound of 1 / lambda_min that holds at least ~0.4 x 31 / K of it (fsnap_chol_probe_gram_k) -- so the estimate
                // 1 / theta is scaled by 120 / K: what the host layer divides by its margin of 10 is then below lambda_min.
                // No sweep with the factor (round 6's first form ran 2 ... 5 Lanczos steps of 0.06 ... 0.40 ms each here).
                // Same bits on every rank: deterministic kernels on bit-identical statistics.
                fsnap::CondEstimate ce;
                if (lstsq && host_out) {
                    if (ctx->probe_n != n) {
                        ctx->probe_gram.assign((size_t)NPR * NPR, 0.0);
                        std::vector<double> row(NPR);
                        for (int r = 0; r < n; ++r) {
                            for (int p = 0; p < NPR; ++p) row[p] = fsnap::chol_probe(r, p + 1);
                            for (int p = 0; p < NPR; ++p)
                                for (int q = 0; q < NPR; ++q) ctx->probe_gram[(size_t)p * NPR + q] += row[p] * row[q];
                        }
                        ctx->probe_n = n;
                    }
                    const double theta = fsnap_gen_eig_max(host_out + head, ctx->probe_gram.data(), NPR);
                    ce.steps = 1;
                    ce.lambda_min = theta > 0.0 ? (1.0 / theta) * (n > 120 ? 120.0 / n : 1.0) : 0.0;
                }
                const double rc_est = (ce.steps && ce.lambda_min < mp) ? ce.lambda_min : mp;
                fsnap_cond_note(mp, ce.lambda_min, ce.steps, 1);
                if (!ce.steps || ce.lambda_min > 64.0 * n * std::numeric_limits<double>::epsilon()) {
                    if (rank) *rank = n;
                    if (rcond_est) *rcond_est = rc_est;
                    // the factor of these statistics stays on the device for further right-hand sides.  Only for the
                    // context's OWN statistics buffer: every launch that rewrites it clears the tag, while a caller-owned
                    // device buffer can change behind the library's back (a torch tensor accumulated between solves)
                    if (d_packed == (const double*)ctx->packed.p) {
                        ctx->chol_factor_of = d_packed;
                        ctx->chol_factor_K = K;
                        ctx->chol_factor_alpha = alpha;
                        ctx->chol_factor_rcond = rc_est;
                        ctx->chol_factor_piv = mp;
                        ctx->chol_factor_lam = ce.lambda_min;
                    }
                    return FSNAP_OK;
                }
                // every pivot passed and the factor still says lambda_min is at the rounding level of the statistics: for
                // a probe that is "unresolved" (the caller goes to the rows); the plain kinds take the general host path
                if (kind >= FSNAP_SOLVE_LSTSQ_PROBE) {
                    for (int i = 0; i < n; ++i) beta[i] = 0.0;
                    if (rank) *rank = -1;
                    if (rcond_est) *rcond_est = rc_est > 0.0 ? rc_est : 0.0;
                    return FSNAP_OK;
                }
            }
        }
        // A probe (FSNAP_SOLVE_*_PROBE: "come straight back when no Cholesky factorisation resolves the system") whose
        // factorisation met a non-positive pivot, or one below the host path's own acceptance threshold 64 n eps, is unresolved
        // for the host Cholesky too -- the same algorithm on the same matrix.  Downloading G (20 MB at K = 1595) for two more
        // failing factorisations cost 36 ms of an 87 ms ill-conditioned SVD fit at 15 213 x 1 595.
        if (kind >= FSNAP_SOLVE_LSTSQ_PROBE && !(status & 1) &&
            ((status & 2) || !(mp > 64.0 * n * std::numeric_limits<double>::epsilon()))) {
            for (int i = 0; i < n; ++i) beta[i] = 0.0;
            if (rank) *rank = -1;
            if (rcond_est) *rcond_est = (mp > 0.0 && mp < 1.0e300) ? mp : 0.0;
            return FSNAP_OK;
        }
        // ill-conditioned / indefinite / non-finite: the general host path decides
    }
    // statistics already mirrored in page-locked host memory by the reduction kernel: wait for it (polling the
    // event: the blocking wait's wake-up latency is several microseconds) and solve
    if (ctx->mirror_of == d_packed && ctx->mirror && K == ctx->mirror_K) {
        int wrc;
        if ((wrc = fsnap::wait_stream(ctx, ctx->mirror_ev, "statistics mirror"))) return wrc;
        const double* Gm = ctx->mirror;
        // tagged with (context, fill count of the mirror): the refinement solves of a fit reuse the factor of its first solve
        const int rcm = fsnap_solve_diag_tagged(kind, param, K, Gm, rhs ? rhs : Gm + K * K, Gm + K * K + K + 3, beta, rank, rcond_est,
                                                ctx->mirror_upper ? 1 : 0, ctx, ctx->mirror_gen);
        if (rcm) ctx->fail(rcm, "fsnap_solve: numerical status %d", rcm);
        return rcm;
    }
    // general path: statistics to the host (page-locked staging), full solver
    const size_t need = (size_t)(K * K + K) * 8;
    if (ctx->pinned_bytes < need) {
        if (ctx->pinned) (void)hipHostFree(ctx->pinned);
        ctx->pinned = nullptr;
        ctx->pinned_bytes = 0;
        if (hipHostMalloc((void**)&ctx->pinned, need, hipHostMallocDefault) != hipSuccess)
            return ctx->fail(FSNAP_E_NOMEM, "hipHostMalloc(%zu) failed", need);
        ctx->pinned_bytes = need;
    }
    FSNAP_HIP(hipMemcpyAsync(ctx->pinned, d_packed, need, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpy(G)");
    {
        int wrc;
        if ((wrc = fsnap::wait_stream(ctx, nullptr, "statistics download"))) return wrc;
    }
    const double* G = ctx->pinned;
    const int rc = fsnap_solve(kind, param, K, G, rhs ? rhs : G + K * K, beta, rank, rcond_est);
    if (rc) ctx->fail(rc, "fsnap_solve: numerical status %d", rc);
    return rc;
}

int fsnap_fit_resident(fsnap_ctx* ctx, int kind, double param, double* beta, int* rank, double* rcond_est,
                       double** d_packed) {
    if (!ctx) return FSNAP_E_ARG;
    double* dp = nullptr;
    int rc = fsnap_normal_eq_resident(ctx, &dp);
    if (rc) return rc;
    if (d_packed) *d_packed = dp;
    // tiled kernel + host factorisation (144 < K < DEVICE_CHOL_MIN_K): a copy kernel fills the page-locked mirror the host solve polls
    // -- the D2H copy it replaces cost 10 us of launch latency + 7.7 us on the ACE shape (13 035 x 142: 95 us per fit)
    if (ctx->mirror_of != dp && ctx->K > 128 && (rc = fsnap_mirror_packed(ctx, dp, ctx->K))) return rc;     // (no-op from DEVICE_CHOL_MIN_K on)
    return fsnap_solve_device_rhs(ctx, kind, param, ctx->K, dp, nullptr, beta, rank, rcond_est);
}

int fsnap_fit_dist(fsnap_ctx* ctx, int kind, double param, int64_t K, double* beta, int* rank, double* rcond_est,
                   double** d_packed) {
    if (!ctx) return FSNAP_E_ARG;
    if (d_packed) *d_packed = nullptr;
    if (!beta || K <= 0) return ctx->fail(FSNAP_E_ARG, "fsnap_fit_dist: bad argument");
    // no silent single-GPU fallback: a job that lost its communicator (context re-created after ParallelTools.free())
    // would otherwise fit every rank's own shard and publish rank 0's as the global fit
    if (!ctx->comm) return ctx->fail(FSNAP_E_STATE, "fsnap_fit_dist: no communicator (fsnap_comm_init first; single-GPU fits call fsnap_fit_resident)");
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    const int64_t n = FSNAP_PACKED_LEN(K);
    // the one thing a rank cannot recover from locally: without the buffer it cannot take part in the collective at all
    // (its peers run into FSNAP_COMM_TIMEOUT)
    if (!ctx->packed.ensure((size_t)n * 8) || !fsnap::allreduce_packed_reserve(ctx, K))
        return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(packed) failed");
    double* dp = (double*)ctx->packed.p;
    // Everything that can fail on THIS rank alone happens before the collective -- and must not keep the rank out of
    // it (the peers would wait forever): a rank that fails contributes a buffer of NaNs, every rank then sees
    // non-finite statistics and returns FSNAP_NUM_NONFINITE, the failing rank returns its own error.
    const bool have_rows = ctx->dA && ctx->m > 0;
    int local_rc = FSNAP_OK;
    std::string local_err;
    ctx->cur_events = nullptr;
    if (have_rows && ctx->K != K) {
        local_rc = ctx->fail(FSNAP_E_ARG, "fsnap_fit_dist: K = %lld but the resident rows have %lld columns", (long long)K,
                             (long long)ctx->K);
    } else if (have_rows) {
        local_rc = launch_normal_eq(ctx, dp);
    } else {
        ctx->mirror_of = nullptr;
        if (hipMemsetAsync(dp, 0, (size_t)n * 8, ctx->stream) != hipSuccess)   // a rank without rows
            local_rc = ctx->fail(FSNAP_E_HIP, "hipMemsetAsync(packed) failed");
    }
    if (local_rc != FSNAP_OK) {
        local_err = ctx->err;
        ctx->mirror_of = nullptr;
        ctx->cur_events = nullptr;
        (void)hipMemsetAsync(dp, 0xFF, (size_t)n * 8, ctx->stream);            // all bits set = NaN in every double
    }
    hipEvent_t* evs = ctx->cur_events;
    int rc;
    {
        if ((rc = fsnap::allreduce_packed(ctx, dp, K))) return rc;       // wide systems: the upper triangle only
        if (evs) {
            FSNAP_HIP(hipEventRecord(evs[3], ctx->stream), "hipEventRecord");
            ctx->ring_comm[(ctx->nfit - 1) % fsnap_ctx::RING] = true;
        }
        if (local_rc == FSNAP_OK) {
            if ((rc = fsnap_mirror_packed(ctx, dp, K))) return rc;     // host-factorised orders: page-locked mirror instead of a D2H copy
            if (d_packed) *d_packed = dp;
            rc = fsnap_solve_device_rhs(ctx, kind, param, K, dp, nullptr, beta, rank, rcond_est);
        }
    }
    if (local_rc != FSNAP_OK) return ctx->fail(local_rc, "%s", local_err.c_str());
    if (rc == FSNAP_NUM_NONFINITE)
        ctx->fail(rc, "non-finite statistics after the all-reduce: NaN/Inf in a training row of some rank, or a rank failed "
                      "before the collective (see that rank's error)");
    return rc;
}

int fsnap_dev_alloc(fsnap_ctx* ctx, int64_t nbytes, void** d_ptr) {
    if (!ctx) return FSNAP_E_ARG;
    if (!d_ptr || nbytes <= 0) return ctx->fail(FSNAP_E_ARG, "fsnap_dev_alloc: bad argument");
    *d_ptr = nullptr;
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    const hipError_t e = hipMalloc(d_ptr, (size_t)nbytes);
    if (e != hipSuccess) {
        *d_ptr = nullptr;
        return ctx->hipfail(e, "hipMalloc");
    }
    return FSNAP_OK;
}

int fsnap_dev_free(fsnap_ctx* ctx, void* d_ptr) {
    if (!ctx) return FSNAP_E_ARG;
    if (!d_ptr) return FSNAP_OK;
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    FSNAP_HIP(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");     // nothing in flight may still use it
    FSNAP_HIP(hipFree(d_ptr), "hipFree");
    return FSNAP_OK;
}

int fsnap_dev_upload(fsnap_ctx* ctx, void* d_dst, const void* h_src, int64_t nbytes) {
    if (!ctx) return FSNAP_E_ARG;
    if (!d_dst || !h_src || nbytes <= 0) return ctx->fail(FSNAP_E_ARG, "fsnap_dev_upload: bad argument");
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    ctx->chol_factor_of = nullptr;              // (the destination may be a statistics buffer)
    FSNAP_HIP(hipMemcpyAsync(d_dst, h_src, (size_t)nbytes, hipMemcpyHostToDevice, ctx->stream), "hipMemcpy(H2D)");
    FSNAP_HIP(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    return FSNAP_OK;
}

int fsnap_dev_download(fsnap_ctx* ctx, void* h_dst, const void* d_src, int64_t nbytes) {
    if (!ctx) return FSNAP_E_ARG;
    if (!h_dst || !d_src || nbytes <= 0) return ctx->fail(FSNAP_E_ARG, "fsnap_dev_download: bad argument");
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    FSNAP_HIP(hipMemcpyAsync(h_dst, d_src, (size_t)nbytes, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpy(D2H)");
    FSNAP_HIP(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    return FSNAP_OK;
}

int fsnap_dev_sync(fsnap_ctx* ctx) {
    if (!ctx) return FSNAP_E_ARG;
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    FSNAP_HIP(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    return FSNAP_OK;
}

int fsnap_residual_rhs(fsnap_ctx* ctx, const double* beta, double* s, double* sse) {
    if (!ctx) return FSNAP_E_ARG;
    int rc;
    if ((rc = check_rows(ctx)) || (rc = check_weights(ctx))) return rc;
    if (!beta || !s) return ctx->fail(FSNAP_E_ARG, "fsnap_residual_rhs: NULL argument");
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    const size_t m = (size_t)ctx->m, K = (size_t)ctx->K;
    // K <= 288: kernels 4 + 7 fused, the rows are read once (option fused_residual = 0: the two-kernel form, A/B)
    const bool fused = ctx->opt_fused_residual && K <= 288;
    const int nb = fused ? fsnap::residual_num_blocks(ctx->m, (int)ctx->K) : fsnap::gemv_num_blocks(ctx->m);
    const int nbt = fused ? nb : fsnap::gemvT_num_blocks(ctx->m);
    if (!ctx->beta.ensure(K * 8) || (!fused && !ctx->du.ensure(m * 8)) || !ctx->dspart.ensure((size_t)nbt * K * 8) ||
        !ctx->dsvec.ensure(K * 8) || (sse && !ctx->sse.ensure((size_t)nb * 8)))
        return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(refinement) failed");
    const unsigned char* mask = ctx->dmask;
    if (!mask) {
        if ((rc = ensure_ones(ctx))) return rc;
        mask = (const unsigned char*)ctx->ones.p;
    }
    FSNAP_HIP(hipMemcpyAsync(ctx->beta.p, beta, K * 8, hipMemcpyHostToDevice, ctx->stream), "hipMemcpy(beta)");
    if (fused) {
        FSNAP_HIP(fsnap::launch_residual_rows(ctx->dA, ctx->lda, (const double*)ctx->beta.p, ctx->m, (int)ctx->K, ctx->db, ctx->dw,
                                              mask, (double*)ctx->dspart.p, sse ? (double*)ctx->sse.p : nullptr,
                                              (double*)ctx->dsvec.p, ctx->stream),
                  "launch fsnap_residual_rows_k");
    } else {
        FSNAP_HIP(fsnap::launch_gemv_rows(ctx->dA, ctx->lda, (const double*)ctx->beta.p, ctx->m, (int)ctx->K, nullptr, ctx->db,
                                          ctx->dw, mask, sse ? (double*)ctx->sse.p : nullptr, (double*)ctx->du.p, ctx->stream),
                  "launch fsnap_gemv_rows_k");
        FSNAP_HIP(fsnap::launch_gemvT_rows(ctx->dA, ctx->lda, (const double*)ctx->du.p, ctx->m, (int)ctx->K,
                                           (double*)ctx->dspart.p, (double*)ctx->dsvec.p, ctx->stream),
                  "launch fsnap_gemvT_rows_k");
    }
    FSNAP_HIP(hipMemcpyAsync(s, ctx->dsvec.p, K * 8, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpy(s)");
    if (sse) {
        std::string tmp;
        tmp.resize((size_t)nb * 8);
        FSNAP_HIP(hipMemcpyAsync(&tmp[0], ctx->sse.p, (size_t)nb * 8, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpy(sse)");
        FSNAP_HIP(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
        const double* ps = (const double*)tmp.data();
        long double acc = 0.0L;
        for (int i = 0; i < nb; ++i) acc += ps[i];
        *sse = (double)acc;
    } else {
        FSNAP_HIP(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    }
    return FSNAP_OK;
}

int fsnap_timing(fsnap_ctx* ctx, double* ms, int n) {
    if (!ctx || !ms || n < 0 || n > 8) return FSNAP_E_ARG;
    FSNAP_HIP(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    double out[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float t = 0.f;
    if (ctx->t_syrk && ctx->nfit > 0) {
        hipEvent_t* sl = ctx->ring[(ctx->nfit - 1) % fsnap_ctx::RING];
        if (hipEventElapsedTime(&t, sl[0], sl[1]) == hipSuccess) out[0] = t;
        if (n > 1 && hipEventElapsedTime(&t, sl[1], sl[2]) == hipSuccess) out[1] = t;
    }
    if (n > 2 && ctx->t_upload && hipEventElapsedTime(&t, ctx->ev[3], ctx->ev[4]) == hipSuccess) out[2] = t;
    if (n > 3 && ctx->t_weight && hipEventElapsedTime(&t, ctx->ev[5], ctx->ev[6]) == hipSuccess) out[3] = t;
    if (n > 4 && ctx->t_predict && hipEventElapsedTime(&t, ctx->ev[7], ctx->ev[8]) == hipSuccess) out[4] = t;
    out[5] = ctx->upload_probe_gbps;          // last large fsnap_upload_rows: rate of the probed pageable copy (0: no probe)
    out[6] = ctx->upload_staged ? 1.0 : 0.0;  // ... and whether the rest went through the page-locked double buffer
    for (int i = 0; i < n; ++i) ms[i] = out[i];
    return FSNAP_OK;
}

int fsnap_timing_history(fsnap_ctx* ctx, double* syrk_ms, double* reduce_ms, int n) {
    if (!ctx || !syrk_ms || n < 0) return FSNAP_E_ARG;
    if (n > fsnap_ctx::RING || n > ctx->nfit) return ctx->fail(FSNAP_E_ARG, "fsnap_timing_history: only the last %d fits are kept",
                                                             (int)(ctx->nfit < fsnap_ctx::RING ? ctx->nfit : fsnap_ctx::RING));
    FSNAP_HIP(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    for (int i = 0; i < n; ++i) {
        hipEvent_t* sl = ctx->ring[(ctx->nfit - n + i) % fsnap_ctx::RING];
        float t = 0.f;
        FSNAP_HIP(hipEventElapsedTime(&t, sl[0], sl[1]), "hipEventElapsedTime");
        syrk_ms[i] = t;
        if (reduce_ms) {
            FSNAP_HIP(hipEventElapsedTime(&t, sl[1], sl[2]), "hipEventElapsedTime");
            reduce_ms[i] = t;
        }
    }
    return FSNAP_OK;
}

int fsnap_timing_history_comm(fsnap_ctx* ctx, double* allreduce_ms, int n) {
    if (!ctx || !allreduce_ms || n < 0) return FSNAP_E_ARG;
    if (n > fsnap_ctx::RING || n > ctx->nfit) return ctx->fail(FSNAP_E_ARG, "fsnap_timing_history_comm: only the last %d fits are kept",
                                                             (int)(ctx->nfit < fsnap_ctx::RING ? ctx->nfit : fsnap_ctx::RING));
    int wrc;
    if ((wrc = fsnap::wait_stream(ctx, nullptr, "timing history"))) return wrc;
    for (int i = 0; i < n; ++i) {
        const int64_t k = (ctx->nfit - n + i) % fsnap_ctx::RING;
        float t = 0.f;
        allreduce_ms[i] = -1.0;                         // not a multi-GPU fit
        if (ctx->ring_comm[k]) {
            FSNAP_HIP(hipEventElapsedTime(&t, ctx->ring[k][2], ctx->ring[k][3]), "hipEventElapsedTime");
            allreduce_ms[i] = t;
        }
    }
    return FSNAP_OK;
}

int fsnap_timing_count(fsnap_ctx* ctx, int64_t* sampled, int64_t* launches) {
    if (!ctx) return FSNAP_E_ARG;
    if (sampled) *sampled = ctx->nfit;
    if (launches) *launches = ctx->nlaunch;
    return FSNAP_OK;
}

int fsnap_launch_info(fsnap_ctx* ctx, int64_t* info, int n) {
    if (!ctx || !info || n < 0 || n > 8) return FSNAP_E_ARG;
    int rc;
    if ((rc = check_rows(ctx))) return rc;
    int64_t out[8] = {0, 0, 0, 0, 0, ctx->num_cu, 0, 0};
    if (use_tiled(ctx)) {
        TiledGeometry t;
        if ((rc = plan_tiled(ctx, &t))) return rc;
        out[0] = (int64_t)t.npairs * t.nsplit;
        out[1] = 256;
        out[2] = t.cps / 4;
        out[3] = 4 * t.NSB;
        out[4] = 0;  // 0 = tiled kernel
        out[6] = t.npairs;
        out[7] = t.nsplit;
    } else {
        Geometry g;
        if ((rc = plan_geometry(ctx, &g))) return rc;
        out[0] = g.nblocks;
        out[1] = g.threads;
        out[2] = g.cpw;
        out[3] = g.NB;
        out[4] = g.split;
        out[6] = g.shortk ? 7 : g.quad ? (g.cluster > 1 ? 6 : 5) : g.acc ? 3 : 4;
        out[7] = g.fused_pack ? 1 : 0;  // kernel id: 1 = wave-triangle, 2 = LDS-shared, 3 = one-wave triangle (1A), 4 = wave-triangle on packed weights (1P), 5 = triangle dealt to the four waves of a workgroup (1Q; chunks per WORKGROUP)
    }
    for (int i = 0; i < n; ++i) info[i] = out[i];
    return FSNAP_OK;
}

}  // extern "C"
