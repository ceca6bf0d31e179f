// fsnap_ctx.h — the context object behind the opaque fsnap_ctx of include/fsnap_hip.h, shared by the translation
// units of the C-ABI layer (fsnap_capi.cpp: rows / statistics / solve; fsnap_comm.cpp: RCCL communicator;
// fsnap_rowspace.cpp: row-space least squares).  Internal; not part of the public boundary.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/fsnap_hip.h"

namespace fsnap {

// From this order on fsnap_solve_device factorises on the GPU (blocked kernels 8a-8e), below it on the host.  Measured
// (scripts/chol_large_test.py, round 4, EPYC 9575F host): K = 256 0.118 ms GPU / 0.130 host, 272 0.147 / 0.145, 288 0.147 /
// 0.162, 320 0.147 / 0.203, 352 0.171 / 0.249, 384 0.171 / 0.357; 192 0.091 / 0.073.  (384 until round 4: the device chain was
// 0.30 ms there when the threshold was set.)  Round 5 (one launch per panel, four-wave diagonal block): K = 192 0.076 ms GPU /
// 0.066 host, 224 0.097 / 0.094, 256 0.096 / 0.131, 288 0.117 / 0.170, 320 0.117 / 0.204, 384 0.138 / 0.356 -- the padded order
// (multiple of 64) sets the GPU's time, so the threshold sits at the first order of the 256 bucket the host no longer wins.
constexpr int64_t DEVICE_CHOL_MIN_K = 232;

// last error text of the calling thread when no context is at hand (fsnap_last_error(NULL))
std::string& library_error();

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    bool ensure(size_t n) {
        if (n <= bytes && p) return true;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        if (hipMalloc(&p, n) != hipSuccess) {
            p = nullptr;
            return false;
        }
        bytes = n;
        return true;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }      // every buffer of a context goes with it (fsnap_ctx_destroy selects the device first)
};

struct Comm;       // fsnap_comm.cpp: RCCL communicator of this rank (nullptr = single GPU)
struct RowSpace;   // fsnap_rowspace.cpp: buffers of the row-space least-squares path

}  // namespace fsnap

using fsnap::DevBuf;

struct fsnap_ctx {
    int device = 0;
    int num_cu = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev[10] = {};
    static constexpr int RING = 256;              // event sets of the last RING fits (fsnap_timing_history): before SYRK,
                                                  // after SYRK, after the reduction, after the all-reduce (fsnap_fit_dist)
    hipEvent_t ring[RING][4] = {};
    bool ring_comm[RING] = {};                    // slot's 4th event was recorded (a multi-GPU fit)
    hipEvent_t* cur_events = nullptr;             // event set of the fit being launched (nullptr: not a sampled one)
    double* chol_host = nullptr;                  // page-locked [beta | panel pivots | status] written by the device Cholesky
    size_t chol_host_bytes = 0;
    hipEvent_t chol_ev = nullptr;
    const void* chol_status_word = nullptr;       // device status word the last chain left cleared
    int64_t cpart_key = -1;                       // geometry the tiled kernel's c partials were last cleared for
    const void* cpart_ptr = nullptr;
    int64_t nlaunch = 0;                          // SYRK launches so far (sampled ones: nfit)
    int64_t timing_phase = 0;                     // launches since the option was last set
    int opt_timing_every = 1;                     // bracket every N-th SYRK launch with events (0: none)
    int64_t nfit = 0;                             // fits launched so far
    std::string err;

    // rows
    const double* dA = nullptr;
    const double* db = nullptr;
    int64_t m = 0, K = 0, lda = 0;
    DevBuf ownA, ownb;
    // weights
    const double* dw = nullptr;
    const unsigned char* dmask = nullptr;
    DevBuf ownw, ownmask, ones;
    // workspaces
    DevBuf part, cpart, spart, packed, beta, preds, sse, aw, bw;
    DevBuf quad_flow;            // kernel 1QC: flow-control words of the clusters (zeroed when allocated, never reset)
    unsigned quad_flow_tag = 0;  // + 2^20 per launch of kernel 1QC (wraps)
    DevBuf st_raw, st_plan, st_frac, st_blank;   // staging of fsnap_assemble
    DevBuf fz_rows, fz_spart, fz_part, fz_cpart; // fsnap_assemble_accumulate: per-row scratch (no A) and partials
    DevBuf dsolve;                                // [beta | min pivot | status] of fsnap_solve_device
    DevBuf dchol;                                 // padded work matrix of the blocked device Cholesky
    DevBuf dcat, dstat;                           // fsnap_error_stats: row categories, partial tables + means
    DevBuf wpack, wpack_spart;                    // kernel 1A: packed (w_eff, w_eff b) per row + partial b-only scalars
    bool wpack_valid = false;                     // false after anything that can change b, w or the mask
    const double* wpack_override = nullptr;       // per-row pairs to use INSTEAD of wpack (row-space passes only)
    int64_t dcat_rows = -1;                       // number of rows the categories on the device belong to
    DevBuf du, dspart, dsvec;                     // refinement: row weights u, per-workgroup partials, s
    double* pinned = nullptr;                     // page-locked host staging of the packed statistics: plain (coarse-grained)
                                                  // pinned memory, the target of DMA copies only -- copies into COHERENT
                                                  // host memory were bimodal (2 MB in 0.05 or in 8 ms)
    size_t pinned_bytes = 0;
    double* mirror = nullptr;                     // page-locked host mirror written by the reduction kernel itself
    size_t mirror_bytes = 0;
    const double* mirror_of = nullptr;            // device buffer the mirror currently reflects (nullptr = stale)
    int64_t mirror_K = 0;                         // order of the system in the mirror
    unsigned long long mirror_gen = 0;            // fills of the mirror so far: (context, mirror_gen) tags one content of it
    bool mirror_upper = false;                    // only the upper triangle of the mirror's G is meaningful (kernel 2b)
    hipEvent_t mirror_ev = nullptr;               // recorded after the reduction that filled the mirror
    // options
    int opt_nblocks = 0;      // 0 = auto
    int opt_tiled = 0;        // force the general-K tiled kernel also for K <= 128
    int opt_short = -1;       // kernel 1S for 81 ... 144 columns: -1 = systems of at most SHORT_MAX_ROWS rows, 0 = never, 1 = always
    int opt_nsplit = 0;       // row splits of the tiled kernel (0 = auto)
    int opt_device_solve = 0; // 0 = auto (K >= DEVICE_CHOL_MIN_K on the GPU, blocked kernels), 1 = every K (K <= 128: fsnap_chol_solve_k), 2 = never
    // cached launch plan of the tiled kernel (plan_tiled)
    bool tplan_valid = false;
    int64_t tplan_key[4] = {0, 0, 0, 0};
    int tplan[3] = {0, 0, 0};
    int64_t tplan_cps = 0;
    // timing flags
    bool t_syrk = false, t_upload = false, t_weight = false, t_predict = false;

    // page-locked staging of fsnap_set_weights*: two slots used alternately, an event per slot marks the end of its
    // DMA -- the call returns while the copy is still in flight (no stream synchronisation on the re-weighting path)
    char* wstage[2] = {nullptr, nullptr};
    size_t wstage_bytes[2] = {0, 0};
    hipEvent_t wstage_ev[2] = {nullptr, nullptr};
    int wstage_next = 0;
    // the factor in `dchol`: of which statistics buffer / order / shift / panel-loop form (nullptr = none; cleared by every launch that
    // rewrites statistics or the work matrix)
    const double* chol_factor_of = nullptr;
    int64_t chol_factor_K = 0;
    double chol_factor_alpha = 0.0;
    double chol_factor_rcond = 0.0;        // what the solve that left the factor reported as *rcond_est (min(pivot, lambda_min estimate))
    double chol_factor_piv = 0.0, chol_factor_lam = 0.0;
    std::vector<double> probe_gram;        // B^T B of the probe vectors the device factorisation carries (condition estimate), for probe_n rows
    int probe_n = 0;
    // small host -> device uploads (weights, masks): which way is faster on THIS box is measured on the first calls (see staged_h2d)
    int h2d_method = -1;                   // -1 undecided, 0 = the runtime's pageable copy (waited for), 1 = page-locked staging
    int h2d_probes = 0;
    double h2d_best[2] = {1.0e30, 1.0e30}; // seconds until the data was on the device, best of the probes
    hipEvent_t h2d_ev = nullptr;
    // page-locked double buffer of fsnap_upload_rows (two 16 MiB slots: host threads fill one while the DMA drains the other)
    char* rstage[2] = {nullptr, nullptr};
    hipEvent_t rstage_ev[2] = {nullptr, nullptr};
    bool rstage_busy[2] = {false, false};   // a DMA out of the slot is (possibly) still in flight: wait for its event before refilling
    DevBuf wtrain, wrank;                         // compact training weights and the mask's exclusive prefix sum
    int64_t ntrain_resident = -1;                 // training rows of the resident mask / prefix (-1 = none)

    // multi-GPU / row-space state owned by the other translation units
    fsnap::Comm* comm = nullptr;
    fsnap::RowSpace* rowspace = nullptr;
    int (*dense_pinv)(void*, int64_t, int64_t, const double*, double, const double*, double*, int*) = nullptr;
    void* dense_pinv_user = nullptr;
    int64_t dense_pinv_token = 0;
    DevBuf commbuf;                               // device staging of host-buffer collectives
    DevBuf tribuf;                                // [upper triangle | c | scalars]: all-reduce payload of wide systems
    int opt_reduce_triangle = -1;                 // all-reduce the triangle only: -1 = K >= 256, 0 = never, 1 = always
    int opt_staged_upload = 0;    // fsnap_upload_rows: 0 pageable hipMemcpy (default) | 2 page-locked double buffer | 1 double buffer, probed
    double upload_probe_gbps = 0.0;   // rate at which the host filled the first page-locked slots of the last large upload (GB/s)
    bool upload_staged = false;       // the whole of the last upload went through the double buffer
    int opt_fused_residual = 1;   // fsnap_residual_rhs, K <= 288: 1 one pass over the rows | 0 kernels 4 + 7, two passes (the form of wider systems)
    int opt_fused_pack = 1;   // kernel 1A forms the per-row pairs of its rows in LDS itself (no packing launch) when they fit
    int opt_rowspace_reuse = 0;  // one-shot, set by the caller right before fsnap_lstsq_rows: the statistics of the fit that just ran (still in the page-locked mirror) are those of the rows as they are now -- the first pass starts from them
    int opt_chol_reuse = 1;   // fsnap_solve_device_rhs with a right-hand side of its own (refinement): 1 = forward + backward sweep with the factor the last solve of the same statistics left on the device, 0 = factorise again (A/B)
    int64_t opt_quad_min_rows = -1;   // fewest rows for kernel 1Q (-1 = default)
    int opt_repack = 0;       // 1 = pack (w_eff, w_eff b) on every launch even when b / w / mask are context-owned
    int opt_comm_timeout = 0; // seconds; 0 = FSNAP_COMM_TIMEOUT (default 300): bound of every wait behind a collective of THIS context
    bool comm_broken = false; // a bounded wait behind a collective ran out: the stream may hold a stuck RCCL kernel

    int fail(int code, const char* fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        fsnap::library_error() = buf;
        return code;
    }
    int hipfail(hipError_t e, const char* what) {
        return fail(e == hipErrorOutOfMemory ? FSNAP_E_NOMEM : FSNAP_E_HIP, "%s: %s", what, hipGetErrorString(e));
    }
};


#define FSNAP_HIP(call, what)                                \
    do {                                                     \
        hipError_t _e = (call);                              \
        if (_e != hipSuccess) return ctx->hipfail(_e, what); \
    } while (0)

namespace fsnap {
// fsnap_comm.cpp: seconds a wait behind a collective may take (FSNAP_COMM_TIMEOUT, default 300)
double comm_timeout_s();                      // FSNAP_COMM_TIMEOUT (seconds, default 300)
double comm_timeout_s(const fsnap_ctx* ctx);  // the context's option comm_timeout, else the above
// fsnap_comm.cpp: wait for the context's stream (ev == nullptr) or for an event on it.  Without a communicator this is
// a plain busy poll / hipStreamSynchronize; with one the wait is bounded by FSNAP_COMM_TIMEOUT -- a peer that died
// before its collective leaves this rank's stream stuck in an RCCL kernel -- and runs out with FSNAP_E_HIP + one line
// in fsnap_last_error; the communicator is then aborted instead of destroyed when the context goes away.
int wait_stream(fsnap_ctx* ctx, hipEvent_t ev, const char* what);
// fsnap_comm.cpp: in-place sum over the ranks of packed statistics [G | c | scalars]; wide systems as a triangle (option
// reduce_triangle)
int allreduce_packed(fsnap_ctx* ctx, double* dp, int64_t K);
bool allreduce_packed_reserve(fsnap_ctx* ctx, int64_t K);     // its buffer, to be reserved before a caller's first collective
// fsnap_capi.cpp: statistics of OTHER rows than the resident ones with the resident rows' launch plan -- the passes
// of the row-space solve run the same SYRK kernels on the orthogonalised copy Q (m x K, leading dimension ldq) with
// per-row pairs qpack = (1, w_eff b); d_packed receives [Q^T Q | Q^T b_w | ...]
int normal_eq_launch_on(fsnap_ctx* ctx, const double* Q, int64_t ldq, const double* qpack, double* d_packed);
// fsnap_capi.cpp: makes sure ctx->wpack holds (w_eff, w_eff b) of the current b / w / mask
int wpack_current(fsnap_ctx* ctx);
// fsnap_rowspace.cpp
void rowspace_release(fsnap_ctx* ctx);
}  // namespace fsnap
