// fsnap_comm.cpp — the multi-GPU exchange step of the linear fit behind the C ABI: one RCCL communicator per
// context (one process per GPU), collectives on the context's stream.
//
// Replaces the reference's mpi4py calls on this path:
//   comm.Allreduce(c), comm.Allreduce(d)     examples/library/transpose_trick/example.py:245-246
//   comm.bcast / comm.allgather of the small control data around a fit (fitsnap3lib/parallel_tools.py:426-441, 562-577)
//
// RCCL is loaded with dlopen at the first fsnap_comm_* call, never at library load: a single-GPU process does not
// map librccl at all, and a process that already holds a copy (e.g. the one bundled with PyTorch) reuses it.
// xGMI is point-to-point; the payload of a fit is K*K + K + 3 doubles (132 KB at K = 128), i.e. latency-bound: ONE
// in-place ncclAllReduce per fit, enqueued on the same stream as the kernels (no host synchronisation in between).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <future>
#include <memory>
#include <string>
#include <new>
#include <thread>

#include "fsnap_ctx.h"
#include "fsnap_kernels.h"
#include "fsnap_p2p.h"

namespace fsnap {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string why;
};

static Rccl* rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return &r;
    tried = true;
    // FSNAP_RCCL_PATH: the copy that belongs to the HIP runtime of this process (the ctypes shim points it at the
    // librccl bundled with PyTorch when it mapped PyTorch's libamdhip64, so that one ROCm stack serves the process)
    const char* names[] = {getenv("FSNAP_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);   // LOCAL: its symbols must not interpose on libraries loaded later (PyTorch)
        if (r.handle) break;
    }
    if (!r.handle) {
        r.why = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "not found");
        return &r;
    }
    bool ok = true;
    auto sym = [&](const char* name) -> void* {
        void* p = dlsym(r.handle, name);
        if (!p) {
            ok = false;
            r.why = std::string("librccl has no symbol ") + name;
        }
        return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.CommAbort = (decltype(r.CommAbort))sym("ncclCommAbort");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!ok) {
        dlclose(r.handle);
        r.handle = nullptr;
    }
    return &r;
}

struct Comm {
    ncclComm_t nccl = nullptr;     // transport 1: RCCL
    P2P* p2p = nullptr;            // transport 2: one-shot all-reduce over hipIpc-mapped windows (fsnap_p2p.cpp)
    int nranks = 1, rank = 0;
};

double comm_timeout_s() {
    static const double t = [] {
        const char* e = getenv("FSNAP_COMM_TIMEOUT");
        const double v = e && *e ? atof(e) : 300.0;
        return v > 0.0 ? v : 300.0;
    }();
    return t;
}

double comm_timeout_s(const fsnap_ctx* ctx) { return ctx && ctx->opt_comm_timeout > 0 ? (double)ctx->opt_comm_timeout : comm_timeout_s(); }

int wait_stream(fsnap_ctx* ctx, hipEvent_t ev, const char* what) {
    const bool bounded = ctx->comm != nullptr;
    if (!bounded && !ev) {
        FSNAP_HIP(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
        return FSNAP_OK;
    }
    // busy poll (the blocking wait's wake-up latency is several microseconds); the clock is read every 1024 polls
    std::chrono::steady_clock::time_point deadline;
    bool armed = false;
    for (unsigned spins = 0;; ++spins) {
        const hipError_t q = ev ? hipEventQuery(ev) : hipStreamQuery(ctx->stream);
        if (q == hipSuccess) {
            if (bounded && p2p_failed(ctx->comm->p2p)) {          // the bounded wait INSIDE a peer-to-peer kernel ran out
                ctx->comm_broken = true;
                return ctx->fail(FSNAP_E_HIP,
                                 "rank %d of %d: a peer's statistics did not arrive within %.0f s (FSNAP_COMM_TIMEOUT / option "
                                 "comm_timeout) in the peer-to-peer all-reduce before %s: a peer rank died or never reached it",
                                 ctx->comm->rank, ctx->comm->nranks, comm_timeout_s(ctx), what);
            }
            return FSNAP_OK;
        }
        if (q != hipErrorNotReady) return ctx->hipfail(q, ev ? "hipEventQuery" : "hipStreamQuery");
        if (bounded && (spins & 1023u) == 1023u) {
            const auto now = std::chrono::steady_clock::now();
            if (!armed) {
                deadline = now + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double>(comm_timeout_s(ctx)));
                armed = true;
            } else if (now > deadline) {
                ctx->comm_broken = true;
                return ctx->fail(FSNAP_E_HIP,
                                 "rank %d of %d: %s did not finish within %.0f s (FSNAP_COMM_TIMEOUT / option comm_timeout) behind a "
                                 "collective: a peer rank died or never reached it",
                                 ctx->comm->rank, ctx->comm->nranks, what, comm_timeout_s(ctx));
            }
        }
    }
}

}  // namespace fsnap

using fsnap::Comm;
using fsnap::Rccl;

namespace {

int nccl_fail(fsnap_ctx* ctx, Rccl* r, ncclResult_t e, const char* what) {
    return ctx->fail(FSNAP_E_HIP, "%s: %s", what, r->GetErrorString ? r->GetErrorString(e) : "RCCL error");
}

#define FSNAP_NCCL(call, what)                                      \
    do {                                                            \
        ncclResult_t _e = (call);                                   \
        if (_e != ncclSuccess) return nccl_fail(ctx, r, _e, what);  \
    } while (0)

int need_comm(fsnap_ctx* ctx, Rccl** r) {
    if (!ctx->comm || !(ctx->comm->nccl || ctx->comm->p2p)) return ctx->fail(FSNAP_E_STATE, "no communicator: call fsnap_comm_init first");
    // a bounded wait behind a collective ran out earlier: the ranks are no longer in step (the peer-to-peer transport counts
    // its collectives), and whatever is entered now can only run into the same deadline again
    if (ctx->comm_broken && ctx->comm->p2p)
        return ctx->fail(FSNAP_E_STATE, "the communicator is broken (an earlier collective timed out): destroy the context");
    *r = ctx->comm->p2p ? nullptr : fsnap::rccl();
    return FSNAP_OK;
}

}  // namespace

extern "C" {

int fsnap_comm_id_p2p(char* id) {
    if (!id) return FSNAP_E_ARG;
    return fsnap::p2p_make_id(id);
}

int fsnap_comm_id(char* id) {
    if (!id) return FSNAP_E_ARG;
    const char* tr = getenv("FSNAP_DIST_TRANSPORT");
    if (tr && !strcmp(tr, "p2p")) return fsnap::p2p_make_id(id);
    Rccl* r = fsnap::rccl();
    if (!r->handle) {
        fsnap::library_error() = r->why;
        return FSNAP_E_HIP;
    }
    static_assert(sizeof(ncclUniqueId) == FSNAP_COMM_ID_BYTES, "FSNAP_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");
    ncclUniqueId uid;
    const ncclResult_t e = r->GetUniqueId(&uid);
    if (e != ncclSuccess) {
        fsnap::library_error() = std::string("ncclGetUniqueId: ") + r->GetErrorString(e);
        return FSNAP_E_HIP;
    }
    memcpy(id, &uid, sizeof uid);
    return FSNAP_OK;
}

int fsnap_comm_init(fsnap_ctx* ctx, int nranks, int rank, const char* id) {
    if (!ctx) return FSNAP_E_ARG;
    if (!id || nranks < 1 || rank < 0 || rank >= nranks) return ctx->fail(FSNAP_E_ARG, "fsnap_comm_init: bad argument");
    if (ctx->comm) return ctx->fail(FSNAP_E_STATE, "fsnap_comm_init: this context already has a communicator");
    if (fsnap::p2p_is_id(id)) {               // the transport travels with the id: every rank takes the same one
        FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
        Comm* c = new (std::nothrow) Comm();
        if (!c) return ctx->fail(FSNAP_E_NOMEM, "out of host memory");
        const int rc = fsnap::p2p_init(ctx, nranks, rank, id, &c->p2p);
        if (rc) {
            delete c;
            ctx->comm_broken = false;
            return rc;
        }
        c->nranks = nranks;
        c->rank = rank;
        ctx->comm = c;
        return FSNAP_OK;
    }
    Rccl* r = fsnap::rccl();
    if (!r->handle) return ctx->fail(FSNAP_E_HIP, "%s", r->why.c_str());
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    Comm* c = new (std::nothrow) Comm();
    if (!c) return ctx->fail(FSNAP_E_NOMEM, "out of host memory");
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    // ncclCommInitRank blocks until ALL ranks have joined; with a rank missing (crashed before this point, stale id) it
    // never returns and cannot be cancelled.  It therefore runs in a thread of its own: after FSNAP_COMM_TIMEOUT seconds
    // this call gives up with an error and leaves the thread behind (the process is expected to exit).
    struct Shared {
        std::promise<ncclResult_t> done;
        ncclComm_t nccl = nullptr;
    };
    auto sh = std::make_shared<Shared>();
    std::future<ncclResult_t> fut = sh->done.get_future();
    const int device = ctx->device;
    auto init = r->CommInitRank;
    std::thread([sh, init, uid, nranks, rank, device]() {
        (void)hipSetDevice(device);
        ncclComm_t comm = nullptr;
        const ncclResult_t e = init(&comm, nranks, uid, rank);
        sh->nccl = comm;
        sh->done.set_value(e);
    }).detach();
    if (fut.wait_for(std::chrono::duration<double>(fsnap::comm_timeout_s(ctx))) != std::future_status::ready) {
        delete c;
        return ctx->fail(FSNAP_E_HIP,
                         "ncclCommInitRank: rank %d of %d did not complete within %.0f s (FSNAP_COMM_TIMEOUT): a rank is missing "
                         "or holds a different communicator id",
                         rank, nranks, fsnap::comm_timeout_s(ctx));
    }
    const ncclResult_t e = fut.get();
    c->nccl = sh->nccl;
    if (e != ncclSuccess) {
        delete c;
        return nccl_fail(ctx, r, e, "ncclCommInitRank");
    }
    c->nranks = nranks;
    c->rank = rank;
    ctx->comm = c;
    return FSNAP_OK;
}

int fsnap_comm_destroy(fsnap_ctx* ctx) {
    if (!ctx) return FSNAP_E_ARG;
    if (!ctx->comm) return FSNAP_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->comm->p2p) {
        // a kernel whose wait ran out has left by itself (its wait is bounded too): the stream can always be drained
        if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
        fsnap::p2p_destroy(ctx, ctx->comm->p2p, ctx->comm_broken);
        delete ctx->comm;
        ctx->comm = nullptr;
        ctx->comm_broken = false;
        return FSNAP_OK;
    }
    Rccl* r = fsnap::rccl();
    if (ctx->comm_broken) {
        // a wait behind a collective ran out: the stream holds an RCCL kernel that waits for a dead peer.  Abort makes
        // that kernel leave; synchronising first would hang
        if (ctx->comm->nccl && r->handle && r->CommAbort) (void)r->CommAbort(ctx->comm->nccl);
    } else {
        if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
        if (ctx->comm->nccl && r->handle) (void)r->CommDestroy(ctx->comm->nccl);
    }
    delete ctx->comm;
    ctx->comm = nullptr;
    ctx->comm_broken = false;
    return FSNAP_OK;
}

int fsnap_comm_transport(fsnap_ctx* ctx, int* transport) {
    if (!ctx) return FSNAP_E_ARG;
    if (transport) *transport = !ctx->comm ? 0 : (ctx->comm->p2p ? 2 : 1);
    return FSNAP_OK;
}

int fsnap_comm_info(fsnap_ctx* ctx, int* nranks, int* rank) {
    if (!ctx) return FSNAP_E_ARG;
    if (nranks) *nranks = ctx->comm ? ctx->comm->nranks : 1;
    if (rank) *rank = ctx->comm ? ctx->comm->rank : 0;
    return FSNAP_OK;
}

int fsnap_allreduce_device(fsnap_ctx* ctx, double* d_buf, int64_t n) {
    if (!ctx) return FSNAP_E_ARG;
    if (!d_buf || n <= 0) return ctx->fail(FSNAP_E_ARG, "fsnap_allreduce_device: bad argument");
    Rccl* r;
    int rc;
    if ((rc = need_comm(ctx, &r))) return rc;
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    ctx->chol_factor_of = nullptr;              // (the buffer may be statistics a factor on the device belongs to)
    if (ctx->comm->p2p) return fsnap::p2p_allreduce_device(ctx, ctx->comm->p2p, d_buf, n);
    FSNAP_NCCL(r->AllReduce(d_buf, d_buf, (size_t)n, ncclDouble, ncclSum, ctx->comm->nccl, ctx->stream), "ncclAllReduce");
    return FSNAP_OK;
}

}  // extern "C"

namespace fsnap {
// In-place sum over the ranks of the packed statistics [G | c | scalars] of a K-column system.  Wide systems (option
// reduce_triangle: -1 = from 256 columns on, 0 = never, 1 = always) travel as [upper triangle | c | scalars] --
// K (K + 1) / 2 + K + 3 doubles instead of K^2 + K + 3 -- between a pack and an unpack kernel on the same stream; the
// unpack mirrors the triangle, so the reduced G is symmetric to the last bit either way.
static bool triangle_form(const fsnap_ctx* ctx, int64_t K) {
    return ctx->opt_reduce_triangle == 1 || (ctx->opt_reduce_triangle < 0 && K >= 256);
}

// the buffer of the triangle form, to be reserved where a rank may still fail alone (BEFORE its first collective)
bool allreduce_packed_reserve(fsnap_ctx* ctx, int64_t K) {
    return !triangle_form(ctx, K) || ctx->tribuf.ensure((size_t)(K * (K + 1) / 2 + K + 3) * 8);
}

int allreduce_packed(fsnap_ctx* ctx, double* dp, int64_t K) {
    if (!triangle_form(ctx, K)) return fsnap_allreduce_device(ctx, dp, FSNAP_PACKED_LEN(K));
    const int64_t nt = K * (K + 1) / 2 + K + 3;
    if (!ctx->tribuf.ensure((size_t)nt * 8))          // reserved by the callers before their first collective
        return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(triangle payload) failed");
    FSNAP_HIP(fsnap::launch_tri_pack(dp, (int)K, (double*)ctx->tribuf.p, ctx->stream), "launch fsnap_tri_pack_k");
    const int rc = fsnap_allreduce_device(ctx, (double*)ctx->tribuf.p, nt);
    if (rc) return rc;
    FSNAP_HIP(fsnap::launch_tri_unpack((const double*)ctx->tribuf.p, (int)K, dp, ctx->stream), "launch fsnap_tri_unpack_k");
    return FSNAP_OK;
}
}  // namespace fsnap

extern "C" {

int fsnap_allreduce_host(fsnap_ctx* ctx, double* buf, int64_t n, int op) {
    if (!ctx) return FSNAP_E_ARG;
    if (!buf || n <= 0 || op < 0 || op > 2) return ctx->fail(FSNAP_E_ARG, "fsnap_allreduce_host: bad argument");
    Rccl* r;
    int rc;
    if ((rc = need_comm(ctx, &r))) return rc;
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    if (ctx->comm->p2p) return fsnap::p2p_allreduce_host(ctx, ctx->comm->p2p, buf, n, op);      // host mailboxes: no launch, no staging
    if (!ctx->commbuf.ensure((size_t)n * 8)) return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(collective staging) failed");
    const ncclRedOp_t rop = op == 0 ? ncclSum : (op == 1 ? ncclMax : ncclMin);
    FSNAP_HIP(hipMemcpyAsync(ctx->commbuf.p, buf, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream), "hipMemcpy(H2D)");
    FSNAP_NCCL(r->AllReduce(ctx->commbuf.p, ctx->commbuf.p, (size_t)n, ncclDouble, rop, ctx->comm->nccl, ctx->stream), "ncclAllReduce");
    FSNAP_HIP(hipMemcpyAsync(buf, ctx->commbuf.p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpy(D2H)");
    return fsnap::wait_stream(ctx, nullptr, "all-reduce");
}

int fsnap_bcast_host(fsnap_ctx* ctx, void* buf, int64_t nbytes, int root) {
    if (!ctx) return FSNAP_E_ARG;
    Rccl* r;
    int rc;
    if ((rc = need_comm(ctx, &r))) return rc;
    if (!buf || nbytes <= 0 || root < 0 || root >= ctx->comm->nranks) return ctx->fail(FSNAP_E_ARG, "fsnap_bcast_host: bad argument");
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    if (ctx->comm->p2p) return fsnap::p2p_bcast_host(ctx, ctx->comm->p2p, buf, (size_t)nbytes, root);
    if (!ctx->commbuf.ensure((size_t)nbytes)) return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(collective staging) failed");
    if (ctx->comm->rank == root)
        FSNAP_HIP(hipMemcpyAsync(ctx->commbuf.p, buf, (size_t)nbytes, hipMemcpyHostToDevice, ctx->stream), "hipMemcpy(H2D)");
    FSNAP_NCCL(r->Broadcast(ctx->commbuf.p, ctx->commbuf.p, (size_t)nbytes, ncclUint8, root, ctx->comm->nccl, ctx->stream), "ncclBroadcast");
    if (ctx->comm->rank != root)
        FSNAP_HIP(hipMemcpyAsync(buf, ctx->commbuf.p, (size_t)nbytes, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpy(D2H)");
    return fsnap::wait_stream(ctx, nullptr, "broadcast");
}

int fsnap_allgather_host(fsnap_ctx* ctx, const void* send, int64_t nbytes, void* recv) {
    if (!ctx) return FSNAP_E_ARG;
    Rccl* r;
    int rc;
    if ((rc = need_comm(ctx, &r))) return rc;
    if (!send || !recv || nbytes <= 0) return ctx->fail(FSNAP_E_ARG, "fsnap_allgather_host: bad argument");
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    if (ctx->comm->p2p) return fsnap::p2p_allgather_host(ctx, ctx->comm->p2p, send, (size_t)nbytes, recv);
    const size_t nb = (size_t)nbytes, total = nb * (size_t)ctx->comm->nranks;
    if (!ctx->commbuf.ensure(nb + total)) return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(collective staging) failed");
    char* d_send = (char*)ctx->commbuf.p;
    char* d_recv = d_send + nb;
    FSNAP_HIP(hipMemcpyAsync(d_send, send, nb, hipMemcpyHostToDevice, ctx->stream), "hipMemcpy(H2D)");
    FSNAP_NCCL(r->AllGather(d_send, d_recv, nb, ncclUint8, ctx->comm->nccl, ctx->stream), "ncclAllGather");
    FSNAP_HIP(hipMemcpyAsync(recv, d_recv, total, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpy(D2H)");
    return fsnap::wait_stream(ctx, nullptr, "all-gather");
}

int fsnap_barrier(fsnap_ctx* ctx) {
    if (!ctx) return FSNAP_E_ARG;
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    if (!ctx->comm) {
        FSNAP_HIP(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
        return FSNAP_OK;
    }
    if (ctx->comm->p2p) {
        int rc;
        if ((rc = fsnap::wait_stream(ctx, nullptr, "barrier"))) return rc;
        return fsnap::p2p_barrier(ctx, ctx->comm->p2p);
    }
    double one = 1.0;
    return fsnap_allreduce_host(ctx, &one, 1, 0);      // a 1-element all-reduce + stream synchronisation
}

}  // extern "C"
