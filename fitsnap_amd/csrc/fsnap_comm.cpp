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

#include <cstdlib>
#include <cstring>
#include <new>

#include "fsnap_ctx.h"

namespace fsnap {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
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
    ncclComm_t nccl = nullptr;
    int nranks = 1, rank = 0;
};

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
    if (!ctx->comm || !ctx->comm->nccl) return ctx->fail(FSNAP_E_STATE, "no communicator: call fsnap_comm_init first");
    *r = fsnap::rccl();
    return FSNAP_OK;
}

}  // namespace

extern "C" {

int fsnap_comm_id(char* id) {
    if (!id) return FSNAP_E_ARG;
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
    Rccl* r = fsnap::rccl();
    if (!r->handle) return ctx->fail(FSNAP_E_HIP, "%s", r->why.c_str());
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    Comm* c = new (std::nothrow) Comm();
    if (!c) return ctx->fail(FSNAP_E_NOMEM, "out of host memory");
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    const ncclResult_t e = r->CommInitRank(&c->nccl, nranks, uid, rank);
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
    Rccl* r = fsnap::rccl();
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm->nccl && r->handle) (void)r->CommDestroy(ctx->comm->nccl);
    delete ctx->comm;
    ctx->comm = nullptr;
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
    FSNAP_NCCL(r->AllReduce(d_buf, d_buf, (size_t)n, ncclDouble, ncclSum, ctx->comm->nccl, ctx->stream), "ncclAllReduce");
    return FSNAP_OK;
}

int fsnap_allreduce_host(fsnap_ctx* ctx, double* buf, int64_t n, int op) {
    if (!ctx) return FSNAP_E_ARG;
    if (!buf || n <= 0 || op < 0 || op > 2) return ctx->fail(FSNAP_E_ARG, "fsnap_allreduce_host: bad argument");
    Rccl* r;
    int rc;
    if ((rc = need_comm(ctx, &r))) return rc;
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    if (!ctx->commbuf.ensure((size_t)n * 8)) return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(collective staging) failed");
    const ncclRedOp_t rop = op == 0 ? ncclSum : (op == 1 ? ncclMax : ncclMin);
    FSNAP_HIP(hipMemcpyAsync(ctx->commbuf.p, buf, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream), "hipMemcpy(H2D)");
    FSNAP_NCCL(r->AllReduce(ctx->commbuf.p, ctx->commbuf.p, (size_t)n, ncclDouble, rop, ctx->comm->nccl, ctx->stream), "ncclAllReduce");
    FSNAP_HIP(hipMemcpyAsync(buf, ctx->commbuf.p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpy(D2H)");
    FSNAP_HIP(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    return FSNAP_OK;
}

int fsnap_bcast_host(fsnap_ctx* ctx, void* buf, int64_t nbytes, int root) {
    if (!ctx) return FSNAP_E_ARG;
    Rccl* r;
    int rc;
    if ((rc = need_comm(ctx, &r))) return rc;
    if (!buf || nbytes <= 0 || root < 0 || root >= ctx->comm->nranks) return ctx->fail(FSNAP_E_ARG, "fsnap_bcast_host: bad argument");
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    if (!ctx->commbuf.ensure((size_t)nbytes)) return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(collective staging) failed");
    if (ctx->comm->rank == root)
        FSNAP_HIP(hipMemcpyAsync(ctx->commbuf.p, buf, (size_t)nbytes, hipMemcpyHostToDevice, ctx->stream), "hipMemcpy(H2D)");
    FSNAP_NCCL(r->Broadcast(ctx->commbuf.p, ctx->commbuf.p, (size_t)nbytes, ncclUint8, root, ctx->comm->nccl, ctx->stream), "ncclBroadcast");
    if (ctx->comm->rank != root)
        FSNAP_HIP(hipMemcpyAsync(buf, ctx->commbuf.p, (size_t)nbytes, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpy(D2H)");
    FSNAP_HIP(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    return FSNAP_OK;
}

int fsnap_allgather_host(fsnap_ctx* ctx, const void* send, int64_t nbytes, void* recv) {
    if (!ctx) return FSNAP_E_ARG;
    Rccl* r;
    int rc;
    if ((rc = need_comm(ctx, &r))) return rc;
    if (!send || !recv || nbytes <= 0) return ctx->fail(FSNAP_E_ARG, "fsnap_allgather_host: bad argument");
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    const size_t nb = (size_t)nbytes, total = nb * (size_t)ctx->comm->nranks;
    if (!ctx->commbuf.ensure(nb + total)) return ctx->fail(FSNAP_E_NOMEM, "hipMalloc(collective staging) failed");
    char* d_send = (char*)ctx->commbuf.p;
    char* d_recv = d_send + nb;
    FSNAP_HIP(hipMemcpyAsync(d_send, send, nb, hipMemcpyHostToDevice, ctx->stream), "hipMemcpy(H2D)");
    FSNAP_NCCL(r->AllGather(d_send, d_recv, nb, ncclUint8, ctx->comm->nccl, ctx->stream), "ncclAllGather");
    FSNAP_HIP(hipMemcpyAsync(recv, d_recv, total, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpy(D2H)");
    FSNAP_HIP(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    return FSNAP_OK;
}

int fsnap_barrier(fsnap_ctx* ctx) {
    if (!ctx) return FSNAP_E_ARG;
    FSNAP_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    if (!ctx->comm) {
        FSNAP_HIP(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
        return FSNAP_OK;
    }
    double one = 1.0;
    return fsnap_allreduce_host(ctx, &one, 1, 0);      // a 1-element all-reduce + stream synchronisation
}

}  // extern "C"
